#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in match9 match9old; do
  [ -f openpano_amd/variants/libopenpano_hip_$v.so ] || continue
  echo "== $v"
  OPENPANO_TRACE_LIB=$v timeout 600 python scripts/match_trace.py 2>&1 | grep "^K=\|residents"
done
