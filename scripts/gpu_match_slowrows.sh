#!/bin/bash
# exact-scan (slow path) row counts of the matcher, old vs new trace builds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in match9old match9; do
  export OPENPANO_HIP_LIB=$PWD/openpano_amd/variants/libopenpano_hip_$v.so
  echo "== $v"
  python bench.py --no-cpu-baseline --no-e2e --no-blend --no-ingest --steps 1 --warmup 1 2>&1 >/dev/null | grep "match trace" | sort | uniq -c | sort -rn | head -6
done
