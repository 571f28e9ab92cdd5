#!/bin/bash
# time the SIFT stages with each variant library under openpano_amd/variants (timing experiments)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "" $(ls openpano_amd/variants/libopenpano_hip_${1:-}*.so 2>/dev/null); do
  if [ -n "$lib" ]; then export OPENPANO_HIP_LIB=$PWD/$lib; else unset OPENPANO_HIP_LIB; fi
  python bench.py --no-cpu-baseline --no-e2e --no-config5 --no-blend --no-ingest ${2:---no-match} --steps 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-product}', 'ms_per_step %.4f' % d['ms_per_step'], {k: v for k, v in d['stage_ms'].items()}, d.get('match', {}).get('stage_ms') if d.get('match') else '')
"
done
