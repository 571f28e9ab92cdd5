#!/bin/bash
# config-5 match stage with each variant library (timing experiments)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "" $(ls openpano_amd/variants/libopenpano_hip_${1:-}*.so 2>/dev/null); do
  if [ -n "$lib" ]; then export OPENPANO_HIP_LIB=$PWD/$lib; else unset OPENPANO_HIP_LIB; fi
  python bench.py --no-cpu-baseline --no-e2e --no-blend --no-ingest --steps 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-product}', 'c4', d['match']['stage_ms'].get('matcher mfma forward'), d['match']['stage_ms'].get('matcher mfma reverse'), 'c5', d['config5']['match_stage_ms'].get('matcher mfma forward'), d['config5']['match_stage_ms'].get('matcher mfma reverse'))
"
done
