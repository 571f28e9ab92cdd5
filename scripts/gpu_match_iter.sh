#!/bin/bash
# matcher iteration: parity tests, per-phase trace (variant match9 if built), bench match + config5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
[ -f openpano_amd/variants/libopenpano_hip_match9.so ] && timeout 300 python scripts/match_trace.py 2>&1 | grep "K="
bash scripts/gpu_variants_c5.sh ${1:-none}
