cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=openpano_amd/variants
( timeout 500 python scripts/match_ab.py --steps 12 --c5-images 32 $V/libopenpano_hip_ring4.so $V/libopenpano_hip_ring3.so ) > gpurun_out/r05c_match_ab.txt 2>&1
grep -E "call|Error|error" gpurun_out/r05c_match_ab.txt | head -20
for R in 4 3; do
  ( OPENPANO_HIP_LIB=$PWD/$V/libopenpano_hip_ring$R.so timeout 600 python -m pytest tests/test_gpu_match.py -m gpu -q -x ) > gpurun_out/r05c_pytest_ring$R.log 2>&1
  tail -3 gpurun_out/r05c_pytest_ring$R.log
done
( timeout 300 python scripts/sift_ab.py --steps 60 $V/libopenpano_hip_pyrtrace.so ) > gpurun_out/r05c_pyrtrace.txt 2>&1
grep -E "step|trace|per step" gpurun_out/r05c_pyrtrace.txt
