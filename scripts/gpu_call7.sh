cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=openpano_amd/variants
( timeout 300 python scripts/blend_ab.py --steps 20 $V/libopenpano_hip_blendvm.so $V/libopenpano_hip_blendvm.so ) > gpurun_out/r05_blend_ab.txt 2>&1
grep -E "crc|rror|fault" gpurun_out/r05_blend_ab.txt
( OPENPANO_HIP_LIB=$PWD/$V/libopenpano_hip_blendvm.so timeout 250 python -m pytest tests/test_gpu_blend.py -m gpu -q -x --timeout 100 ) > gpurun_out/r05_pytest_blendvm.log 2>&1
tail -3 gpurun_out/r05_pytest_blendvm.log
( timeout 200 python scripts/sift_ab.py --steps 60 $V/libopenpano_hip_pyrtrace.so ) > gpurun_out/r05_pyrtrace_v2.txt 2>&1
grep -E "trace|per step" gpurun_out/r05_pyrtrace_v2.txt
