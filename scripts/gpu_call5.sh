cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=openpano_amd/variants
( timeout 300 python scripts/sift_ab.py --steps 100 $V/libopenpano_hip_pyrvm.so $V/libopenpano_hip_pyrvm.so ) > gpurun_out/r05f_sift_ab.txt 2>&1
grep -E "step|fault" gpurun_out/r05f_sift_ab.txt
( timeout 200 python scripts/sift_ab.py --steps 30 --config5 --images 32 $V/libopenpano_hip_pyrvm.so ) > gpurun_out/r05f_sift_ab_c5.txt 2>&1
grep -E "step|fault" gpurun_out/r05f_sift_ab_c5.txt
( OPENPANO_HIP_LIB=$PWD/$V/libopenpano_hip_pyrvm.so timeout 600 python -m pytest tests/test_gpu_sift.py tests/test_config_variants.py -m gpu -q -x ) > gpurun_out/r05f_pytest_pyrvm.log 2>&1
tail -3 gpurun_out/r05f_pytest_pyrvm.log
