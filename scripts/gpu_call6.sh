cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=openpano_amd/variants
nproc; uptime
( OPENPANO_HIP_LIB=$PWD/$V/libopenpano_hip_pyrvm.so timeout 170 python -m pytest tests/test_gpu_sift.py tests/test_config_variants.py -m gpu -q -x --durations=12 --timeout 40 ) > gpurun_out/r05g_pytest_pyrvm.log 2>&1
tail -25 gpurun_out/r05g_pytest_pyrvm.log
