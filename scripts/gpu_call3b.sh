cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=openpano_amd/variants
rocm-smi --showproductname 2>/dev/null | head -8
( timeout 300 python scripts/sift_ab.py --steps 20 ) > gpurun_out/r05c_sift_product.txt 2>&1
grep -E "step|fault" gpurun_out/r05c_sift_product.txt
( timeout 500 python scripts/match_ab.py --steps 12 --c5-images 32 ) > gpurun_out/r05c_match_product.txt 2>&1
grep -E "call|fault" gpurun_out/r05c_match_product.txt | head
( timeout 500 python scripts/match_ab.py --steps 12 --c5-images 32 $V/libopenpano_hip_ring4.so $V/libopenpano_hip_ring3.so ) > gpurun_out/r05c_match_ab.txt 2>&1
grep -E "call|fault" gpurun_out/r05c_match_ab.txt | head -20
( timeout 300 python scripts/sift_ab.py --steps 60 $V/libopenpano_hip_pyrtrace.so ) > gpurun_out/r05c_pyrtrace.txt 2>&1
grep -E "step|trace|per step|fault" gpurun_out/r05c_pyrtrace.txt
