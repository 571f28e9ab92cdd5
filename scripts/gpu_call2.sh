cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=openpano_amd/variants
( timeout 400 python scripts/sift_ab.py --steps 100 --json gpurun_out/r05b_sift_ab.json $V/libopenpano_hip_rowmerge.so $V/libopenpano_hip_pyrfma.so $V/libopenpano_hip_seg32.so $V/libopenpano_hip_seg48.so $V/libopenpano_hip_descw7.so $V/libopenpano_hip_descw8.so $V/libopenpano_hip_rowmerge.so ) > gpurun_out/r05b_sift_ab.txt 2>&1
grep step gpurun_out/r05b_sift_ab.txt
( timeout 200 python scripts/sift_ab.py --steps 40 --config5 --images 32 $V/libopenpano_hip_rowmerge.so ) > gpurun_out/r05b_sift_ab_c5.txt 2>&1
grep step gpurun_out/r05b_sift_ab_c5.txt
( timeout 300 python scripts/ba_probe.py 4 8 16 32 ) > gpurun_out/r05b_ba_probe.txt 2>&1
grep -E "best of 3|calls 111" gpurun_out/r05b_ba_probe.txt
( timeout 600 python -m pytest tests -m gpu -q -x -k "pairwise_table or rehearsal or camera or blend_trig" ) > gpurun_out/r05b_pytest_sel.log 2>&1
tail -3 gpurun_out/r05b_pytest_sel.log
( timeout 600 python scripts/scale_rehearsal.py r05b ) > gpurun_out/r05b_rehearsal.txt 2>&1
tail -4 gpurun_out/r05b_rehearsal.txt
