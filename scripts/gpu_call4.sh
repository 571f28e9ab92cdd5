cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=openpano_amd/variants
for v in product ring4_NONE ring4_OP_KO_DMA ring4_OP_KO_LDS ring4_OP_KO_TOPK ring4_OP_KO_BARRIER; do
  lib=$PWD/$V/libopenpano_hip_$v.so; [ $v = product ] && lib=$PWD/openpano_amd/libopenpano_hip.so
  rm -rf /tmp/prof_$v
  ( OPENPANO_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o m -- python scripts/match_ab.py --steps 4 --c5-images 32 ) > /tmp/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; [ -n "$f" ] && grep -E "k_match_sweep|k_match_slow" "$f" | cut -d, -f1-8 | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-150
done > gpurun_out/r05d_match_knockouts.txt 2>&1
cat gpurun_out/r05d_match_knockouts.txt
