#!/bin/bash
# rocprofv3 SQ counter passes over the SIFT step of one or more library builds (GPU box):
#   scripts/gpu_pmc_sift.sh <tag> <lib.so | product> ...   -> gpurun_out/<tag>_<name>_sq{1,2}/ + a per-kernel summary on stdout
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for lib in "$@"; do
  name=$(basename $lib .so); name=${name#libopenpano_hip_}
  if [ $lib = product ]; then arg=""; else arg="--no-product $lib"; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/${tag}_${name}_sq1 -o pmc -- python scripts/sift_ab.py --steps 2 $arg > gpurun_out/${tag}_${name}_sq1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/${tag}_${name}_sq2 -o pmc -- python scripts/sift_ab.py --steps 2 $arg > gpurun_out/${tag}_${name}_sq2.log 2>&1
  python scripts/pmc_sift_summary.py ${tag} ${name}
done
