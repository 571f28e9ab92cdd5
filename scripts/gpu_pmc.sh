#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 PMC passes over the bench's kernels.
# Counters are collected in their own runs (with --kernel-trace only), one --pmc set per pass:
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots), MI355X_MICROARCH.md "rocprofv3 PMC slots".
# Usage: scripts/gpu_pmc.sh <tag> [bench args]   -> gpurun_out/<tag>_pmc_{fetch,write,sq}/...csv
tag=${1:-r01}; shift
args=${@:---steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-configs}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${tag}_pmc_${name} -o pmc -- python bench.py $args \
    > gpurun_out/${tag}_pmc_${name}.json 2> gpurun_out/${tag}_pmc_${name}.err
  echo "[pmc] pass $name rc=$?"; find gpurun_out/${tag}_pmc_${name} -name "*counter_collection.csv" | head -2
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE
# matrix pipe: the sweeps issue v_mfma_f32_32x32x16_bf16 (MOPS_BF16 x 512 = flop; BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) = MfmaUtil of counter_defs.yaml)
run mfma SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
python scripts/pmc_summary.py $tag
