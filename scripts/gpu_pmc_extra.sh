#!/bin/bash
# Extra PMC passes (matrix-pipe occupancy, LDS waits, L2 hit rates) over the bench's kernels, for
# planning the next round of kernel work.  Usage: scripts/gpu_pmc_extra.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
args="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-ingest --no-blend"
run() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${tag}_pmcx_${name} -o pmc -- python bench.py $args \
    > gpurun_out/${tag}_pmcx_${name}.json 2> gpurun_out/${tag}_pmcx_${name}.err
  echo "[pmcx] pass $name rc=$?"; }
run mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
run l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
python - <<PY
import csv, glob, json, os, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/${tag}_pmcx_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", row.get("Kernel_Name", "")); k = m.group(1) if m else row.get("Kernel_Name", "")[:40]
        a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
out = {k: {c: s / max(n, 1) for c, (s, n) in cs.items()} for k, cs in acc.items()}
json.dump(out, open("gpurun_out/${tag}_pmc_extra.json", "w"), indent=1, sort_keys=True)
for k in ("k_pyramid_rows", "k_descriptor", "k_match_sweep", "k_grey_octaves"):
    print(k, {c: round(v) for c, v in out.get(k, {}).items()})
PY
