#!/bin/bash
# rocprofv3 --kernel-trace --stats of the SIFT step (GPU box):  scripts/gpu_trace_sift.sh <tag> [sift_ab args]  -> gpurun_out/<tag>_trace/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o sift -- python scripts/sift_ab.py --steps 10 "$@" > gpurun_out/${tag}_trace.log 2>&1
f=$(find gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"k_\w+", r["Name"])
    print("%-28s calls %5s  avg %9.2f us  total %6.2f %%" % (m.group(0) if m else r["Name"][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
