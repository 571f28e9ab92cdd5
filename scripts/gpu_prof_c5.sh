#!/bin/bash
# kernel stats of the matcher on the config-5 job (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c5prof
rocprofv3 --kernel-trace --stats -d gpurun_out/c5prof -o c5 --output-format csv -- python bench.py --no-cpu-baseline --no-e2e --no-blend --no-ingest --steps 2 --warmup 1 > gpurun_out/c5prof/bench.json 2> gpurun_out/c5prof/bench.err
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/c5prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if "match" in n or "split" in n:
        agg[(n.split("(")[0], r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(agg.items()):
    v2 = sorted(v)
    print("%-28s grid %-10s n %-3d median %.4f ms  max %.4f" % (k[0], k[1], len(v), v2[len(v2) // 2], v2[-1]))
PY
