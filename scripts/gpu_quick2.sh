#!/bin/bash
# quick GPU check: selected tests + a short bench; usage: scripts/gpu_quick2.sh "<pytest args>" "<bench args>" "<json keys to print>"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest $1 -x -q 2>&1 | tail -8
timeout 600 python bench.py $2 2>gpurun_out/quick_bench.err > gpurun_out/quick_bench.json; echo "[bench rc=$?]"; tail -3 gpurun_out/quick_bench.err
python - "$3" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "stage_ms", d["stage_ms"])
for k in sys.argv[1].split(","):
    if k and k in d:
        v = d[k]
        if isinstance(v, dict):
            v = {a: b for a, b in v.items() if a not in ("cpu_baseline", "roofline", "workload")}
        print(k, json.dumps(v)[:1500])
PY
