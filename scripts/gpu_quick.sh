#!/bin/bash
# quick GPU iteration: SIFT/match parity tests + a short bench (no CPU baseline)
# Usage: scripts/gpu_quick.sh <tag> [pytest -k expr]
tag=${1:-q}; kexpr=${2:-"sift or match"}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -4
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ingest --no-e2e > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || tail -5 gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print("value %.4g kp/s  ms/step %.3f"%(d["value"], d["ms_per_step"]))
print("stage_ms", d["stage_ms"])
m=d.get("match") or {}
print("match ms/step", m.get("ms_per_step"), m.get("stage_ms"))
PY
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print("ransac", d.get("ransac"))
b=d.get("blend") or {}
for k,v in b.items(): print("blend", k, v["ms_per_blend"], v["canvas"], v["stage_ms"], v["roofline"]["frac"])
PY
