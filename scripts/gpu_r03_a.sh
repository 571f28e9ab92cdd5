#!/bin/bash
# Round-3 full pass: whole GPU suite + default bench (device-resident match lists -> RANSAC, config-5 parity block).
tag=${1:-r03k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/${tag}_pytest.log 2>&1
echo "[pytest rc=$?]"; tail -16 gpurun_out/${tag}_pytest.log
( time timeout 900 python bench.py ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "[bench rc=$?]"; tail -8 gpurun_out/${tag}_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
    print("value %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]), d["stage_ms"])
    print("match", d["match"]["ms_per_step"], d["match"]["stage_ms"], "frac", d["match"]["roofline"]["frac"])
    print("ransac", d["ransac"]["ms_per_step"], d["ransac"]["stage_ms"])
    print("config5", d["config5"]["phase_ms"], d["config5"]["match_stage_ms"], d["config5"]["match_roofline"]["frac"], d["config5"].get("parity"))
    print("e2e", d["stitch_e2e"]["ms_total"], d["stitch_e2e"]["stage_ms"])
    print("parity", d.get("parity_checked"), "gpu/cpu", d.get("gpu_over_cpu"))
except Exception as e:
    print("parse failed", e)
PY
