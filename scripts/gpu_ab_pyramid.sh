#!/bin/bash
# Run on the GPU box (via gpurun): A/B of the two scale-space kernels on the bench workload in one
# session (OPENPANO_PYRAMID=tiles forces the tiled kernel K3 for the shipped Gaussian bank) and a
# kernel trace of the whole-pipeline section.   Usage: scripts/gpu_ab_pyramid.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for mode in rows tiles; do
  OPENPANO_PYRAMID=$mode python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ingest --no-match --no-blend > gpurun_out/${tag}_ab_${mode}.json 2>/dev/null
done
python - <<PY
import json
out = {}
for mode in ("rows", "tiles"):
    d = json.load(open("gpurun_out/${tag}_ab_%s.json" % mode))
    out[mode] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "stage_ms": d["stage_ms"], "roofline": d["roofline"]}
json.dump(out, open("gpurun_out/${tag}_pyramid_ab.json", "w"), indent=1)
print(json.dumps({k: (v["ms_per_step"], v["stage_ms"]["build pyramid"]) for k, v in out.items()}))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_e2e_prof -o e2e -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest \
  > gpurun_out/${tag}_e2e_under_rocprof.json 2> gpurun_out/${tag}_e2e_prof.err
f=$(find gpurun_out/${tag}_e2e_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-120 "$f" | head -8
