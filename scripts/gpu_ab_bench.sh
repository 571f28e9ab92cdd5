#!/bin/bash
# same-box A/B of bench.py sections between the product library and a variant build (OPENPANO_HIP_LIB):
#   scripts/gpu_ab_bench.sh <tag> <variant .so> [bench flags]
tag=$1; var=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for rep in 1 2; do
  for which in product variant; do
    if [ $which = variant ]; then export OPENPANO_HIP_LIB=$PWD/$var; else unset OPENPANO_HIP_LIB; fi
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ingest --no-e2e --no-config5 "$@" > gpurun_out/${tag}_${which}_$rep.json 2> gpurun_out/${tag}_${which}_$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_${which}_$rep.json"))
out = ["$which $rep", "step %.4f" % d["ms_per_step"]]
if d.get("match"): out.append("match %.3f" % d["match"]["ms_per_step"])
if d.get("ransac"): out.append("ransac %.3f %s" % (d["ransac"]["ms_per_step"], d["ransac"]["stage_ms"]))
for k, v in (d.get("blend") or {}).items(): out.append("%s %s" % (k, v["stage_ms"]))
print("  ".join(out))
PY
  done
done
