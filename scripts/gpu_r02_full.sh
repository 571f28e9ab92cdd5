#!/bin/bash
# Round-2 full GPU pass: whole GPU suite, default bench, one-rank RCCL bench, rocprofv3 kernel stats, PMC passes.
# Usage: scripts/gpu_r02_full.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/${tag}_pytest.log 2>&1
echo "[pytest rc=$?]"; tail -14 gpurun_out/${tag}_pytest.log
( time timeout 900 python bench.py ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "[bench rc=$?]"; tail -6 gpurun_out/${tag}_bench.err
( OPENPANO_FORCE_DIST=1 timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-e2e --no-blend --no-ingest ) > gpurun_out/${tag}_bench_forcedist.json 2> gpurun_out/${tag}_bench_forcedist.err
echo "[forcedist rc=$?]"
( timeout 600 python bench.py --texture natural --steps 10 --no-cpu-baseline --no-e2e --no-blend --no-ingest --no-config5 ) > gpurun_out/${tag}_bench_natural.json 2> gpurun_out/${tag}_bench_natural.err
echo "[natural rc=$?]"
( timeout 600 python bench.py --scaling strong --steps 10 --no-cpu-baseline --no-e2e --no-blend --no-ingest --no-config5 ) > gpurun_out/${tag}_bench_strong1.json 2> gpurun_out/${tag}_bench_strong1.err
echo "[strong N=1 rc=$?]"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o sift -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 \
  > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-150 "$f" | head -16
bash scripts/gpu_pmc.sh ${tag} --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 2>&1 | tail -25
python - <<PY
import json
for name in ("bench", "bench_forcedist", "bench_natural", "bench_strong1"):
    try:
        d = json.loads(open("gpurun_out/${tag}_%s.json" % name).read().strip().splitlines()[-1])
        print(name, "value %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]), d["stage_ms"])
        print("  match", {k: d["match"].get(k) for k in ("ms_per_step", "descriptor_allgather_ms", "match_results_gather_ms", "allgather_bytes_per_rank")}, "frac", d["match"]["roofline"]["frac"])
        print("  ransac", d["ransac"]["ms_per_step"], d["ransac"]["stage_ms"])
        if "strong_config4" in d: print("  strong_config4", d["strong_config4"]["phase_ms"], d["strong_config4"]["job_wall_ms"])
        if "config5" in d: print("  config5", d["config5"]["phase_ms"], d["config5"]["match_roofline"]["frac"])
        if "stitch_e2e" in d: print("  e2e", d["stitch_e2e"]["ms_total"], d["stitch_e2e"]["stage_ms"])
        if "cpu_baseline" in d and d["cpu_baseline"]: print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("threads_sweep"), "gpu/cpu", d.get("gpu_over_cpu"), "parity", d.get("parity_checked"))
        if "blend" in d: print("  blend", {k: (v["ms_per_blend"], v["roofline"]["frac"]) for k, v in d["blend"].items()})
        if "protocol" in d: print("  protocol", d["protocol"]["value_mat32f"], d["protocol"]["value_uint8"])
    except Exception as e:
        print(name, "parse failed", e)
PY
