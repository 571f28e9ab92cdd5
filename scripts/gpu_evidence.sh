#!/bin/bash
# Evidence pass for the CURRENT build of the library (everything lands under gpurun_out/<tag>_*; copy what is kept to profiles/):
#   1 whole GPU suite                         -> <tag>_pytest.log
#   2 default bench (parity + CPU baselines)  -> <tag>_bench.json
#   3 rocprofv3 --kernel-trace --stats of a config-4 bench run (numpy-made inputs: no torch kernels in the table) -> <tag>_prof/
#   4 the four PMC passes of scripts/gpu_pmc.sh (pmc json carries the library hash; bench.py replays traffic only for the same hash)
#   5 one-rank RCCL run (OPENPANO_FORCE_DIST=1): the exchange / gather code on the nccl backend
#   6 micro-benchmarks DESIGN.md quotes: mfma_power, mfma_valu_overlap, lds_atomic_order, the sweep's phase trace (if that variant is built)
#   7 the SIFT step on 38 / 19 / 10 / 5 of the config-4 images (one rank's share at N = 1 / 2 / 4 / 8)  -> <tag>_sift_shares.txt
#   8 the whole config-5 job against the oracle, all 8128 pairs (OPENPANO_FULL_C5=1; minutes of host time) -> <tag>_config5_all_pairs.txt
#     + the record the test writes itself, tied to the library's hash -> config5_all_pairs.json (copy to profiles/config5_all_pairs_latest.json)
#   9 one-device rehearsal of the 1 / 2 / 4 / 8-rank strong-scaled jobs, config 4 and config 5 (scripts/scale_rehearsal.py)
#     -> <tag>_scale_rehearsal.json (copy to profiles/scale_rehearsal_latest.json: bench.py --gpus N prints it as `predicted`)
#  10 per-dispatch matrix-pipe counters of the config-5 sweeps -> <tag>_config5_mfma.json (copy to profiles/config5_mfma_latest.json)
# Usage: scripts/gpu_evidence.sh <tag> [sections, default "1 2 3 4 5 6 7"]
tag=${1:-r05}; what=${2:-"1 2 3 4 5 6 7"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
has() { [[ " $what " == *" $1 "* ]]; }
if has 4; then        # first: the bench of the same call (section 2) replays these counters when the library hash matches
  bash scripts/gpu_pmc.sh ${tag} --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest --no-configs 2>&1 | tail -30
  [ -f gpurun_out/${tag}_pmc.json ] && cp gpurun_out/${tag}_pmc.json profiles/pmc_latest.json
fi
if has 1; then
  ( time timeout 1500 python -m pytest tests -m gpu -q --durations=6 ) > gpurun_out/${tag}_pytest.log 2>&1
  echo "[pytest rc=$?]"; tail -12 gpurun_out/${tag}_pytest.log
fi
if has 2; then
  ( time timeout 900 python bench.py ) > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
  echo "[bench rc=$?]"; tail -4 gpurun_out/${tag}_bench.err
  cp gpurun_out/bench_detail.json gpurun_out/${tag}_bench.json          # the full record; the line is its digest (bench_line.py)
  # the driver's own invocation as well: fewer steps, colder clock
  ( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/${tag}_bench_line_driver_style.json 2>> gpurun_out/${tag}_bench.err
  cp gpurun_out/bench_detail.json gpurun_out/${tag}_bench_driver_style.json
fi
if has 3; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o sift -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest --no-configs \
    > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err
  f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv && cut -c1-170 "$f" | head -22
fi

if has 5; then
  ( OPENPANO_FORCE_DIST=1 timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-e2e --no-blend --no-ingest --no-configs ) > gpurun_out/${tag}_bench_line_forcedist.json 2> gpurun_out/${tag}_bench_forcedist.err
  echo "[forcedist rc=$?]"; cp gpurun_out/bench_detail.json gpurun_out/${tag}_bench_forcedist.json
fi
if has 6; then
  for b in mfma_power mfma_valu_overlap lds_atomic_order; do
    [ -x scripts/ubench/$b ] && ( echo "== scripts/ubench/$b"; timeout 120 scripts/ubench/$b ) > gpurun_out/${tag}_ubench_$b.txt 2>&1
    tail -8 gpurun_out/${tag}_ubench_$b.txt
  done


fi
if has 7; then
  ( echo "# scripts/sift_ab.py --steps 60 --images N: the SIFT step of one rank's share of BASELINE config 4 under strong scaling (N = 38 / 19 / 10 / 5 images = 1 / 2 / 4 / 8 ranks), one MI355X"
    for n in 38 19 10 5; do echo "images $n"; timeout 200 python scripts/sift_ab.py --steps 60 --images $n 2>&1 | grep step; done ) > gpurun_out/${tag}_sift_shares.txt
  cat gpurun_out/${tag}_sift_shares.txt
fi
if has 8; then
  ( time OPENPANO_FULL_C5=1 timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k whole_match_job ) > gpurun_out/${tag}_config5_all_pairs.txt 2>&1
  tail -6 gpurun_out/${tag}_config5_all_pairs.txt
fi
if has 9; then
  ( timeout 600 python scripts/scale_rehearsal.py ${tag} ) > gpurun_out/${tag}_rehearsal.txt 2>&1
  tail -3 gpurun_out/${tag}_rehearsal.txt
fi
if has 10; then      # per-dispatch matrix-pipe counters of the config-5 sweeps -> <tag>_config5_mfma.json (copy to profiles/config5_mfma_latest.json)
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc_c5mfma -o pmc -- \
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-blend --no-ingest --no-configs > gpurun_out/${tag}_pmc_c5mfma.json 2> gpurun_out/${tag}_pmc_c5mfma.err
  python scripts/pmc_config5_mfma.py gpurun_out/${tag}_pmc_c5mfma gpurun_out/${tag}_config5_mfma.json | tail -30
fi
python - <<PY
import json
for name in ("bench", "bench_forcedist"):
    try:
        d = json.loads(open("gpurun_out/${tag}_%s.json" % name).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "not parsed:", e); continue
    print(name, "value %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]), d["stage_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"),
          "line bytes", len(open("gpurun_out/${tag}_%s.json" % name.replace("bench", "bench_line")).read()))
    m = d.get("match") or {}
    print("  match", m.get("ms_per_step"), m.get("stage_ms"), "allgather", m.get("descriptor_allgather_ms"), "gather", m.get("match_results_gather_ms"))
    print("  ransac", (d.get("ransac") or {}).get("ms_per_step"), (d.get("ransac") or {}).get("stage_ms"))
    if "config5" in d: print("  config5", d["config5"]["phase_ms"], d["config5"]["match_roofline"]["frac"], (d["config5"].get("parity") or {}).get("ok"))
    if "blend" in d: print("  blend", {k: (round(v["ms_per_blend"], 3), round(v["roofline"]["frac"], 3)) for k, v in d["blend"].items()})
    if "protocol" in d: print("  protocol", d["protocol"]["ms_per_step_mat32f"], d["protocol"]["ms_per_step_uint8"])
    if d.get("cpu_baseline"): print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "gpu/cpu", d.get("gpu_over_cpu"), "parity", d.get("parity_checked"))
PY
