#!/bin/bash
# Round-2 first GPU pass: the whole GPU test suite (incl. the whole-job parity tests), the bench
# (default + one-rank RCCL path), a kernel trace.   Usage: scripts/gpu_r02_a.sh <tag>
tag=${1:-r02a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/${tag}_pytest.log 2>&1
echo "[pytest rc=$?]"; tail -25 gpurun_out/${tag}_pytest.log
( time timeout 900 python bench.py ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "[bench rc=$?]"; tail -12 gpurun_out/${tag}_bench.err; cut -c1-3000 gpurun_out/${tag}_bench.json
( time OPENPANO_FORCE_DIST=1 timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-e2e --no-blend --no-ingest ) > gpurun_out/${tag}_bench_forcedist.json 2> gpurun_out/${tag}_bench_forcedist.err
echo "[forcedist rc=$?]"; tail -5 gpurun_out/${tag}_bench_forcedist.err; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_bench_forcedist.json").read().strip().splitlines()[-1])
    print({k: d["match"].get(k) for k in ("descriptor_allgather_ms", "match_results_gather_ms", "allgather_bytes_per_rank", "ms_per_step")})
    print("config5", {k: d.get("config5", {}).get(k) for k in ("phase_ms", "keypoints_per_image", "image_pairs", "matches")})
except Exception as e:
    print("forcedist parse failed", e)
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o sift -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 \
  > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-170 "$f" | head -30
