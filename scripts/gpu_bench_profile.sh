#!/bin/bash
# Run on the GPU box (via gpurun): headline bench + rocprofv3 kernel trace of the same command
# (+ PMC passes with scripts/gpu_pmc.sh).   Usage: scripts/gpu_bench_profile.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -3 gpurun_out/${tag}_bench.err; cat gpurun_out/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o sift -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e \
  > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -24
bash scripts/gpu_pmc.sh ${tag} --steps 2 --warmup 1 --no-cpu-baseline --no-e2e
