#!/bin/bash
# Run on the GPU box (via gpurun): headline bench + rocprofv3 kernel trace of the same command.
# Usage: scripts/gpu_bench_profile.sh <tag>   -> gpurun_out/<tag>_*.{json,csv}
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -3 gpurun_out/${tag}_bench.err; cat gpurun_out/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o sift -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline \
  > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err
find gpurun_out/${tag}_prof -name "*stats*" | head
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
