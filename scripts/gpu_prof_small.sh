#!/bin/bash
# GPU-box helper: small-size bench + rocprofv3 kernel stats (writes under gpurun_out/)
set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
NS=${1:-128}
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 --n-side $NS --no-cpu-baseline > gpurun_out/bench_small.log 2>&1
tail -4 gpurun_out/bench_small.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -- python bench.py --steps 2 --warmup 1 --n-side $NS --no-cpu-baseline > gpurun_out/prof_small.log 2>&1
find gpurun_out/prof_small -name "*kernel_stats*" | head
