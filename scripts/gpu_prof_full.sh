#!/bin/bash
# GPU-box helper: full-size bench + rocprofv3 kernel stats (writes under gpurun_out/<tag>)
set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-prof_full}
shift
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/$TAG.log 2>&1
tail -3 gpurun_out/$TAG.log
find gpurun_out/$TAG -name "*kernel_stats*" | head
