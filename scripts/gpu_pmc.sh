#!/bin/bash
# GPU-box helper: HBM-side traffic counters of the hot kernels (separate --pmc passes, kernel-trace only)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc/$C -- python scripts/bench_kernels.py attn conv > gpurun_out/pmc/$C.log 2>&1
  echo "pass $C done"
done
find gpurun_out/pmc -name "*counter_collection*" | head
