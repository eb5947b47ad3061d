#!/bin/bash
# GPU-box helper: let PyTorch TunableOp pick the hipBLASLt/rocBLAS solution per GEMM shape of the workload.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=1
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
export PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_results.csv
( while true; do sleep 45; echo "[tune] alive $(date +%s)"; done ) &
KA=$!
python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_tuned.log 2> gpurun_out/bench_tuned.err
kill $KA
grep -E "^\{|bench\]" gpurun_out/bench_tuned.log gpurun_out/bench_tuned.err | cut -c1-300
ls -la gpurun_out/tunableop_results*.csv | head
