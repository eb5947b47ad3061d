#!/bin/bash
# GPU-box helper: power and shader clock (rocm-smi) while one kernel family runs back to back -- is a kernel limited by the chip's power budget?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r3pw; mkdir -p $O
python scripts/conv_walk_probe.py 3000 > $O/conv.log 2>&1 &
PID=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -6
  echo ---
  sleep 1.5
done
wait $PID
tail -4 $O/conv.log
