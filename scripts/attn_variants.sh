#!/bin/bash
# GPU-box helper: rebuild attention_mfma.hip with different -D switches and time the dec0 kernels
cd ${GRAFT_REPO_ROOT:-.}
for flags in "" "-DFA_DKV_WAVES=8" "-DFA_DKV_WAVES=8 -DFA_BQ2=32" "-DFA_DKV_WAVES=2"; do
  touch scenesplat_amd/csrc/attention_mfma.hip
  SS_EXTRA_HIPCC_FLAGS="$flags" python -m scenesplat_amd.build > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== flags: [$flags]"
  timeout -k 10 200 python scripts/bench_kernels.py attn 2>&1 | grep "attn L"
done
touch scenesplat_amd/csrc/attention_mfma.hip; python -m scenesplat_amd.build > /dev/null 2>&1
