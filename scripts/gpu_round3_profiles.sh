#!/bin/bash
# GPU-box helper (round 3): the evidence copied into profiles/ -- kernel stats of the bench command with the per-launch-shape
# durations of every roofline kernel, SQ counters of the attention kernels (ubench harness), the full bench line.
# Counters and traces in separate rocprofv3 runs (kernel-trace only with --pmc).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r3p
rm -rf $O; mkdir -p $O
# (no replay/eager calibration in the traced run: its 16 extra steps would be in the per-step averages; the replay is what is traced)
SS_BENCH_CALIBRATE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 10 --warmup 3 --no-pmc --no-secondary --no-cpu-baseline > $O/stats.log 2>&1
python - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r3p"
# bench.py starts a small child (the mfma_peak probe): the python process is the one with the LARGEST trace
stats = max(glob.glob(O + "/stats/*/*kernel_stats.csv"), key=os.path.getsize)
trace = max(glob.glob(O + "/stats/*/*kernel_trace.csv"), key=os.path.getsize)
steps = 13.0
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
out = ["source: SS_BENCH_CALIBRATE=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-pmc --no-secondary --no-cpu-baseline",
       f"GPU kernel time {tot / 1e6 / steps:.2f} ms/step, {calls / steps:.0f} launches/step (over {steps:g} steps incl. warm-up; the roofline probes and", 
       "the plan builds of the side stream are in the same trace)", "", "| ms/step | % | calls/step | avg us | kernel |", "|---|---|---|---|---|"]
for r in rows[:45]:
    out.append(f"| {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {float(r['Percentage']):.2f} | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | `{r['Name'][:110]}` |")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    k = r["Kernel_Name"].split("(")[0]
    if any(t in k for t in ("k_gemm8", "k_wgrad8", "k_attn", "k_gather_add", "k_segment_reduce", "k_segment_bcast", "k_hm_pack", "k_feat_text_scan", "k_gather_rows", "k_subm_f32")):
        acc[(k[:64], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out += ["", "## Roofline kernels in this trace, per launch shape", "", "| kernel | grid (threads) | launches | mean us (min, max) |", "|---|---|---|---|"]
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:36]:
    out.append(f"| `{k}` | {g} | {len(v)} | {sum(v)/len(v):.1f} ({min(v):.1f}, {max(v):.1f}) |")
open(O + "/r03_kernel_stats.md", "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
PY
rm -rf $O/stats
if [ "$1" != "quick" ]; then
  # SQ counters of the three head-major attention kernels at the dec0 shape (stand-alone harness: fwd, dQ, dK/dV)
  scripts/pmc_ubench.sh attn scripts/ubench/bin/attn_hm_bench 48 > /dev/null 2>&1
  cp gpurun_out/r3/pmc_attn/summary.txt $O/r03_sq_attn.txt 2>/dev/null
  python bench.py > $O/r03_bench_1gpu.json 2> $O/bench.err
  tail -3 $O/bench.err
fi
