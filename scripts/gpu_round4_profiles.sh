#!/bin/bash
# GPU-box helper (round 4): the evidence copied into profiles/ -- kernel stats of the bench command with the per-launch-shape
# durations of every roofline kernel, SQ counters of the attention kernels (ubench harness), the full bench line.
# Counters and traces in separate rocprofv3 runs (kernel-trace only with --pmc).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r4p
rm -rf $O; mkdir -p $O
# (no replay/eager calibration in the traced run: its 16 extra steps would be in the per-step averages; the replay is what is traced)
SS_BENCH_CALIBRATE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 10 --warmup 3 --no-pmc --no-secondary --no-cpu-baseline > $O/stats.log 2>&1
python - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r4p"
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
open(O + "/r04_kernel_stats.md", "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
PY
rm -rf $O/stats
if [ "$1" != "quick" ]; then
  # SQ counters of the three head-major attention kernels at the dec0 shape (stand-alone harness: fwd, dQ, dK/dV)
  scripts/pmc_ubench.sh attn scripts/ubench/bin/attn_hm_bench 48 > /dev/null 2>&1
  cp gpurun_out/r3/pmc_attn/summary.txt $O/r04_sq_attn.txt 2>/dev/null
  python bench.py > $O/r04_bench_1gpu.json 2> $O/bench.err
  tail -3 $O/bench.err
fi
# ---- round 4 additions: the Mix3D (duplicate-voxel) regime and the pointops replacements, kernel rows of each ----
for t in mix3d pointops; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$t -- python scripts/${t}_trace.py > $O/$t.log 2>&1
  python - "$t" <<'PY'
import csv, glob, os, sys
t = sys.argv[1]; O = "gpurun_out/r4p"
stats = max(glob.glob(O + "/%s/*/*kernel_stats.csv" % t), key=os.path.getsize)
rows = list(csv.DictReader(open(stats)))
head = open(O + "/%s.log" % t).read().strip().splitlines()[-1]
out = ["source: rocprofv3 --kernel-trace --stats -- python scripts/%s_trace.py" % t, head, "", "| total ms | calls | avg us | kernel |", "|---|---|---|---|"]
for r in rows[:40]:
    out.append("| %.3f | %d | %.1f | `%s` |" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
if t == "mix3d":
    tap = [r for r in rows if "k_gather_rows" in r["Name"]]
    out += ["", "per-tap path markers: k_gather_rows launches = %d (the per-tap fp32 conv issues one per tap and call: 125 + 27 x 4 per step if it ran)" % sum(int(r["Calls"]) for r in tap),
            "duplicate-voxel kernels: " + ", ".join("%s x %s" % (r["Name"].split("(")[0][-40:], r["Calls"]) for r in rows if "k_dup_" in r["Name"] or "k_subm_f32" in r["Name"])]
open(O + "/r04_%s_kernels.md" % t, "w").write("\n".join(out) + "\n")
print("\n".join(out[:14]))
PY
  rm -rf $O/$t
done
