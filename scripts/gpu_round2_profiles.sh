#!/bin/bash
# GPU-box helper (round 2): the evidence copied into profiles/ -- kernel stats of the bench command, SQ counters of the
# attention kernels, the full bench line.  Counters and traces in separate rocprofv3 runs (kernel-trace only with --pmc).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r2p
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 10 --warmup 3 --no-pmc --no-secondary --no-cpu-baseline > $O/stats.log 2>&1
python scripts/summarize_prof.py $O/stats 13 $O/r02_kernel_stats.md > /dev/null
python - <<'PY' >> gpurun_out/r2p/r02_kernel_stats.md
import csv, glob, collections
f = glob.glob("gpurun_out/r2p/stats/*/*kernel_trace.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if any(t in k for t in ("k_gemm8", "k_wgrad8<true>", "k_attn_fwd_mfma32", "k_attn_bwd", "k_feat_text_scan", "k_gather_rows")):
        acc[(k[:60], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("\n## Roofline kernels in this trace (per launch shape)\n\n| kernel | grid | launches | mean us (min, max) |\n|---|---|---|---|")
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"| `{k}` | {g} | {len(v)} | {sum(v)/len(v):.1f} ({min(v):.1f}, {max(v):.1f}) |")
PY
# host launch calls per timed step: HIP API trace of a 10-step and a 50-step run, differenced (graph replay vs eager launches)
for mode in on off; do
  for st in 10 50; do
    rocprofv3 --hip-trace --stats --output-format csv -d $O/hip_${mode}_$st -- python bench.py --steps $st --warmup 3 --graph $mode --no-pmc --no-secondary --no-cpu-baseline > $O/hip_${mode}_$st.log 2>&1
  done
done
python - <<'PY' > gpurun_out/r2p/r02_host_launches.md
import csv, glob, re
def counts(d):
    best = {}
    for f in glob.glob(d + "/*/*hip_api_stats.csv") or glob.glob(d + "/*/*_hip_stats.csv") or glob.glob(d + "/*/*hip*stats*.csv"):
        out = {}
        for r in csv.DictReader(open(f)):
            out[r["Name"]] = int(r["Calls"])
        if sum(out.values()) > sum(best.values()):      # the python process (bench.py also starts the small mfma_peak probe)
            best = out
    return best
def ms(log):
    m = re.search(r"timed \d+ steps: ([0-9.]+) ms/step \(host enqueue ([0-9.]+)", open(log).read())
    return m.groups() if m else ("?", "?")
print("# Host-side HIP API calls per timed step (rocprofv3 --hip-trace --stats; 50-step run minus 10-step run, / 40)\n")
print("The profiler serialises the process, so the ms/step under it are not the metric; the call counts are exact.\n")
for mode, name in (("on", "hipGraph replay (bench default at N=1)"), ("off", "eager launches (--graph off; the N>1 path)")):
    a, b = counts("gpurun_out/r2p/hip_%s_10" % mode), counts("gpurun_out/r2p/hip_%s_50" % mode)
    per = {k: (b.get(k, 0) - a.get(k, 0)) / 40.0 for k in b}
    tot = sum(v for k, v in per.items() if re.search(r"Launch|Memcpy|Memset", k))
    print("## %s\n" % name)
    print("launch-type calls (kernel launches, graph launches, async copies / memsets) per step: **%.0f**\n" % tot)
    print("| calls/step | API |\n|---|---|")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
        if v >= 0.5:
            print("| %.1f | `%s` |" % (v, k))
    print()
PY
rm -rf $O/hip_on_10 $O/hip_on_50 $O/hip_off_10 $O/hip_off_50      # (traces are large: gpurun merges back at most 64 MiB)
rm -rf gpurun_out/pmc_sq
bash scripts/gpu_pmc_sq.sh attn > $O/pmc_sq.log 2>&1
python scripts/summarize_pmc.py gpurun_out/pmc_sq k_attn > $O/r02_sq_attn.txt 2>&1
rm -rf $O/stats gpurun_out/pmc_sq
python bench.py > $O/r02_bench_1gpu.json 2> $O/bench.err
tail -3 $O/bench.err
head -c 400 $O/r02_bench_1gpu.json
