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
rm -rf gpurun_out/pmc_sq
bash scripts/gpu_pmc_sq.sh attn > $O/pmc_sq.log 2>&1
python scripts/summarize_pmc.py gpurun_out/pmc_sq k_attn > $O/r02_sq_attn.txt 2>&1
python bench.py > $O/r02_bench_1gpu.json 2> $O/bench.err
tail -3 $O/bench.err
head -c 400 $O/r02_bench_1gpu.json
