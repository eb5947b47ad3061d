#!/bin/bash
# GPU-box helper: anatomy of ONE replayed bench step from the rocprofv3 kernel trace -- span, busy time and idle gaps of the compute
# queue, kernels ranked, and the tail of small launches.  (The plan build runs on a side stream: other queue ids.)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r4s
rm -rf $O; mkdir -p $O
SS_BENCH_CALIBRATE=0 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python bench.py --steps 6 --warmup 2 --no-pmc --no-secondary --no-cpu-baseline > $O/run.log 2>&1
python - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/r4s"
trace = max(glob.glob(O + "/tr/*/*kernel_trace.csv"), key=os.path.getsize)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
# the replayed steps: find the repeating marker kernel (k_attn_hm_fwd<48> first launch of each step = 2 per step)
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_attn_hm_fwd<48>")]
print("queue column:", qkey, "| kernels:", len(rows), "| k_attn_hm_fwd<48> launches:", len(marks))
# a step = from one k_cast_bf16_group (first kernel of the captured step) to the next; take the last complete one
starts = [i for i, r in enumerate(rows) if "k_cast_bf16_group" in r["Kernel_Name"]]
print("k_cast_bf16_group launches:", len(starts))
s0, s1 = starts[-2], starts[-1]
step = rows[s0:s1]
byq = collections.Counter(r[qkey] for r in step)
mainq = byq.most_common(1)[0][0]
print("step kernels:", len(step), "by queue:", dict(byq))
main = [r for r in step if r[qkey] == mainq]
t0, t1 = int(main[0]["Start_Timestamp"]), int(main[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in main)
gaps = []
for x, y in zip(main[:-1], main[1:]):
    g = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
    if g > 0: gaps.append((g, x["Kernel_Name"].split("(")[0][-40:], y["Kernel_Name"].split("(")[0][-40:]))
print("main queue: span %.2f ms, busy %.2f ms, idle %.2f ms over %d gaps (mean %.1f us)" % ((t1 - t0) / 1e6, busy / 1e6, sum(g for g, _, _ in gaps) / 1e6, len(gaps), sum(g for g, _, _ in gaps) / max(1, len(gaps)) / 1e3))
for g in sorted(gaps, reverse=True)[:12]:
    print("   gap %.1f us between %s -> %s" % (g[0] / 1e3, g[1], g[2]))
# the seam between two replays: what runs (on every queue) from the last backward kernel to the first kernel of the next graph
seam0 = max(i for i, r in enumerate(step) if "k_subm_f32_wgrad" in r["Kernel_Name"] or "k_subm_wgrad" in r["Kernel_Name"])
base = int(step[seam0]["End_Timestamp"])
print("seam (us after the end of the last backward kernel; queue; duration; kernel):")
for r in step[max(0, seam0 - 3):] + rows[s1:s1 + 6]:
    print("   %9.1f  q%s  %7.1f  %s" % ((int(r["Start_Timestamp"]) - base) / 1e3, r[qkey], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][-70:]))
acc = collections.defaultdict(lambda: [0, 0])
for r in main:
    k = r["Kernel_Name"].split("(")[0][-60:]
    acc[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); acc[k][1] += 1
print("top kernels of the step (main queue):")
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
    print("   %8.1f us %4d  %s" % (t / 1e3, c, k))
small = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in main if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 20000]
print("launches shorter than 20 us: %d, %.2f ms in total" % (len(small), sum(small) / 1e6))
sm = collections.defaultdict(lambda: [0, 0])
for r in main:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d < 20000:
        k = r["Kernel_Name"].split("(")[0][-56:]
        sm[k][0] += d; sm[k][1] += 1
for k, (tt, c) in sorted(sm.items(), key=lambda kv: -kv[1][0])[:32]:
    print("   small: %7.1f us %4d  %s" % (tt / 1e3, c, k))
oth = [r for r in step if r[qkey] != mainq]
if oth:
    offs = sorted((int(r["Start_Timestamp"]) - t0) / 1e6 for r in oth)
    print("other-queue kernels start at (ms after the step's first kernel): min %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f  (step span %.2f)" % (
        offs[0], offs[len(offs) // 10], offs[len(offs) // 2], offs[len(offs) * 9 // 10], offs[-1], (t1 - t0) / 1e6))
    hist = collections.Counter(int(o // 2.0) * 2 for o in offs)
    print("   per 2-ms bin:", " ".join("%d:%d" % (k, hist[k]) for k in sorted(hist)))
oa = collections.defaultdict(lambda: [0, 0])
for r in oth:
    k = r["Kernel_Name"].split("(")[0][-56:]
    oa[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); oa[k][1] += 1
for k, (tt, c) in sorted(oa.items(), key=lambda kv: -kv[1][0])[:16]:
    print("   side: %7.1f us %4d  %s" % (tt / 1e3, c, k))
print("other queues: %d kernels, %.2f ms" % (len(oth), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in oth) / 1e6))
PY
rm -rf $O/tr
