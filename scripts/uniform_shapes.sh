#!/bin/bash
# GPU-box helper: per-launch-shape durations of the conv / attention / GEMM kernels on the uniform-102400 stress fixture
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r4u
rm -rf $O; mkdir -p $O
SS_BENCH_CALIBRATE=0 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python bench.py --fixture uniform --steps 6 --warmup 3 --no-pmc --no-secondary --no-cpu-baseline > $O/run.log 2>&1
python - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/r4u"
trace = max(glob.glob(O + "/tr/*/*kernel_trace.csv"), key=os.path.getsize)
acc = collections.defaultdict(list)
tot = collections.defaultdict(float)
for r in csv.DictReader(open(trace)):
    k = r["Kernel_Name"].split("(")[0][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc[(k, r["Grid_Size_X"])].append(d); tot[k] += d
steps = 9.0 + 3      # 3 warm-up (+ up to 3 extra for the capture) + 6 timed
out = ["uniform-102400: kernels by total time, per launch shape (us; ~%d steps in the trace)" % steps, "", "| kernel | grid (threads) | launches | mean us | total ms |", "|---|---|---|---|---|"]
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:60]:
    out.append("| `%s` | %s | %d | %.1f | %.2f |" % (k, g, len(v), sum(v) / len(v), sum(v) / 1e3))
open(O + "/r04_uniform_shapes.md", "w").write("\n".join(out) + "\n")
print("\n".join(out[:50]))
PY
rm -rf $O/tr
