#!/bin/bash
# GPU-box helper: which tensors do the many small bf16 copy kernels of the bench step move?  (grid sizes + neighbours in the trace)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r3c
rm -rf $O; mkdir -p $O
SS_BENCH_CALIBRATE=0 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python bench.py --steps 4 --warmup 2 --no-pmc --no-secondary --no-cpu-baseline > $O/run.log 2>&1
python - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/r3c"
trace = max(glob.glob(O + "/tr/*/*kernel_trace.csv"), key=os.path.getsize)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][-70:] for r in rows]
sizes = collections.Counter()
ctx = collections.Counter()
for i, r in enumerate(rows):
    if "bfloat16_copy_kernel" in r["Kernel_Name"]:
        sizes[(r["Grid_Size_X"], r["Stream_Id"] if "Stream_Id" in r else r.get("Queue_Id", "?"))] += 1
        prev = next((names[j] for j in range(i - 1, max(0, i - 6), -1) if "bfloat16_copy" not in names[j]), "?")
        nxt = next((names[j] for j in range(i + 1, min(len(rows), i + 6)) if "bfloat16_copy" not in names[j]), "?")
        ctx[(prev[-50:], nxt[-50:])] += 1
print("total kernels", len(rows))
print("bf16 copy grid sizes (threads, stream/queue): count")
for k, v in sizes.most_common(25):
    print("  ", k, v)
print("contexts (previous other kernel -> next other kernel): count")
for k, v in ctx.most_common(25):
    print("  ", k, v)
PY
rm -rf $O/tr
