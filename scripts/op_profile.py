#!/usr/bin/env python
"""torch.profiler over eager training steps of the bench model: aten ops by device time and input shape -- finds what launches
the many small elementwise / copy kernels between the HIP ops.   python scripts/op_profile.py [filter substring]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

RUNTIME.update(bench_runtime())
model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).cuda().train()
data = {k: v.cuda() for k, v in room_chunk(256, 0, lang_dim=0).items()}
cot = torch.randn(len(data["feat"]), 768, device="cuda").to(torch.bfloat16)


def step():
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"]))
    torch.autograd.backward(out.feat, grad_tensors=cot.to(out.feat.dtype))


for _ in range(3):
    step()
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dev = getattr(e, "self_device_time_total", None)
    if dev is None:
        dev = getattr(e, "self_cuda_time_total", 0)
    if dev <= 0 or flt not in e.key:
        continue
    rows.append((dev / STEPS, e.count / STEPS, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print("%9s %7s  %-38s %s" % ("us/step", "calls", "op", "input shapes"))
for r in rows[:70]:
    print("%9.1f %7.1f  %-38s %s" % r)
print("total self device time of listed ops: %.2f ms/step" % (sum(r[0] for r in rows) / 1e3))
