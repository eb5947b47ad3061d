#!/usr/bin/env python
"""Diagnostic: host (enqueue) time of one step split into plan build / forward / backward, and the launch count of each
phase's Python side (number of native C-ABI calls + torch ops is not visible here; use rocprofv3 --kernel-trace for that)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

RUNTIME.update(bench_runtime())
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).to(dev).train()
data = {k: v.to(dev) for k, v in room_chunk(n_side=256, seed=0, lang_dim=0).items()}
n = data["feat"].shape[0]
cot = torch.randn(n, LANG_PTV3["dec_channels"][0], device=dev)
side = torch.cuda.Stream()
acc = dict(plan=0.0, fwd=0.0, bwd=0.0, total=0.0)
import gc
for it in range(13):
    if it == 3:
        gc.collect(); gc.freeze()
        for k in acc: acc[k] = 0.0
        torch.cuda.synchronize(); t_all = time.perf_counter()
    t0 = time.perf_counter()
    plan = model.prepare_plan(data, stream=side)
    t1 = time.perf_counter()
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"], plan=plan))
    t2 = time.perf_counter()
    torch.autograd.backward(out.feat, grad_tensors=cot.to(out.feat.dtype))
    t3 = time.perf_counter()
    acc["plan"] += t1 - t0; acc["fwd"] += t2 - t1; acc["bwd"] += t3 - t2
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print("per step over 10 steps: plan %.1f ms, forward %.1f ms, backward %.1f ms host; wall %.1f ms" % (
    acc["plan"] * 100, acc["fwd"] * 100, acc["bwd"] * 100, tot * 100))
if "--profile" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(3):
            plan = model.prepare_plan(data, stream=side)
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"], plan=plan))
            torch.autograd.backward(out.feat, grad_tensors=cot.to(out.feat.dtype))
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
