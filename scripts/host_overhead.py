#!/usr/bin/env python
"""Host-side cost of one step: CPU time to ENQUEUE forward / backward vs GPU time (GPU box helper)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd.pointcept_api import MODELS, RUNTIME
from scenesplat_amd import native as nv
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

RUNTIME["attn_impl"] = nv.ATTN_MFMA; RUNTIME["conv_dtype"] = torch.bfloat16
model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).cuda().train()
data = {k: v.cuda() for k, v in room_chunk(256, 0, lang_dim=0).items()}
cot = torch.randn(len(data["feat"]), 768, device="cuda")

def step(timing=None):
    model.zero_grad(set_to_none=True)
    t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"]))
    t1 = time.perf_counter()
    torch.autograd.backward(out.feat, grad_tensors=cot)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if timing is not None:
        timing.append((t1 - t0, t2 - t1, t3 - t2))

for _ in range(3): step()
tm = []
for _ in range(5): step(tm)
f = sum(t[0] for t in tm) / 5 * 1e3; b = sum(t[1] for t in tm) / 5 * 1e3; s = sum(t[2] for t in tm) / 5 * 1e3
print(f"enqueue forward {f:.1f} ms, enqueue backward {b:.1f} ms, final sync wait {s:.1f} ms, total {f+b+s:.1f} ms")
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
