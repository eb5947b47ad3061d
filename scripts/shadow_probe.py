#!/usr/bin/env python
"""Which launches re-cast the bf16 parameter shadows?  Counts the group-cast / multi-tensor-copy calls of one forward after an
"optimizer step" and lists the parameters that miss the one-launch path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd import native as nv, functional as SF
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

RUNTIME.update(bench_runtime())
model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).cuda().train()
data = {k: v.cuda() for k, v in room_chunk(64, 0, lang_dim=0).items()}
calls = {"group": 0, "foreach": 0, "foreach_tensors": 0}
g0, f0 = nv.cast_bf16_group, torch._foreach_copy_


def g1(a, b):
    calls["group"] += 1
    return g0(a, b)


def f1(d, s, *a, **k):
    calls["foreach"] += 1; calls["foreach_tensors"] += len(d)
    bad = [(tuple(x.shape), x.dtype, x.is_contiguous(), tuple(y.shape), y.dtype, y.is_contiguous()) for x, y in zip(s, d)][:5]
    print("foreach_copy of", len(d), "tensors, e.g.", bad)
    return f0(d, s, *a, **k)


nv.cast_bf16_group, torch._foreach_copy_ = g1, f1
for it in range(2):
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"]))
    out.feat.float().sum().backward()
    print("step", it, calls)
