#!/usr/bin/env python
"""Diagnostic: which Python lines issue the small torch kernels (casts, fills, adds) of one eager step?"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk
RUNTIME.update(bench_runtime())
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).to(dev).train()
data = {k: v.to(dev) for k, v in room_chunk(n_side=256, seed=0, lang_dim=0).items()}
n = data["feat"].shape[0]
cot = torch.randn(n, LANG_PTV3["dec_channels"][0], device=dev).to(torch.bfloat16)
plan = model.prepare_plan(data)
def fb():
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"], plan=plan))
    torch.autograd.backward(out.feat, grad_tensors=cot)
for _ in range(2):
    fb()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter()
want = ("copy_", "_to_copy", "fill_", "zero_", "add", "add_", "mul", "mul_", "sum", "cat", "zeros", "index_add_", "clone", "div", "sub", "empty_like", "zeros_like", "index", "index_select")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in want:
            fr = [f for f in traceback.extract_stack() if "scenesplat_amd" in f.filename and "small_ops" not in f.filename]
            where = "%s:%d" % (fr[-1].filename.split("scenesplat_amd/")[-1], fr[-1].lineno) if fr else "?"
            numel = max([a.numel() for a in args if isinstance(a, torch.Tensor)] or [0])
            cnt[(name, where, "big" if numel > 1_000_000 else "small")] += 1
        return func(*args, **(kwargs or {}))
torch.autograd.set_multithreading_enabled(False)
with Log():
    fb()
torch.cuda.synchronize()
for (name, where, sz), c in cnt.most_common(70):
    print("%4d  %-12s %-6s %s" % (c, name, sz, where))
