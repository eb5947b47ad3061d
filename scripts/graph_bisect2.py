#!/usr/bin/env python
"""Diagnostic: replicate the steady-state test flow with switches.  usage: graph_bisect2.py [noref] [nomul] [nocheck] [sameperm]"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = set(sys.argv[1:])
import torch
from scenesplat_amd import native as nv
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.steady_state import SteadyStateStep
from scenesplat_amd.synthetic import room_chunk
RUNTIME.update(bench_runtime())
TINY = dict(in_channels=11, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2),
            enc_depths=(1, 1, 1), enc_channels=(16, 32, 48), enc_num_head=(1, 2, 3), enc_patch_size=(64, 64, 16),
            dec_depths=(1, 1), dec_channels=(48, 32), dec_num_head=(1, 2), dec_patch_size=(64, 64))
torch.manual_seed(11)
model = MODELS.build(dict(type="PT-v3m1", **TINY, drop_path=0.0, shuffle_orders=True)).cuda().train()
d = {k: v.cuda() for k, v in room_chunk(n_side=40, seed=3, lang_dim=0).items()}
n = d["feat"].shape[0]

def fn(plan, t):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=t["feat"], grid_coord=d["grid_coord"], offset=d["offset"], plan=plan))
    torch.autograd.backward(out.feat, grad_tensors=t["cot"])
    return {"feat": out.feat}

params = list(model.parameters())
steady = SteadyStateStep(fn, params, warmup=1)
if "nocheck" in flags:
    steady._checked = {None}   # (obsolete switch of the first experiments)
g = torch.Generator(device="cuda").manual_seed(1)
perms0 = model.draw_perms()
for it in range(5):
    feat = torch.randn(n, 11, device="cuda", generator=g)
    cot = torch.randn(n, 48, device="cuda", generator=g).to(torch.bfloat16)
    perms = perms0 if "sameperm" in flags else model.draw_perms()
    if "noref" not in flags:
        model.zero_grad(set_to_none=True)
        ref = fn(model.prepare_plan(d, perms=perms), dict(feat=feat, cot=cot))["feat"].float().clone()
        rg = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    print("it", it, "calling steady", flush=True)
    out = steady(model.prepare_plan(d, perms=perms), dict(feat=feat, cot=cot))["feat"].float()
    torch.cuda.synchronize()
    print("it", it, "replays", steady.replays, "eager", steady.eager_steps, "refused", steady.refused, flush=True)
    if "nomul" not in flags:
        with torch.no_grad():
            for p in params:
                p.mul_(1.0 + 0.1 * (it % 2 * 2 - 1))
print("flow done", sorted(flags), flush=True)
