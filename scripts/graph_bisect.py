#!/usr/bin/env python
"""Diagnostic: which part of the tiny PT-v3m1 step breaks hipGraph capture?  usage: graph_bisect.py MODE"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode == "nogroup":
    os.environ["SS_WGRAD_GROUP_MAX"] = "0"
import torch
from scenesplat_amd import native as nv
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.synthetic import room_chunk
RUNTIME.update(bench_runtime())
if mode == "simt":
    RUNTIME["attn_impl"] = nv.ATTN_SIMT
if mode == "fp32conv":
    RUNTIME["conv_dtype"] = torch.float32
TINY = dict(in_channels=11, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2),
            enc_depths=(1, 1, 1), enc_channels=(16, 32, 48), enc_num_head=(1, 2, 3), enc_patch_size=(64, 64, 16),
            dec_depths=(1, 1), dec_channels=(48, 32), dec_num_head=(1, 2), dec_patch_size=(64, 64))
torch.manual_seed(11)
model = MODELS.build(dict(type="PT-v3m1", **TINY, drop_path=0.0, shuffle_orders=True)).cuda().train()
n_side = 80 if mode == "bigger" else 40
d = {k: v.cuda() for k, v in room_chunk(n_side=n_side, seed=3, lang_dim=0).items()}
n = d["feat"].shape[0]
cot = torch.randn(n, 48, device="cuda").to(torch.bfloat16)
plan = model.prepare_plan(d)
feat_in = d["feat"]
print(mode, "n =", n, [lv.n for lv in plan.levels], flush=True)

def fn():
    if mode == "fwd_nograd":
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return model(dict(feat=d["feat"], grid_coord=d["grid_coord"], offset=d["offset"], plan=plan)).feat
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=feat_in, grid_coord=d["grid_coord"], offset=d["offset"], plan=plan))
    if mode != "fwd_grad":
        torch.autograd.backward(out.feat, grad_tensors=cot)
    return out.feat

for _ in range(2):
    model.zero_grad(set_to_none=True); fn()
torch.cuda.synchronize()
if "fresh" in mode:
    plan = model.prepare_plan(d)
if "own" in mode:
    plan = plan.own_storage()
if "clone" in mode:
    feat_in = d["feat"].clone(); cot = cot.clone()
if "pgrad" in mode:
    for p_ in model.parameters():
        p_.grad = None
torch.cuda.synchronize()
model.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
pool = nv.DescriptorPool(); nv.CAPTURE_POOL = pool
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g.capture_begin()
    out = fn()
    print(mode, "recorded", flush=True)
    g.capture_end()
print(mode, "capture ended", flush=True)
g.replay(); torch.cuda.synchronize()
print(mode, "replayed OK", bool(torch.isfinite(out.float()).all()), flush=True)
