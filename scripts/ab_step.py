#!/usr/bin/env python
"""A/B of a runtime switch inside ONE process (interleaved blocks; run-to-run noise of separate processes is +-2 ms):
   python scripts/ab_step.py attn_hm|group_cast|shadows|im2col|... [blocks] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd import native as nv, functional as SF
from scenesplat_amd.pointcept_api import MODELS, RUNTIME
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

which = sys.argv[1] if len(sys.argv) > 1 else "attn_hm"
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
from scenesplat_amd.pointcept_api import bench_runtime
RUNTIME.update(bench_runtime())
model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).cuda().train()
data = {k: v.cuda() for k, v in room_chunk(256, 0, lang_dim=0).items()}
cot = torch.randn(len(data["feat"]), 768, device="cuda").to(torch.bfloat16)
side = torch.cuda.Stream()
state = {"plan": model.prepare_plan(data, stream=side)}


def setter(on):
    if which == "shadows":
        RUNTIME["param_shadows"] = on
    elif which == "im2col":
        SF.CONV_IM2COL_MAX_SITES = 8192 if on else 0
    elif which == "wgrad_rows":
        SF.LINEAR_WGRAD_MIN_ROWS = 2048 if on else 1024
    elif which == "im2col_big":
        SF.CONV_IM2COL_MAX_SITES = 2048 if on else 8192
    elif which == "coarse15":
        import scenesplat_amd.plan as P
        P.CONV_COARSE_BITS = 17 if on else 15
    elif which == "coarse11":
        import scenesplat_amd.plan as P
        P.CONV_COARSE_BITS = 11 if on else 13
    elif which == "blaslt":
        torch.backends.cuda.preferred_blas_library("cublaslt" if on else "cublas")
    elif which == "rowscale":
        if on:
            model.__dict__.pop("_draw_row_scales", None)
        else:
            model.__dict__["_draw_row_scales"] = lambda levels, device: None
    elif which == "ln_seam":
        RUNTIME["fuse_ln_seam"] = on
    elif which == "rb_hash":
        nv.RULEBOOK_HASHED = on
    elif which == "attn_hm":
        RUNTIME["attn_headmajor"] = on
    elif which == "group_cast":
        SF.SHADOW_GROUP_CAST = on
    elif which == "conv_f32":
        SF.CONV_F32_MFMA = on
    elif which == "conv_walk":
        SF.CONV_WALK_RULEBOOK = on
    elif which == "gelu":
        SF.GELU_HIP = on
    elif which == "wgrad_xcd":        # XCD-aware tile order of the weight-gradient kernels (wgrad8.hip: w8_xcd_order)
        nv.lib().ss_wgrad_set_xcd_order(1 if on else 0)
    elif which == "hm_rows":          # the qkv projection's head-major epilogue also on levels of 1,024 .. 4,095 rows (enc3: 1,600)
        SF.HM_FUSED_MIN_ROWS = 1024 if on else 4096
    elif which == "hm_ch":            # ... and from 64 channels on (enc1: 25,600 x 64)
        SF.HM_FUSED_MIN_CHANNELS = 64 if on else 128
    elif which == "mask_small":
        import scenesplat_amd.plan as P
        P.CONV_MASK_MIN_SITES = 4096 if on else 16384
    else:
        raise SystemExit("unknown switch")


_params = [p for p in model.parameters()]


def step():
    if which == "group_cast":
        with torch.no_grad():
            torch._foreach_add_(_params, 0.0)          # an "optimizer step": every shadow is re-cast by the next forward
    model.zero_grad(set_to_none=True)
    plan, state["plan"] = state["plan"], None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"], plan=plan))
    state["prev"] = plan
    torch.autograd.backward(out.feat, grad_tensors=cot.to(out.feat.dtype))
    state["plan"] = model.prepare_plan(data, stream=side)


for on in (True, False):
    setter(on)
    for _ in range(3):
        step()
torch.cuda.synchronize()
res = {True: [], False: []}
for b in range(blocks):
    on = (b % 2 == 0)
    setter(on)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    res[on].append((time.perf_counter() - t0) / steps * 1e3)
print(which, "ON ", " ".join(f"{v:.1f}" for v in res[True]), " mean %.2f" % (sum(res[True]) / len(res[True])))
print(which, "OFF", " ".join(f"{v:.1f}" for v in res[False]), " mean %.2f" % (sum(res[False]) / len(res[False])))
