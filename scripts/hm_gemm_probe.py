#!/usr/bin/env python
"""Timing of the qkv projection forms at the dec0 shape: hipBLASLt (F.linear), the pipeline GEMM (row-major), the pipeline GEMM with
the head-major epilogue (identity and curve-order row index)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd import native as nv
from scenesplat_amd.plan import build_plan
from scenesplat_amd.synthetic import room_chunk


def t(fn, it=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


d = room_chunk(256, 0, lang_dim=0)
plan = build_plan(d["grid_coord"].cuda(), d["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), (2, 2, 2))
lv = plan.levels[0]
win = lv.window(0, 1024)
n, C, H = lv.n, 768, 16
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(3 * C, C, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
b = torch.randn(3 * C, device="cuda", generator=g)
fl = 2.0 * n * C * 3 * C
for name, fn in (("hipBLASLt F.linear", lambda: torch.nn.functional.linear(x, w, b.to(torch.bfloat16))),
                 ("gemm8 row-major", lambda: nv.linear_fwd(x, w, b)),
                 ("gemm8 head-major (curve order)", lambda: nv.linear_fwd_headmajor(x, win, w, b, H, 0.2))):
    ms = t(fn)
    print("%-34s %.3f ms  %.0f TFLOP/s" % (name, ms, fl / ms / 1e9))
import types
ident = types.SimpleNamespace(gidx=torch.arange(n, device="cuda", dtype=torch.int32), n=n, n_pad=n)
ms = t(lambda: nv.linear_fwd_headmajor(x, ident, w, b, H, 0.2))
print("%-34s %.3f ms  %.0f TFLOP/s" % ("gemm8 head-major (identity rows)", ms, fl / ms / 1e9))
