#!/usr/bin/env python
"""Per-shape timing of the CPE convolutions of the LangPretrainer PTv3 at room-102400 (fwd / dgrad share a kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from scenesplat_amd import native as nv
from scenesplat_amd.plan import build_plan
from scenesplat_amd.synthetic import room_chunk
from bench_kernels import ev

data = room_chunk(256, 0, lang_dim=0)
plan = build_plan(data["grid_coord"].cuda(), data["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), (2, 2, 2, 2))
g = torch.Generator(device="cuda").manual_seed(0)
tot = 0.0
for li, C, calls in [(0, 32, 2), (0, 768, 2), (1, 64, 2), (1, 512, 2), (2, 128, 2), (2, 256, 2), (3, 256, 8), (4, 512, 2)]:
    lv = plan.levels[li]; n = lv.n
    nbr = lv.neighbors(3); perm = lv.conv_rowperm(); blocks = lv.conv_blocks(3)
    x = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(C, 27, C, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    go = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
    tf = ev(lambda: nv.subm_conv_fwd(x, w, None, nbr, perm), 10, 3)
    tw = ev(lambda: nv.subm_conv_wgrad(x, go, nbr, perm, blocks), 10, 3)
    step = calls * (2 * tf + tw)
    tot += step
    print(f"L{li} n={n} C={C}: fwd {tf*1e3:.0f} us  wgrad {tw*1e3:.0f} us  x{calls} blocks -> {step:.2f} ms/step", flush=True)
print(f"total conv {tot:.2f} ms/step", flush=True)
