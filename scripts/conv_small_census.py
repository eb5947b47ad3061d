#!/usr/bin/env python
"""fwd+bwd time of the CPE conv autograd op at the small levels (im2col + library GEMM vs implicit-GEMM kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from scenesplat_amd import functional as SF
from scenesplat_amd.plan import build_plan
from scenesplat_amd.synthetic import room_chunk
from bench_kernels import ev

data = room_chunk(256, 0, lang_dim=0)
plan = build_plan(data["grid_coord"].cuda(), data["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), (2, 2, 2, 2))
g = torch.Generator(device="cuda").manual_seed(0)
for li, C in [(1, 64), (2, 128), (2, 256), (3, 256), (4, 512)]:
    lv = plan.levels[li]; n = lv.n
    nbr = lv.neighbors(3); perm = lv.conv_rowperm()
    x = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(C, 3, 3, 3, C, device="cuda", generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(C, device="cuda", generator=g).requires_grad_(True)
    go = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
    res = {}
    for mode, cap in (("implicit", 0), ("im2col", 1 << 30)):
        SF.CONV_IM2COL_MAX_SITES = cap
        def f():
            y = SF.subm_conv3d(x, w, b, nbr, False, torch.bfloat16, perm, lambda: lv.conv_blocks(3))
            y.backward(go)
        res[mode] = ev(f, 10, 3)
    print(f"L{li} n={n} C={C}: implicit {res['implicit']*1e3:.0f} us  im2col {res['im2col']*1e3:.0f} us (fwd+bwd incl. host)", flush=True)
