#!/usr/bin/env python
"""Dispatch thresholds of the conv kernels: 128x128 register-staged kernel vs 256x256 LDS-DMA pipeline on the shapes of
the uniform stress fixture and the room (fwd; wgrad old vs pipe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from scenesplat_amd import native as nv
from scenesplat_amd.plan import build_plan
from scenesplat_amd.synthetic import room_chunk, uniform_chunk
from bench_kernels import ev

g = torch.Generator(device="cuda").manual_seed(0)
for name, data in (("uniform", uniform_chunk()), ("room", room_chunk(256, 0, lang_dim=0))):
    plan = build_plan(data["grid_coord"].cuda(), data["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), (2, 2, 2))
    for li, C in [(0, 32), (0, 64), (1, 64), (1, 128), (2, 128), (2, 256), (3, 256)]:
        lv = plan.levels[li]; n = lv.n
        nbr = lv.neighbors(3); perm = lv.conv_rowperm(); blocks = lv.conv_blocks(3)
        x = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(C, 27, C, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        go = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
        pairs = (nbr >= 0).sum().item() / n
        os.environ.pop("SS_CONV_PIPE", None)
        t_dispatch = ev(lambda: nv.subm_conv_fwd(x, w, None, nbr, perm), 10, 3)
        t_pipe = ev(lambda: nv.subm_conv_fwd_pipe(x, w, None, nbr, perm), 10, 3) if nv.lib().ss_gemm8_ok(n, C, C, 27) else float("nan")
        tw_dispatch = ev(lambda: nv.subm_conv_wgrad(x, go, nbr, perm, blocks), 10, 3)
        tw_pipe = ev(lambda: nv.subm_conv_wgrad_pipe(x, go, nbr, perm, blocks), 10, 3)
        print(f"{name} L{li} n={n} C={C} pairs/site={pairs:.2f}: fwd dispatch {t_dispatch*1e3:.0f} us pipe {t_pipe*1e3:.0f} us | "
              f"wgrad dispatch {tw_dispatch*1e3:.0f} us pipe {tw_pipe*1e3:.0f} us", flush=True)
