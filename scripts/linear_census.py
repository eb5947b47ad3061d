#!/usr/bin/env python
"""Per-shape timing of the nn.Linear weight gradients of the LangPretrainer PTv3 at room-102400:
pipeline kernel (ss_linear_wgrad, with the bias sums) vs hipBLASLt (dy^T x + column sum)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from scenesplat_amd import native as nv
from bench_kernels import ev

g = torch.Generator(device="cuda").manual_seed(0)
tot_p = tot_h = 0.0
for (m, C, blocks) in [(102400, 32, 2), (102400, 768, 2), (25600, 64, 2), (25600, 512, 2), (6400, 128, 2), (6400, 256, 2)]:
    for (k, n, name) in [(C, C, "cpe/proj x2"), (C, 3 * C, "qkv"), (C, 4 * C, "fc1"), (4 * C, C, "fc2")]:
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        dy = torch.randn(m, n, device="cuda", generator=g).to(torch.bfloat16)
        tp = ev(lambda: nv.linear_wgrad(x, dy, True), 10, 3)
        th = ev(lambda: ((dy.t() @ x).float(), dy.sum(0, dtype=torch.float32)), 10, 3)
        mult = blocks * (2 if k == n else 1)
        tot_p += tp * mult; tot_h += th * mult
        print(f"m={m} {k}->{n} ({name}): pipe {tp*1e3:.0f} us  hipBLASLt+sum {th*1e3:.0f} us   x{mult}", flush=True)
print(f"total per step: pipe {tot_p:.2f} ms  hipBLASLt {tot_h:.2f} ms", flush=True)
