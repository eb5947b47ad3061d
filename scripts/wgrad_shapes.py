#!/usr/bin/env python
"""Diagnostic: time nn.Linear weight gradients of the dec0 shapes (m = 102,400).  SS_WGRAD_MINPER overrides the share length."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scenesplat_amd import native as nv
m = 102400
shapes = [(768, 2304), (768, 3072), (3072, 768), (768, 768)]
if len(sys.argv) > 1:
    shapes = [shapes[int(sys.argv[1])]]
for k, n in shapes:
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    dy = torch.randn(m, n, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        nv.linear_wgrad(x, dy, True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        nv.linear_wgrad(x, dy, True)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print("minper=%s  %4d -> %4d: %.1f us  %.0f TFLOP/s" % (os.environ.get("SS_WGRAD_MINPER", "auto"), k, n, ms * 1e3, 2.0 * m * k * n / ms / 1e9))
