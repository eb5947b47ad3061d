#!/usr/bin/env python
"""BASELINE config 5: 1M-Gaussian open-vocab scan (feat x 160 text embeddings -> sigmoid -> max/argmax), 1 GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd import native as nv
n, D, C = 1_000_000, 768, 160
g = torch.Generator(device="cuda").manual_seed(0)
feat = torch.nn.functional.normalize(torch.randn(n, D, device="cuda", generator=g), dim=1).to(torch.bfloat16)
text = torch.nn.functional.normalize(torch.randn(C, D, device="cuda", generator=g), dim=1).to(torch.bfloat16)
for _ in range(3): nv.feat_text_scan(feat, text)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): nv.feat_text_scan(feat, text)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
by = n * D * 2 + n * 8
print(f"scan 1M x 768 x 160: {ms:.3f} ms  {by/ms/1e6:.0f} GB/s algorithmic ({by/ms/1e6/8000*100:.0f}% of 8 TB/s), {2*n*D*C/ms/1e9:.0f} TFLOP/s")
s.record()
for _ in range(5):
    lg = torch.sigmoid(feat @ text.t()); mp, am = lg.max(1)
e.record(); torch.cuda.synchronize()
print(f"  torch mm+sigmoid+max: {s.elapsed_time(e)/5:.3f} ms")
