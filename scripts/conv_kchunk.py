#!/usr/bin/env python
"""Diagnostic: dec0 / dec1 conv forward time under SS_CONV_KCHUNK (read once per process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scripts.roofline_probes import build, time_probe
p = build(which=("conv_fwd",))[0]
ms = time_probe(p, iters=20, warmup=3)
print("kchunk=%s conv_fwd dec0: %.3f ms  %.0f TFLOP/s" % (os.environ.get("SS_CONV_KCHUNK", "default"), ms, p["flops"] / ms / 1e9), flush=True)
