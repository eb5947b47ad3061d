#!/usr/bin/env python
"""Diagnostic: dec0 conv weight-gradient time under SS_WGRAD_MINPER (K-tiles per share; read once per process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.roofline_probes import build, time_probe
p = build(which=("conv_wgrad",))[0]
ms = time_probe(p, iters=20, warmup=3)
print("minper=%s conv_wgrad dec0: %.3f ms  %.0f TFLOP/s" % (os.environ.get("SS_WGRAD_MINPER", "auto(160)"), ms, p["flops"] / ms / 1e9), flush=True)
