#!/usr/bin/env python
"""Diagnostic: time the open-vocabulary scan probe of bench.py (1 M x 768 x 160 classes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.roofline_probes import build, time_probe
p = build(which=("scan",))[0]
ms = time_probe(p, iters=20, warmup=3)
print("scan v1=%s flags=%s: %.3f ms  %.0f GB/s (%.1f %% of 8 TB/s)" % (os.environ.get("SS_SCAN_V1", "0"), os.environ.get("SS_EXTRA_HIPCC_FLAGS", ""), ms, p["bytes"] / ms / 1e6, p["bytes"] / ms / 1e6 / 80), flush=True)
