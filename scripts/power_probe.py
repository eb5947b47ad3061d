#!/usr/bin/env python
"""Run one roofline probe back to back for a few seconds while a thread polls rocm-smi: socket power and shader clock under that
kernel alone.   python scripts/power_probe.py attn_fwd|attn_bwd|conv_fwd|conv_wgrad|unpool_fwd|gather_hbm [seconds]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import roofline_probes as rp

name = sys.argv[1] if len(sys.argv) > 1 else "attn_fwd"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
p = rp.build(which={name})[0]
ms = rp.time_probe(p, iters=20, warmup=5)
samples, stop = [], threading.Event()


def poll():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = re.search(r"Power \(W\): ([0-9.]+)", out); sc = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            samples.append((time.time(), float(pw.group(1)) if pw else None, int(sc.group(1)) if sc else None))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), None, None))
        time.sleep(0.5)


th = threading.Thread(target=poll, daemon=True); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50):
        p["run"]()
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop.set(); th.join(timeout=5)
sus = e0.elapsed_time(e1) / n
mid = [s for s in samples if s[1] is not None and t0 + 1.5 < s[0] < t0 + secs - 0.5]
pw = sum(s[1] for s in mid) / max(1, len(mid)); sc = sum(s[2] for s in mid if s[2]) / max(1, len([s for s in mid if s[2]]))
unit = "TFLOP/s" if p["bound"] == "mfma" else "GB/s"
ach = (p["flops"] / (sus * 1e-3) / 1e12) if p["bound"] == "mfma" else (p["bytes"] / (sus * 1e-3) / 1e9)
print("%-11s burst %.3f ms | sustained %.3f ms over %d launches = %.0f %s | socket power %.0f W, sclk %.0f MHz (%d samples)" % (
    name, ms, sus, n, ach, unit, pw, sc, len(mid)))
