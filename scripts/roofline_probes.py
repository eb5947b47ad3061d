#!/usr/bin/env python
"""Isolated launches of the kernels bench.py prices against a roofline, at the shapes of the metric's workload.

Two uses:
  * imported by bench.py: `build(...)` returns the probes, `time_probe` times one with HIP events on the launch stream;
  * run as a program under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (bench.py starts those passes as CHILD
    processes after its timed region): launches every probe a few times so the counter CSV holds per-launch bytes of
    exactly the kernels, shapes and build that were just timed.

Algorithmic work per launch (DESIGN.md section 4 / SURVEY 8d):
  conv_fwd / conv_wgrad  dec0 submanifold conv, n = 102,400 sites, C = 768: 2 * pairs * C^2 FLOP; bytes 2 n C 2 + 27 C^2 2
  attn_fwd / attn_bwd    dec0 window attention (head-major kernels, csrc/attention_hm.hip): 100 windows x 16 heads, K = 1024,
                         d = 48: 4 K^2 d per (window, head) FLOP forward, 2.5 x backward (5 products, recompute not counted);
                         bytes: q, k, v read + out written (fwd); + dO, O read, dO copy, dq, dk, dv written (bwd)
  unpool_fwd / unpool_bwd  the gather / scatter launches of the step at the dec0 unpooling seam (grid-pool scatter of the north
                         star): n (3 * 1536 + 4) bytes forward, n 1540 + n/4 1540 bytes backward
  gather_hbm             row gather of 819,200 x 768 bf16 rows (the config-3 batch: 2.5 GB working set, far beyond the
                         256 MiB Infinity Cache): n (2 * 1536 + 4) bytes
  scan                   config 5: 1,000,000 x 768 bf16 unit rows x 160 text rows -> sigmoid -> max/argmax: n (1536 + 8) bytes
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak


def build(n_side=256, which=None):
    """-> list of dict(name, kernel (substring of the HIP kernel name), run, bound, flops | bytes, note)."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.plan import build_plan
    from scenesplat_amd.synthetic import room_chunk
    data = room_chunk(n_side, 0, lang_dim=0)
    plan = build_plan(data["grid_coord"].cuda(), data["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), (2, 2, 2))
    lv = plan.levels[0]
    C, H, K = 768, 16, 1024
    d = C // H
    g = torch.Generator(device="cuda").manual_seed(0)
    want = lambda k: which is None or k in which
    probes = []
    if want("conv_fwd") or want("conv_wgrad"):
        nbr, perm, blocks, walk = lv.neighbors(3), lv.conv_rowperm(), lv.conv_blocks(3), lv.neighbors_walk(3)   # as the step's launches
        x = torch.randn(lv.n, C, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(C, 27, C, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        go = torch.randn(lv.n, C, device="cuda", generator=g).to(torch.bfloat16)
        pairs = int((nbr >= 0).sum().item())
        fl = 2.0 * pairs * C * C
        by = lv.n * C * 2 * 2 + 27 * C * C * 2
        note = "n=%d, C=%d, 27 taps, %.2f pairs/site" % (lv.n, C, pairs / lv.n)
        if want("conv_fwd"):
            probes.append(dict(name="conv_fwd", kernel="k_gemm8", run=lambda: nv.subm_conv_fwd(x, w, None, nbr, perm, nbr_walk=walk), bound="mfma",
                               flops=fl, bytes=by, note="k_gemm8<true> subm conv fwd/dgrad (dec0: %s)" % note))
        if want("conv_wgrad"):
            probes.append(dict(name="conv_wgrad", kernel="k_wgrad8", run=lambda: nv.subm_conv_wgrad(x, go, nbr, perm, blocks, nbr_walk=walk), bound="mfma",
                               flops=fl, bytes=2 * lv.n * C * 2 + 27 * C * C * 4, note="k_wgrad8<true> subm conv wgrad (dec0: %s)" % note))
    if want("attn_fwd") or want("attn_bwd"):
        # the kernels the step runs since round 3: head-major, window-ordered q / k / v (csrc/attention_hm.hip)
        win = lv.window(0, K)
        qkv = torch.randn(lv.n, 3 * C, device="cuda", generator=g).to(torch.bfloat16)
        dout = torch.randn(lv.n, C, device="cuda", generator=g).to(torch.bfloat16)
        sc = d ** -0.5
        hm = nv.headmajor_pack(qkv, win, H, 3, sc * nv.LOG2E)
        out, nlse2 = nv.window_attn_hm_fwd(hm, win, H)
        fl = win.num_windows * H * 4.0 * K * K * d
        note = "%d windows x %d heads, K=%d, d=%d" % (win.num_windows, H, K, d)
        if want("attn_fwd"):
            probes.append(dict(name="attn_fwd", kernel="k_attn_hm_fwd", run=lambda: nv.window_attn_hm_fwd(hm, win, H), bound="mfma",
                               flops=fl, bytes=lv.n * C * 2 * 4, note="head-major window attention forward k_attn_hm_fwd<48> (dec0: %s)" % note))
        if want("attn_bwd"):
            probes.append(dict(name="attn_bwd", kernel="k_attn_hm_d", run=lambda: nv.window_attn_hm_bwd(hm, out, dout, nlse2, win, H, sc),
                               bound="mfma", flops=2.5 * fl, bytes=lv.n * C * 2 * 8,
                               note="head-major window attention backward, k_attn_hm_dq<48> + k_attn_hm_dkv<48> (dec0: %s)" % note))
    if want("unpool_fwd") or want("unpool_bwd"):
        # the gather / scatter launches the STEP runs at the dec0 unpooling seam (ptv3:471-482): parent + child[cluster] forward
        # (k_gather_add_v), segment sum of the gradient back to the 25,600 coarse sites backward (k_segment_reduce_v)
        l1 = plan.levels[1]
        skip = torch.randn(lv.n, C, device="cuda", generator=g).to(torch.bfloat16)
        up = torch.randn(l1.n, C, device="cuda", generator=g).to(torch.bfloat16)
        if want("unpool_fwd"):
            probes.append(dict(name="unpool_fwd", kernel="k_gather_add", run=lambda: nv.gather_add_rows(skip, up, l1.cluster), bound="hbm",
                               bytes=lv.n * (3 * C * 2 + 4), note="dec0 unpool forward: skip + up[cluster], %d x %d bf16 (the step's own launch)" % (lv.n, C)))
        if want("unpool_bwd"):
            probes.append(dict(name="unpool_bwd", kernel="k_segment_reduce", run=lambda: nv.segment_reduce(skip, l1.indices, l1.idx_ptr, l1.n, False),
                               bound="hbm", bytes=lv.n * (C * 2 + 4) + l1.n * (C * 2 + 4),
                               note="dec0 unpool backward: segment sum %d -> %d x %d bf16 (the step's own launch)" % (lv.n, l1.n, C)))
    if want("gather_hbm"):
        nb = 8 * lv.n
        src = torch.randn(nb, C, device="cuda", generator=g).to(torch.bfloat16)
        idx = torch.randperm(nb, device="cuda", generator=g).to(torch.int32)
        dst = torch.empty_like(src)
        probes.append(dict(name="gather_hbm", kernel="k_gather_rows", run=lambda: nv.gather_rows(src, idx, out=dst), bound="hbm",
                           bytes=nb * (2 * C * 2 + 4), note="gather_rows(%d x %d bf16, random permutation; %.2f GB working set)" % (nb, C, 2 * nb * C * 2 / 1e9)))
    if want("scan"):
        n, cls = 1_000_000, int(os.environ.get("SS_PROBE_SCAN_CLASSES", "160"))     # (env: diagnostic only)
        feat = torch.nn.functional.normalize(torch.randn(n, C, device="cuda", generator=g), dim=1).to(torch.bfloat16)
        text = torch.nn.functional.normalize(torch.randn(cls, C, device="cuda", generator=g), dim=1).to(torch.bfloat16)
        probes.append(dict(name="scan", kernel="k_feat_text_scan", run=lambda: nv.feat_text_scan(feat, text), bound="hbm",
                           bytes=n * (C * 2 + 8), flops=2.0 * n * C * cls,
                           note="open-vocabulary scan (config 5): %d x %d bf16 x %d classes -> sigmoid -> max/argmax" % (n, C, cls)))
    return probes


def time_probe(p, iters=5, warmup=2):
    """Average launch duration in ms: HIP events on the stream the kernel is launched on (torch's current stream)."""
    for _ in range(warmup):
        p["run"]()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        p["run"]()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def sustained_probe(p, secs=2.5):
    """Launch the probe back to back for `secs` seconds while a thread polls rocm-smi: -> dict(ms, socket_power_w, sclk_mhz, launches) or
    None when rocm-smi is not there.  What the chip holds under THIS kernel alone (the 1,400 W cap pulls the clock of the matrix kernels)."""
    import re
    import shutil
    import subprocess
    import threading
    import time
    if shutil.which("rocm-smi") is None:
        return None
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                pw = re.search(r"Power \(W\): ([0-9.]+)", out); sc = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
                if pw and sc:
                    samples.append((time.time(), float(pw.group(1)), int(sc.group(1))))
            except Exception:  # noqa: BLE001 -- a missing or slow tool must not take the bench line down
                return
            time.sleep(0.3)
    th = threading.Thread(target=poll, daemon=True); th.start()
    for _ in range(5):
        p["run"]()
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(25):
            p["run"]()
        n += 25
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join(timeout=12)
    mid = [s_ for s_ in samples if t0 + 0.8 < s_[0] < t0 + secs]
    if not mid:
        return None
    return dict(ms=e0.elapsed_time(e1) / n, launches=n, socket_power_w=sum(s_[1] for s_ in mid) / len(mid), sclk_mhz=sum(s_[2] for s_ in mid) / len(mid))


def roofline_entry(p, ms, traffic=None):
    if p["bound"] == "mfma":
        ach = p["flops"] / (ms * 1e-3) / 1e12
        ent = dict(bound="mfma", kernel=p["note"], achieved=ach, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s", frac=ach / MFMA_BF16_PEAK_TF,
                   algorithmic_flops=p["flops"], algorithmic_bytes=p["bytes"])
    else:
        ach = p["bytes"] / (ms * 1e-3) / 1e9
        ent = dict(bound="hbm", kernel=p["note"], achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                   algorithmic_bytes=p["bytes"])
    ent["ms"] = ms
    ent["traffic"] = traffic
    ent["traffic_unit"] = "bytes/launch beyond the XCD L2s (rocprofv3 FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE, measured in this run)"
    return ent


if __name__ == "__main__":
    # PMC pass: every probe a few times (no timing here; the counters are per launch)
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or None
    for p in build(which=names):
        for _ in range(3):
            p["run"]()
        torch.cuda.synchronize()
        print("probe", p["name"], "done", flush=True)
