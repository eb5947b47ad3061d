#!/usr/bin/env python
"""Per-launch time of the first-stage conv kernels (fp32 MFMA, csrc/subm_f32.hip) against the bf16x3 split path they replace,
on level 0 of the bench workload (102,400 sites): stem k = 5 (11 -> 32) and cpe k = 3 (32 -> 32), forward / dgrad / wgrad."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd import native as nv, functional as SF
from scenesplat_amd.plan import build_plan
from scenesplat_amd.synthetic import room_chunk

data = room_chunk(256, 0, lang_dim=0)
lv = build_plan(data["grid_coord"].cuda(), data["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), ()).levels[0]
n = lv.n


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for cin, k in ((32, 3), (11, 5)):
    nbr, rp, blocks, walk = lv.neighbors(k), lv.conv_rowperm(), lv.conv_blocks(k), lv.neighbors_walk(k)
    pairs = int((nbr >= 0).sum())
    cp = 16 if cin <= 16 else 32
    x = torch.randn(n, cp, device="cuda"); x[:, cin:] = 0
    g = torch.randn(n, 32, device="cuda")
    w = torch.randn(32, k ** 3, cin, device="cuda") * 0.1
    wq = nv.subm_f32_weight_layout(w)
    t_f = timeit(lambda: nv.subm_f32_fwd(x, wq, None, walk, rp))
    t_w = timeit(lambda: nv.subm_f32_wgrad(x, g, walk, rp, blocks, cin))
    flop = 2.0 * pairs * cp * 32
    print("cin %d k %d: %d pairs (%.1f per site) | fp32-MFMA fwd %.1f us (%.2f TFLOP/s, gather %.0f GB/s)  wgrad %.1f us" % (
        cin, k, pairs, pairs / n, t_f, flop / t_f / 1e6, (pairs * cp * 4 + n * 128) / t_f / 1e3, t_w))
    for mode in (True, False):
        SF.CONV_F32_MFMA = mode
        xg = x[:, :cin].clone().requires_grad_(cin == 32); wg = w.reshape(32, k, k, k, cin).clone().requires_grad_(True)
        y = SF.subm_conv3d(xg, wg, None, nbr, False, "bf16x3", rp, lambda: blocks, lambda: walk)
        t_fw = timeit(lambda: SF.subm_conv3d(xg, wg, None, nbr, False, "bf16x3", rp, lambda: blocks, lambda: walk))
        t_all = timeit(lambda: torch.autograd.backward(SF.subm_conv3d(xg, wg, None, nbr, False, "bf16x3", rp, lambda: blocks, lambda: walk), grad_tensors=g))
        print("   %s: autograd forward %.1f us, forward+backward %.1f us (all launches incl. layout / split helpers)" % (
            "fp32-MFMA" if mode else "bf16x3   ", t_fw, t_all))

# how dense are the (site tile, tap) products the kernels issue?  (zeros are multiplied for sites of an active tile without the pair)
for k in (3, 5):
    nbr, rp = lv.neighbors(k), lv.conv_rowperm()
    has = (nbr >= 0)
    has = has[:, rp.long()] if rp is not None else has
    pairs = int(has.sum())
    for tile in (16, 32, 64):
        m = (n // tile) * tile
        act = has[:, :m].reshape(has.shape[0], m // tile, tile).any(2)
        print("k %d tile %d sites: %.1f active taps per tile of %d, density of issued products %.2f" % (
            k, tile, float(act.sum()) / (m // tile), has.shape[0], pairs / (float(act.sum()) * tile)))
