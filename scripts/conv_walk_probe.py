#!/usr/bin/env python
"""dec0 / dec1 conv forward and weight gradient on the pipeline kernels: plain rulebook (scattered words) vs walk-order rulebook;
forward outputs must be bit-identical, weight gradients equal up to the order of their fp32 atomics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd import native as nv
from scenesplat_amd.plan import build_plan
from scenesplat_amd.synthetic import room_chunk

data = room_chunk(256, 0, lang_dim=0)
plan = build_plan(data["grid_coord"].cuda(), data["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), (2, 2, 2))


REPS, WARM = (int(sys.argv[1]), int(sys.argv[1]) // 4) if len(sys.argv) > 1 else (20, 3)     # e.g. 300: sustained (power-limited) clocks


def timeit(fn, reps=None):
    reps = REPS if reps is None else reps
    for _ in range(WARM):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for li, C in ((0, 768), (1, 512)):
    lv = plan.levels[li]
    n = lv.n
    nbr, rp, walk = lv.neighbors(3), lv.conv_rowperm(), lv.neighbors_walk(3)
    x = torch.randn(n, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(C, 27, C, device="cuda") * 0.02).to(torch.bfloat16)
    assert nv.subm_conv_fwd_uses_pipe(n, C, C, 27), (n, C)
    a = nv.subm_conv_fwd(x, w, None, nbr, rp)
    b = nv.subm_conv_fwd(x, w, None, nbr, rp, nbr_walk=walk)
    print("level %d n %d C %d: identical %s | plain %.1f us, walk-order %.1f us" % (
        li, n, C, torch.equal(a, b), timeit(lambda: nv.subm_conv_fwd(x, w, None, nbr, rp)),
        timeit(lambda: nv.subm_conv_fwd(x, w, None, nbr, rp, nbr_walk=walk))))
    g = torch.randn(n, C, device="cuda").to(torch.bfloat16)
    blocks = lv.conv_blocks(3)
    da = nv.subm_conv_wgrad(x, g, nbr, rp, blocks)
    db = nv.subm_conv_wgrad(x, g, nbr, rp, blocks, nbr_walk=walk)
    rel = float((da - db).norm() / da.norm())
    print("   wgrad: rel diff %.1e | plain %.1f us, walk-order %.1f us" % (
        rel, timeit(lambda: nv.subm_conv_wgrad(x, g, nbr, rp, blocks)), timeit(lambda: nv.subm_conv_wgrad(x, g, nbr, rp, blocks, nbr_walk=walk))))
