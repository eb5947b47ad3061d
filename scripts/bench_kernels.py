#!/usr/bin/env python
"""Isolated timings of the hot HIP kernels at the dec0 shapes of room-102400 (GPU box helper)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd import native as nv
from scenesplat_amd.plan import build_plan
from scenesplat_amd.synthetic import room_chunk


def ev(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    which = sys.argv[1:] or ["attn", "conv"]
    data = room_chunk(256, 0, lang_dim=0)
    plan = build_plan(data["grid_coord"].cuda(), data["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), (2, 2, 2))
    g = torch.Generator(device="cuda").manual_seed(0)
    for li, (C, H) in enumerate([(768, 16), (512, 16), (256, 16)]):
        lv = plan.levels[li]
        n = lv.n
        if "attn" in which:
            win = lv.window(0, 1024)
            qkv = torch.randn(n, 3 * C, device="cuda", generator=g).to(torch.bfloat16)
            dout = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
            sc = (C // H) ** -0.5
            o, lse = nv.window_attn_fwd(qkv, win, H, sc, nv.ATTN_MFMA)
            tf = ev(lambda: nv.window_attn_fwd(qkv, win, H, sc, nv.ATTN_MFMA))
            tb = ev(lambda: nv.window_attn_bwd(qkv, o, dout, lse, win, H, sc, nv.ATTN_MFMA))
            fl = win.num_windows * H * 4.0 * 1024 * 1024 * (C // H)
            print(f"attn L{li} n={n} C={C} d={C//H}: fwd {tf:.3f} ms ({fl/tf/1e9:.0f} TF/s)  bwd {tb:.3f} ms ({2.5*fl/tb/1e9:.0f} TF/s alg)", flush=True)
        if "conv" in which:
            nbr = lv.neighbors(3); perm = lv.conv_rowperm()
            x = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
            w = (torch.randn(C, 27, C, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
            gout = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
            act = (nbr >= 0).float().sum().item() / n
            tf = ev(lambda: nv.subm_conv_fwd(x, w, None, nbr, perm), 5, 2)
            blocks = lv.conv_blocks(3)
            tw = ev(lambda: nv.subm_conv_wgrad(x, gout, nbr, perm, blocks), 5, 2)
            fl = 2.0 * n * act * C * C
            if nv.lib().ss_wgrad8_ok(n, C, C, 27):
                refw = nv.subm_conv_wgrad(x, gout, nbr, perm, blocks)
                gotw = nv.subm_conv_wgrad_pipe(x, gout, nbr, perm, blocks)
                errw = (gotw - refw).abs().max().item() / refw.abs().max().item()
                tw2 = ev(lambda: nv.subm_conv_wgrad_pipe(x, gout, nbr, perm, blocks), 5, 2)
                print(f"   pipe kernel: wgrad {tw2:.3f} ms ({fl/tw2/1e9:.0f} TF/s)  rel err vs dispatch {errw:.2e}", flush=True)
            if nv.lib().ss_gemm8_ok(n, C, C, 27):
                ref = nv.subm_conv_fwd(x, w, None, nbr, perm).float()
                got = nv.subm_conv_fwd_pipe(x, w, None, nbr, perm).float()
                err = (got - ref).abs().max().item() / ref.abs().max().item()
                tp = ev(lambda: nv.subm_conv_fwd_pipe(x, w, None, nbr, perm), 5, 2)
                print(f"   pipe kernel: fwd {tp:.3f} ms ({fl/tp/1e9:.0f} TF/s)  rel err vs dispatch {err:.2e}", flush=True)
            print(f"conv L{li} n={n} C={C} taps/site={act:.2f}: fwd {tf:.3f} ms ({fl/tf/1e9:.0f} TF/s)  wgrad {tw:.3f} ms ({fl/tw/1e9:.0f} TF/s)", flush=True)


def gemm():
    g = torch.Generator(device="cuda").manual_seed(0)
    for (m, k, n) in [(102400, 768, 2304), (102400, 768, 768), (102400, 768, 3072), (102400, 3072, 768),
                      (25600, 512, 1536), (25600, 512, 2048), (25600, 2048, 512), (6400, 256, 1024), (102400, 256, 768)]:
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        b = torch.randn(n, device="cuda", generator=g)
        bb = b.to(torch.bfloat16)
        ref = torch.nn.functional.linear(x, w, bb).float()
        got = nv.linear_fwd(x, w, b).float()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        t0 = ev(lambda: torch.nn.functional.linear(x, w, bb), 10, 3)
        t1 = ev(lambda: nv.linear_fwd(x, w, b), 10, 3)
        dy = torch.randn(m, n, device="cuda", generator=g).to(torch.bfloat16)
        wt = w.t().contiguous()
        t2 = ev(lambda: dy @ w, 10, 3)                       # dgrad, NN
        t3 = ev(lambda: nv.linear_fwd(dy, wt), 10, 3)        # dgrad on the pipe kernel with W^T
        t4 = ev(lambda: dy.t() @ x, 10, 3)                   # wgrad, TN
        refw = (dy.t() @ x).float()
        gotw = nv.linear_wgrad(x, dy)
        errw = (gotw - refw).abs().max().item() / refw.abs().max().item()
        t5 = ev(lambda: nv.linear_wgrad(x, dy), 10, 3)
        print(f"   wgrad pipe {t5:.3f} ms ({2.0*m*k*n/t5/1e9:.0f} TF/s) rel err {errw:.2e}", flush=True)
        fl = 2.0 * m * k * n
        print(f"   dgrad hipBLASLt {t2:.3f} ms ({fl/t2/1e9:.0f} TF/s) pipe(W^T) {t3:.3f} ms ({fl/t3/1e9:.0f} TF/s) | wgrad hipBLASLt {t4:.3f} ms ({fl/t4/1e9:.0f} TF/s)", flush=True)
        print(f"gemm m={m} k={k} n={n}: hipBLASLt {t0:.3f} ms ({fl/t0/1e9:.0f} TF/s)  pipe {t1:.3f} ms ({fl/t1/1e9:.0f} TF/s)  rel err {err:.2e}", flush=True)


if __name__ == "__main__":
    if "gemm" in sys.argv[1:]:
        gemm()
        sys.argv = [a for a in sys.argv if a != "gemm"]
        if len(sys.argv) == 1:
            sys.exit(0)
    main()
