#!/usr/bin/env python
"""Sweep SS_WGRAD_MINPER for the pipeline weight-gradient kernel (one process per value: the env is read once)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(here))
    import torch
    from scenesplat_amd import native as nv
    from scenesplat_amd.plan import build_plan
    from scenesplat_amd.synthetic import room_chunk
    from bench_kernels import ev
    g = torch.Generator(device="cuda").manual_seed(0)
    out = []
    data = room_chunk(256, 0, lang_dim=0)
    plan = build_plan(data["grid_coord"].cuda(), data["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), (2, 2, 2))
    for li, C in enumerate([768, 512, 256]):
        lv = plan.levels[li]; n = lv.n
        nbr = lv.neighbors(3); perm = lv.conv_rowperm(); blocks = lv.conv_blocks(3)
        x = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
        go = torch.randn(n, C, device="cuda", generator=g).to(torch.bfloat16)
        out.append(f"convL{li}:{ev(lambda: nv.subm_conv_wgrad_pipe(x, go, nbr, perm, blocks), 5, 2):.3f}")
    for (m, k, n) in [(102400, 768, 768), (102400, 768, 3072), (25600, 512, 1536), (102400, 256, 768), (6400, 256, 1024)]:
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        dy = torch.randn(m, n, device="cuda", generator=g).to(torch.bfloat16)
        out.append(f"lin{m}x{k}x{n}:{ev(lambda: nv.linear_wgrad(x, dy), 10, 3):.3f}")
    print(os.environ.get("SS_WGRAD_MINPER", "auto"), " ".join(out), flush=True)
else:
    sys.path.insert(0, here)
    for per in ["auto", "8", "16", "32", "64", "128", "256"]:
        env = dict(os.environ)
        if per != "auto":
            env["SS_WGRAD_MINPER"] = per
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
