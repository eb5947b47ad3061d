#!/usr/bin/env python
"""Diagnostic: nn.Linear dgrad dx = dy @ W on hipBLASLt, W (n_out, k_in) row-major (NN form) vs a transposed contiguous copy (NT form)."""
import torch
import torch.nn.functional as F
def ev(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
tot = [0.0, 0.0]
for m, k, n in [(102400, 768, 2304), (102400, 768, 768), (102400, 768, 3072), (102400, 3072, 768), (25600, 512, 1536), (25600, 512, 512), (25600, 512, 2048), (25600, 2048, 512)]:
    dy = torch.randn(m, n, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    wt = w.t().contiguous()
    t_nn = ev(lambda: dy @ w)
    t_nt = ev(lambda: F.linear(dy, wt))
    err = ((dy @ w).float() - F.linear(dy, wt).float()).abs().max().item()
    tot[0] += t_nn; tot[1] += t_nt
    print("m=%6d k_in=%4d n_out=%4d: NN %.3f ms  NT %.3f ms  (%.0f -> %.0f TFLOP/s)  max diff %.3g" % (m, k, n, t_nn, t_nt, 2.0 * m * k * n / t_nn / 1e9, 2.0 * m * k * n / t_nt / 1e9, err), flush=True)
print("sum NN %.3f ms  NT %.3f ms" % tuple(tot))
