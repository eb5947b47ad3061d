"""GPU parity: float kernels (rows, CSR pooling, submanifold conv, window attention) vs the
oracle and the reference golden vectors."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as oops
from oracle import ptv3 as optv3

pytestmark = pytest.mark.gpu
ORD = ("z", "z-trans", "hilbert", "hilbert-trans")


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


@pytest.mark.parametrize("C,dtype", [(11, torch.float32), (32, torch.float32), (768, torch.bfloat16), (6, torch.bfloat16)])
def test_gather_scatter_rows(C, dtype):
    from scenesplat_amd import native as nv
    g = torch.Generator().manual_seed(C)
    src = torch.randn(1000, C, generator=g).to(dtype).cuda()
    idx = torch.randint(-1, 1000, (1777,), generator=g, dtype=torch.int32).cuda()
    out = nv.gather_rows(src, idx)
    ref = torch.where((idx >= 0).unsqueeze(1), src[idx.clamp(min=0).long()], torch.zeros_like(src[:1]))
    assert torch.equal(out, ref)
    perm = torch.randperm(1000, generator=g).to(torch.int32).cuda()
    perm[::7] = -1
    dst = torch.zeros_like(src)
    nv.scatter_rows(src, perm, dst)
    ref = torch.zeros_like(src)
    keep = perm >= 0
    ref[perm[keep].long()] = src[keep]
    assert torch.equal(dst, ref)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_segment_mean_and_unpool_against_oracle(dtype, tol):
    from scenesplat_amd import functional as SF
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(1)
    gc = torch.randint(0, 40, (5000, 3), generator=g)
    gc = torch.unique(gc, dim=0)
    gc = gc[torch.randperm(len(gc), generator=g)]
    offs = torch.tensor([len(gc) // 2, len(gc)])
    plan = build_plan(gc.cuda(), offs.cuda(), ORD, (2,))
    ref = optv3.build_levels(gc.numpy(), offs.numpy(), ORD, (2,))
    fine, coarse = plan.levels
    C = 24
    x = torch.randn(fine.n, C, generator=g)
    up = torch.randn(coarse.n, C, generator=g)
    xg = x.to(dtype).cuda().requires_grad_(True)
    y = SF.segment_mean(xg, coarse)
    xo = x.clone().requires_grad_(True)
    yo = oops.segment_csr(xo[torch.as_tensor(ref[1].indices)], ref[1].idx_ptr, "mean")
    assert torch.allclose(y.float().cpu(), yo, atol=tol, rtol=tol)
    cot = torch.randn(coarse.n, C, generator=g)
    (y.float() * cot.cuda()).sum().backward(); (yo * cot).sum().backward()
    assert torch.allclose(xg.grad.float().cpu(), xo.grad, atol=tol, rtol=tol)
    # unpool
    sg, ug = x.to(dtype).cuda().requires_grad_(True), up.to(dtype).cuda().requires_grad_(True)
    z = SF.unpool_add(sg, ug, coarse)
    so, uo = x.clone().requires_grad_(True), up.clone().requires_grad_(True)
    zo = so + uo[torch.as_tensor(ref[1].cluster)]
    assert torch.allclose(z.float().cpu(), zo, atol=tol * 4, rtol=tol)
    cot = torch.randn(fine.n, C, generator=g)
    (z.float() * cot.cuda()).sum().backward(); (zo * cot).sum().backward()
    assert torch.allclose(sg.grad.float().cpu(), so.grad, atol=tol, rtol=tol)
    assert torch.allclose(ug.grad.float().cpu(), uo.grad, atol=tol * 8, rtol=tol * 2)


@pytest.mark.parametrize("k,cin,cout,dup", [(3, 16, 16, False), (5, 11, 32, False), (3, 8, 8, True)])
def test_subm_conv_against_oracle(k, cin, cout, dup):
    from scenesplat_amd import functional as SF
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(k + cin)
    gc = torch.randint(0, 12, (1500, 3), generator=g)
    if not dup:
        gc = torch.unique(gc, dim=0)
        gc = gc[torch.randperm(len(gc), generator=g)]
    n = len(gc)
    offs = torch.tensor([n // 3, n])
    plan = build_plan(gc.cuda(), offs.cuda(), ORD, ())
    lv = plan.levels[0]
    assert lv.has_duplicates == dup
    batch = np.repeat([0, 1], [n // 3, n - n // 3])
    nbr = oops.neighbor_table(gc.numpy(), batch, k)
    x = torch.randn(n, cin, generator=g); w = torch.randn(cout, k, k, k, cin, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    cot = torch.randn(n, cout, generator=g)
    xo, wo, bo = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = oops.subm_conv3d(xo, wo, bo, nbr)
    (yo * cot).sum().backward()
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = SF.subm_conv3d(xg, wg, bg, lv.neighbors(k), lv.has_duplicates, torch.float32)
    (y * cot.cuda()).sum().backward()
    assert torch.allclose(y.cpu(), yo, atol=2e-4, rtol=1e-4)
    assert torch.allclose(xg.grad.cpu(), xo.grad, atol=2e-4, rtol=1e-4)
    assert torch.allclose(wg.grad.cpu(), wo.grad, atol=1e-3, rtol=1e-4)
    assert torch.allclose(bg.grad.cpu(), bo.grad, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("k,cin,cout,dup", [(3, 32, 32, False), (3, 96, 160, False), (5, 11, 32, False), (3, 256, 256, False),
                                             (3, 64, 8, False), (3, 32, 48, True), (5, 16, 16, True)])
def test_subm_conv_fused_mfma_against_oracle(k, cin, cout, dup):
    """bf16 MFMA implicit-GEMM conv (fwd, dgrad, wgrad) vs the fp32 oracle on bf16-rounded operands."""
    from scenesplat_amd import functional as SF
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(k * 1000 + cin + cout)
    gc = torch.randint(0, 14, (2600, 3), generator=g)
    if not dup:
        gc = torch.unique(gc, dim=0)         # dup=True keeps the repeated voxels (Mix3D-style batches)
    gc = gc[torch.randperm(len(gc), generator=g)]
    n = len(gc)
    offs = torch.tensor([n // 3, n])
    plan = build_plan(gc.cuda(), offs.cuda(), ORD, ())
    lv = plan.levels[0]
    assert lv.has_duplicates == dup
    batch = np.repeat([0, 1], [n // 3, n - n // 3])
    nbr = oops.neighbor_table(gc.numpy(), batch, k)
    rb = lambda t: t.to(torch.bfloat16).float()
    x = rb(torch.randn(n, cin, generator=g)); w = rb(torch.randn(cout, k, k, k, cin, generator=g) * 0.2)
    b = torch.randn(cout, generator=g); cot = rb(torch.randn(n, cout, generator=g))
    xo, wo, bo = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = oops.subm_conv3d(xo, wo, bo, nbr)
    (yo * cot).sum().backward()
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = SF.subm_conv3d(xg, wg, bg, lv.neighbors(k), lv.has_duplicates, torch.bfloat16, lv.conv_rowperm())
    assert y.dtype == torch.bfloat16                     # under autocast the next op is a bf16 GEMM
    with torch.no_grad():                                # outside autocast (evaluator call form): fp32 out for the fp32 Linear
        assert SF.subm_conv3d(xg, wg, bg, lv.neighbors(k), lv.has_duplicates, torch.bfloat16, lv.conv_rowperm()).dtype == torch.float32
    (y.float() * cot.cuda()).sum().backward()
    def rel(a, r): return ((a.float().cpu() - r).norm() / r.norm()).item()
    assert rel(y, yo) < 6e-3, rel(y, yo)                 # bf16 output rounding
    assert rel(xg.grad, xo.grad) < 6e-3, rel(xg.grad, xo.grad)
    assert rel(wg.grad, wo.grad) < 2e-3, rel(wg.grad, wo.grad)   # fp32 accumulate / output
    assert rel(bg.grad, bo.grad) < 2e-3


@pytest.mark.parametrize("n_pts,order", [(3000, "z"), (70000, "mask"), (70000, "z")])
def test_subm_block_lists_and_row_order(n_pts, order, monkeypatch):
    """Active 64-site block lists (ss_subm_block_lists) against a numpy restatement, for the plain curve walk and the
    neighbour-mask regrouped walk; the walk order must be a permutation of the sites."""
    from scenesplat_amd import native as nv, plan as P
    monkeypatch.setattr(P, "CONV_ORDER", order)
    monkeypatch.setattr(P, "CONV_MASK_MIN_SITES", 1000)
    g = torch.Generator().manual_seed(n_pts)
    gc = torch.unique(torch.cat([torch.randint(0, 96, (n_pts, 2), generator=g), torch.randint(0, 3, (n_pts, 1), generator=g)], 1), dim=0)
    gc = gc[torch.randperm(len(gc), generator=g)]
    n = len(gc)
    plan = P.build_plan(gc.cuda(), torch.tensor([n]).cuda(), ORD, ())
    lv = plan.levels[0]
    nbr = lv.neighbors(3); perm = lv.conv_rowperm()
    assert sorted(perm.cpu().tolist()) == list(range(n))
    cnt, lst = nv.subm_block_lists(nbr, perm)
    a = (nbr[:, perm.long()] >= 0).cpu().numpy()
    nb = (n + 63) // 64
    a = np.concatenate([a, np.zeros((27, nb * 64 - n), bool)], 1).reshape(27, nb, 64).any(2)
    assert cnt.cpu().tolist() == a.sum(1).tolist()
    for t in range(27):
        assert lst[t, :int(cnt[t])].cpu().tolist() == np.nonzero(a[t])[0].tolist()
    if order == "mask":      # regrouping must not issue more tap work per 256-site tile than the plain walk
        zperm = lv.order[list(lv.curve_names).index("z")]
        def issued(p):
            b = (nbr[:, p.long()] >= 0).cpu().numpy()
            nt = (n + 255) // 256
            b = np.concatenate([b, np.zeros((27, nt * 256 - n), bool)], 1).reshape(27, nt, 256).any(2)
            return int(b.sum())
        assert issued(perm) <= issued(zperm)


@pytest.mark.parametrize("k,cin,cout,dup,f32out", [(3, 64, 256, False, False), (3, 128, 72, False, True), (5, 64, 48, False, False),
                                                    (3, 192, 260, True, False)])
def test_subm_conv_pipeline_kernel_against_oracle(k, cin, cout, dup, f32out):
    """The 256x256 LDS-DMA pipeline kernel (gemm8.hip, GATHER form) called directly: ragged site and channel tails,
    k=5 needs taps <= 27 so it must be refused there."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(k * 77 + cin + cout)
    gc = torch.randint(0, 14, (2600, 3), generator=g)
    if not dup:
        gc = torch.unique(gc, dim=0)
    gc = gc[torch.randperm(len(gc), generator=g)]
    n = len(gc)
    offs = torch.tensor([n // 3, n])
    plan = build_plan(gc.cuda(), offs.cuda(), ORD, ())
    lv = plan.levels[0]
    batch = np.repeat([0, 1], [n // 3, n - n // 3])
    rb = lambda t: t.to(torch.bfloat16).float()
    x = rb(torch.randn(n, cin, generator=g)); w = rb(torch.randn(cout, k, k, k, cin, generator=g) * 0.2)
    b = torch.randn(cout, generator=g)
    wk = w.reshape(cout, k ** 3, cin).to(torch.bfloat16).cuda()
    if k == 5:
        assert not nv.lib().ss_gemm8_ok(n, cin, cout, k ** 3)
        with pytest.raises(RuntimeError):
            nv.subm_conv_fwd_pipe(x.to(torch.bfloat16).cuda(), wk, b.cuda(), lv.neighbors(k), lv.conv_rowperm())
        return
    nbr = oops.neighbor_table(gc.numpy(), batch, k)
    yo = oops.subm_conv3d(x, w, b, nbr)
    y = nv.subm_conv_fwd_pipe(x.to(torch.bfloat16).cuda(), wk, b.cuda(), lv.neighbors(k), lv.conv_rowperm(),
                              torch.float32 if f32out else torch.bfloat16)
    rel = ((y.float().cpu() - yo).norm() / yo.norm()).item()
    assert rel < (2e-5 if f32out else 6e-3), rel
    # identity walk order (rowperm = NULL) gives the same rows
    y2 = nv.subm_conv_fwd_pipe(x.to(torch.bfloat16).cuda(), wk, b.cuda(), lv.neighbors(k), None, torch.float32)
    assert ((y2.cpu() - yo).norm() / yo.norm()).item() < 2e-5


@pytest.mark.parametrize("m,k,n,bias", [(1, 64, 4, True), (255, 64, 256, False), (257, 128, 260, True), (1000, 192, 768, True),
                                        (3000, 4096, 36, False)])
def test_linear_pipeline_kernel_against_fp64(m, k, n, bias):
    """ss_linear_fwd (gemm8.hip, plain form) vs an fp64 matmul of the same bf16 operands; fp32 output is exact to
    accumulation order, bf16 output to one rounding."""
    from scenesplat_amd import native as nv
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16); w = (torch.randn(n, k, generator=g) * 0.1).to(torch.bfloat16)
    b = torch.randn(n, generator=g) if bias else None
    ref = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    y32 = nv.linear_fwd(x.cuda(), w.cuda(), b.cuda() if bias else None, torch.float32).cpu().double()
    y16 = nv.linear_fwd(x.cuda(), w.cuda(), b.cuda() if bias else None, torch.bfloat16).cpu().double()
    scale = ref.abs().max().item()
    assert (y32 - ref).abs().max().item() < 2e-5 * scale
    assert (y16 - ref).abs().max().item() < 5e-3 * scale
    with pytest.raises(RuntimeError):       # K must be a multiple of 64
        nv.linear_fwd(x[:, :40].contiguous().cuda(), w[:, :40].contiguous().cuda())


@pytest.mark.parametrize("m,k,n,bias", [(5000, 64, 192, True), (4100, 136, 72, False), (300, 64, 64, True)])
def test_linear_autograd_matches_torch_autocast(m, k, n, bias):
    """SF.linear under bf16 autocast: same forward / dx as torch's autocast F.linear, weight and bias gradients equal to
    the fp64 result of the same bf16 operands (they are fp32-accumulated and never rounded to bf16)."""
    from scenesplat_amd import functional as SF
    g = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, k, generator=g).cuda(); w = (torch.randn(n, k, generator=g) * 0.1).cuda(); b = torch.randn(n, generator=g).cuda() if bias else None
    cot = torch.randn(m, n, generator=g).cuda().to(torch.bfloat16)
    outs = []
    for fn in (SF.linear, torch.nn.functional.linear):
        xi, wi = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        bi = b.clone().requires_grad_(True) if bias else None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn(xi, wi, bi)
        y.backward(cot)
        outs.append((y, xi.grad, wi.grad, bi.grad if bias else None))
    (y0, dx0, dw0, db0), (y1, dx1, dw1, db1) = outs
    assert y0.dtype == torch.bfloat16 and torch.equal(y0, y1)
    assert torch.allclose(dx0, dx1, atol=1e-2, rtol=1e-2)
    assert dw0.dtype == torch.float32
    ref_dw = cot.double().t() @ x.to(torch.bfloat16).double()
    tol = 1e-4 if m >= SF.LINEAR_WGRAD_MIN_ROWS else 1e-2      # below the threshold hipBLASLt's bf16 result is used
    assert (dw0.double() - ref_dw).abs().max().item() < tol * ref_dw.abs().max().item() + 1e-5
    assert (dw1.double() - ref_dw).abs().max().item() < 1e-2 * ref_dw.abs().max().item()
    if bias:
        ref_db = cot.double().sum(0)
        assert (db0.double() - ref_db).abs().max().item() < 1e-4 * ref_db.abs().max().item() + 1e-5


def _attn_case(golden_dir, name):
    fx = np.load(os.path.join(golden_dir, "attention.npz"))
    C, H, K, oi = [int(v) for v in fx[f"{name}_cfg"]]
    sd = {k[len(name) + 4:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(name + "_sd_")}
    return fx, C, H, K, oi, sd


@pytest.mark.parametrize("name", ["h2d16", "h2d48"])
@pytest.mark.parametrize("impl,dtype,tol", [("simt", torch.float32, 3e-5), ("simt", torch.bfloat16, 3e-2),
                                            ("mfma", torch.bfloat16, 3e-2)])
def test_window_attention_matches_reference_module(golden_dir, name, impl, dtype, tol):
    """SerializedAttention (qkv Linear -> windows -> proj Linear) against the reference module's
    output and gradients (non-flash math, padded tail window)."""
    from scenesplat_amd import functional as SF, native as nv
    from scenesplat_amd.plan import build_plan
    fx, C, H, K, oi, sd = _attn_case(golden_dir, name)
    plan = build_plan(dev(fx[f"{name}_gc"]), dev(fx[f"{name}_offset"]), ORD, ())
    win = plan.levels[0].window(oi, K)
    assert win.n_pad > win.n            # the fixture exercises duplicate padding
    x = dev(fx[f"{name}_x"]).requires_grad_(True)
    p = {k: v.cuda().requires_grad_(True) for k, v in sd.items()}
    qkv = F.linear(x, p["qkv.weight"], p["qkv.bias"])
    a = SF.window_attention(qkv.to(dtype), win, H, (C // H) ** -0.5, nv.ATTN_SIMT if impl == "simt" else nv.ATTN_MFMA)
    y = F.linear(a.float(), p["proj.weight"], p["proj.bias"])
    (y * dev(fx[f"{name}_cot"])).sum().backward()
    ref_y = torch.from_numpy(fx[f"{name}_y"])
    scale = ref_y.abs().max().item()
    assert (y.detach().cpu() - ref_y).abs().max() <= tol * max(1.0, scale)
    ref_dx = torch.from_numpy(fx[f"{name}_dx"])
    assert (x.grad.cpu() - ref_dx).abs().max() <= tol * max(1.0, ref_dx.abs().max().item()) * 2
    for k in p:
        r = torch.from_numpy(fx[f"{name}_grad_{k}"])
        assert (p[k].grad.cpu() - r).norm() <= tol * 4 * r.norm() + 1e-5, k


@pytest.mark.parametrize("impl", ["simt", "mfma"])
@pytest.mark.parametrize("L", [1, 5, 63, 64, 65, 200])
def test_window_attention_short_windows(L, impl):
    """An element with <= K points is one short window (varlen semantics, ptv3:135-136)."""
    from scenesplat_amd import functional as SF, native as nv
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(L)
    n = L + 300
    gc = torch.stack([torch.randperm(n, generator=g), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1)
    offs = torch.tensor([L, n])
    plan = build_plan(gc.cuda(), offs.cuda(), ("hilbert",), ())
    lv = plan.levels[0]
    win = lv.window(0, 256)
    H, d = 2, 16
    qkv = torch.randn(n, 3 * H * d, generator=g)
    if impl == "mfma":
        qkv = qkv.to(torch.bfloat16).float()
        out = SF.window_attention(qkv.cuda().to(torch.bfloat16), win, H, d ** -0.5, nv.ATTN_MFMA).float().cpu()
    else:
        out = SF.window_attention(qkv.cuda(), win, H, d ** -0.5, nv.ATTN_SIMT).cpu()
    ref = optv3.build_levels(gc.numpy(), offs.numpy(), ("hilbert",), ())[0]
    pad, unpad, cu = ref.padding(256)
    o = oops.window_attention(qkv[torch.as_tensor(ref.order[0][pad])], cu, H, d ** -0.5)
    o = o[torch.as_tensor(unpad[ref.inverse[0]])]
    tol = 2e-2 if impl == "mfma" else 2e-5
    assert torch.allclose(out, o, atol=tol, rtol=tol)


@pytest.mark.parametrize("H,d,K,counts", [(4, 48, 1024, [2500]), (2, 16, 1024, [1100, 900]), (2, 32, 256, [700, 300, 40]),
                                           (1, 64, 128, [333])])
def test_window_attention_mfma_matches_simt_fwd_bwd(H, d, K, counts):
    """MFMA kernels against the fp32-math SIMT kernels on identical bf16 inputs (asymmetric random
    data, padded tail windows, multi-tile windows): forward, dQ, dK, dV incl. borrowed-slot fix-up."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(H * d)
    n = sum(counts)
    gc = torch.stack([torch.randperm(n, generator=g), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1)
    plan = build_plan(gc.cuda(), torch.tensor(counts).cumsum(0).cuda(), ("hilbert", "z"), ())
    win = plan.levels[0].window(1, K)
    C = H * d
    qkv = (torch.randn(n, 3 * C, generator=g) * 1.5).to(torch.bfloat16).cuda()
    dout = torch.randn(n, C, generator=g).to(torch.bfloat16).cuda()
    scale = d ** -0.5
    o_s, lse_s = nv.window_attn_fwd(qkv, win, H, scale, nv.ATTN_SIMT)
    o_m, lse_m = nv.window_attn_fwd(qkv, win, H, scale, nv.ATTN_MFMA)
    assert torch.allclose(lse_m, lse_s, atol=2e-2, rtol=1e-3)
    assert (o_m.float() - o_s.float()).abs().max() < 3e-2
    g_s = nv.window_attn_bwd(qkv, o_s, dout, lse_s, win, H, scale, nv.ATTN_SIMT).float()
    g_m = nv.window_attn_bwd(qkv, o_s, dout, lse_s, win, H, scale, nv.ATTN_MFMA).float()
    for name, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        a, b = g_m[:, sl], g_s[:, sl]
        rel = (a - b).norm() / b.norm()
        assert rel < 2e-2, (name, rel.item())
        assert (a - b).abs().max() < 0.05 * b.abs().max() + 1e-2, name


@pytest.mark.parametrize("C", [16, 48, 256, 768])
@pytest.mark.parametrize("xdt,ydt,hdt", [(torch.float32, torch.float32, torch.float32), (torch.float32, torch.bfloat16, torch.bfloat16),
                                         (torch.bfloat16, torch.bfloat16, torch.bfloat16)])
def test_fused_add_layernorm_against_torch(C, xdt, ydt, hdt):
    """csrc/norm.hip vs the plain fp32 PyTorch composition x + s*y -> LayerNorm, forward and backward
    (incl. the bf16-copy gradient path and the DropPath row scale)."""
    from scenesplat_amd import functional as SF
    g = torch.Generator().manual_seed(C)
    n = 1037
    x = torch.randn(n, C, generator=g).to(xdt); y = torch.randn(n, C, generator=g).to(ydt)
    rs = (torch.rand(n, generator=g) < 0.7).float() / 0.7
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    cx, ch, cc = torch.randn(n, C, generator=g), torch.randn(n, C, generator=g), torch.randn(n, C, generator=g)
    # reference
    xr, yr, gr, br = (t.float().clone().requires_grad_(True) for t in (x, y, gam, bet))
    vo = xr + rs[:, None] * yr
    ho = F.layer_norm(vo, (C,), gr, br, 1e-5)
    (vo * cx + ho * ch + vo.to(torch.bfloat16).float() * cc).sum().backward()   # bf16 copy: straight-through for the gradient
    xg, yg, gg, bg = (t.cuda().requires_grad_(True) for t in (x, y, gam, bet))
    xo, h, xc = SF.add_layer_norm(xg, yg, rs.cuda(), gg, bg, 1e-5, True, hdt)
    assert xo.dtype == torch.float32 and h.dtype == hdt and xc.dtype == torch.bfloat16
    (xo * cx.cuda() + h.float() * ch.cuda() + xc.float() * cc.cuda()).sum().backward()
    tol = 1e-5 if hdt == torch.float32 else 2e-2
    assert torch.allclose(xo.cpu(), vo.detach(), atol=1e-6)
    assert torch.allclose(h.float().cpu(), ho.detach(), atol=tol, rtol=tol)
    assert torch.allclose(xc.float().cpu(), vo.detach(), atol=1e-2, rtol=8e-3)     # bf16 rounding of the fused sum (fma contraction may flip an ulp)
    gtol = 2e-5 if ydt == torch.float32 else 3e-2
    assert torch.allclose(xg.grad.float().cpu(), xr.grad, atol=gtol if xdt == torch.float32 else 3e-2, rtol=1e-2)
    assert torch.allclose(yg.grad.float().cpu(), yr.grad, atol=gtol, rtol=1e-2)
    ptol = 1e-3 if hdt == torch.float32 else 8e-3      # bf16 h => the incoming gradient is rounded to bf16 by autograd
    assert (gg.grad.cpu() - gr.grad).norm() <= ptol * gr.grad.norm() + 1e-4
    assert (bg.grad.cpu() - br.grad).norm() <= ptol * br.grad.norm() + 1e-4
    # plain LN (no add), bf16 in / bf16 out as used for cpe.2
    t = torch.randn(n, C, generator=g).to(ydt)
    tr, gr, br = t.float().clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    (F.layer_norm(tr, (C,), gr, br, 1e-5) * ch).sum().backward()
    tg, gg, bg = t.cuda().requires_grad_(True), gam.cuda().requires_grad_(True), bet.cuda().requires_grad_(True)
    o = SF.layer_norm(tg, gg, bg, 1e-5)
    assert o.dtype == ydt
    (o.float() * ch.cuda()).sum().backward()
    assert torch.allclose(tg.grad.float().cpu(), tr.grad, atol=gtol, rtol=2e-2)
    assert (gg.grad.cpu() - gr.grad).norm() <= (2e-3 if ydt == torch.float32 else 1e-2) * gr.grad.norm() + 1e-4


@pytest.mark.parametrize("C,act", [(32, True), (768, True), (48, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("training", [True, False])
def test_fused_batchnorm_gelu_against_torch(C, act, dtype, training):
    from scenesplat_amd import functional as SF
    g = torch.Generator().manual_seed(C + int(act))
    n = 2111
    x = (torch.randn(n, C, generator=g) * 1.7 + 0.3).to(dtype)
    cot = torch.randn(n, C, generator=g)
    def mk():
        bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01)
        with torch.no_grad():
            bn.weight.copy_(1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(1)))
            bn.bias.copy_(0.1 * torch.randn(C, generator=torch.Generator().manual_seed(2)))
            bn.running_mean.copy_(0.2 * torch.randn(C, generator=torch.Generator().manual_seed(3)))
            bn.running_var.copy_(1 + 0.2 * torch.rand(C, generator=torch.Generator().manual_seed(4)))
        return bn.train(training)
    ref, dev = mk(), mk().cuda()
    xr = x.float().clone().requires_grad_(True)
    yr = ref(xr)
    if act:
        yr = F.gelu(yr)
    (yr * cot).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = SF.batch_norm_act(xg, dev, act)
    assert y.dtype == dtype
    (y.float() * cot.cuda()).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(y.float().cpu(), yr.detach(), atol=tol, rtol=tol)
    assert torch.allclose(xg.grad.float().cpu(), xr.grad, atol=tol * 2, rtol=5e-2 if dtype == torch.bfloat16 else 1e-3)
    rt = 2e-4 if dtype == torch.float32 else 8e-3
    assert (dev.weight.grad.cpu() - ref.weight.grad).norm() <= rt * ref.weight.grad.norm() + 1e-4
    assert (dev.bias.grad.cpu() - ref.bias.grad).norm() <= rt * ref.bias.grad.norm() + 1e-4
    assert torch.allclose(dev.running_mean.cpu(), ref.running_mean, atol=1e-5, rtol=1e-4)
    assert torch.allclose(dev.running_var.cpu(), ref.running_var, atol=1e-5, rtol=1e-4)
    assert int(dev.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.parametrize("C", [6, 20, 100, 160, 200])
def test_open_vocab_scan_against_reference_math(C):
    """feat x text^T -> sigmoid -> max/argmax (evaluator.py:793-800) and fragment accumulation (test.py:343-351),
    with the reference's own SigLIP2 text-embedding shapes (6..200 x 768)."""
    from scenesplat_amd import native as nv
    g = torch.Generator().manual_seed(C)
    n, D = 3001, 768
    feat = F.normalize(torch.randn(n, D, generator=g), dim=1).to(torch.bfloat16)
    text = F.normalize(torch.randn(C, D, generator=g), dim=1).to(torch.bfloat16)
    logits = feat.float() @ text.float().t()
    probs = torch.sigmoid(logits)
    mp, am = nv.feat_text_scan(feat.cuda(), text.cuda())
    rmax, rarg = probs.max(1)
    assert torch.allclose(mp.cpu(), rmax, atol=2e-4)
    agree = (am.cpu().long() == rarg)
    assert agree.float().mean() > 0.999 and torch.allclose(probs[torch.arange(n), am.cpu().long()], rmax, atol=2e-4)
    idx = torch.randperm(5000, generator=g)[:n].to(torch.int32)
    pred = torch.zeros(5000, C, device="cuda")
    nv.feat_text_scan(feat.cuda(), text.cuda(), want_max=False, idx=idx.cuda(), pred_accum=pred)
    nv.feat_text_scan(feat.cuda(), text.cuda(), want_max=False, idx=idx.cuda(), pred_accum=pred)
    ref = torch.zeros(5000, C); ref[idx.long()] = 2 * probs
    assert torch.allclose(pred.cpu(), ref, atol=5e-4)


@pytest.mark.parametrize("cout,taps,cin", [(32, 27, 32), (48, 125, 16), (256, 27, 72), (1, 1, 1)])
def test_subm_weight_mirror(cout, taps, cin):
    from scenesplat_amd import native as nv
    w = torch.randn(cout, taps, cin, generator=torch.Generator().manual_seed(cout + cin)).to(torch.bfloat16).cuda()
    assert torch.equal(nv.subm_weight_mirror(w), w.flip(1).permute(2, 1, 0).contiguous())


@pytest.mark.parametrize("reduce", ["min", "max"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_segment_minmax_against_oracle(reduce, dtype):
    """reduce='min'/'max' of torch_scatter.segment_csr (row a14) on a pooled level: values and the gradient routing
    to the arg rows, against the oracle's scatter_reduce restatement (distinct values -> no ties)."""
    from scenesplat_amd import functional as SF
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(11)
    gc = torch.unique(torch.randint(0, 24, (3000, 3), generator=g), dim=0)
    n = len(gc)
    plan = build_plan(gc.cuda(), torch.tensor([n // 2, n]).cuda(), ORD, (2,))
    fine, coarse = plan.levels
    C = 20
    # all distinct and never exactly 0: torch's scatter_reduce backward counts the zero-initialised `self` as a tie
    # even with include_self=False (an artefact of the oracle's restatement, not of segment_csr)
    x = -(torch.randperm(n * C, generator=g).float().reshape(n, C) + 1) / (n * C)
    if dtype == torch.bfloat16:      # bf16-exact values that stay distinct inside every cluster (<= 8 rows)
        idx0 = coarse.indices.long().cpu(); ptr0 = coarse.idx_ptr.cpu()[: coarse.n + 1].long()
        seg = torch.repeat_interleave(torch.arange(coarse.n), ptr0[1:] - ptr0[:-1])
        pos = torch.empty(n, dtype=torch.long); pos[idx0] = torch.arange(n) - ptr0[seg]
        x = -(((pos.unsqueeze(1) * 7 + torch.arange(C).unsqueeze(0) * 3) % 11) + 1).float() / 8.0
    xg = x.to(dtype).cuda().requires_grad_(True)
    y = SF.segment_minmax(xg, coarse, reduce == "max")
    idx = coarse.indices.long().cpu(); ptr = coarse.idx_ptr.cpu()[: coarse.n + 1]
    xo = x.clone().requires_grad_(True)
    yo = oops.segment_csr(xo[idx], ptr, reduce)
    assert torch.equal(y.float().cpu(), yo.detach().to(dtype).float())
    cot = torch.randn(coarse.n, C, generator=g).to(dtype)
    y.backward(cot.cuda()); yo.backward(cot.float())
    assert torch.equal(xg.grad.float().cpu(), xo.grad.to(dtype).float())


@pytest.mark.parametrize("C,xdt,tdt,hdt", [(32, torch.float32, torch.bfloat16, torch.bfloat16), (768, torch.float32, torch.bfloat16, torch.bfloat16),
                                           (260, torch.float32, torch.float32, torch.float32), (1024, torch.bfloat16, torch.bfloat16, torch.float32)])
def test_fused_ln_add_ln_against_torch(C, xdt, tdt, hdt):
    """x + LN0(t) -> LN1 in one kernel each way vs the torch composition in fp32 (same rounded inputs)."""
    from scenesplat_amd import functional as SF
    g = torch.Generator().manual_seed(C)
    n = 777
    x = torch.randn(n, C, generator=g).to(xdt); t = (torch.randn(n, C, generator=g) * 2 + 0.5).to(tdt)
    ln0, ln1 = torch.nn.LayerNorm(C, eps=1e-5), torch.nn.LayerNorm(C, eps=1e-5)
    for ln in (ln0, ln1):
        ln.weight.data = torch.randn(C, generator=g) * 0.5 + 1.0; ln.bias.data = torch.randn(C, generator=g) * 0.3
    cx, ch = torch.randn(n, C, generator=g), torch.randn(n, C, generator=g).to(hdt)
    # reference in fp32
    xr, tr = x.float().clone().requires_grad_(True), t.float().clone().requires_grad_(True)
    xo_r = xr + ln0(tr); h_r = ln1(xo_r)
    ((xo_r * cx).sum() + (h_r * ch.float()).sum()).backward()
    ref = [xo_r, h_r, xr.grad, tr.grad, ln0.weight.grad.clone(), ln0.bias.grad.clone(), ln1.weight.grad.clone(), ln1.bias.grad.clone()]
    for ln in (ln0, ln1):
        ln.zero_grad()
    l0, l1 = ln0.cuda(), ln1.cuda()
    xg, tg = x.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    xo, h = SF.ln_add_ln(xg, tg, l0, l1, hdt)
    assert xo.dtype == torch.float32 and h.dtype == hdt
    torch.autograd.backward([xo, h], [cx.cuda(), ch.cuda()])
    got = [xo, h, xg.grad, tg.grad, l0.weight.grad, l0.bias.grad, l1.weight.grad, l1.bias.grad]
    lo = hdt == torch.bfloat16 or xdt == torch.bfloat16 or tdt == torch.bfloat16
    for i, (a, b) in enumerate(zip(got, ref)):
        tol = 2e-2 if (lo and i in (1, 2, 3)) else 2e-4
        err = (a.float().cpu() - b.detach()).abs().max().item() / max(1.0, b.abs().max().item())
        assert err < tol, (i, err)


def test_window_attention_long_window_falls_back_to_simt_kernels():
    """Windows longer than the MFMA kernels' LDS index copy (2048 rows) run on the SIMT kernels even when the caller asks
    for the MFMA implementation: forward and backward must agree with an explicit SIMT call (bf16 in both)."""
    from scenesplat_amd import functional as SF, native as nv
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(9)
    n, K, H, d = 5200, 4096, 2, 16
    gc = torch.stack([torch.randperm(n, generator=g), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1)
    plan = build_plan(gc.cuda(), torch.tensor([n]).cuda(), ("z",), ())
    win = plan.levels[0].window(0, K)
    assert win.max_window == K
    qkv = torch.randn(n, 3 * H * d, generator=g).to(torch.bfloat16).cuda()
    cot = torch.randn(n, H * d, generator=g).to(torch.bfloat16).cuda()
    res = []
    for impl in (nv.ATTN_MFMA, nv.ATTN_SIMT):
        q = qkv.clone().requires_grad_(True)
        o = SF.window_attention(q, win, H, d ** -0.5, impl)
        o.backward(cot)
        res.append((o.detach().float(), q.grad.float()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("dup", [False, True])
@pytest.mark.parametrize("k", [3, 5])
def test_rulebook_hashed_equals_sorted_lookup(k, dup, monkeypatch):
    """ss_subm_rulebook_hashed (hash table) against ss_subm_rulebook (binary search in the sorted z keys): bit-identical
    tables, including duplicate voxels (lowest row wins) and two batch elements; the latter is pinned on the oracle."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(k + dup)
    gc = torch.randint(0, 20, (5000, 3), generator=g)
    if not dup:
        gc = torch.unique(gc, dim=0)
        gc = gc[torch.randperm(len(gc), generator=g)]
    n = len(gc)
    offs = torch.tensor([n // 3, n])
    tabs = []
    for hashed in (False, True):
        monkeypatch.setattr(nv, "RULEBOOK_HASHED", hashed)
        plan = build_plan(gc.cuda(), offs.cuda(), ORD, ())
        tabs.append(plan.levels[0].neighbors(k).cpu())
    assert torch.equal(tabs[0], tabs[1])
    batch = np.repeat([0, 1], [n // 3, n - n // 3])
    assert np.array_equal(tabs[1].numpy(), np.asarray(oops.neighbor_table(gc.numpy(), batch, k)).T)


def test_row_keep_scales_are_bernoulli_over_keep_and_follow_the_seed():
    """DropPath row scales (timm DropPath on (n, C) rows, ptv3:333-336): values in {0, 1/keep}, keep rate within 5 sigma per
    segment, same seed -> same masks, other seed -> other masks, neighbouring rows uncorrelated."""
    from scenesplat_amd import native as nv
    n = 400_003
    keep = torch.cat([torch.full((n // 2,), 0.9), torch.full((n - n // 2,), 0.7)]).cuda()
    s0 = torch.tensor([1234567], dtype=torch.int64, device="cuda")
    a = nv.row_keep_scales(keep, s0); b = nv.row_keep_scales(keep, s0)
    c = nv.row_keep_scales(keep, torch.tensor([7654321], dtype=torch.int64, device="cuda"))
    assert torch.equal(a, b) and not torch.equal(a, c)
    on = a > 0
    assert torch.allclose(a[on], (1.0 / keep)[on])
    for sl, k in ((slice(0, n // 2), 0.9), (slice(n // 2, n), 0.7)):
        m = sl.stop - sl.start
        rate = float(on[sl].float().mean())
        assert abs(rate - k) < 5 * (k * (1 - k) / m) ** 0.5, (rate, k)
    x = on[: n // 2].float() - 0.9
    corr = float((x[1:] * x[:-1]).mean() / (0.9 * 0.1))
    assert abs(corr) < 0.02, corr
    d = nv.row_keep_scales(keep)                       # seed from torch's generator: differs call to call
    e = nv.row_keep_scales(keep)
    assert not torch.equal(d, e)
    assert nv.row_keep_scales(keep[:0]).numel() == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gelu_kernels_match_the_exact_erf_form(dtype):
    """nn.GELU() of the MLP (ptv3:225-248): forward and backward of the HIP kernel pair against torch's fp32 CPU evaluation of the same
    operands (the oracle's activation), rounded once to the storage type: at most one unit in the last place of that type apart, and
    through the autograd Function incl. a ragged tail and extreme arguments."""
    from scenesplat_amd import functional as SF
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3001, 77, generator=g) * 3.0
    x.view(-1)[:8] = torch.tensor([0.0, -0.0, 1e-8, -1e-8, 12.0, -12.0, 40.0, -40.0])
    dy = torch.randn(3001, 77, generator=g)
    xs, dys = x.to(dtype), dy.to(dtype)
    xr = xs.float().clone().requires_grad_(True)
    yr = torch.nn.functional.gelu(xr)
    yr.backward(dys.float())
    xg = xs.detach().cuda().requires_grad_(True)
    y = SF.gelu(xg)
    assert type(y.grad_fn).__name__ == "_GeluBackward" and y.dtype == dtype
    y.backward(dys.cuda())
    # one unit in the last place of the storage type (relative 2^-7 for bf16); fp32: erff of the device library and the host's differ by
    # a few fp32 ulps of a value near 1, which is an ABSOLUTE error of ~1e-6 where 1 + erf cancels (x < -3)
    eps, atol = (2.0 ** -7, 1e-5) if dtype == torch.bfloat16 else (2.0 ** -18, 4e-6)
    for got, ref in ((y, yr), (xg.grad, xr.grad)):
        got, ref = got.detach().float().cpu(), ref.detach()
        assert torch.isfinite(got).all()
        err = (got - ref.to(dtype).float()).abs()
        worst = float((err - eps * ref.abs()).max())
        assert worst <= atol, worst
