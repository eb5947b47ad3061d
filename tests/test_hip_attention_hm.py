"""GPU parity of the head-major window attention (round 3; csrc/attention_hm.hip, gemm8.hip's head-major epilogue):
the LDS-DMA kernels against the oracle's fp32 attention (ptv3:190-206, pinned to the reference by tests/test_oracle.py) on the
same bf16-rounded operands, against the round-2 kernels, and the fused projection + attention function against autocast
PyTorch.  Shapes: the dec0 shape (16 heads x 48, K = 1024) with a borrowed tail, short windows, every head width."""
import importlib.util
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as oops
from oracle import ptv3 as optv3

pytestmark = pytest.mark.gpu


def _prod():
    spec = importlib.util.spec_from_file_location("prod_inputs", os.path.join(os.path.dirname(__file__), "golden", "prod_inputs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _cosd(a, b):
    return 1 - F.cosine_similarity(a.double().cpu(), b.double().cpu(), dim=1)


def _dequantised(qkv16, win, H, scale):
    """(hm, qkv_eff): the head-major operand packed from a bf16 (n, 3C) projection, and the (n, 3C) fp32 projection the
    kernels effectively see -- k, v unchanged, q = the packed q~ / (scale * log2 e).  (Packing from bf16 rounds q a second
    time; the production path writes q~ from the projection's fp32 accumulators, one rounding.)"""
    from scenesplat_amd import native as nv
    hm = nv.headmajor_pack(qkv16.cuda().contiguous(), win, H, 3, scale * nv.LOG2E)
    n, C3 = qkv16.shape
    C = C3 // 3
    sidx = win.sidx.cpu().long()
    canon = (sidx >= 0).nonzero().squeeze(1)
    qt = hm[0].float().cpu().permute(1, 0, 2).reshape(win.n_pad, C)          # (slot, H * D)
    eff = qkv16.float().clone()
    eff[sidx[canon], :C] = qt[canon] / (scale * nv.LOG2E)
    return hm, eff


def _oracle(qkv, gc, counts, orders, oi, K, H, scale, dout):
    lvo = optv3.build_levels(gc.numpy(), np.asarray(counts).cumsum(), orders, ())[0]
    pad, unpad, cu = lvo.padding(K)
    order = torch.as_tensor(lvo.order[oi][pad])
    inverse = torch.as_tensor(unpad[lvo.inverse[oi]])
    q = qkv.float().clone().requires_grad_(True)
    ref = oops.window_attention(q[order], cu, H, scale)[inverse]
    (ref * dout.float()).sum().backward()
    return ref.detach(), q.grad


def test_headmajor_attention_d48_k1024_against_oracle():
    """k_attn_hm_fwd / _dq / _dkv <48> against the oracle on the SAME bf16-rounded qkv (two full windows + a tail topped up with
    472 borrowed points): outputs, dq, dk, dv incl. the borrowed-slot fix-up."""
    from scenesplat_amd import functional as SF
    from scenesplat_amd.plan import build_plan
    mp = _prod()
    gc, x, cot, sd = mp.att_inputs()
    n, K, H, C = mp.ATT["n"], mp.ATT["K"], mp.ATT["H"], mp.ATT["C"]
    plan = build_plan(gc.cuda(), torch.tensor([n]).cuda(), mp.ORD, ())
    win = plan.levels[0].window(mp.ATT["order_index"], K)
    assert win.num_windows == 3 and win.n_pad == 3072
    qkv16 = F.linear(x, sd["qkv.weight"], sd["qkv.bias"]).to(torch.bfloat16)
    dout16 = cot.to(torch.bfloat16)
    scale = (C // H) ** -0.5
    q = qkv16.cuda().requires_grad_(True)
    out = SF.window_attention_hm(q, win, H, scale)
    out.backward(dout16.cuda())
    _, eff = _dequantised(qkv16, win, H, scale)
    ref, gref = _oracle(eff, gc, [n], mp.ORD, mp.ATT["order_index"], K, H, scale, dout16)
    cd = _cosd(out.float(), ref)
    print("hm attn fwd d48/K1024 vs oracle: max cosine distance %.2e, rel %.2e" % (cd.max(), _rel(out.float(), ref)))
    assert cd.max() < 2e-5 and _rel(out.float(), ref) < 6e-3
    for nm, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        r = _rel(q.grad[:, sl].float(), gref[:, sl])
        print("hm attn bwd %s rel err %.2e" % (nm, r))
        assert r < 8e-3, (nm, r)


@pytest.mark.parametrize("C,H", [(768, 16), (64, 4), (128, 4), (256, 4)])
def test_headmajor_forward_rescale_branch_forced(C, H):
    """The forward moves its running shift lazily (past 2^6 only) and starts the score accumulators from the negated shift:
    force the rare branch late in a window and in the borrowed tail (scores that exceed everything before them), full-tensor
    comparison with the oracle (cdna guide rule 26)."""
    from scenesplat_amd import functional as SF
    from scenesplat_amd.plan import build_plan
    mp = _prod()
    gc, _, _, _ = mp.att_inputs()
    n, K = mp.ATT["n"], mp.ATT["K"]
    d = C // H
    g = torch.Generator().manual_seed(C)
    qkv = torch.randn(n, 3 * C, generator=g) * 0.5
    lvo = optv3.build_levels(gc.numpy(), np.array([n]), mp.ORD, ())[0]
    pad, unpad, cu = lvo.padding(K)
    order = torch.as_tensor(lvo.order[1][pad])
    for kslot, qslots, hd in ((700, (5, 40, 333, 1000), 1), (1023, (5, 77), 1), (2048 + 900, (2048 + 17,), 0)):
        krow = int(order[kslot])
        for qs in qslots:
            qrow = int(order[qs])
            qv = qkv[qrow, hd * d:(hd + 1) * d]
            qkv[krow, C + hd * d:C + (hd + 1) * d] += 6.0 * qv * (d ** 0.5) / qv.norm()
    qkv16 = qkv.to(torch.bfloat16)
    scale = d ** -0.5
    plan = build_plan(gc.cuda(), torch.tensor([n]).cuda(), mp.ORD, ())
    win = plan.levels[0].window(1, K)
    from scenesplat_amd import native as nv
    hm, eff = _dequantised(qkv16, win, H, scale)
    out, nlse2 = nv.window_attn_hm_fwd(hm, win, H)
    inverse = torch.as_tensor(unpad[lvo.inverse[1]])
    ref = oops.window_attention(eff[order], cu, H, scale)[inverse]
    cd = _cosd(out.float(), ref)
    err = (out.float().cpu() - ref).abs().max()
    print("hm forced rescale d=%d: max cosine distance %.2e, max abs err %.2e" % (d, cd.max(), err))
    # -log2 sum exp2 against the oracle's log-sum-exp on window 0
    q = eff[order[:K]].reshape(K, 3, H, d)
    sc = torch.einsum("qhd,khd->hqk", q[:, 0], q[:, 1]) * scale
    ref_lse = torch.logsumexp(sc, dim=-1)                                    # (H, K)
    assert torch.allclose(-nlse2[:, :K].cpu() * 0.6931471805599453, ref_lse, atol=2e-3, rtol=1e-4)
    assert cd.max() < 3e-5 and err < 3e-2


@pytest.mark.parametrize("L", [1, 5, 31, 63, 64, 65, 129, 200])
def test_headmajor_short_windows(L):
    """An element with <= K points is one short window (varlen semantics, ptv3:135-136): every tail length of the 128-key
    ring slots, forward and backward."""
    from scenesplat_amd import functional as SF
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(L)
    n = L + 300
    gc = torch.stack([torch.randperm(n, generator=g), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1)
    counts = [L, 300]
    plan = build_plan(gc.cuda(), torch.tensor(counts).cumsum(0).cuda(), ("hilbert",), ())
    win = plan.levels[0].window(0, 256)
    H, d = 2, 16
    qkv = torch.randn(n, 3 * H * d, generator=g).to(torch.bfloat16)
    dout = torch.randn(n, H * d, generator=g).to(torch.bfloat16)
    q = qkv.cuda().requires_grad_(True)
    out = SF.window_attention_hm(q, win, H, d ** -0.5)
    out.backward(dout.cuda())
    _, eff = _dequantised(qkv, win, H, d ** -0.5)
    ref, gref = _oracle(eff, gc, counts, ("hilbert",), 0, 256, H, d ** -0.5, dout)
    assert torch.allclose(out.float().cpu(), ref, atol=2e-2, rtol=2e-2)
    assert _rel(q.grad.float(), gref) < 1.5e-2


@pytest.mark.parametrize("H,d,K,counts", [(4, 48, 1024, [2500]), (2, 16, 1024, [1100, 900]), (2, 32, 256, [700, 300, 40]),
                                           (1, 64, 128, [333])])
def test_headmajor_matches_round2_kernels(H, d, K, counts):
    """Head-major kernels against the round-2 MFMA kernels and the fp32-math SIMT kernels on identical bf16 inputs (padded
    tails, borrowed slots, several batch elements)."""
    from scenesplat_amd import functional as SF, native as nv
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
    g_s = nv.window_attn_bwd(qkv, o_s, dout, lse_s, win, H, scale, nv.ATTN_SIMT).float()
    q = qkv.clone().requires_grad_(True)
    o_h = SF.window_attention_hm(q, win, H, scale)
    o_h.backward(dout)
    assert (o_h.float() - o_s.float()).abs().max() < 3e-2
    for name, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        a, b = q.grad[:, sl].float(), g_s[:, sl]
        rel = (a - b).norm() / b.norm()
        assert rel < 1.5e-2, (name, rel.item())
        assert (a - b).abs().max() < 0.05 * b.abs().max() + 1e-2, name


@pytest.mark.parametrize("n,C,H,K", [(2600, 768, 16, 1024), (5000, 256, 16, 1024), (1500, 64, 4, 256), (900, 32, 2, 128)])
def test_fused_projection_attention_matches_autocast_reference(n, C, H, K):
    """SF.qkv_window_attention (the projection writing head-major q / k / v from its own epilogue -- or, for narrow levels, the
    fp32 library GEMM + pack kernel -- then the LDS-DMA attention) against F.linear + the oracle attention in fp32: output,
    dx, dW, db."""
    from scenesplat_amd import functional as SF
    from scenesplat_amd.plan import build_plan
    g = torch.Generator().manual_seed(n + C)
    gc = torch.stack([torch.randperm(n, generator=g), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1)
    plan = build_plan(gc.cuda(), torch.tensor([n]).cuda(), ("z", "hilbert"), ())
    win = plan.levels[0].window(1, K)
    x = torch.randn(n, C, generator=g).to(torch.bfloat16)
    w = (torch.randn(3 * C, C, generator=g) * C ** -0.5)
    b = torch.randn(3 * C, generator=g) * 0.1
    dout = torch.randn(n, C, generator=g).to(torch.bfloat16)
    scale = (C // H) ** -0.5
    xg = x.cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = SF.qkv_window_attention(xg, wg, bg, win, H, scale)
    out.backward(dout.cuda())
    # reference: bf16-rounded operands, fp32 math
    xr = x.float().requires_grad_(True)
    wr = w.to(torch.bfloat16).float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    qkv = F.linear(xr, wr, br)
    lvo = optv3.build_levels(gc.numpy(), np.array([n]), ("z", "hilbert"), ())[0]
    pad, unpad, cu = lvo.padding(K)
    order = torch.as_tensor(lvo.order[1][pad])
    inverse = torch.as_tensor(unpad[lvo.inverse[1]])
    ref = oops.window_attention(qkv[order], cu, H, scale)[inverse]
    (ref * dout.float()).sum().backward()
    cd = _cosd(out.float(), ref.detach())
    print("fused qkv+attn C=%d: max cosine distance %.2e; dx %.2e dW %.2e db %.2e" % (
        C, cd.max(), _rel(xg.grad.float(), xr.grad), _rel(wg.grad, wr.grad), _rel(bg.grad, br.grad)))
    assert cd.max() < 3e-5
    assert _rel(xg.grad.float(), xr.grad) < 1.5e-2
    assert _rel(wg.grad, wr.grad) < 1.5e-2
    assert _rel(bg.grad, br.grad) < 1.5e-2


def test_headmajor_projection_epilogue_is_the_pack_of_the_fp32_projection():
    """ss_linear_fwd_headmajor == ss_headmajor_pack(fp32 x @ W.T + b) bit for bit up to the GEMMs' accumulation order (one bf16
    rounding either way), at the dec0 width with a borrowed tail."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.plan import build_plan
    n, C, H, K = 3000, 768, 16, 1024
    g = torch.Generator().manual_seed(7)
    gc = torch.stack([torch.randperm(n, generator=g), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1)
    plan = build_plan(gc.cuda(), torch.tensor([n]).cuda(), ("z",), ())
    win = plan.levels[0].window(0, K)
    x = torch.randn(n, C, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(torch.bfloat16).cuda()
    b = (torch.randn(3 * C, generator=g) * 0.1).cuda()
    s0 = (C // H) ** -0.5 * nv.LOG2E
    hm = nv.linear_fwd_headmajor(x, win, w, b, H, s0)
    ref = nv.headmajor_pack((x.float() @ w.float().t() + b).contiguous(), win, H, 3, s0)
    assert hm.shape == (3, H, win.n_pad, C // H)
    diff = (hm.float() - ref.float()).abs()
    assert diff.max() <= 2 ** -6 * ref.float().abs().max()          # at most one bf16 ulp apart
    assert (diff > 0).float().mean() < 0.05


def test_headmajor_kernels_race_screen_bitwise_repeats():
    """The three kernels keep DMA pieces in flight across barriers (3-slot ring, counted vmcnt): a read placed one barrier too early
    passes every reference check whenever the DMA happens to land first.  Screen: the benchmark's dec0 shape (100 windows x 16 heads,
    K = 1024) and a ragged one (short windows, borrowed tail), 12 back-to-back repeats each on a busy chip: forward outputs, -lse and all
    three gradients must be BIT-IDENTICAL every time (the kernels have no atomics: any difference is a race)."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.plan import build_plan
    from scenesplat_amd.synthetic import room_chunk
    g = torch.Generator(device="cuda").manual_seed(3)
    cases = []
    d = room_chunk(256, 0, lang_dim=0)
    plan = build_plan(d["grid_coord"].cuda(), d["offset"].cuda(), ("z", "hilbert"), ())
    cases.append((plan.levels[0].window(1, 1024), 768, 16))
    n = 5000
    gc = torch.stack([torch.randperm(n), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1)
    plan2 = build_plan(gc.cuda(), torch.tensor([130, 1900, 5000]).cuda(), ("z",), ())
    cases.append((plan2.levels[0].window(0, 1024), 128, 4))
    for win, C, H in cases:
        qkv = torch.randn(win.n, 3 * C, device="cuda", generator=g).to(torch.bfloat16)
        dout = torch.randn(win.n, C, device="cuda", generator=g).to(torch.bfloat16)
        sc = (C // H) ** -0.5
        hm = nv.headmajor_pack(qkv, win, H, 3, sc * nv.LOG2E)
        ref = None
        for it in range(12):
            out, nl = nv.window_attn_hm_fwd(hm, win, H)
            dq = nv.window_attn_hm_bwd(hm, out, dout, nl, win, H, sc)
            cur = (out.clone(), nl.clone(), dq.clone())
            if ref is None:
                ref = cur
                assert torch.isfinite(out.float()).all() and torch.isfinite(dq.float()).all()
            else:
                for a, b, nm in zip(ref, cur, ("out", "neg_lse2", "dqkv")):
                    assert torch.equal(a, b), (C, it, nm)
