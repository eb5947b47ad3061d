"""First-stage submanifold conv in exact fp32 on v_mfma_f32_32x32x2_f32 (csrc/subm_f32.hip) against the oracle's fp32 conv
(oracle/ops.py:subm_conv3d, restating spconv.SubMConv3d as the reference uses it in pointcept/models/modules.py:64-75 and
point_transformer_v3m1_base.py:271-289, 572-590).  fp32 products, fp32 accumulate: the only difference from the oracle is the
order of the sums, so the bar is fp32 rounding (1e-5 relative over the tensor), not a bf16 tolerance."""
import numpy as np
import pytest
import torch

from oracle import ops as oops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _level(n_side, seed=0, min_sites=0):
    from scenesplat_amd.plan import build_plan
    from scenesplat_amd.synthetic import room_chunk
    data = room_chunk(n_side, seed, lang_dim=0)
    lv = build_plan(data["grid_coord"].cuda(), data["offset"].cuda(), ("z", "z-trans", "hilbert", "hilbert-trans"), ()).levels[0]
    assert lv.n >= min_sites
    return data, lv


@pytest.mark.parametrize("cin,k,n_side", [(32, 3, 64), (11, 5, 64), (6, 5, 48), (16, 3, 48), (32, 3, 180)])
def test_fp32_mfma_conv_matches_the_oracle_forward_and_backward(cin, k, n_side):
    from scenesplat_amd import functional as SF
    data, lv = _level(n_side)
    n = lv.n
    g = torch.Generator().manual_seed(100 + cin + k)
    x = torch.randn(n, cin, generator=g); w = torch.randn(32, k, k, k, cin, generator=g) * (k ** 3 * cin / 3) ** -0.5
    b = torch.randn(32, generator=g) * 0.02; cot = torch.randn(n, 32, generator=g)
    need_dx = cin == 32
    xo, wo, bo = x.clone().requires_grad_(need_dx), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    nbro = oops.neighbor_table(data["grid_coord"].numpy(), np.zeros(n, np.int64), k)
    yo = oops.subm_conv3d(xo, wo, bo, nbro)
    (yo * cot).sum().backward()
    xg, wg, bg = x.cuda().requires_grad_(need_dx), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    walk_fn = (lambda: lv.neighbors_walk(k)) if n_side != 48 else None          # None: the Function derives it itself
    y = SF.subm_conv3d(xg, wg, bg, lv.neighbors(k), False, "bf16x3", lv.conv_rowperm(), lambda: lv.conv_blocks(k), walk_fn)
    assert type(y.grad_fn).__name__ == "_SubMConv3dF32Backward", type(y.grad_fn).__name__
    (y * cot.cuda()).sum().backward()
    err = dict(y=_rel(y, yo), dw=_rel(wg.grad, wo.grad), db=_rel(bg.grad, bo.grad))
    if need_dx:
        err["dx"] = _rel(xg.grad, xo.grad)
    print("fp32-MFMA conv cin=%d k=%d n=%d: %s" % (cin, k, n, " ".join("%s %.1e" % kv for kv in err.items())))
    assert max(err.values()) < 1e-5, err
    # per-row check of the forward: no row may be off by more than fp32 summation noise of its own magnitude
    d = (y.detach().cpu() - yo.detach()).abs().amax(1) / yo.detach().abs().amax(1).clamp_min(1e-3)
    assert float(d.max()) < 2e-5


def test_fp32_mfma_conv_identity_order_and_ragged_tail():
    """rowperm = None (identity walk) and n not a multiple of 32 / 64 / 128: the tail waves and partial blocks."""
    from scenesplat_amd import native as nv
    data, lv = _level(40)
    n = (lv.n // 128) * 128 - 51
    assert n > 1000 and n % 32 and n % 64
    # a rulebook over the first n sites only (neighbours beyond n removed)
    nbr_full = lv.neighbors(3)
    nbr = nbr_full[:, :n].clone(); nbr[nbr >= n] = -1
    nbr = nbr.contiguous()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 32, generator=g); w = torch.randn(32, 27, 32, generator=g) * 0.1; cot = torch.randn(n, 32, generator=g)
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yo = oops.subm_conv3d(xo, wo.reshape(32, 3, 3, 3, 32), None, nbr.t().cpu().numpy())
    (yo * cot).sum().backward()
    xc, gc = x.cuda(), cot.cuda()
    assert nv.subm_walk_rulebook(nbr, None) is nbr
    y = nv.subm_f32_fwd(xc, nv.subm_f32_weight_layout(w.cuda()), None, nbr, None)
    dx = nv.subm_f32_fwd(gc, nv.subm_f32_weight_layout(w.cuda(), mirror=True), None, nbr, None)
    dw = nv.subm_f32_wgrad(xc, gc, nbr, None, nv.subm_block_lists(nbr, None), 32)
    assert _rel(y, yo) < 1e-5 and _rel(dx, xo.grad) < 1e-5 and _rel(dw, wo.grad) < 1e-5


def test_fp32_mfma_conv_is_bit_reproducible_in_the_forward():
    """The forward has no atomics: two launches give identical bits (the weight gradient accumulates with fp32 atomics and is
    reproducible only to rounding)."""
    from scenesplat_amd import native as nv
    data, lv = _level(48)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(lv.n, 32, generator=g).cuda(); wq = nv.subm_f32_weight_layout((torch.randn(32, 27, 32, generator=g) * 0.1).cuda())
    assert torch.equal(lv.neighbors_walk(3), lv.neighbors(3)[:, lv.conv_rowperm().long()])
    a = nv.subm_f32_fwd(x, wq, None, lv.neighbors_walk(3), lv.conv_rowperm())
    b = nv.subm_f32_fwd(x, wq, None, lv.neighbors_walk(3), lv.conv_rowperm())
    assert torch.equal(a, b)


def test_fp32_mfma_conv_refuses_what_it_does_not_cover():
    from scenesplat_amd import native as nv
    lib = nv.lib
    assert lib().ss_subm_f32_ok(32, 32) == 1 and lib().ss_subm_f32_ok(16, 32) == 1
    assert lib().ss_subm_f32_ok(32, 64) == 0 and lib().ss_subm_f32_ok(24, 32) == 0
    with pytest.raises(RuntimeError):
        nv.subm_f32_weight_layout(torch.zeros(64, 27, 32, device="cuda"))
