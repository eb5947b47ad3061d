"""GPU parity: libs/pointops, pointops2, pointgroup_ops replacements vs the numpy oracle
("parity unpinned": the reference has no fixtures for these; see oracle/pointops.py)."""
import numpy as np
import pytest
import torch

from oracle import pointops as opo

pytestmark = pytest.mark.gpu


def data(seed=0, counts=(300, 500, 41)):
    g = np.random.default_rng(seed)
    n = sum(counts)
    xyz = g.random((n, 3), dtype=np.float32)
    offset = np.cumsum(counts)
    return xyz, offset


def cu(a, dt=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).cuda()
    return t.to(dt) if dt is not None else t


@pytest.mark.parametrize("k", [1, 3, 16, 40])
def test_knn_query(k):
    from scenesplat_amd import pointops as po
    xyz, off = data()
    new_xyz, noff = data(1, (120, 80, 30))
    idx, dist = po.knn_query(k, cu(xyz), cu(off), cu(new_xyz), cu(noff))
    ridx, rd2 = opo.knn_query(k, xyz, off, new_xyz, noff)
    assert np.allclose(dist.cpu().numpy() ** 2, rd2, rtol=1e-4, atol=1e-7)
    assert np.array_equal(idx.cpu().numpy(), ridx)        # continuous random coords: no distance ties
    # self query, padding when the segment is smaller than k
    idx2, d2 = po.knn_query(50, cu(xyz), cu(off))
    r2, rd = opo.knn_query(50, xyz, off)
    assert np.array_equal(idx2.cpu().numpy(), r2)
    assert (idx2[-41:, 41:] == -1).all()


def test_ball_and_random_ball_query():
    from scenesplat_amd import pointops as po
    xyz, off = data(2)
    idx, dist = po.ball_query(16, 0.25, 0.05, cu(xyz), cu(off))
    ridx, rd2 = opo.ball_query(16, 0.25, 0.05, xyz, off)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.allclose(dist.cpu().numpy() ** 2, rd2, rtol=1e-4)
    order = np.concatenate([np.random.default_rng(3).permutation(np.arange(s, e)) for s, e in zip([0] + off[:-1].tolist(), off)])
    idx, dist = po.random_ball_query(8, 0.3, 0.0, cu(xyz), cu(off), order=cu(order, torch.int32))
    ridx, rd2 = opo.random_ball_query(8, 0.3, 0.0, order, xyz, off)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.allclose(dist.cpu().numpy() ** 2, rd2, rtol=1e-4)


def test_farthest_point_sampling():
    from scenesplat_amd import pointops as po
    xyz, off = data(4, (1500, 700, 9))
    noff = np.cumsum([100, 33, 9])
    idx = po.farthest_point_sampling(cu(xyz), cu(off), cu(noff))
    assert np.array_equal(idx.cpu().numpy(), opo.farthest_point_sampling(xyz, off, noff))


def test_grouping_subtraction_aggregation_interpolation():
    from scenesplat_amd import pointops as po
    g = np.random.default_rng(5)
    n, m, ns, c, wc = 200, 150, 8, 12, 4
    feat = g.standard_normal((n, c)).astype(np.float32)
    idx = g.integers(-1, n, (m, ns))
    f = cu(feat).requires_grad_(True)
    out = po.grouping2(f, cu(idx, torch.int32))
    assert np.allclose(out.detach().cpu().numpy(), opo.grouping(feat, idx), atol=1e-6)
    cot = g.standard_normal((m, ns, c)).astype(np.float32)
    (out * cu(cot)).sum().backward()
    ref = np.zeros_like(feat); np.add.at(ref, idx[idx >= 0], cot[idx >= 0])
    assert np.allclose(f.grad.cpu().numpy(), ref, atol=1e-4)
    # subtraction
    idx = g.integers(0, n, (n, ns))
    a, b = cu(feat).requires_grad_(True), cu(feat[::-1].copy()).requires_grad_(True)
    out = po.subtraction(a, b, cu(idx, torch.int32))
    assert np.allclose(out.detach().cpu().numpy(), opo.subtraction(feat, feat[::-1], idx), atol=1e-6)
    cot = g.standard_normal((n, ns, c)).astype(np.float32)
    (out * cu(cot)).sum().backward()
    assert np.allclose(a.grad.cpu().numpy(), cot.sum(1), atol=1e-4)
    ref = np.zeros_like(feat); np.add.at(ref, idx, -cot)
    assert np.allclose(b.grad.cpu().numpy(), ref, atol=1e-4)
    # aggregation (torch autograd of the same formula as reference for the backward)
    pos = g.standard_normal((n, ns, c)).astype(np.float32); w = g.standard_normal((n, ns, wc)).astype(np.float32)
    ti, tp, tw = cu(feat).requires_grad_(True), cu(pos).requires_grad_(True), cu(w).requires_grad_(True)
    out = po.aggregation(ti, tp, tw, cu(idx, torch.int32))
    assert np.allclose(out.detach().cpu().numpy(), opo.aggregation(feat, pos, w, idx), atol=1e-4)
    cot = g.standard_normal((n, c)).astype(np.float32)
    (out * cu(cot)).sum().backward()
    ri, rp, rw = (torch.tensor(x, requires_grad=True) for x in (feat, pos, w))
    ro = ((ri[torch.as_tensor(idx)] + rp) * rw[:, :, torch.arange(c) % wc]).sum(1)
    (ro * torch.tensor(cot)).sum().backward()
    for a_, b_ in ((ti, ri), (tp, rp), (tw, rw)):
        assert torch.allclose(a_.grad.cpu(), b_.grad, atol=1e-4)
    # interpolation
    xyz, off = data(6, (120, 80)); new_xyz, noff = data(7, (60, 40))
    fin = g.standard_normal((200, c)).astype(np.float32)
    tf = cu(fin).requires_grad_(True)
    out = po.interpolation(cu(xyz), cu(new_xyz), tf, cu(off), cu(noff), 3)
    ridx, rd2 = opo.knn_query(3, xyz, off, new_xyz, noff)
    wgt = opo.interpolation_weights(np.sqrt(rd2))
    assert np.allclose(out.detach().cpu().numpy(), opo.interpolation(fin, ridx, wgt), atol=1e-4)
    out.sum().backward()
    ref = np.zeros_like(fin); np.add.at(ref, ridx, np.repeat(wgt[:, :, None], c, 2))
    assert np.allclose(tf.grad.cpu().numpy(), ref, atol=1e-4)


def test_attention_relation_fusion_and_pointops2_steps():
    from scenesplat_amd import pointops as po
    g = np.random.default_rng(8)
    n, m, h, c = 90, 400, 3, 8
    q, k, v = (g.standard_normal((n, h, c)).astype(np.float32) for _ in range(3))
    w = g.standard_normal(c).astype(np.float32)
    it, ir = g.integers(0, n, m), g.integers(0, n, m)
    tq, tk, tw = cu(q).requires_grad_(True), cu(k).requires_grad_(True), cu(w).requires_grad_(True)
    out = po.attention_relation_step(tq, tk, tw, cu(it, torch.int32), cu(ir, torch.int32))
    assert np.allclose(out.detach().cpu().numpy(), opo.attention_relation(q, k, w, it, ir), atol=1e-4)
    cot = g.standard_normal((m, h)).astype(np.float32)
    (out * cu(cot)).sum().backward()
    rq, rk, rw = (torch.tensor(x, requires_grad=True) for x in (q, k, w))
    ((rq[it] * rk[ir] * rw).sum(-1) * torch.tensor(cot)).sum().backward()
    for a_, b_ in ((tq, rq), (tk, rk), (tw, rw)):
        assert torch.allclose(a_.grad.cpu(), b_.grad, atol=2e-4)
    # step1 (no channel weight) and step2 / fusion
    out1 = po.attention_step1(cu(q), cu(k), cu(it, torch.int32), cu(ir, torch.int32))
    assert np.allclose(out1.cpu().numpy(), opo.attention_relation(q, k, None, it, ir), atol=1e-4)
    attn = g.standard_normal((m, h)).astype(np.float32)
    ta, tv = cu(attn).requires_grad_(True), cu(v).requires_grad_(True)
    out2 = po.attention_step2(ta, tv, cu(it, torch.int32), cu(ir, torch.int32))
    assert np.allclose(out2.detach().cpu().numpy(), opo.attention_fusion(attn, v, it, ir, n), atol=1e-4)
    cot = g.standard_normal((n, h, c)).astype(np.float32)
    (out2 * cu(cot)).sum().backward()
    ra, rv = torch.tensor(attn, requires_grad=True), torch.tensor(v, requires_grad=True)
    ro = torch.zeros(n, h, c).index_add(0, torch.as_tensor(it), ra[:, :, None] * rv[ir])
    (ro * torch.tensor(cot)).sum().backward()
    assert torch.allclose(ta.grad.cpu(), ra.grad, atol=2e-4) and torch.allclose(tv.grad.cpu(), rv.grad, atol=2e-4)
    # relative position encoding
    L = 7
    table = g.standard_normal((L, h, c, 3)).astype(np.float32)
    rel = g.integers(0, L, (m, 3))
    tq, tt = cu(q).requires_grad_(True), cu(table).requires_grad_(True)
    o = po.dot_prod_with_idx(tq, cu(it, torch.int32), tt, cu(rel, torch.int32))
    assert np.allclose(o.detach().cpu().numpy(), opo.rpe_dot_prod(q, it, table, rel), atol=1e-4)
    o.sum().backward()
    rq, rt = torch.tensor(q, requires_grad=True), torch.tensor(table, requires_grad=True)
    sum((rq[it] * rt[rel[:, d], :, :, d]).sum() for d in range(3)).backward()
    assert torch.allclose(tq.grad.cpu(), rq.grad, atol=2e-4) and torch.allclose(tt.grad.cpu(), rt.grad, atol=2e-4)
    ta, tv, tt = cu(attn).requires_grad_(True), cu(v).requires_grad_(True), cu(table).requires_grad_(True)
    o = po.attention_step2_with_rel_pos_value(ta, tv, cu(it, torch.int32), cu(ir, torch.int32), tt, cu(rel, torch.int32))
    assert np.allclose(o.detach().cpu().numpy(), opo.rpe_attn_step2(attn, v, it, ir, table, rel, n), atol=1e-4)
    (o * cu(cot)).sum().backward()
    ra, rv, rt = (torch.tensor(x, requires_grad=True) for x in (attn, v, table))
    tsum = sum(rt[rel[:, d], :, :, d] for d in range(3))
    ro = torch.zeros(n, h, c).index_add(0, torch.as_tensor(it), ra[:, :, None] * (rv[ir] + tsum))
    (ro * torch.tensor(cot)).sum().backward()
    for a_, b_ in ((ta, ra), (tv, rv), (tt, rt)):
        assert torch.allclose(a_.grad.cpu(), b_.grad, atol=3e-4)


def test_pointgroup_ballquery_and_bfs_cluster():
    from scenesplat_amd import pointops as po
    xyz, off = data(9, (400, 250))
    bidx = np.repeat([0, 1], [400, 250]); boff = np.array([0, 400, 650])
    idx, start_len = po.ballquery_batch_p(cu(xyz), cu(bidx, torch.int32), cu(boff, torch.int32), 0.12, 2)   # forces one regrow
    ridx, rsl = opo.ballquery_batch_p(xyz, bidx, boff, 0.12)
    assert np.array_equal(start_len.cpu().numpy(), rsl) and np.array_equal(idx.cpu().numpy(), ridx)
    labels = (xyz[:, 0] > 0.5).astype(np.int32)
    cidx, coff = po.bfs_cluster(torch.as_tensor(labels), idx.cpu(), start_len.cpu(), 10)
    ri, ro = opo.bfs_cluster(labels, ridx, rsl, 10)
    assert np.array_equal(cidx.numpy(), ri) and np.array_equal(coff.numpy(), ro)


def test_neighbor_voting_and_majority_vote():
    from scenesplat_amd import pointops as po
    g = np.random.default_rng(11)
    n, k, C = 3000, 25, 20
    xyz = g.random((n, 3), dtype=np.float32)
    lab = g.integers(0, C, n).astype(np.int32)
    lab[g.random(n) < 0.1] = -1
    valid = g.random(n) < 0.8
    out = po.neighbor_voting(cu(xyz), cu(lab), cu(valid), k, -1, C).cpu().numpy()
    vi = np.nonzero(valid)[0]
    d = ((xyz[:, None, :] - xyz[None, vi, :]) ** 2).sum(-1)
    nn = np.argsort(d, 1, kind="stable")[:, :k]
    nl = lab[vi][nn]
    ref = np.full(n, -1, np.int32)
    for i in range(n):          # pointcept/utils/misc.py:33-48
        cnt = np.bincount(nl[i][nl[i] >= 0], minlength=C)
        if cnt.max() > 0:
            ref[i] = int(np.argmax(cnt))
    assert (out == ref).mean() > 0.999        # fp distance ties at the k-th neighbour may differ


def test_gpu_grid_sample_train_semantics():
    from scenesplat_amd.gpu_transforms import grid_sample_train
    g = torch.Generator().manual_seed(0)
    coord = torch.rand(50000, 3, generator=g) * torch.tensor([4.0, 3.0, 1.0])
    gs = 0.05
    res = grid_sample_train(coord.cuda(), gs, return_inverse=True)
    gc_all = np.floor(coord.numpy() / gs).astype(np.int64)
    gc_all -= gc_all.min(0)
    uniq = np.unique(gc_all, axis=0)
    sel = res["idx_unique"].cpu().numpy()
    assert len(sel) == len(uniq)                                            # one point per occupied voxel
    assert np.array_equal(np.unique(gc_all[sel], axis=0), uniq)             # every voxel represented exactly once
    assert np.array_equal(res["grid_coord"].cpu().numpy(), gc_all[sel])
    inv = res["inverse"].cpu().numpy()
    assert np.array_equal(gc_all[sel][inv], gc_all)                         # inverse maps each point to its voxel's row
    res2 = grid_sample_train(coord.cuda(), gs)
    assert not np.array_equal(sel, res2["idx_unique"].cpu().numpy())         # random representative


def test_pointops2_v2_v3_call_forms():
    """pointops2's _v2 / _v3 call forms (CSR query offsets, fused q/k table dot products) against the plain-torch
    statement of the same quantities (libs/pointops2/functions/pointops.py:166-957)."""
    from scenesplat_amd import pointops as po
    g = np.random.default_rng(12)
    n, h, c, L = 60, 2, 8, 5
    counts = g.integers(0, 9, n)                       # pairs per query, sorted by query (some queries have none)
    m = int(counts.sum())
    i0 = np.repeat(np.arange(n), counts)
    offs = np.concatenate([[0], np.cumsum(counts)])
    i1 = g.integers(0, n, m)
    q, k, v = (g.standard_normal((n, h, c)).astype(np.float32) for _ in range(3))
    tq_, tk_ = (g.standard_normal((L, h, c, 3)).astype(np.float32) for _ in range(2))
    rel = g.integers(0, L, (m, 3))
    attn = g.standard_normal((m, h)).astype(np.float32)
    n_max = int(counts.max())
    o1 = po.attention_step1_v2(cu(q), cu(k), cu(i1, torch.int32), cu(offs, torch.int32), n_max)
    assert np.allclose(o1.cpu().numpy(), opo.attention_relation(q, k, None, i0, i1), atol=1e-4)
    o2 = po.attention_step2_v2(cu(attn), cu(v), cu(i0, torch.int32), cu(i1, torch.int32))
    assert np.allclose(o2.cpu().numpy(), opo.attention_fusion(attn, v, i0, i1, n), atol=1e-4)
    ref = opo.rpe_dot_prod(q, i0, tq_, rel) + opo.rpe_dot_prod(k, i1, tk_, rel)
    tq, tk, ttq, ttk = (cu(x).requires_grad_(True) for x in (q, k, tq_, tk_))
    o3 = po.dot_prod_with_idx_v2(tq, cu(i0, torch.int32), tk, cu(i1, torch.int32), ttq, ttk, cu(rel, torch.int32))
    assert np.allclose(o3.detach().cpu().numpy(), ref, atol=1e-4)
    o4 = po.dot_prod_with_idx_v3(cu(q), cu(offs, torch.int32), n_max, cu(k), cu(i1, torch.int32), cu(tq_), cu(tk_), cu(rel, torch.int32))
    assert np.allclose(o4.cpu().numpy(), ref, atol=1e-4)
    cot = g.standard_normal((m, h)).astype(np.float32)
    (o3 * cu(cot)).sum().backward()
    rq, rk, rtq, rtk = (torch.tensor(x, requires_grad=True) for x in (q, k, tq_, tk_))
    r = sum((rq[i0] * rtq[rel[:, d], :, :, d]).sum(-1) + (rk[i1] * rtk[rel[:, d], :, :, d]).sum(-1) for d in range(3))
    (r * torch.tensor(cot)).sum().backward()
    for a_, b_ in ((tq, rq), (tk, rk), (ttq, rtq), (ttk, rtk)):
        assert torch.allclose(a_.grad.cpu(), b_.grad, atol=3e-4)
    o5 = po.attention_step2_with_rel_pos_value_v2(cu(attn), cu(v), cu(offs, torch.int32), n_max, cu(i1, torch.int32), cu(tq_),
                                                  cu(rel, torch.int32))
    assert np.allclose(o5.cpu().numpy(), opo.rpe_attn_step2(attn, v, i0, i1, tq_, rel, n), atol=1e-4)


# ---- exact kNN on the hash grid (csrc/knn_grid.hip) against the brute-force kernel ---------------------------------------------
def _room_cloud(n_side, seed, batch=1):
    from scenesplat_amd.synthetic import room_chunk
    d = room_chunk(n_side, seed, lang_dim=0, batch=batch)
    return d["coord"].numpy().astype(np.float32), d["offset"].numpy()


@pytest.mark.parametrize("case", ["uniform", "room2", "queries_outside", "k_exceeds_segment", "outliers", "k1", "k64"])
def test_knn_grid_equals_the_brute_force_kernel(case):
    """The ring search returns the brute-force kernel's neighbours bit for bit -- indices AND squared distances (same fp32
    distance expression) -- on volumetric and surface clouds, several batch elements, queries outside the data's box, segments with
    fewer than k points (-1 padding) and isolated outliers (the wave-wide fallback scan)."""
    from scenesplat_amd import pointops as po
    g = np.random.default_rng(7)
    k, new_xyz, noff = 25, None, None
    if case == "uniform":
        xyz = g.random((20000, 3), dtype=np.float32) * np.array([4.0, 3.0, 2.0], np.float32); off = np.array([12000, 20000])
    elif case == "room2":
        xyz, off = _room_cloud(96, 1, batch=2)
    elif case == "queries_outside":
        xyz, off = _room_cloud(64, 2)
        new_xyz = np.concatenate([xyz[:3000] + 0.01, g.random((500, 3), dtype=np.float32) * 8 - 3]).astype(np.float32)
        noff = np.array([len(new_xyz)])
    elif case == "k_exceeds_segment":
        xyz = g.random((6000, 3), dtype=np.float32); off = np.array([5980, 6000]); k = 40          # the second element holds 20 points
    elif case == "outliers":
        xyz, off = _room_cloud(64, 3)
        xyz = np.concatenate([xyz, np.array([[40, 40, 40], [-30, 2, 1], [2, 55, -9]], np.float32)]); off = np.array([len(xyz)])
    elif case == "k1":
        xyz, off = _room_cloud(64, 4); k = 1
    else:
        xyz = g.random((9000, 3), dtype=np.float32); off = np.array([9000]); k = 64
    args = (cu(xyz), cu(off)) + (() if new_xyz is None else (cu(new_xyz), cu(noff)))
    bi, bd = po.knn_query(k, *args, impl="brute")
    gi, gd = po.knn_query(k, *args, impl="grid")
    assert torch.equal(gi, bi), (case, int((gi != bi).sum()))
    assert torch.equal(gd, bd)
    if case == "k_exceeds_segment":
        assert bool((gi[5980:, 20:] == -1).all()) and bool((gi[5980:, :20] >= 5980).all())
    # "auto" picks the grid from KNN_GRID_MIN_POINTS candidates on
    ai, _ = po.knn_query(k, *args)
    assert torch.equal(ai, bi)


def test_neighbor_voting_on_the_grid_matches_brute_force_and_scales():
    """evaluator.py:697-739 (k = 25 over the valid Gaussians, all Gaussians queried): same labels through either kNN; and the
    1,000,000-Gaussian case the review asks for runs (timing is reported by bench.py's secondary line, not asserted here)."""
    from scenesplat_amd import pointops as po
    xyz, _ = _room_cloud(160, 5)                       # 40,000 Gaussians
    n = len(xyz)
    g = torch.Generator().manual_seed(1)
    labels = torch.randint(0, 20, (n,), generator=g).int().cuda()
    valid = (torch.rand(n, generator=g) < 0.9).cuda()
    a = po.neighbor_voting(cu(xyz), labels, valid, 25, -1, 20)
    old = po.KNN_GRID_MIN_POINTS
    try:
        po.KNN_GRID_MIN_POINTS = 1 << 40
        b = po.neighbor_voting(cu(xyz), labels, valid, 25, -1, 20)
    finally:
        po.KNN_GRID_MIN_POINTS = old
    assert torch.equal(a, b)
    big, _ = _room_cloud(800, 6)                        # 1,000,000 Gaussians
    nb = len(big)
    lab = torch.randint(0, 160, (nb,), generator=g).int().cuda()
    val = (torch.rand(nb, generator=g) < 0.9).cuda()
    out = po.neighbor_voting(cu(big), lab, val, 25, -1, 160)
    torch.cuda.synchronize()
    assert out.shape == (nb,) and int(out.min()) >= 0 and int(out.max()) < 160


@pytest.mark.parametrize("case", ["room", "uniform2", "crowded", "queries_outside"])
def test_ball_query_on_the_grid_equals_the_brute_force_kernel(case):
    """ss_ball_grid_query (one wave per query, candidates sorted in LDS) against ss_ball_query (one thread per query, 16 KiB of global
    scratch each): identical indices and squared distances -- all candidates when they fit nsample, the strided subsample of the
    sorted list otherwise, and the reference's first-2048-in-index-order rule when a ball holds more than 2048 candidates."""
    from scenesplat_amd import pointops as po
    g = np.random.default_rng(11)
    new_xyz = noff = None
    if case == "room":
        xyz, off = _room_cloud(96, 1, batch=2); ns, rmax, rmin = 16, 0.1, 0.0
    elif case == "uniform2":
        xyz = g.random((15000, 3), dtype=np.float32); off = np.array([9000, 15000]); ns, rmax, rmin = 32, 0.08, 0.02
    elif case == "crowded":
        xyz = (g.random((12000, 3), dtype=np.float32) * 0.2).astype(np.float32); off = np.array([12000]); ns, rmax, rmin = 24, 0.15, 0.0   # > 2048 candidates per ball
    else:
        xyz, off = _room_cloud(64, 2); ns, rmax, rmin = 8, 0.07, 0.01
        new_xyz = np.concatenate([xyz[:2000] + 0.003, g.random((300, 3), dtype=np.float32) * 6 - 2]).astype(np.float32); noff = np.array([len(new_xyz)])
    args = (cu(xyz), cu(off)) + (() if new_xyz is None else (cu(new_xyz), cu(noff)))
    bi, bd = po.ball_query(ns, rmax, rmin, *args, impl="brute")
    gi, gd = po.ball_query(ns, rmax, rmin, *args, impl="grid")
    assert torch.equal(gi, bi), (case, int((gi != bi).sum()))
    assert torch.equal(gd, bd)
    if case == "crowded":
        assert int((bi >= 0).all(1).sum()) == len(bi)            # every ball overflows nsample: the strided subsample everywhere
