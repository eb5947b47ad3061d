"""Python surface of libs/pointops, libs/pointops2 and libs/pointgroup_ops on the HIP library.

Function names, argument order and return values follow the reference wrappers
(libs/pointops/functions/{query,sampling,grouping,interpolation,subtraction,aggregation,attention,utils}.py,
libs/pointops2/functions/pointops.py, libs/pointgroup_ops/functions/functions.py) so that
`import pointops` call sites in evaluator hooks / legacy backbones can be pointed here unchanged.
fp32 features, int32 indices, cumulative `offset` vectors; GPU tensors only.
"""
import ctypes

import torch
from torch.autograd import Function

from . import native as nv
from ._lib import check, NativeError

_p, _s, _req = nv._p, nv._stream, nv._req


def _f(t, name):
    return _req(t.contiguous(), torch.float32, name)


def _i(t, name):
    return _req(t.to(torch.int32).contiguous(), torch.int32, name)


# ---- utils (libs/pointops/functions/utils.py) --------------------------------------------------
def offset2batch(offset):
    return nv.offsets_to_batch(_i(offset, "offset"), int(offset[-1])).long()


def batch2offset(batch):
    b = batch.long()
    nb = int(b.max()) + 1 if b.numel() else 0
    return torch.zeros(nb, dtype=torch.long, device=b.device).scatter_add_(0, b, torch.ones_like(b)).cumsum(0).int()


# ---- queries -----------------------------------------------------------------------------------
KNN_GRID_MIN_POINTS = 4096       # below: the brute-force scan (one LDS tile pass) is faster than building a grid


def knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None, impl="auto"):
    """-> idx (m, nsample) int32 (-1 pad), dist (m, nsample) f32 (sqrt of squared distance).
    impl: "brute" = the O(m n) LDS-tiled scan (the reference's algorithm, knn_query_cuda_kernel.cu:60-104); "grid" = the exact
    ring search on a uniform hash grid (csrc/knn_grid.hip, nsample <= 64); "auto" = grid from KNN_GRID_MIN_POINTS candidates on.
    Both return the same neighbours (same fp32 distance expression; equal distances: the smaller index first)."""
    self_query = new_xyz is None or new_offset is None
    if self_query:
        new_xyz, new_offset = xyz, offset
    xyz, new_xyz = _f(xyz, "xyz"), _f(new_xyz, "new_xyz")
    off, noff = _i(offset, "offset"), _i(new_offset, "new_offset")
    m = new_xyz.shape[0]
    if impl == "grid" or (impl == "auto" and nsample <= 64 and xyz.shape[0] >= KNN_GRID_MIN_POINTS):
        if nsample > 64:
            raise RuntimeError("knn_query: the grid search holds one list entry per lane (nsample <= 64)")
        idx, d2 = _knn_grid(nsample, xyz, off, new_xyz, noff, self_query)
        return idx, torch.sqrt(d2)
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    d2 = torch.empty((m, nsample), dtype=torch.float32, device=xyz.device)
    check(nv.lib().ss_knn_query(m, nsample, _p(xyz), _p(new_xyz), _p(off), _p(noff), off.numel(), _p(idx), _p(d2), _s()),
          "ss_knn_query")
    return idx, torch.sqrt(d2)


def _knn_grid(nsample, xyz, off, new_xyz, noff, self_query, cell=None, return_info=False, ball=None):
    """Build the hash grid over xyz (cell size from the data: ~2/3 nsample points per OCCUPIED cell, refined at most twice from
    the occupied-cell count the build reports) and run the ring search.  Host reads: the bounding box and that count."""
    import ctypes
    n, m, dev = xyz.shape[0], new_xyz.shape[0], xyz.device
    lib = nv.lib()
    lo, hi = xyz.amin(0), xyz.amax(0)
    box = torch.cat([lo, hi]).cpu().tolist()
    origin, ext = box[:3], [max(box[3 + a] - box[a], 1e-6) for a in range(3)]
    target = max(4.0, 0.66 * nsample)
    if cell is None:
        h = max((ext[0] * ext[1] * ext[2] / max(n, 1) * target) ** (1.0 / 3.0), max(ext) / 60000.0, 1e-6)
        tries = 3
    else:
        h, tries = max(float(cell), max(ext) / 60000.0), 1
    ws = nv._ws(lib.ss_knn_grid_workspace_bytes(n), dev)
    keys = torch.empty((1, n), dtype=torch.int64, device=dev)
    ncell = torch.empty(1, dtype=torch.int32, device=dev)
    nb = off.numel()
    for it in range(tries):
        check(lib.ss_knn_grid_keys(_p(xyz), _p(off), nb, n, origin[0], origin[1], origin[2], h, _p(keys), _s()), "ss_knn_grid_keys")
        order, _, skeys = nv.argsort_i64(keys, 63, want_inverse=False)
        check(lib.ss_knn_grid_build(_p(xyz), _p(skeys), _p(order), n, _p(ws), ws.numel(), _p(ncell), _s()), "ss_knn_grid_build")
        if it + 1 == tries:
            break
        occ = n / max(1, int(ncell.item()))
        if occ > 1.6 * target:         # surfaces: occupancy grows with h^2
            h2 = h * max(0.25, (target / occ) ** 0.5)
        elif occ < 0.4 * target:
            h2 = h * min(4.0, (target / occ) ** (1.0 / 3.0))
        else:
            break
        h2 = max(h2, max(ext) / 60000.0)
        if abs(h2 - h) < 1e-3 * h:
            break
        h = h2
    dims = [min(65535, int(ext[a] / h) + 1) for a in range(3)]
    qorder = order[0]
    if not self_query:
        qkeys = torch.empty((1, m), dtype=torch.int64, device=dev)
        check(lib.ss_knn_grid_keys(_p(new_xyz), _p(noff), noff.numel(), m, origin[0], origin[1], origin[2], h, _p(qkeys), _s()), "ss_knn_grid_keys")
        qorder = nv.argsort_i64(qkeys, 63, want_inverse=False, want_sorted=False)[0][0]
    idx = torch.empty((m, nsample), dtype=torch.int32, device=dev)
    d2 = torch.empty((m, nsample), dtype=torch.float32, device=dev)
    if ball is not None:
        check(lib.ss_ball_grid_query(m, nsample, ball[0], ball[1], _p(xyz), _p(new_xyz), _p(qorder), _p(off), _p(noff), nb, origin[0], origin[1],
                                     origin[2], h, dims[0], dims[1], dims[2], n, _p(ws), _p(idx), _p(d2), _s()), "ss_ball_grid_query")
        return idx, d2
    check(lib.ss_knn_grid_query(m, nsample, _p(new_xyz), _p(qorder), _p(off), _p(noff), nb, origin[0], origin[1], origin[2], h,
                                dims[0], dims[1], dims[2], n, _p(ws), _p(idx), _p(d2), _s()), "ss_knn_grid_query")
    if return_info:
        return idx, d2, dict(cell=h, dims=dims, occupied_cells=int(ncell.item()), points_per_cell=n / max(1, int(ncell.item())))
    return idx, d2


def ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None, impl="auto"):
    """impl: "brute" = one thread per query scanning its batch element (the reference's algorithm); "grid" = one wave per query on the
    hash grid of csrc/knn_grid.hip (cell = max_radius: 27 cells hold the ball); "auto" = grid from KNN_GRID_MIN_POINTS candidates on.
    Same result (candidates sorted ascending; equal distances: the smaller index first)."""
    self_query = new_xyz is None or new_offset is None
    if self_query:
        new_xyz, new_offset = xyz, offset
    assert min_radius < max_radius
    xyz, new_xyz = _f(xyz, "xyz"), _f(new_xyz, "new_xyz")
    off, noff = _i(offset, "offset"), _i(new_offset, "new_offset")
    m = new_xyz.shape[0]
    if impl == "grid" or (impl == "auto" and xyz.shape[0] >= KNN_GRID_MIN_POINTS):
        idx, d2 = _knn_grid(nsample, xyz, off, new_xyz, noff, self_query, cell=float(max_radius), ball=(float(min_radius), float(max_radius)))
        return idx, torch.sqrt(d2)
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    d2 = torch.empty((m, nsample), dtype=torch.float32, device=xyz.device)
    ws = nv._ws(nv.lib().ss_ball_query_workspace_bytes(m), xyz.device)
    check(nv.lib().ss_ball_query(m, nsample, float(min_radius), float(max_radius), _p(xyz), _p(new_xyz), _p(off), _p(noff),
                                 off.numel(), _p(idx), _p(d2), _p(ws), ws.numel(), _s()), "ss_ball_query")
    return idx, torch.sqrt(d2)


def random_ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None, order=None):
    if new_xyz is None or new_offset is None:
        new_xyz, new_offset = xyz, offset
    assert min_radius < max_radius
    xyz, new_xyz = _f(xyz, "xyz"), _f(new_xyz, "new_xyz")
    off, noff = _i(offset, "offset"), _i(new_offset, "new_offset")
    if order is None:   # per-batch random permutation (functions/query.py:48-54)
        parts, s = [], 0
        for e in off.tolist():
            parts.append(torch.randperm(e - s, dtype=torch.int32, device=xyz.device) + s)
            s = e
        order = torch.cat(parts)
    order = _i(order, "order")
    m = new_xyz.shape[0]
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    d2 = torch.empty((m, nsample), dtype=torch.float32, device=xyz.device)
    check(nv.lib().ss_random_ball_query(m, nsample, float(min_radius), float(max_radius), _p(order), _p(xyz), _p(new_xyz),
                                        _p(off), _p(noff), off.numel(), _p(idx), _p(d2), _s()), "ss_random_ball_query")
    return idx, torch.sqrt(d2)


def farthest_point_sampling(xyz, offset, new_offset):
    """-> idx (m) int32 (functions/sampling.py:7-24)."""
    xyz = _f(xyz, "xyz")
    off, noff = _i(offset, "offset"), _i(new_offset, "new_offset")
    m = int(noff[-1])
    idx = torch.zeros(m, dtype=torch.int32, device=xyz.device)
    tmp = torch.full((xyz.shape[0],), 1e10, dtype=torch.float32, device=xyz.device)
    check(nv.lib().ss_farthest_point_sampling(off.numel(), _p(xyz), _p(off), _p(noff), _p(tmp), _p(idx), _s()),
          "ss_farthest_point_sampling")
    return idx


# ---- grouping / interpolation / subtraction / aggregation ------------------------------------------
class Grouping(Function):
    @staticmethod
    def forward(ctx, input, idx):
        input, idx = _f(input, "input"), _i(idx, "idx")
        m, ns = idx.shape
        n, c = input.shape
        out = torch.empty((m, ns, c), dtype=torch.float32, device=input.device)
        check(nv.lib().ss_grouping_fwd(m, ns, c, _p(input), _p(idx), _p(out), _s()), "ss_grouping_fwd")
        ctx.n = n
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        m, ns, c = g.shape
        gi = torch.zeros((ctx.n, c), dtype=torch.float32, device=g.device)
        check(nv.lib().ss_grouping_bwd(m, ns, c, _p(_f(g, "g")), _p(idx), _p(gi), _s()), "ss_grouping_bwd")
        return gi, None


def grouping2(input, idx):
    return Grouping.apply(input, idx)


def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    """functions/grouping.py:36-58 semantics: -1 neighbours give zero rows; optional relative xyz first."""
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and feat.is_contiguous()
    g = Grouping.apply(feat, idx)
    if with_xyz:
        mask = torch.sign(idx + 1).unsqueeze(-1).float()
        gx = (Grouping.apply(xyz, idx) - new_xyz.unsqueeze(1)) * mask
        return torch.cat((gx, g), -1)
    return g


def query_and_group(xyz, new_xyz, feat, idx, offset, new_offset, nsample=None, with_xyz=True):
    if idx is None:
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def knn_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, nsample=None, with_xyz=False):
    if idx is None:
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def ball_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, max_radio=None, min_radio=0,
                         nsample=None, with_xyz=False):
    if idx is None:
        idx, _ = ball_query(nsample, max_radio, min_radio, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


class Interpolation2(Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        idx, dist = knn_query(k, xyz, offset, new_xyz, new_offset)
        dist_recip = 1.0 / (dist + 1e-8)
        weight = (dist_recip / torch.sum(dist_recip, dim=1, keepdim=True)).contiguous()
        input = _f(input, "input")
        n, c, m = new_xyz.shape[0], input.shape[1], input.shape[0]
        out = torch.empty((n, c), dtype=torch.float32, device=input.device)
        check(nv.lib().ss_interpolation_fwd(n, c, k, _p(input), _p(idx), _p(weight), _p(out), _s()), "ss_interpolation_fwd")
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, weight = ctx.saved_tensors
        n, c = g.shape
        gi = torch.zeros((ctx.m, c), dtype=torch.float32, device=g.device)
        check(nv.lib().ss_interpolation_bwd(n, c, ctx.k, _p(_f(g, "g")), _p(idx), _p(weight), _p(gi), _s()), "ss_interpolation_bwd")
        return None, None, gi, None, None, None


interpolation2 = Interpolation2.apply


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """functions/interpolation.py:26-40 (inverse-distance weights over k nearest)."""
    return Interpolation2.apply(xyz, new_xyz, feat, offset, new_offset, k)


class Subtraction(Function):
    @staticmethod
    def forward(ctx, input1, input2, idx):
        input1, input2, idx = _f(input1, "input1"), _f(input2, "input2"), _i(idx, "idx")
        n, c = input1.shape
        ns = idx.shape[-1]
        out = torch.empty((n, ns, c), dtype=torch.float32, device=input1.device)
        check(nv.lib().ss_subtraction_fwd(n, ns, c, _p(input1), _p(input2), _p(idx), _p(out), _s()), "ss_subtraction_fwd")
        ctx.save_for_backward(idx)
        ctx.n2 = input2.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        n, ns, c = g.shape
        g1 = torch.empty((n, c), dtype=torch.float32, device=g.device)
        g2 = torch.zeros((ctx.n2, c), dtype=torch.float32, device=g.device)
        check(nv.lib().ss_subtraction_bwd(n, ns, c, _p(idx), _p(_f(g, "g")), _p(g1), _p(g2), _s()), "ss_subtraction_bwd")
        return g1, g2, None


subtraction = Subtraction.apply


class Aggregation(Function):
    @staticmethod
    def forward(ctx, input, position, weight, idx):
        input, position, weight, idx = _f(input, "input"), _f(position, "position"), _f(weight, "weight"), _i(idx, "idx")
        n, ns, c = position.shape
        wc = weight.shape[-1]
        out = torch.empty((n, c), dtype=torch.float32, device=input.device)
        check(nv.lib().ss_aggregation_fwd(n, ns, c, wc, _p(input), _p(position), _p(weight), _p(idx), _p(out), _s()),
              "ss_aggregation_fwd")
        ctx.save_for_backward(input, position, weight, idx)
        return out

    @staticmethod
    def backward(ctx, g):
        input, position, weight, idx = ctx.saved_tensors
        n, ns, c = position.shape
        wc = weight.shape[-1]
        gi = torch.zeros_like(input); gp = torch.empty_like(position); gw = torch.zeros_like(weight)
        check(nv.lib().ss_aggregation_bwd(n, ns, c, wc, _p(input), _p(position), _p(weight), _p(idx), _p(_f(g, "g")), _p(gi),
                                          _p(gp), _p(gw), _s()), "ss_aggregation_bwd")
        return gi, gp, gw, None


aggregation = Aggregation.apply


# ---- attention relation / fusion (pointops) and step1 / step2 (pointops2) --------------------------
class AttentionRelationStep(Function):
    @staticmethod
    def forward(ctx, query, key, weight, index_target, index_refer):
        query, key = _f(query, "query"), _f(key, "key")
        weight = _f(weight, "weight") if weight is not None else None
        it, ir = _i(index_target, "index_target"), _i(index_refer, "index_refer")
        n, g, c = query.shape
        m = it.shape[0]
        out = torch.empty((m, g), dtype=torch.float32, device=query.device)
        check(nv.lib().ss_attention_relation_fwd(m, g, c, _p(query), _p(key), _p(weight), _p(it), _p(ir), _p(out), _s()),
              "ss_attention_relation_fwd")
        ctx.has_w = weight is not None
        ctx.save_for_backward(query, key, weight if weight is not None else query.new_empty(0), it, ir)
        return out

    @staticmethod
    def backward(ctx, g_):
        query, key, weight, it, ir = ctx.saved_tensors
        n, g, c = query.shape
        m = it.shape[0]
        gq, gk = torch.zeros_like(query), torch.zeros_like(key)
        gw = torch.zeros_like(weight) if ctx.has_w else None
        check(nv.lib().ss_attention_relation_bwd(m, g, c, _p(query), _p(gq), _p(key), _p(gk), _p(weight if ctx.has_w else None),
                                                 _p(gw), _p(it), _p(ir), _p(_f(g_, "g")), _s()), "ss_attention_relation_bwd")
        return gq, gk, gw, None, None


attention_relation_step = AttentionRelationStep.apply


class AttentionFusionStep(Function):
    @staticmethod
    def forward(ctx, weight, value, index_target, index_refer):
        weight, value = _f(weight, "weight"), _f(value, "value")
        it, ir = _i(index_target, "index_target"), _i(index_refer, "index_refer")
        n, g, c = value.shape
        m = it.shape[0]
        out = torch.zeros((n, g, c), dtype=torch.float32, device=value.device)
        check(nv.lib().ss_attention_fusion_fwd(m, g, c, _p(weight), _p(value), _p(it), _p(ir), _p(out), _s()),
              "ss_attention_fusion_fwd")
        ctx.save_for_backward(weight, value, it, ir)
        return out

    @staticmethod
    def backward(ctx, g_):
        weight, value, it, ir = ctx.saved_tensors
        n, g, c = value.shape
        m = it.shape[0]
        gw, gv = torch.empty_like(weight), torch.zeros_like(value)
        check(nv.lib().ss_attention_fusion_bwd(m, g, c, _p(weight), _p(gw), _p(value), _p(gv), _p(it), _p(ir), _p(_f(g_, "g")),
                                               _s()), "ss_attention_fusion_bwd")
        return gw, gv, None, None


attention_fusion_step = AttentionFusionStep.apply


def attention_step1(q, k, index0, index1):
    """pointops2 attention_step1(_v2): attn (M, h) = q[index0] . k[index1] per head (q, k: (N, h, C//h))."""
    return AttentionRelationStep.apply(q, k, None, index0, index1)


def attention_step2(attn, v, index0, index1):
    """pointops2 attention_step2: out[index0[m]] += attn[m] * v[index1[m]]."""
    return AttentionFusionStep.apply(attn, v, index0, index1)


class DotProdWithIdx(Function):
    @staticmethod
    def forward(ctx, q, index, table, rel_idx):
        q, table, index, rel_idx = _f(q, "q"), _f(table, "table"), _i(index, "index"), _i(rel_idx, "rel_idx")
        n, h, hd = q.shape
        m = index.shape[0]
        out = torch.empty((m, h), dtype=torch.float32, device=q.device)
        check(nv.lib().ss_rpe_dot_prod_fwd(n, m, h, hd, _p(q), _p(index), _p(table), _p(rel_idx), _p(out), _s()), "ss_rpe_dot_prod_fwd")
        ctx.save_for_backward(q, index, table, rel_idx)
        return out

    @staticmethod
    def backward(ctx, g):
        q, index, table, rel_idx = ctx.saved_tensors
        n, h, hd = q.shape
        m = index.shape[0]
        gq, gt = torch.zeros_like(q), torch.zeros_like(table)
        check(nv.lib().ss_rpe_dot_prod_bwd(n, m, h, hd, _p(_f(g, "g")), _p(q), _p(index), _p(table), _p(rel_idx), _p(gq), _p(gt),
                                           _s()), "ss_rpe_dot_prod_bwd")
        return gq, None, gt, None


dot_prod_with_idx = DotProdWithIdx.apply


class AttentionStep2WithRelPosValue(Function):
    @staticmethod
    def forward(ctx, attn, v, index0, index1, table, rel_idx):
        attn, v, table = _f(attn, "attn"), _f(v, "v"), _f(table, "table")
        i0, i1, rel_idx = _i(index0, "index0"), _i(index1, "index1"), _i(rel_idx, "rel_idx")
        n, h, hd = v.shape
        m = i0.shape[0]
        out = torch.zeros((n, h, hd), dtype=torch.float32, device=v.device)
        check(nv.lib().ss_rpe_attn_step2_fwd(n, m, h, hd, _p(attn), _p(v), _p(i0), _p(i1), _p(table), _p(rel_idx), _p(out), _s()),
              "ss_rpe_attn_step2_fwd")
        ctx.save_for_backward(attn, v, i0, i1, table, rel_idx)
        return out

    @staticmethod
    def backward(ctx, g):
        attn, v, i0, i1, table, rel_idx = ctx.saved_tensors
        n, h, hd = v.shape
        m = i0.shape[0]
        ga, gv, gt = torch.zeros_like(attn), torch.zeros_like(v), torch.zeros_like(table)
        check(nv.lib().ss_rpe_attn_step2_bwd(n, m, h, hd, _p(_f(g, "g")), _p(i0), _p(i1), _p(attn), _p(v), _p(table), _p(rel_idx),
                                             _p(ga), _p(gv), _p(gt), _s()), "ss_rpe_attn_step2_bwd")
        return ga, gv, None, None, gt, None


attention_step2_with_rel_pos_value = AttentionStep2WithRelPosValue.apply


# ---- pointops2 v2 / v3 call forms (libs/pointops2/functions/pointops.py:166-957) ------------------------------------
# The reference's _v2 / _v3 kernels compute the SAME quantities as the v1 ones; they only take the query side as CSR
# offsets (pairs sorted by query, `index0_offsets` (N+1), `n_max` = longest row -- a launch-geometry hint of the CUDA
# kernels) and fuse the q- and k-table dot products.  Here they are call-compatible fronts of the same HIP kernels.
def _index_from_offsets(offsets, m):
    offsets = offsets.to(torch.int64)
    counts = offsets[1:] - offsets[:-1]
    idx = torch.repeat_interleave(torch.arange(counts.numel(), device=offsets.device), counts, output_size=int(m))
    return idx.to(torch.int32)


def attention_step1_v2(q, k, index1, index0_offsets, n_max=None):
    """attn (M, h) = q[index0[m]] . k[index1[m]] with index0 given as CSR offsets (pointops.py:166-254)."""
    return attention_step1(q, k, _index_from_offsets(index0_offsets, index1.shape[0]), index1)


def attention_step2_v2(attn, v, index0, index1):
    """Same contract as attention_step2 (pointops.py:334-400 launches the v1 kernel)."""
    return attention_step2(attn, v, index0, index1)


def dot_prod_with_idx_v2(q, index_q, k, index_k, table_q, table_k, rel_idx):
    """out (M, h) = q[index_q] . table_q[rel_idx] + k[index_k] . table_k[rel_idx] (pointops.py:472-625)."""
    return dot_prod_with_idx(q, index_q, table_q, rel_idx) + dot_prod_with_idx(k, index_k, table_k, rel_idx)


def dot_prod_with_idx_v3(q, index_q_offsets, n_max, k, index_k, table_q, table_k, rel_idx):
    """dot_prod_with_idx_v2 with the query index as CSR offsets (pointops.py:628-751)."""
    index_q = _index_from_offsets(index_q_offsets, index_k.shape[0])
    return dot_prod_with_idx_v2(q, index_q, k, index_k, table_q, table_k, rel_idx)


def attention_step2_with_rel_pos_value_v2(attn, v, index0_offsets, n_max, index1, table, rel_idx):
    """attention_step2_with_rel_pos_value with index0 as CSR offsets (pointops.py:850-957)."""
    index0 = _index_from_offsets(index0_offsets, index1.shape[0])
    return attention_step2_with_rel_pos_value(attn, v, index0, index1, table, rel_idx)


# pointops2 spellings of the shared ops (pointops.py:30,52,86,1054,1107,1190)
furthestsampling = farthest_point_sampling


# ---- pointgroup_ops (functions/functions.py) ----------------------------------------------------------
def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive):
    """-> idx (nActive) int32, start_len (n, 2) int32.  Two-pass and deterministic; the buffer is sized
    n*meanActive like the reference and grown once if the pair count exceeds it (functions.py:26-35)."""
    coords = _f(coords, "coords")
    bi, bo = _i(batch_idxs, "batch_idxs"), _i(batch_offsets, "batch_offsets")
    n = coords.shape[0]
    while True:
        idx = torch.zeros(n * meanActive, dtype=torch.int32, device=coords.device)
        start_len = torch.zeros((n, 2), dtype=torch.int32, device=coords.device)
        total = torch.zeros(1, dtype=torch.int32, device=coords.device)
        check(nv.lib().ss_ballquery_batch_p(n, int(meanActive), float(radius), _p(coords), _p(bi), _p(bo), _p(idx), _p(start_len),
                                            _p(total), _s()), "ss_ballquery_batch_p")
        nactive = int(total.item())
        if nactive <= n * meanActive:
            break
        meanActive = int(nactive // n + 1)
    return idx[:nactive], start_len


def bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold):
    """CPU int32 tensors in, CPU tensors out: cluster_idxs (sumNPoint, 2), cluster_offsets (nCluster + 1)."""
    sl, bq, st = (t.to(torch.int32).contiguous().cpu() for t in (semantic_label, ball_query_idxs, start_len))
    n = sl.shape[0]
    cidx = torch.zeros((max(n, 1), 2), dtype=torch.int32)
    coff = torch.zeros(n + 2, dtype=torch.int32)
    nc, npts = ctypes.c_int32(0), ctypes.c_int32(0)
    rc = nv.lib().ss_bfs_cluster(_p(sl), _p(bq), _p(st), n, int(threshold), _p(cidx), n, _p(coff), n + 1,
                                 ctypes.addressof(nc), ctypes.addressof(npts))
    if rc != 0:
        raise NativeError(f"ss_bfs_cluster failed: status {rc}")
    return cidx[:npts.value].clone(), coff[:nc.value + 1].clone()


# ---- evaluator helpers -----------------------------------------------------------------
def majority_vote(nn_idx, labels, ignore_label, num_classes):
    """out[i] = most frequent valid label among labels[nn_idx[i]] (ties -> smallest label)."""
    nn_idx, labels = _i(nn_idx, "nn_idx"), _i(labels, "labels")
    m, k = nn_idx.shape
    out = torch.empty(m, dtype=torch.int32, device=nn_idx.device)
    check(nv.lib().ss_majority_vote(_p(nn_idx), _p(labels), m, k, int(ignore_label), int(num_classes), _p(out), _s()),
          "ss_majority_vote")
    return out


def neighbor_voting(coords, initial_labels, valid_mask, vote_k, ignore_label, num_classes, query_coords=None):
    """GPU form of LangPretrainZeroShotSemSegEval._neighbor_voting (engines/hooks/evaluator.py:697-739, top-1):
    kNN (k = vote_k) among the valid Gaussians + majority vote, instead of CPU cKDTree + numba."""
    valid = valid_mask.bool()
    vc = coords[valid].float().contiguous()
    q = (coords if query_coords is None else query_coords).float().contiguous()
    if vc.shape[0] == 0:
        return torch.full((q.shape[0],), ignore_label, dtype=torch.int32, device=coords.device)
    k = min(int(vote_k), vc.shape[0])
    off = torch.tensor([vc.shape[0]], dtype=torch.int32, device=coords.device)
    noff = torch.tensor([q.shape[0]], dtype=torch.int32, device=coords.device)
    idx, _ = knn_query(k, vc, off, q, noff)
    return majority_vote(idx, initial_labels[valid], ignore_label, num_classes)
