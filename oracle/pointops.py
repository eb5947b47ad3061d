"""Oracle (test infrastructure): numpy restatements of libs/pointops, pointops2, pointgroup_ops.

"parity unpinned": the CUDA sources cannot be built here (no nvcc) and the reference ships no
fixtures for them; each function follows the .cu file cited (paths under
/root/reference/libs).  Pure loops / vectorised numpy, small cases only."""
import numpy as np


def _segments(offset):
    offset = np.asarray(offset).astype(np.int64)
    return np.concatenate([[0], offset[:-1]]), offset


def _batch_of(i, offset):
    return int(np.searchsorted(np.asarray(offset), i, side="right"))


def knn_query(nsample, xyz, offset, new_xyz=None, new_offset=None):
    """pointops/src/knn_query/knn_query_cuda_kernel.cu:60-104 -> idx (-1 pad), dist2 (1e10 pad), ascending."""
    if new_xyz is None:
        new_xyz, new_offset = xyz, offset
    st, en = _segments(offset)
    m = len(new_xyz)
    idx = np.full((m, nsample), -1, np.int64); d2 = np.full((m, nsample), 1e10, np.float32)
    for q in range(m):
        b = _batch_of(q, new_offset)
        seg = np.arange(st[b], en[b])
        dd = ((xyz[seg].astype(np.float32) - new_xyz[q].astype(np.float32)) ** 2).sum(1, dtype=np.float32)
        o = np.argsort(dd, kind="stable")[:nsample]
        idx[q, :len(o)] = seg[o]; d2[q, :len(o)] = dd[o]
    return idx, d2


def ball_query(nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None):
    """ball_query_cuda_kernel.cu:58-123 with true nearest-first ordering and real distances."""
    if new_xyz is None:
        new_xyz, new_offset = xyz, offset
    st, en = _segments(offset)
    m = len(new_xyz)
    idx = np.full((m, nsample), -1, np.int64); d2 = np.full((m, nsample), 1e10, np.float32)
    mn2, mx2 = np.float32(min_radius) ** 2, np.float32(max_radius) ** 2
    for q in range(m):
        b = _batch_of(q, new_offset)
        seg = np.arange(st[b], en[b])
        dd = ((xyz[seg].astype(np.float32) - new_xyz[q].astype(np.float32)) ** 2).sum(1, dtype=np.float32)
        keep = (dd <= 1e-5) | ((dd >= mn2) & (dd < mx2))
        seg, dd = seg[keep][:2048], dd[keep][:2048]
        o = np.argsort(dd, kind="stable")
        seg, dd = seg[o], dd[o]
        if len(seg) <= nsample:
            idx[q, :len(seg)] = seg; d2[q, :len(seg)] = dd
        else:
            sep = np.float32(len(seg)) / np.float32(nsample)
            k = (sep * np.arange(nsample, dtype=np.float32)).astype(np.int64)
            idx[q] = seg[k]; d2[q] = dd[k]
    return idx, d2


def random_ball_query(nsample, max_radius, min_radius, order, xyz, offset, new_xyz=None, new_offset=None):
    """random_ball_query_cuda_kernel.cu:58-108: first nsample hits along `order`."""
    if new_xyz is None:
        new_xyz, new_offset = xyz, offset
    st, en = _segments(offset)
    m = len(new_xyz)
    idx = np.full((m, nsample), -1, np.int64); d2 = np.full((m, nsample), 1e10, np.float32)
    mn2, mx2 = np.float32(min_radius) ** 2, np.float32(max_radius) ** 2
    for q in range(m):
        b = _batch_of(q, new_offset)
        cnt = 0
        for i in range(st[b], en[b]):
            o = order[i]
            dd = ((xyz[o].astype(np.float32) - new_xyz[q].astype(np.float32)) ** 2).sum(dtype=np.float32)
            if dd <= 1e-5 or (mn2 <= dd < mx2):
                idx[q, cnt] = o; d2[q, cnt] = dd; cnt += 1
                if cnt >= nsample:
                    break
    return idx, d2


def farthest_point_sampling(xyz, offset, new_offset):
    """sampling_cuda_kernel.cu:14-129: first sample = first point; arg-max of running min distance."""
    st, en = _segments(offset)
    ost, oen = _segments(new_offset)
    out = np.zeros(int(np.asarray(new_offset)[-1]), np.int64)
    for b in range(len(st)):
        if oen[b] <= ost[b]:
            continue
        pts = xyz[st[b]:en[b]].astype(np.float32)
        tmp = np.full(len(pts), 1e10, np.float32)
        cur = 0
        out[ost[b]] = st[b]
        for j in range(ost[b] + 1, oen[b]):
            d = ((pts - pts[cur]) ** 2).sum(1, dtype=np.float32)
            tmp = np.minimum(tmp, d)
            cur = int(np.argmax(tmp))
            out[j] = st[b] + cur
    return out


def grouping(inp, idx):
    out = inp[np.maximum(idx, 0)]
    out[idx < 0] = 0
    return out


def subtraction(in1, in2, idx):
    return in1[:, None, :] - in2[idx]


def aggregation(inp, pos, w, idx):
    c, wc = inp.shape[1], w.shape[-1]
    wfull = w[:, :, np.arange(c) % wc]
    return ((inp[idx] + pos) * wfull).sum(1)


def interpolation_weights(dist, eps=1e-8):
    r = 1.0 / (dist + eps)
    return r / r.sum(1, keepdims=True)


def interpolation(inp, idx, w):
    return (inp[idx] * w[:, :, None]).sum(1)


def attention_relation(q, k, w, it, ir):
    ww = 1.0 if w is None else w[None, None, :]
    return (q[it] * k[ir] * ww).sum(-1)


def attention_fusion(w, v, it, ir, n):
    out = np.zeros((n,) + v.shape[1:], np.float64)
    np.add.at(out, it, w[:, :, None] * v[ir])
    return out


def rpe_dot_prod(q, index, table, rel_idx):
    m = len(index)
    out = np.zeros((m, q.shape[1]), np.float64)
    for d in range(3):
        out += (q[index] * table[rel_idx[:, d], :, :, d]).sum(-1)
    return out


def rpe_attn_step2(attn, v, i0, i1, table, rel_idx, n):
    t = sum(table[rel_idx[:, d], :, :, d] for d in range(3))
    out = np.zeros((n,) + v.shape[1:], np.float64)
    np.add.at(out, i0, attn[:, :, None] * (v[i1] + t))
    return out


def ballquery_batch_p(xyz, batch_idxs, batch_offsets, radius):
    n = len(xyz)
    idx, start_len = [], np.zeros((n, 2), np.int64)
    r2 = np.float32(radius) ** 2
    for i in range(n):
        b = batch_idxs[i]
        seg = np.arange(batch_offsets[b], batch_offsets[b + 1])
        dd = ((xyz[seg].astype(np.float32) - xyz[i].astype(np.float32)) ** 2).sum(1, dtype=np.float32)
        hit = seg[dd < r2][:1000]
        start_len[i] = (len(idx), len(hit))
        idx.extend(hit.tolist())
    return np.array(idx, np.int64), start_len


def bfs_cluster(semantic_label, ball_query_idxs, start_len, threshold):
    """pointgroup_ops/src/bfs_cluster.cpp:53-137"""
    n = len(semantic_label)
    visited = np.zeros(n, bool)
    clusters = []
    for i in range(n):
        if visited[i]:
            continue
        cc, queue = [i], [i]
        visited[i] = True
        while queue:
            cur = queue.pop(0)
            s, ln = start_len[cur]
            for j in ball_query_idxs[s:s + ln]:
                if semantic_label[j] != semantic_label[cur] or visited[j]:
                    continue
                visited[j] = True; cc.append(int(j)); queue.append(int(j))
        if len(cc) >= threshold:
            clusters.append(cc)
    offs = np.concatenate([[0], np.cumsum([len(c) for c in clusters])]).astype(np.int64)
    idxs = np.array([[ci, p] for ci, c in enumerate(clusters) for p in c], np.int64).reshape(-1, 2)
    return idxs, offs
