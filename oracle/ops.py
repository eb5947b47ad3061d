"""Oracle (test infrastructure): pure-PyTorch fp32 CPU restatements of the three
third-party ops PTv3 calls, autograd-capable.

  submanifold conv  <- spconv.SubMConv3d call sites ptv3:278-284, 499-506 (spconv is an
                       un-vendored wheel: arithmetic "parity unpinned", definition
                       out[i] = b + sum_t W_t . in[j(i,t)], j = site at grid(i)+delta_t in the
                       same batch element; weight layout (Cout, kx, ky, kz, Cin))
  segment_csr       <- torch_scatter.segment_csr call sites ptv3:416-421 (un-vendored:
                       "parity unpinned"; out[i] = reduce(src[ptr[i]:ptr[i+1]]))
  window attention  <- the reference's own non-flash branch ptv3:190-206 generalised to
                       cu_seqlens segments exactly like flash_attn_varlen (ptv3:208-214)
"""
import numpy as np
import torch


def neighbor_table(grid_coord, batch, kernel_size):
    """(n, k^3) int64 table, -1 where no site.  Tap t = (ix*k + iy)*k + iz with offset
    (ix-k//2, iy-k//2, iz-k//2) over (x, y, z).  Duplicate voxels: the winner is the
    lowest row index (including for the centre tap)."""
    gc = np.asarray(grid_coord).astype(np.int64)
    b = np.asarray(batch).astype(np.int64)
    k = int(kernel_size); h = k // 2

    def pack(bb, x, y, z):
        return (bb << 51) | ((x + 1) << 34) | ((y + 1) << 17) | (z + 1)

    keys = pack(b, gc[:, 0], gc[:, 1], gc[:, 2])
    srt = np.argsort(keys, kind="stable")
    skeys = keys[srt]
    n = len(keys)
    nbr = np.full((n, k ** 3), -1, dtype=np.int64)
    t = 0
    for ix in range(k):
        for iy in range(k):
            for iz in range(k):
                x = gc[:, 0] + ix - h; y = gc[:, 1] + iy - h; z = gc[:, 2] + iz - h
                ok = (x >= 0) & (y >= 0) & (z >= 0)
                q = pack(b, np.maximum(x, 0), np.maximum(y, 0), np.maximum(z, 0))
                pos = np.searchsorted(skeys, q, side="left")
                posc = np.minimum(pos, n - 1)
                hit = ok & (pos < n) & (skeys[posc] == q)
                nbr[hit, t] = srt[posc[hit]]
                t += 1
    return nbr


def subm_conv3d(feat, weight, bias, nbr):
    """feat (n,Cin) f32, weight (Cout,k,k,k,Cin), bias (Cout)|None, nbr (n,k^3)."""
    n = feat.shape[0]
    cout = weight.shape[0]
    taps = nbr.shape[1]
    w = weight.reshape(cout, taps, -1)
    out = feat.new_zeros(n, cout)
    nbr_t = torch.as_tensor(nbr)
    for t in range(taps):
        j = nbr_t[:, t]
        rows = torch.nonzero(j >= 0, as_tuple=True)[0]
        if rows.numel() == 0:
            continue
        out = out.index_add(0, rows, feat[j[rows]] @ w[:, t, :].t())
    if bias is not None:
        out = out + bias
    return out


def segment_csr(src, indptr, reduce="mean"):
    """out[i] = reduce(src[indptr[i]:indptr[i+1]]); empty segments give 0."""
    indptr = torch.as_tensor(indptr, dtype=torch.int64)
    counts = indptr[1:] - indptr[:-1]
    seg = torch.repeat_interleave(torch.arange(len(counts)), counts)
    shape = (len(counts),) + tuple(src.shape[1:])
    if reduce in ("sum", "mean"):
        out = src.new_zeros(shape).index_add(0, seg, src)
        if reduce == "mean":
            out = out / counts.clamp(min=1).to(src.dtype).reshape(-1, *([1] * (src.dim() - 1)))
        return out
    if reduce in ("max", "min"):
        idx = seg.reshape(-1, *([1] * (src.dim() - 1))).expand_as(src)
        out = src.new_zeros(shape).scatter_reduce(0, idx, src, "a" + reduce, include_self=False)
        return out
    raise ValueError(reduce)


def window_attention(qkv, cu_seqlens, num_heads, scale, max_windows_per_chunk=8):
    """qkv (total, 3*C) already in window order; segments [cu[j], cu[j+1]) attend
    within themselves: softmax(q*scale @ k^T) @ v, no mask, no bias (ptv3:190-206).
    Returns (total, C)."""
    total, c3 = qkv.shape
    C = c3 // 3; H = num_heads; d = C // H
    cu = [int(v) for v in np.asarray(cu_seqlens)]
    outs = []
    j = 0
    while j < len(cu) - 1:
        # batch consecutive windows of equal length to bound memory
        L = cu[j + 1] - cu[j]
        e = j
        while e < len(cu) - 1 and cu[e + 1] - cu[e] == L and e - j < max_windows_per_chunk:
            e += 1
        blk = qkv[cu[j]:cu[e]].reshape(e - j, L, 3, H, d).permute(2, 0, 3, 1, 4)
        q, k, v = blk[0], blk[1], blk[2]
        attn = torch.softmax((q * scale) @ k.transpose(-2, -1), dim=-1)
        outs.append((attn @ v).transpose(1, 2).reshape(-1, C))
        j = e
    return torch.cat(outs, 0)
