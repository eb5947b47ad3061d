"""Oracle (test infrastructure): space-filling-curve keys, orders, window padding,
grid-pool partition.  numpy int64, scalar-spec restatement.

Follows (reference file:line, under /root/reference):
  pointcept/models/utils/serialization/default.py:8-24   encode()
  pointcept/models/utils/serialization/z_order.py:40-50  bit interleave
  pointcept/models/utils/serialization/hilbert.py:91-198 Skilling transpose->Hilbert
  pointcept/models/utils/structure.py:47-102             Point.serialization
  pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:114-170  padding
  pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:371-444  pooling
"""
import numpy as np

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")


def z_order_key(x, y, z, depth):
    """bit i of x -> 3i+2, y -> 3i+1, z -> 3i (z_order.py:40-50)."""
    x = x.astype(np.int64); y = y.astype(np.int64); z = z.astype(np.int64)
    key = np.zeros_like(x)
    for i in range(depth):
        key |= ((x >> i) & 1) << (3 * i + 2)
        key |= ((y >> i) & 1) << (3 * i + 1)
        key |= ((z >> i) & 1) << (3 * i)
    return key


def hilbert_key(x, y, z, depth):
    """Skilling (2004) transpose form, 3 dims (hilbert.py:156-181)."""
    X = [x.astype(np.int64).copy(), y.astype(np.int64).copy(), z.astype(np.int64).copy()]
    q = 1 << (depth - 1) if depth > 0 else 0
    while q > 1:
        p = q - 1
        for i in range(3):
            on = (X[i] & q) != 0
            # bit on: invert the low bits of dim 0; bit off: swap low bits of dim 0 and dim i
            t = np.where(on, 0, (X[0] ^ X[i]) & p)
            X[0] = np.where(on, X[0] ^ p, X[0] ^ t)
            X[i] = X[i] ^ t if i != 0 else X[0]
        q >>= 1
    g = z_order_key(X[0], X[1], X[2], depth)  # MSB-first interleave == Morton interleave
    h = g.copy()
    s = 1
    while s < 3 * depth:
        h ^= h >> s
        s <<= 1
    return h


def encode(grid_coord, batch, depth, order):
    """code = (batch << 3*depth) | key  (default.py:8-24)."""
    gc = np.asarray(grid_coord).astype(np.int64)
    x, y, z = gc[:, 0], gc[:, 1], gc[:, 2]
    if order == "z":
        key = z_order_key(x, y, z, depth)
    elif order == "z-trans":
        key = z_order_key(y, x, z, depth)
    elif order == "hilbert":
        key = hilbert_key(x, y, z, depth)
    elif order == "hilbert-trans":
        key = hilbert_key(y, x, z, depth)
    else:
        raise NotImplementedError(order)
    if batch is not None:
        key = (np.asarray(batch).astype(np.int64) << (3 * depth)) | key
    return key


def serialization_depth(grid_coord):
    """int(grid_coord.max()).bit_length()  (structure.py:64-67)."""
    return int(np.asarray(grid_coord).max()).bit_length()


def serialize(grid_coord, batch, orders=ORDERS, depth=None):
    """Returns (code, order, inverse, depth), each (len(orders), N) int64.
    argsort ties are broken by row index (stable); the reference's torch.argsort is
    unspecified on ties, which only occur for duplicate voxels (structure.py:85-92)."""
    if depth is None:
        depth = serialization_depth(grid_coord)
    code = np.stack([encode(grid_coord, batch, depth, o) for o in orders])
    order = np.stack([np.argsort(c, kind="stable") for c in code]).astype(np.int64)
    inverse = np.empty_like(order)
    ar = np.arange(code.shape[1], dtype=np.int64)
    for k in range(code.shape[0]):
        inverse[k, order[k]] = ar
    return code, order, inverse, depth


def offset2bincount(offset):
    offset = np.asarray(offset).astype(np.int64)
    return np.diff(offset, prepend=0)


def offset2batch(offset):
    bc = offset2bincount(offset)
    return np.repeat(np.arange(len(bc), dtype=np.int64), bc)


def padding(offset, patch_size):
    """pad / unpad / cu_seqlens of SerializedAttention.get_padding_and_inverse
    (point_transformer_v3m1_base.py:114-170).  A batch element with more than K points
    is padded to a multiple of K by borrowing the last K-r slots of the previous
    window; an element with <= K points is one short window."""
    K = int(patch_size)
    bc = offset2bincount(offset)
    bc_pad = np.where(bc > K, (bc + K - 1) // K * K, bc)
    _off = np.concatenate([[0], np.cumsum(bc)])
    _off_pad = np.concatenate([[0], np.cumsum(bc_pad)])
    pad = np.arange(_off_pad[-1], dtype=np.int64)
    unpad = np.arange(_off[-1], dtype=np.int64)
    cu = []
    for i in range(len(bc)):
        unpad[_off[i]:_off[i + 1]] += _off_pad[i] - _off[i]
        if bc[i] != bc_pad[i]:
            r = bc[i] % K
            pad[_off_pad[i + 1] - K + r:_off_pad[i + 1]] = \
                pad[_off_pad[i + 1] - 2 * K + r:_off_pad[i + 1] - K]
        pad[_off_pad[i]:_off_pad[i + 1]] -= _off_pad[i] - _off[i]
        cu.append(np.arange(_off_pad[i], _off_pad[i + 1], K, dtype=np.int32))
    cu_seqlens = np.concatenate(cu + [np.array([_off_pad[-1]], dtype=np.int32)]).astype(np.int32)
    return pad, unpad, cu_seqlens


def pool_partition(code, stride_depth=1):
    """Grid-pool partition of SerializedPooling.forward (ptv3:384-398).
    code: (k, N) int64 serialization codes.  Returns
      cluster (N)      pooling_inverse: rank of code[0]>>3d among its unique values
      indices (N)      rows sorted by cluster (stable)
      idx_ptr (n'+1)   CSR pointer
      head (n')        indices[idx_ptr[:-1]]
      new_code (k,n')  code[:, head] >> 3d
    """
    c = code >> (3 * stride_depth)
    _, cluster, counts = np.unique(c[0], return_inverse=True, return_counts=True)
    cluster = cluster.astype(np.int64)
    indices = np.argsort(cluster, kind="stable").astype(np.int64)
    idx_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    head = indices[idx_ptr[:-1]]
    return cluster, indices, idx_ptr, head, c[:, head]
