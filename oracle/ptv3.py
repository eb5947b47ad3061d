"""Oracle (test infrastructure): functional fp32 CPU restatement of the reference
PointTransformerV3 forward (PT-v3m1) driven by a plain state dict.

Follows /root/reference/pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py
  Embedding :485-515, Block :318-338, SerializedAttention :172-222, MLP :225-248,
  SerializedPooling :371-444, SerializedUnpooling :471-482, PointTransformerV3 :699-714
and pointcept/models/modules.py:58-91 (PointSequential feature mirroring, incl. the
"stale CPE input" of the first decoder block of every stage: the conv reads
sparse_conv_feat.features, which after unpooling still holds proj_skip(skip) only).

State-dict keys are the reference's.  drop_path is 0 (identity) and
the per-level curve permutation of shuffle_orders is an explicit argument so results are
deterministic.  Autograd works through every float op, so gradients w.r.t. the state-dict
tensors and the input features are the oracle for the backward pass too.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from . import serialization as ser

DEFAULT_CFG = dict(
    in_channels=11,
    order=("z", "z-trans", "hilbert", "hilbert-trans"),
    stride=(2, 2, 2),
    enc_depths=(2, 2, 2, 6),
    enc_channels=(32, 64, 128, 256),
    enc_num_head=(2, 4, 8, 16),
    enc_patch_size=(1024, 1024, 1024, 1024),
    dec_depths=(2, 2, 2),
    dec_channels=(768, 512, 256),
    dec_num_head=(16, 16, 16),
    dec_patch_size=(1024, 1024, 1024),
    mlp_ratio=4,
)

BN_EPS, BN_MOM = 1e-3, 0.01  # ptv3:581


class Level:
    """Integer structure of one resolution level (everything Point carries besides feat)."""

    def __init__(self, grid_coord, batch, code, order, inverse, depth):
        self.grid_coord, self.batch = grid_coord, batch
        self.code, self.order, self.inverse, self.depth = code, order, inverse, depth
        self.n = len(batch)
        self.offset = np.cumsum(np.bincount(batch, minlength=int(batch.max()) + 1))
        self.nbr = {}
        self.pads = {}
        # set on pooled levels
        self.cluster = self.indices = self.idx_ptr = None

    def neighbors(self, k):
        if k not in self.nbr:
            self.nbr[k] = ops.neighbor_table(self.grid_coord, self.batch, k)
        return self.nbr[k]

    def padding(self, K):
        if K not in self.pads:
            self.pads[K] = ser.padding(self.offset, K)
        return self.pads[K]


def build_levels(grid_coord, offset, orders, strides, perms=None):
    """perms: optional list (len = 1 + len(strides)) of curve permutations emulating
    shuffle_orders (structure.py:94-98, ptv3:408-412)."""
    gc = np.asarray(grid_coord).astype(np.int64)
    batch = ser.offset2batch(offset)
    code, order, inverse, depth = ser.serialize(gc, batch, orders)
    if perms is not None:
        p = np.asarray(perms[0]); code, order, inverse = code[p], order[p], inverse[p]
    levels = [Level(gc, batch, code, order, inverse, depth)]
    for s, stride in enumerate(strides):
        prev = levels[-1]
        pd = (int(stride) - 1).bit_length()
        if pd > prev.depth:
            pd = 0
        cluster, indices, idx_ptr, head, ncode = ser.pool_partition(prev.code, pd)
        norder = np.stack([np.argsort(c, kind="stable") for c in ncode]).astype(np.int64)
        ninv = np.empty_like(norder)
        for k in range(len(ncode)):
            ninv[k, norder[k]] = np.arange(ncode.shape[1])
        if perms is not None:
            p = np.asarray(perms[s + 1]); ncode, norder, ninv = ncode[p], norder[p], ninv[p]
        lv = Level(prev.grid_coord[head] >> pd, prev.batch[head], ncode, norder, ninv, prev.depth - pd)
        lv.cluster, lv.indices, lv.idx_ptr = cluster, indices, idx_ptr
        levels.append(lv)
    return levels


def _bn(x, sd, prefix, training, stats_out):
    rm = sd[prefix + "running_mean"].detach().clone()
    rv = sd[prefix + "running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[prefix + "weight"], sd[prefix + "bias"], training, BN_MOM, BN_EPS)
    if training and stats_out is not None:
        stats_out[prefix + "running_mean"] = rm
        stats_out[prefix + "running_var"] = rv
    return y


def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"], sd[prefix + "bias"], 1e-5)


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + "weight"], sd[prefix + "bias"])


def attention(x, sd, prefix, lv, order_index, num_heads, patch_size):
    """SerializedAttention.forward with cu_seqlens semantics (ptv3:172-222)."""
    C = x.shape[1]
    pad, unpad, cu = lv.padding(patch_size)
    order = torch.as_tensor(lv.order[order_index][pad])
    inverse = torch.as_tensor(unpad[lv.inverse[order_index]])
    qkv = _lin(x, sd, prefix + "qkv.")[order]
    scale = (C // num_heads) ** -0.5
    feat = ops.window_attention(qkv, cu, num_heads, scale)
    return _lin(feat[inverse], sd, prefix + "proj.")


def block(x, conv_in, sd, prefix, lv, order_index, num_heads, patch_size):
    """Block.forward (ptv3:318-338); conv_in is sparse_conv_feat.features."""
    cpe = ops.subm_conv3d(conv_in, sd[prefix + "cpe.0.weight"], sd[prefix + "cpe.0.bias"], lv.neighbors(3))
    cpe = _ln(_lin(cpe, sd, prefix + "cpe.1."), sd, prefix + "cpe.2.")
    x = x + cpe
    x = x + attention(_ln(x, sd, prefix + "norm1.0."), sd, prefix + "attn.", lv, order_index, num_heads, patch_size)
    h = _ln(x, sd, prefix + "norm2.0.")
    h = _lin(F.gelu(_lin(h, sd, prefix + "mlp.0.fc1.")), sd, prefix + "mlp.0.fc2.")
    return x + h


def forward(sd, cfg, feat, grid_coord, offset, bn_training=False, perms=None, stats_out=None,
            levels=None, taps=None):
    """Returns (N, dec_channels[0]) features.  taps: optional dict filled with
    intermediate tensors (for per-module parity checks)."""
    cfg = {**DEFAULT_CFG, **cfg}
    orders = cfg["order"]; no = len(orders)
    S = len(cfg["enc_depths"])
    if levels is None:
        levels = build_levels(grid_coord, offset, orders, cfg["stride"], perms)
    lv = levels[0]
    x = ops.subm_conv3d(feat, sd["embedding.stem.conv.weight"], None, lv.neighbors(5))
    x = F.gelu(_bn(x, sd, "embedding.stem.norm.", bn_training, stats_out))
    if taps is not None:
        taps["embedding"] = x
    skips = []
    for s in range(S):
        if s > 0:
            skips.append(x)
            lv = levels[s]
            p = f"enc.enc{s}.down."
            y = _lin(x, sd, p + "proj.")[torch.as_tensor(lv.indices)]
            y = ops.segment_csr(y, lv.idx_ptr, "mean")
            x = F.gelu(_bn(y, sd, p + "norm.0.", bn_training, stats_out))
        for i in range(cfg["enc_depths"][s]):
            x = block(x, x, sd, f"enc.enc{s}.block{i}.", lv, i % no, cfg["enc_num_head"][s], cfg["enc_patch_size"][s])
        if taps is not None:
            taps[f"enc{s}"] = x
    for s in reversed(range(S - 1)):
        child = levels[s + 1]
        lv = levels[s]
        p = f"dec.dec{s}.up."
        up = F.gelu(_bn(_lin(x, sd, p + "proj.0."), sd, p + "proj.1.", bn_training, stats_out))
        skip = F.gelu(_bn(_lin(skips[s], sd, p + "proj_skip.0."), sd, p + "proj_skip.1.", bn_training, stats_out))
        x = skip + up[torch.as_tensor(child.cluster)]
        conv_in = skip  # stale sparse_conv_feat (modules.py:64-75, ptv3:476-478)
        for i in range(cfg["dec_depths"][s]):
            x = block(x, conv_in if i == 0 else x, sd, f"dec.dec{s}.block{i}.", lv, i % no,
                      cfg["dec_num_head"][s], cfg["dec_patch_size"][s])
        if taps is not None:
            taps[f"dec{s}"] = x
    return x


def init_state_dict(cfg, seed=0, dtype=torch.float32):
    """Random state dict with the reference's key set/shapes.
    Values are NOT the reference's init distribution; used for parity of forward math."""
    cfg = {**DEFAULT_CFG, **cfg}
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def lin(p, o, i):
        sd[p + "weight"] = torch.randn(o, i, generator=g, dtype=dtype) * (i ** -0.5)
        sd[p + "bias"] = torch.randn(o, generator=g, dtype=dtype) * 0.02

    def bn(p, c):
        sd[p + "weight"] = 1 + 0.1 * torch.randn(c, generator=g, dtype=dtype)
        sd[p + "bias"] = 0.1 * torch.randn(c, generator=g, dtype=dtype)
        sd[p + "running_mean"] = 0.1 * torch.randn(c, generator=g, dtype=dtype)
        sd[p + "running_var"] = 1 + 0.1 * torch.rand(c, generator=g, dtype=dtype)
        sd[p + "num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    def ln(p, c):
        sd[p + "weight"] = 1 + 0.1 * torch.randn(c, generator=g, dtype=dtype)
        sd[p + "bias"] = 0.1 * torch.randn(c, generator=g, dtype=dtype)

    def blk(p, c):
        sd[p + "cpe.0.weight"] = torch.randn(c, 3, 3, 3, c, generator=g, dtype=dtype) * ((9 * c) ** -0.5)
        sd[p + "cpe.0.bias"] = torch.randn(c, generator=g, dtype=dtype) * 0.02
        lin(p + "cpe.1.", c, c); ln(p + "cpe.2.", c); ln(p + "norm1.0.", c)
        lin(p + "attn.qkv.", 3 * c, c); lin(p + "attn.proj.", c, c); ln(p + "norm2.0.", c)
        hid = int(c * cfg["mlp_ratio"])
        lin(p + "mlp.0.fc1.", hid, c); lin(p + "mlp.0.fc2.", c, hid)

    ec, dc = list(cfg["enc_channels"]), list(cfg["dec_channels"]) + [cfg["enc_channels"][-1]]
    c0 = ec[0]
    sd["embedding.stem.conv.weight"] = torch.randn(c0, 5, 5, 5, cfg["in_channels"], generator=g, dtype=dtype) * ((25 * cfg["in_channels"]) ** -0.5)
    bn("embedding.stem.norm.", c0)
    for s in range(len(ec)):
        if s > 0:
            lin(f"enc.enc{s}.down.proj.", ec[s], ec[s - 1]); bn(f"enc.enc{s}.down.norm.0.", ec[s])
        for i in range(cfg["enc_depths"][s]):
            blk(f"enc.enc{s}.block{i}.", ec[s])
    for s in reversed(range(len(ec) - 1)):
        lin(f"dec.dec{s}.up.proj.0.", dc[s], dc[s + 1]); bn(f"dec.dec{s}.up.proj.1.", dc[s])
        lin(f"dec.dec{s}.up.proj_skip.0.", dc[s], ec[s]); bn(f"dec.dec{s}.up.proj_skip.1.", dc[s])
        for i in range(cfg["dec_depths"][s]):
            blk(f"dec.dec{s}.block{i}.", dc[s])
    return sd
