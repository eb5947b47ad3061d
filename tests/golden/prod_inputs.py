"""Seeded inputs, weights and checksum helpers of the production-shape fixtures (attention_prod.npz,
ptv3_lang_prod.npz).  Pure numpy / torch: imported by make_golden_prod.py (container, next to the reference)
and by the tests (CPU oracle tests and GPU parity tests), so neither stores inputs or parameters."""
import numpy as np
import torch


def room(n_side, seed):
    """room fixture scaled down: floor n x n + two walls n x h, rows shuffled."""
    h = max(2, n_side * 72 // 256)
    xs, ys = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij")
    floor = np.stack([xs.ravel(), ys.ravel(), np.zeros(n_side * n_side, int)], 1)
    yy, zz = np.meshgrid(np.arange(n_side), np.arange(1, h + 1), indexing="ij")
    wa = np.stack([np.zeros(yy.size, int), yy.ravel(), zz.ravel()], 1)
    wb = np.stack([np.full(yy.size, n_side - 1), yy.ravel(), zz.ravel()], 1)
    gc = np.concatenate([floor, wa, wb]).astype(np.int64)
    g = torch.Generator().manual_seed(seed)
    return gc[torch.randperm(len(gc), generator=g).numpy()]


ORD = ("z", "z-trans", "hilbert", "hilbert-trans")
ATT = dict(C=768, H=16, K=1024, n=2600, order_index=2)
LANG_SIDE = 64                   # room(64): 4096 floor + 2 x 64 x 18 wall voxels = 6,400
POOL_SEED = 77                   # torch.manual_seed before each forward: SerializedPooling's randperm draws
GRAD_KEYS = ("embedding.stem.conv.weight", "enc.enc1.down.proj.weight", "enc.enc3.block5.attn.qkv.weight",
             "enc.enc3.block2.cpe.0.weight", "dec.dec2.block1.mlp.0.fc2.weight", "dec.dec1.block0.cpe.0.weight",
             "dec.dec1.up.proj_skip.0.weight", "dec.dec0.up.proj.1.weight", "dec.dec0.block0.cpe.0.weight",
             "dec.dec0.block0.attn.qkv.weight", "dec.dec0.block0.attn.qkv.bias", "dec.dec0.block1.cpe.0.weight",
             "dec.dec0.block1.cpe.1.weight", "dec.dec0.block1.attn.proj.weight", "dec.dec0.block1.mlp.0.fc1.weight",
             "dec.dec0.block1.mlp.0.fc2.weight", "dec.dec0.block1.norm2.0.weight", "dec.dec0.block1.norm1.0.bias")


def proj(t, k=4, seed=12345):
    """(rows, ...) -> (rows, k) fp32: product of the flattened rows with a seeded Gaussian matrix."""
    t2 = t.detach().reshape(t.shape[0], -1).double() if t.dim() > 1 else t.detach().reshape(-1, 1).double()
    r = torch.randn(t2.shape[1], k, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
    return (t2 @ r).float()


def row_subset(n, m, seed=4321):
    return torch.randperm(n, generator=torch.Generator().manual_seed(seed))[:m].sort().values


def att_inputs():
    """Seeded inputs / weights of the attention fixture (shared with tests)."""
    C, n = ATT["C"], ATT["n"]
    g = torch.Generator().manual_seed(21)
    gc = torch.from_numpy(room(48, 2)[:n])
    x = torch.randn(n, C, generator=g)
    cot = torch.randn(n, C, generator=g)
    sd = {"qkv.weight": torch.randn(3 * C, C, generator=g) * C ** -0.5, "qkv.bias": torch.randn(3 * C, generator=g) * 0.02,
          "proj.weight": torch.randn(C, C, generator=g) * C ** -0.5, "proj.bias": torch.randn(C, generator=g) * 0.02}
    return gc, x, cot, sd


def lang_inputs():
    """Seeded inputs of the full-model fixture (weights: oracle.ptv3.init_state_dict(cfg, seed=5))."""
    gc = torch.from_numpy(room(LANG_SIDE, 3))
    n = len(gc)
    g = torch.Generator().manual_seed(6)
    color = torch.rand(n, 3, generator=g) * 2 - 1
    opacity = torch.rand(n, 1, generator=g)
    quat = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1)
    scale = torch.rand(n, 3, generator=g) * 1.5
    feat = torch.cat([color, opacity, quat, scale], 1)
    cot = torch.randn(n, 768, generator=g)
    return gc, feat, cot


FULL_SIDE = 256                  # the benchmark's room-102400 workload itself


def full_inputs():
    """Seeded inputs of the FULL-SIZE fixture ptv3_lang_full.npz: one 102,400-Gaussian room chunk (the shape bench.py
    times), weights oracle.ptv3.init_state_dict(cfg, seed=5)."""
    gc = torch.from_numpy(room(FULL_SIDE, 0))
    n = len(gc)
    g = torch.Generator().manual_seed(16)
    color = torch.rand(n, 3, generator=g) * 2 - 1
    opacity = torch.rand(n, 1, generator=g)
    quat = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1)
    scale = torch.rand(n, 3, generator=g) * 1.5
    return gc, torch.cat([color, opacity, quat, scale], 1)


def full_cotangent(n, seed=17):
    """Seeded cotangent of the full-size backward fixture (ptv3_lang_full_bwd.npz): d loss / d feat, (n, 768)."""
    return torch.randn(n, 768, generator=torch.Generator().manual_seed(seed))
