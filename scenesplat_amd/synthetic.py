"""Synthetic 3DGS chunks with the hot-path input contract (SURVEY 8a row a0 / 8d "room").

room(n_side): floor n x n voxels + two walls n x (n*72//256) -> unique voxels, rows shuffled so
memory order != curve order.  n_side=256 gives exactly 102,400 Gaussians (stage sizes after
stride-2 pooling: 102400 / 25600 / 6400 / 1600)."""
import numpy as np
import torch


def room_grid(n_side, seed=0, walls=True):
    """walls=False: the floor alone -- "room-4096" of BASELINE config 1 is room_grid(64, walls=False): 64 x 64 = 4,096 voxels."""
    h = max(2, n_side * 72 // 256) if walls else 0
    xs, ys = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij")
    floor = np.stack([xs.ravel(), ys.ravel(), np.zeros(n_side * n_side, int)], 1)
    yy, zz = np.meshgrid(np.arange(n_side), np.arange(1, h + 1), indexing="ij")
    wa = np.stack([np.zeros(yy.size, int), yy.ravel(), zz.ravel()], 1)
    wb = np.stack([np.full(yy.size, n_side - 1), yy.ravel(), zz.ravel()], 1)
    gc = np.concatenate([floor, wa, wb]).astype(np.int64)
    perm = torch.randperm(len(gc), generator=torch.Generator().manual_seed(seed)).numpy()
    return gc[perm]


def room_chunk(n_side=256, seed=0, lang_dim=768, num_classes=20, batch=1, walls=True):
    """Input dict (CPU tensors) of `batch` chunks: coord, grid_coord, feat (11 = color3, opacity1,
    quat4, scale3), lang_feat (unit rows), valid_feat_mask (90 %), segment (-1 = ignore), offset."""
    parts = []
    for b in range(batch):
        g = torch.Generator().manual_seed(seed + b)
        gc = torch.from_numpy(room_grid(n_side, seed + b, walls))
        n = len(gc)
        coord = gc.float() * 0.02 + torch.rand(n, 3, generator=g) * 0.02
        color = torch.rand(n, 3, generator=g) * 2 - 1
        opacity = torch.rand(n, 1, generator=g)
        quat = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1)
        scale = torch.rand(n, 3, generator=g) * 1.5
        feat = torch.cat([color, opacity, quat, scale], 1)
        lang = torch.nn.functional.normalize(torch.randn(n, lang_dim, generator=g), dim=1) if lang_dim else None
        mask = torch.rand(n, generator=g) < 0.9
        seg = ((gc[:, 0] // max(1, n_side // 8)) + 8 * (gc[:, 1] // max(1, n_side // 8))) % num_classes
        seg[torch.rand(n, generator=g) < 0.1] = -1
        parts.append(dict(coord=coord, grid_coord=gc, feat=feat, lang_feat=lang, valid_feat_mask=mask, segment=seg))
    out = {k: torch.cat([p[k] for p in parts]) for k in parts[0] if parts[0][k] is not None}
    out["offset"] = torch.cumsum(torch.tensor([len(p["coord"]) for p in parts]), 0)
    return out


def uniform_chunk(n=102400, extent=(300, 300, 150), seed=0):
    """SURVEY 8d secondary fixture "uniform-102400": n unique voxels uniform in a box (depth 9) -- the stress case:
    ~1-2 occupied neighbours per site, irregular neighbour masks, pooled levels that barely shrink, window tails."""
    g = torch.Generator().manual_seed(seed)
    ex = torch.tensor(extent)
    tot = int(ex.prod())
    lin = torch.randperm(tot, generator=g)[:n]
    gc = torch.stack([lin // (extent[1] * extent[2]), (lin // extent[2]) % extent[1], lin % extent[2]], 1).long()
    coord = gc.float() * 0.02 + torch.rand(n, 3, generator=g) * 0.02
    feat = torch.cat([torch.rand(n, 3, generator=g) * 2 - 1, torch.rand(n, 1, generator=g),
                      torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1), torch.rand(n, 3, generator=g) * 1.5], 1)
    return dict(coord=coord, grid_coord=gc, feat=feat, offset=torch.tensor([n]))


LANG_PTV3 = dict(  # configs/concat_dataset/lang-pretrain-concat-scan-ppv2-matt-mcmc-wo-normal-contrastive.py:20-54
    in_channels=11, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2, 2),
    enc_depths=(2, 2, 2, 6), enc_channels=(32, 64, 128, 256), enc_num_head=(2, 4, 8, 16),
    enc_patch_size=(1024, 1024, 1024, 1024), dec_depths=(2, 2, 2), dec_channels=(768, 512, 256),
    dec_num_head=(16, 16, 16), dec_patch_size=(1024, 1024, 1024), mlp_ratio=4, qkv_bias=True, qk_scale=None,
    attn_drop=0.0, proj_drop=0.0, drop_path=0.3, shuffle_orders=True, pre_norm=True, enable_rpe=False,
    enable_flash=True, upcast_attention=False, upcast_softmax=False, cls_mode=False)
