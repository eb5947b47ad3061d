"""GPU-side data fast path: GridSample(mode="train") on the device.

Reference: pointcept/datasets/transform.py:1182-1300 -- voxelise at `grid_size`, keep ONE random point per
occupied voxel, return its integer grid coordinate (and the point->voxel inverse).  The CPU version hashes
(FNV-1a), argsorts and uniques with numpy per sample inside dataloader workers; here the voxel key is the exact
packed coordinate (no hash collisions), sorted with the library's radix argsort and segmented with the grid-pool
partition kernel -- the same two kernels the PTv3 plan uses."""
import torch

from . import native as nv


@torch.no_grad()
def grid_sample_train(coord, grid_size, generator=None, return_inverse=False):
    """coord (N, 3) float GPU tensor -> dict(idx_unique (n,), grid_coord (n, 3) int32 [, inverse (N,)])."""
    if not coord.is_cuda:
        raise RuntimeError("grid_sample_train: GPU tensor required (no CPU fallback)")
    gc = torch.floor(coord / grid_size).to(torch.int64)
    gc = gc - gc.amin(0, keepdim=True)
    if int(gc.max()) >= (1 << 21):
        raise ValueError("grid extent exceeds 21 bits per axis")
    key = ((gc[:, 0] << 42) | (gc[:, 1] << 21) | gc[:, 2]).unsqueeze(0).contiguous()
    order, _, _ = nv.argsort_i64(key, 63, want_inverse=False, want_sorted=False)
    order = order[0]
    cluster, idx_ptr, head, n_out = nv.pool_partition(key[0], order, 0)
    n = int(n_out.item())
    ptr = idx_ptr[:n + 1].long()
    count = ptr[1:] - ptr[:-1]
    r = torch.randint(0, 1 << 30, (n,), device=coord.device, generator=generator)
    idx_unique = order[(ptr[:-1] + r % count)].long()          # one random member per voxel (transform.py:1263-1267)
    out = dict(idx_unique=idx_unique, grid_coord=gc[idx_unique].to(torch.int32), count=count)
    if return_inverse:
        out["inverse"] = cluster.long()
    return out
