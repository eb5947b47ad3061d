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


@torch.no_grad()
def grid_sample_test(coord, grid_size, return_inverse=False):
    """GridSample(mode="test") on the device (pointcept/datasets/transform.py:1302-1330): the fragment generator of the
    open-vocabulary tester (engines/test.py:300-351).  A voxel with c points contributes its (i mod c)-th member to fragment i,
    i = 0 .. max(c) - 1, so every fragment holds ONE point per occupied voxel and the fragments together cover every point.
    -> dict(index (P, n_vox) int64: row p = the `index` of fragment p; grid_coord (n_vox, 3) int32 (the same for every fragment:
    members of a voxel share it); count (n_vox,) [, inverse (N,)]).  Members are enumerated in row order (stable sort; the
    reference's numpy argsort leaves the order inside a voxel unspecified)."""
    if not coord.is_cuda:
        raise RuntimeError("grid_sample_test: GPU tensor required (no CPU fallback)")
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
    parts = int(count.max())
    sel = ptr[:-1].unsqueeze(0) + torch.arange(parts, device=coord.device).unsqueeze(1) % count.unsqueeze(0)       # (P, n_vox)
    index = order.long()[sel]
    out = dict(index=index, grid_coord=gc[index[0]].to(torch.int32), count=count)
    if return_inverse:
        out["inverse"] = cluster.long()
    return out


@torch.no_grad()
def open_vocab_fragments(model, data_dict, text_embeddings, grid_size, chunk_size=600000, topk=None, confidence_threshold=0.1,
                         ignore_index=-1, feat_keys=None):
    """The fragment loop of ZeroShotSemSegTester.test (pointcept/engines/test.py:300-378) on the device: GridSample(mode="test")
    fragments -> model(fragment, chunk_size)["point_feat"]["feat"] -> pred[index] += sigmoid(feat text^T) (fused scan +
    accumulate, csrc/scan.hip) -> per point either the top-k classes (ScanNet++: `pred.topk(3)`, test.py:371-374) or
    arg-max with the confidence threshold (test.py:376-378).  data_dict: coord (N, 3), feat (N, C) (or the `feat_keys` columns to
    concatenate, as Collect does), all on the GPU.  -> (pred_labels (N,) | (N, k) int64, pred (N, classes) f32)."""
    coord = data_dict["coord"]
    feat = data_dict["feat"] if feat_keys is None else torch.cat([data_dict[k].float() for k in feat_keys], 1)
    frag = grid_sample_test(coord, grid_size)
    text = text_embeddings.to(torch.bfloat16).contiguous()
    pred = torch.zeros((coord.shape[0], text.shape[0]), dtype=torch.float32, device=coord.device)
    nvox = frag["index"].shape[1]
    off = torch.tensor([nvox], device=coord.device)
    for p in range(frag["index"].shape[0]):
        idx = frag["index"][p]
        inp = dict(coord=coord[idx], grid_coord=frag["grid_coord"], feat=feat[idx].contiguous(), offset=off)
        f = model(inp, chunk_size=chunk_size)["point_feat"]["feat"]
        nv.feat_text_scan(f, text, want_max=False, idx=idx.to(torch.int32).contiguous(), pred_accum=pred)
    if topk is not None:
        return pred.topk(int(topk), dim=1)[1], pred
    mx, am = pred.max(1)
    am = am.clone()
    am[mx < confidence_threshold] = ignore_index
    return am, pred


def _take_rows(t, idx32):
    """t[idx] for a per-point tensor: wide contiguous rows go through the library's row-gather kernel."""
    if t.dim() == 2 and t.is_contiguous() and (t.shape[1] * t.element_size()) % 16 == 0 and t.dtype in (torch.float32, torch.bfloat16, torch.float16):
        return nv.gather_rows(t, idx32)
    return t[idx32.long()]


# the per-point entries SphereCrop crops (transform.py:1511-1546); every other entry of the dict is passed through untouched
SPHERE_CROP_KEYS = ("coord", "origin_coord", "grid_coord", "color", "quat", "scale", "opacity", "normal", "lang_feat",
                    "valid_feat_mask", "segment", "instance", "displacement", "strength")


@torch.no_grad()
def sphere_crop(data_dict, point_max=80000, sample_rate=None, mode="random", generator=None, center_index=None, extra_keys=()):
    """SphereCrop(mode="random" | "center") on the device (pointcept/datasets/transform.py:1420-1548): when the sample has
    more than point_max points keep the point_max points nearest to a centre point -- a uniformly random point, or the
    middle row -- in ascending distance order (ties by row index).  The distance sort is the library's stable radix
    argsort on the float bit pattern (squared distances are non-negative, so the IEEE bits are order-preserving).
    Cropped: exactly the keys the reference enumerates (SPHERE_CROP_KEYS) plus `extra_keys`; anything else -- also a tensor
    whose first dimension happens to equal N -- is passed through, as in the reference.  center_index overrides the random
    draw (replaying a recorded centre: tests/golden/transforms.npz)."""
    if mode not in ("random", "center"):
        raise NotImplementedError('sphere_crop: mode must be "random" or "center" (mode "all" is the tester\'s CPU fragment loop)')
    coord = data_dict["coord"]
    if not coord.is_cuda:
        raise RuntimeError("sphere_crop: GPU tensors required (no CPU fallback)")
    n = coord.shape[0]
    pm = int(sample_rate * n) if sample_rate is not None else int(point_max)
    if n <= pm:
        return data_dict
    if center_index is not None:
        ci = int(center_index)
    elif mode == "random":
        ci = int(torch.randint(0, n, (1,), generator=generator, device=coord.device if generator is not None and generator.device.type == "cuda" else "cpu"))
    else:
        ci = n // 2
    d2 = (coord - coord[ci]).square().sum(1).float().contiguous()
    key = d2.view(torch.int32).to(torch.int64).unsqueeze(0).contiguous()
    order, _, _ = nv.argsort_i64(key, 32, want_inverse=False, want_sorted=False)
    idx = order[0, :pm].contiguous()
    out = dict(data_dict)
    for k in tuple(SPHERE_CROP_KEYS) + tuple(extra_keys):
        v = data_dict.get(k)
        if isinstance(v, torch.Tensor):
            if v.shape[0] != n:
                raise ValueError(f"sphere_crop: '{k}' has {v.shape[0]} rows, coord has {n}")
            out[k] = _take_rows(v, idx)
    return out


@torch.no_grad()
def collect(data_dict, keys, offset_keys_dict=None, **kwargs):
    """Collect (transform.py:320-352): pick `keys`, record `offset` = number of points of `coord`, and build every
    `<name>_keys=(...)` entry as the float32 column-concatenation of the listed tensors, e.g.
    feat_keys=("color", "opacity", "quat", "scale") -> feat (N, 11) for the SceneSplat language configs."""
    if isinstance(keys, str):
        keys = [keys]
    offset_keys = dict(offset="coord") if offset_keys_dict is None else offset_keys_dict
    data = {k: data_dict[k] for k in keys if k in data_dict}
    for k, v in offset_keys.items():
        data[k] = torch.tensor([data_dict[v].shape[0]], device=data_dict[v].device)
    for name, ks in kwargs.items():
        data[name.replace("_keys", "")] = torch.cat([data_dict[k].float() for k in ks], dim=1)
    return data


@torch.no_grad()
def point_collate(batch, mix_prob=0.0, rng=None):
    """point_collate_fn (pointcept/datasets/utils.py:8-48) for a list of per-sample dicts of device tensors: tensors are
    concatenated along the points, every key containing "offset" becomes the running total, strings are listed, scalars
    stacked; with probability mix_prob (host RNG, as the reference) Mix3D merges neighbouring samples pairwise by keeping
    every second offset (the merged element then holds duplicate voxels, which the plan tolerates)."""
    import random
    if not batch or not isinstance(batch[0], dict):
        raise TypeError("point_collate expects a list of dicts")
    out = {}
    for key in batch[0]:
        vals = [d[key] for d in batch]
        v0 = vals[0]
        if isinstance(v0, torch.Tensor):
            out[key] = torch.cat([v if v.dim() > 0 else v.reshape(1) for v in vals])
        elif isinstance(v0, str):
            out[key] = list(vals)
        else:
            out[key] = torch.as_tensor(vals)
    for key in out:
        if "offset" in key:
            out[key] = torch.cumsum(out[key], dim=0)
    if "offset" in out and (rng or random).random() < mix_prob:
        off = out["offset"]
        out["offset"] = torch.cat([off[1:-1:2], off[-1:]], dim=0)
    return out
