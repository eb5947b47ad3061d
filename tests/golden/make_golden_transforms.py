"""Golden vectors of the data-path transforms from the REFERENCE's own Python (container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_transforms.py

Imports /root/reference/pointcept/datasets/transform.py and datasets/utils.py (plain numpy / torch; the package __init__ files
are not run, same recipe as make_golden.py) and records what GridSample(mode="train"), SphereCrop, Collect and
point_collate_fn produce on seeded inputs -> tests/golden/transforms.npz (data only; no reference source).

What is comparable with a GPU implementation that draws its own random numbers:
  GridSample(train)   the SET of occupied voxels (grid_coord rows, sorted), the points-per-voxel counts, the point -> voxel
                      partition (`inverse`, up to a renumbering); which member represents a voxel is random by definition
  SphereCrop          mode="center": the kept rows IN ORDER (deterministic); mode="random": replayed with the centre the
                      reference drew (stored)
  Collect             keys, feature concatenation, offset
  point_collate_fn    concatenation, running offsets, the Mix3D offset merge (host RNG replayed through random.seed)
"""
import importlib
import os
import random
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
R = "/root/reference/"


def sample(n, seed):
    """A seeded SceneSplat-like sample (numpy, as the datasets hand it to the transforms): a slab of Gaussians."""
    g = np.random.RandomState(seed)
    coord = (g.rand(n, 3) * np.array([4.0, 3.0, 1.5])).astype(np.float32)
    return dict(coord=coord, color=(g.rand(n, 3) * 2 - 1).astype(np.float32), opacity=g.rand(n, 1).astype(np.float32),
                quat=g.randn(n, 4).astype(np.float32), scale=g.rand(n, 3).astype(np.float32),
                segment=g.randint(-1, 20, n).astype(np.int64), lang_feat=g.randn(n, 16).astype(np.float32),
                valid_feat_mask=(g.rand(n) < 0.9).astype(np.int64), name="scene%d" % seed)


def main():
    import make_golden as mg
    mg.stubpkg("pointcept", R + "pointcept")
    mg.stubpkg("pointcept.utils", R + "pointcept/utils")
    mg.stubpkg("pointcept.datasets", R + "pointcept/datasets")
    T = importlib.import_module("pointcept.datasets.transform")
    U = importlib.import_module("pointcept.datasets.utils")
    fx = {}

    # ---- GridSample(train) at the SceneSplat grid size (2 cm), keys of the language configs ----
    d = sample(6000, 1)
    fx["gs_seed"], fx["gs_n"], fx["gs_grid"] = np.int64(1), np.int64(6000), np.float64(0.1)
    keys = ("coord", "color", "opacity", "quat", "scale", "segment", "lang_feat", "valid_feat_mask")
    np.random.seed(11)
    out = T.GridSample(grid_size=0.1, hash_type="fnv", mode="train", keys=keys, return_grid_coord=True, return_inverse=True)(dict(d))
    gc = out["grid_coord"]
    order = np.lexsort((gc[:, 2], gc[:, 1], gc[:, 0]))
    fx["gs_grid_coord_sorted"] = gc[order].astype(np.int32)
    fx["gs_inverse"] = out["inverse"].astype(np.int64)                    # voxel id per input point (reference numbering)
    fx["gs_n_out"] = np.int64(len(gc))
    # the kept coordinate of every voxel lies in that voxel (self-check of the recorded run)
    assert np.array_equal(np.floor(out["coord"] / 0.1).astype(int) - np.floor(d["coord"] / 0.1).astype(int).min(0), gc)

    # ---- GridSample(test): the tester's fragment generator (transform.py:1302-1330) on the same sample ----
    parts = T.GridSample(grid_size=0.1, hash_type="fnv", mode="test", keys=keys, return_grid_coord=True)(dict(d))
    fx["gst_nparts"] = np.int64(len(parts))
    fx["gst_index"] = np.stack([p_["index"] for p_ in parts]).astype(np.int32)            # (parts, n_vox): the `index` of every fragment
    assert all(np.array_equal(p_["grid_coord"], parts[0]["grid_coord"]) for p_ in parts) and fx["gst_index"].shape[1] == len(gc)

    # ---- SphereCrop: center (deterministic) and random (centre replayed) ----
    d = sample(5000, 2)
    fx["sc_seed"], fx["sc_n"], fx["sc_point_max"] = np.int64(2), np.int64(5000), np.int64(1800)
    out = T.SphereCrop(point_max=1800, mode="center")(dict(d))
    fx["sc_center_coord"] = out["coord"]
    fx["sc_center_segment"] = out["segment"]
    fx["sc_center_lang"] = out["lang_feat"]
    np.random.seed(5)
    ci = np.random.randint(d["coord"].shape[0])          # what SphereCrop(mode="random") draws first
    np.random.seed(5)
    out = T.SphereCrop(point_max=1800, mode="random")(dict(d))
    fx["sc_random_center_index"] = np.int64(ci)
    fx["sc_random_coord"] = out["coord"]
    fx["sc_random_opacity"] = out["opacity"]
    out = T.SphereCrop(sample_rate=0.25, mode="center")(dict(d))
    fx["sc_rate_coord"] = out["coord"]
    small = sample(100, 3)
    out = T.SphereCrop(point_max=1800, mode="center")(dict(small))
    assert out["coord"].shape[0] == 100                   # fewer points than point_max: untouched

    # ---- Collect with the feat_keys of the SceneSplat language configs ----
    d = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sample(700, 4).items()}
    d["grid_coord"] = torch.floor(d["coord"] / 0.02).long()
    out = T.Collect(keys=("coord", "grid_coord", "segment", "lang_feat", "valid_feat_mask", "name"),
                    feat_keys=("color", "opacity", "quat", "scale"))(dict(d))
    fx["co_seed"], fx["co_n"] = np.int64(4), np.int64(700)
    fx["co_keys"] = np.array(sorted(out.keys()))
    fx["co_feat"] = out["feat"].numpy()
    fx["co_offset"] = out["offset"].numpy()

    # ---- point_collate_fn: plain and Mix3D ----
    batch = []
    for s, n in ((5, 300), (6, 450), (7, 280), (8, 510)):
        d = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sample(n, s).items()}
        d["grid_coord"] = torch.floor(d["coord"] / 0.02).long()
        batch.append(T.Collect(keys=("coord", "grid_coord", "segment", "name"), feat_keys=("color", "opacity", "quat", "scale"))(d))
    fx["pc_sizes"] = np.array([300, 450, 280, 510]); fx["pc_seeds"] = np.array([5, 6, 7, 8])
    out = U.point_collate_fn([dict(b) for b in batch], mix_prob=0.0)
    fx["pc_offset"] = out["offset"].numpy(); fx["pc_feat_sum"] = out["feat"].double().sum(0).numpy()
    fx["pc_coord"] = out["coord"].numpy(); fx["pc_names"] = np.array(out["name"])
    random.seed(3)
    r0 = random.random()                                  # the draw point_collate_fn compares with mix_prob
    random.seed(3)
    out = U.point_collate_fn([dict(b) for b in batch], mix_prob=1.0)
    fx["pc_mix_draw"] = np.float64(r0)
    fx["pc_mix_offset"] = out["offset"].numpy()
    odd = U.point_collate_fn([dict(b) for b in batch[:3]], mix_prob=1.0)
    fx["pc_mix_offset_odd"] = odd["offset"].numpy()
    np.savez_compressed(os.path.join(HERE, "transforms.npz"), **fx)
    print("transforms.npz", os.path.getsize(os.path.join(HERE, "transforms.npz")) // 1024, "KiB;",
          "GridSample kept %d of 6000, SphereCrop random centre row %d, Mix3D offsets %s / %s" %
          (fx["gs_n_out"], ci, fx["pc_mix_offset"].tolist(), fx["pc_mix_offset_odd"].tolist()))


if __name__ == "__main__":
    main()
