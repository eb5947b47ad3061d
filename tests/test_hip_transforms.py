"""GPU data fast path (scenesplat_amd/gpu_transforms.py) against OUTPUTS OF THE REFERENCE'S OWN transforms
(tests/golden/transforms.npz: pointcept/datasets/transform.py GridSample / SphereCrop / Collect and datasets/utils.py
point_collate_fn run in the build container by tests/golden/make_golden_transforms.py)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sample(n, seed):
    g = np.random.RandomState(seed)
    coord = (g.rand(n, 3) * np.array([4.0, 3.0, 1.5])).astype(np.float32)
    return dict(coord=coord, color=(g.rand(n, 3) * 2 - 1).astype(np.float32), opacity=g.rand(n, 1).astype(np.float32),
                quat=g.randn(n, 4).astype(np.float32), scale=g.rand(n, 3).astype(np.float32),
                segment=g.randint(-1, 20, n).astype(np.int64), lang_feat=g.randn(n, 16).astype(np.float32),
                valid_feat_mask=(g.rand(n) < 0.9).astype(np.int64), name="scene%d" % seed)


def _cuda(d):
    return {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in d.items()}


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "transforms.npz"))


def test_grid_sample_train_occupies_the_reference_voxels(fx):
    from scenesplat_amd.gpu_transforms import grid_sample_train
    d = _sample(int(fx["gs_n"]), int(fx["gs_seed"]))
    out = grid_sample_train(torch.from_numpy(d["coord"]).cuda(), float(fx["gs_grid"]), return_inverse=True)
    gc = out["grid_coord"].cpu().numpy()
    order = np.lexsort((gc[:, 2], gc[:, 1], gc[:, 0]))
    assert np.array_equal(gc[order], fx["gs_grid_coord_sorted"])                 # the same set of occupied voxels, bit for bit
    inv = out["inverse"].cpu().numpy()
    assert len(np.unique(np.stack([inv, fx["gs_inverse"]], 1), axis=0)) == int(fx["gs_n_out"])    # the same point -> voxel partition
    # the representative of voxel i is a member of voxel i, and the counts are the partition's
    idx = out["idx_unique"].cpu().numpy()
    assert np.array_equal(inv[idx], np.arange(len(idx)))
    assert np.array_equal(out["count"].cpu().numpy(), np.bincount(inv))


def test_sphere_crop_matches_the_reference_rows(fx):
    from scenesplat_amd.gpu_transforms import sphere_crop
    d = _sample(int(fx["sc_n"]), int(fx["sc_seed"]))
    pm = int(fx["sc_point_max"])
    g = _cuda(d)
    g["not_per_point"] = torch.arange(len(d["coord"]), device="cuda")        # N rows but not a SphereCrop key: passed through
    out = sphere_crop(g, point_max=pm, mode="center")
    assert np.array_equal(out["coord"].cpu().numpy(), fx["sc_center_coord"])
    assert np.array_equal(out["segment"].cpu().numpy(), fx["sc_center_segment"])
    assert np.array_equal(out["lang_feat"].cpu().numpy(), fx["sc_center_lang"])
    assert out["not_per_point"].shape[0] == len(d["coord"]) and out["name"] == d["name"]
    out = sphere_crop(_cuda(d), point_max=pm, mode="random", center_index=int(fx["sc_random_center_index"]))
    assert np.array_equal(out["coord"].cpu().numpy(), fx["sc_random_coord"])
    assert np.array_equal(out["opacity"].cpu().numpy(), fx["sc_random_opacity"])
    out = sphere_crop(_cuda(d), sample_rate=0.25, mode="center")
    assert np.array_equal(out["coord"].cpu().numpy(), fx["sc_rate_coord"])
    small = _cuda(_sample(100, 3))
    assert sphere_crop(small, point_max=pm, mode="center") is small          # fewer points than point_max: untouched


def test_collect_and_point_collate_match_the_reference(fx):
    from scenesplat_amd.gpu_transforms import collect, point_collate
    d = _cuda(_sample(int(fx["co_n"]), int(fx["co_seed"])))
    d["grid_coord"] = torch.floor(d["coord"] / 0.02).long()
    out = collect(d, ("coord", "grid_coord", "segment", "lang_feat", "valid_feat_mask", "name"), feat_keys=("color", "opacity", "quat", "scale"))
    assert sorted(out.keys()) == list(fx["co_keys"])
    assert np.array_equal(out["feat"].cpu().numpy(), fx["co_feat"]) and np.array_equal(out["offset"].cpu().numpy(), fx["co_offset"])
    batch = []
    for s, n in zip(fx["pc_seeds"], fx["pc_sizes"]):
        d = _cuda(_sample(int(n), int(s)))
        d["grid_coord"] = torch.floor(d["coord"] / 0.02).long()
        batch.append(collect(d, ("coord", "grid_coord", "segment", "name"), feat_keys=("color", "opacity", "quat", "scale")))
    out = point_collate([dict(b) for b in batch], mix_prob=0.0)
    assert np.array_equal(out["offset"].cpu().numpy(), fx["pc_offset"]) and np.array_equal(out["coord"].cpu().numpy(), fx["pc_coord"])
    assert np.allclose(out["feat"].double().sum(0).cpu().numpy(), fx["pc_feat_sum"], rtol=1e-12) and list(out["name"]) == list(fx["pc_names"])
    random.seed(3)                                                             # the host draw the reference compared with mix_prob
    out = point_collate([dict(b) for b in batch], mix_prob=1.0)
    assert np.array_equal(out["offset"].cpu().numpy(), fx["pc_mix_offset"])
    out = point_collate([dict(b) for b in batch[:3]], mix_prob=1.0)
    assert np.array_equal(out["offset"].cpu().numpy(), fx["pc_mix_offset_odd"])
    # and the collated batch is a valid model input: the planner accepts it (duplicate voxels of a Mix3D element included)
    from scenesplat_amd.plan import build_plan
    gc = out["grid_coord"] - out["grid_coord"].amin(0, keepdim=True)
    plan = build_plan(gc.cuda(), out["offset"].cuda(), ("z", "hilbert"), (2,))
    assert plan.levels[0].n == int(fx["pc_sizes"][:3].sum())
