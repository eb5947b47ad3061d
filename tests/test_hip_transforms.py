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


def test_grid_sample_test_fragments_cover_the_reference_partition(fx):
    """GridSample(mode="test") (transform.py:1302-1330): as many fragments as the fullest voxel has points, each holding ONE
    point per occupied voxel, together covering every point; fragment i takes member (i mod c) of a voxel with c points.  The
    reference's numpy argsort leaves the order INSIDE a voxel unspecified, so the recorded run is compared per voxel: the
    same member set, enumerated cyclically with the same period."""
    from scenesplat_amd.gpu_transforms import grid_sample_test
    d = _sample(int(fx["gs_n"]), int(fx["gs_seed"]))
    out = grid_sample_test(torch.from_numpy(d["coord"]).cuda(), float(fx["gs_grid"]), return_inverse=True)
    idx = out["index"].cpu().numpy()
    ref = fx["gst_index"]
    assert idx.shape == ref.shape == (int(fx["gst_nparts"]), int(fx["gs_n_out"]))
    inv = out["inverse"].cpu().numpy()                       # our voxel id per point
    cnt = out["count"].cpu().numpy()
    assert cnt.max() == idx.shape[0] and np.array_equal(np.bincount(inv), cnt)
    assert np.array_equal(np.sort(np.unique(idx)), np.arange(len(inv)))            # every point is in some fragment
    for p in range(idx.shape[0]):
        assert np.array_equal(inv[idx[p]], np.arange(idx.shape[1]))                 # fragment p: one member of every voxel, in voxel order
    # cyclic enumeration in row order: member (p mod c) of the voxel's ascending rows
    members = [np.nonzero(inv == v)[0] for v in range(40)]
    for v, m in enumerate(members):
        assert [int(idx[p, v]) for p in range(idx.shape[0])] == [int(m[p % len(m)]) for p in range(idx.shape[0])]
    # the reference's fragments: same voxels (its own numbering), per voxel the same member set with the same period
    rinv = fx["gs_inverse"]                                    # reference voxel id per point
    for p in range(ref.shape[0]):
        assert np.array_equal(np.sort(rinv[ref[p]]), np.arange(ref.shape[1]))
    ours_of = {}                                               # our voxel -> the set of its members over the fragments
    for v in range(idx.shape[1]):
        ours_of[frozenset(idx[:, v].tolist())] = cnt[v]
    rcol = np.argsort(rinv[ref[0]])                            # reference columns by reference voxel id
    for c_ in rcol[:500]:
        col = ref[:, c_]
        key = frozenset(col.tolist())
        assert key in ours_of                                  # the same member set as one of our voxels
        c = ours_of[key]
        assert all(col[p] == col[p % c] for p in range(len(col)))                    # period = the voxel's point count
    gc = out["grid_coord"].cpu().numpy()
    order = np.lexsort((gc[:, 2], gc[:, 1], gc[:, 0]))
    assert np.array_equal(gc[order], fx["gs_grid_coord_sorted"])


def test_open_vocab_fragment_loop_equals_the_plain_torch_loop():
    """engines/test.py:300-378 on the device (gpu_transforms.open_vocab_fragments): fragments -> chunked forward -> fused
    scan + accumulate -> top-3 (ScanNet++ form) or arg-max with the confidence threshold, against the same loop written with
    torch.mm / sigmoid / index_add on the fragments' features."""
    from scenesplat_amd.gpu_transforms import grid_sample_test, open_vocab_fragments
    from scenesplat_amd.pointcept_api import MODELS
    from scenesplat_amd.synthetic import room_chunk
    cfg = dict(in_channels=11, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2), enc_depths=(1, 1, 1), enc_channels=(16, 32, 48),
               enc_num_head=(1, 2, 3), enc_patch_size=(64, 64, 16), dec_depths=(1, 1), dec_channels=(48, 32), dec_num_head=(1, 2), dec_patch_size=(64, 64),
               shuffle_orders=False)
    torch.manual_seed(0)
    model = MODELS.build(dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **cfg), criteria=[])).cuda().eval()
    # (SerializedPooling shuffles its curves with the host RNG: seeded before every loop below)
    d = room_chunk(n_side=32, seed=2, lang_dim=0)
    g = torch.Generator().manual_seed(4)
    # three Gaussians per voxel on a part of the room, two / one elsewhere: 3 fragments
    coord = torch.cat([d["coord"], d["coord"][:900] + 0.004, d["coord"][:300] + 0.008]).cuda()
    feat = torch.cat([d["feat"], d["feat"][:900] * 0.9, d["feat"][:300] * 1.1]).cuda()
    text = torch.nn.functional.normalize(torch.randn(20, 48, generator=g), dim=1).cuda()
    frag = grid_sample_test(coord, 0.02)
    P = frag["index"].shape[0]
    assert P >= 3                                               # (the shifted copies also land in neighbouring voxels)
    torch.manual_seed(9)
    top3, pred = open_vocab_fragments(model, dict(coord=coord, feat=feat), text, 0.02, topk=3)
    torch.manual_seed(9)
    lab, pred2 = open_vocab_fragments(model, dict(coord=coord, feat=feat), text, 0.02, confidence_threshold=0.55)
    assert torch.equal(pred, pred2)
    # the plain loop on the same fragments
    ref = torch.zeros_like(pred)
    torch.manual_seed(9)
    with torch.no_grad():
        for p in range(P):
            idx = frag["index"][p]
            f = model(dict(coord=coord[idx], grid_coord=frag["grid_coord"], feat=feat[idx].contiguous(), offset=torch.tensor([len(idx)]).cuda()),
                      chunk_size=600000)["point_feat"]["feat"]
            ref.index_add_(0, idx, torch.sigmoid(f.to(torch.bfloat16).float() @ text.to(torch.bfloat16).float().t()))
    assert torch.allclose(pred, ref, atol=2e-3, rtol=1e-3)
    assert float((pred.sum(1) > 0).float().mean()) == 1.0       # every Gaussian was in some fragment
    # top-3 and thresholded arg-max follow from pred
    assert torch.equal(top3, pred.topk(3, dim=1)[1])
    mx, am = pred.max(1)
    exp = am.clone(); exp[mx < 0.55] = -1
    assert torch.equal(lab, exp) and int((lab == -1).sum()) > 0 and int((lab >= 0).sum()) > 0
