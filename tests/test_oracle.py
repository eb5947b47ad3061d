"""CPU tests: the oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py) and against independent definitions for the two third-party
ops whose arithmetic is 'parity unpinned'."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import losses as olosses
from oracle import ops as oops
from oracle import ptv3 as optv3
from oracle import serialization as oser

ORD = ("z", "z-trans", "hilbert", "hilbert-trans")


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("depth", [1, 2, 5, 8, 9, 13, 16])
def test_codes_match_reference(golden_dir, depth):
    fx = load(golden_dir, "serialization.npz")
    gc, b = fx[f"gc_d{depth}"], fx[f"b_d{depth}"]
    for o in ORD:
        assert np.array_equal(oser.encode(gc, b, depth, o), fx[f"code_d{depth}_{o}"]), (depth, o)


def test_codes_appendix_a_literals(golden_dir):
    fx = load(golden_dir, "serialization.npz")
    gc = fx["gc_A"]
    lit = {"z": [2591635, 21191003, 2083255], "z-trans": [2590629, 25375981, 2074039],
           "hilbert": [16615868, 32706338, 1479605], "hilbert-trans": [16614844, 24651392, 1507091]}
    for o in ORD:
        code = oser.encode(gc, np.zeros(len(gc), np.int64), 9, o)
        assert np.array_equal(code, fx[f"code_A_{o}"])
        assert code[:3].tolist() == lit[o]
    assert oser.encode(gc, None, 9, "z").max() == 117672634


def test_point_serialization_matches_reference(golden_dir):
    fx = load(golden_dir, "serialization.npz")
    batch = oser.offset2batch(fx["room_offset"])
    code, order, inverse, depth = oser.serialize(fx["room_gc"], batch, ORD)
    assert depth == int(fx["room_depth"])
    assert np.array_equal(code, fx["room_code"])
    assert np.array_equal(order, fx["room_order"])      # unique voxels => order is unique
    assert np.array_equal(inverse, fx["room_inverse"])


def test_padding_matches_reference(golden_dir):
    fx = load(golden_dir, "padding.npz")
    for ci in range(int(fx["ncases"])):
        pad, unpad, cu = oser.padding(fx[f"c{ci}_offset"], int(fx[f"c{ci}_K"]))
        assert np.array_equal(pad, fx[f"c{ci}_pad"]), ci
        assert np.array_equal(unpad, fx[f"c{ci}_unpad"]), ci
        assert np.array_equal(cu, fx[f"c{ci}_cu"]) and cu.dtype == np.int32, ci
    # SURVEY Appendix A.1 literal
    pad, unpad, cu = oser.padding([10, 13, 19], 4)
    assert pad.tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 6, 7, 10, 11, 12, 13, 14, 15, 16, 17, 18, 15, 16]
    assert cu.tolist() == [0, 4, 8, 12, 15, 19, 23]


@pytest.mark.parametrize("name", ["h2d16", "h2d48"])
def test_attention_matches_reference(golden_dir, name):
    fx = load(golden_dir, "attention.npz")
    C, H, K, oi = [int(v) for v in fx[f"{name}_cfg"]]
    sd = {k[len(name) + 4:]: torch.from_numpy(fx[k]).requires_grad_(True) for k in fx.files if k.startswith(name + "_sd_")}
    x = torch.from_numpy(fx[f"{name}_x"]).requires_grad_(True)
    levels = optv3.build_levels(fx[f"{name}_gc"], fx[f"{name}_offset"], ORD, ())
    y = optv3.attention(x, sd, "", levels[0], oi, H, K)
    (y * torch.from_numpy(fx[f"{name}_cot"])).sum().backward()
    assert torch.allclose(y, torch.from_numpy(fx[f"{name}_y"]), atol=2e-5, rtol=1e-4)
    assert torch.allclose(x.grad, torch.from_numpy(fx[f"{name}_dx"]), atol=2e-5, rtol=1e-4)
    for k in sd:
        assert torch.allclose(sd[k].grad, torch.from_numpy(fx[f"{name}_grad_{k}"]), atol=5e-5, rtol=1e-4), k


def tiny_cfg(fx):
    cfg = {}
    for k in fx.files:
        if k.startswith("cfg_"):
            v = fx[k]
            cfg[k[4:]] = tuple(v.tolist()) if v.ndim else v.item()
    return cfg


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_tiny_ptv3_matches_reference(golden_dir, mode):
    fx = load(golden_dir, "ptv3_tiny.npz")
    cfg = tiny_cfg(fx)
    sd = optv3.init_state_dict(cfg, seed=11)
    chk = np.array([float(v.double().abs().sum()) for k, v in sorted(sd.items())])
    assert np.allclose(chk, fx["sd_checksum"], rtol=1e-12), "seeded init drifted: regenerate golden"
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    feat = torch.from_numpy(fx["feat"]).requires_grad_(True)
    stats = {}
    # SerializedPooling always shuffles the curve order (ptv3:408-412): replay its RNG draws
    torch.manual_seed(77)
    perms = [np.arange(4)] + [torch.randperm(4).numpy() for _ in cfg["stride"]]
    y = optv3.forward(sd, cfg, feat, fx["gc"], fx["offset"], bn_training=(mode == "train"), stats_out=stats,
                      perms=perms)
    (y * torch.from_numpy(fx["cot"])).sum().backward()
    ref = torch.from_numpy(fx[f"{mode}_y"])
    assert torch.allclose(y, ref, atol=2e-4, rtol=1e-3), (y - ref).abs().max()
    cos = F.cosine_similarity(y, ref, dim=1)
    assert (1 - cos).max() < 1e-6
    assert torch.allclose(feat.grad, torch.from_numpy(fx[f"{mode}_dfeat"]), atol=1e-3, rtol=1e-2)
    for k in fx.files:
        if k.startswith(f"{mode}_grad_"):
            g, r = sd[k[len(mode) + 6:]].grad, torch.from_numpy(fx[k])
            assert (g - r).norm() <= 2e-3 * r.norm() + 1e-5, (k, (g - r).norm(), r.norm())
    if mode == "train":
        for k in fx.files:
            if k.startswith("train_stat_"):
                assert torch.allclose(stats[k[11:]], torch.from_numpy(fx[k]), atol=1e-5, rtol=1e-4), k


def test_stale_cpe_call_order(golden_dir):
    """SURVEY finding 11 / Appendix A.6: conv call order by indice_key in the reference."""
    fx = load(golden_dir, "ptv3_tiny.npz")
    assert fx["conv_call_keys"].tolist() == ["stem", "stage0", "stage1", "stage2", "stage2",
                                             "stage1", "stage0", "stage0"]


def test_losses_match_reference(golden_dir):
    fx = load(golden_dir, "losses.npz")
    pred0, tgt = torch.from_numpy(fx["pred"]), torch.from_numpy(fx["tgt"])
    mask, seg = torch.from_numpy(fx["mask"]), torch.from_numpy(fx["seg"])
    crit = [dict(type="CosineSimilarity", reduction="mean", loss_weight=1.0),
            dict(type="L2Loss", reduction="mean", loss_weight=1.0),
            dict(type="AggregatedContrastiveLoss", temperature=0.2, reduction="mean", loss_weight=0.02,
                 schedule="last_75")]
    assert np.allclose(olosses.cosine_similarity_loss(pred0, tgt, mask).numpy(), fx["loss_cos"], rtol=1e-6)
    assert np.allclose(olosses.l2_loss(pred0, tgt, mask).numpy(), fx["loss_l2"], rtol=1e-6)
    for ep in (0.1, 0.5):
        pred = pred0.clone().requires_grad_(True)
        torch.manual_seed(123)
        # lang_head normalises first; pred rows are already unit so this is the identity up to rounding
        loss = olosses.lang_head(pred, tgt, mask, seg, ep, crit)
        loss.backward()
        assert np.allclose(loss.detach().numpy(), fx[f"loss_ep{ep}"], rtol=2e-6), ep
    # gradient of the criteria themselves (no normalize in front), as the reference computed it
    for ep in (0.1, 0.5):
        pred = pred0.clone().requires_grad_(True)
        torch.manual_seed(123)
        loss = (olosses.cosine_similarity_loss(pred, tgt, mask) + olosses.l2_loss(pred, tgt, mask)
                + olosses.aggregated_contrastive_loss(pred, mask, seg, ep, 0.2, 0.02, "last_75"))
        loss.backward()
        assert np.allclose(pred.grad.numpy(), fx[f"dpred_ep{ep}"], atol=1e-7, rtol=1e-4), ep


def test_subm_conv_against_dense_conv3d():
    """'parity unpinned' op: cross-check the restated submanifold conv against a dense
    torch conv3d evaluated at the active sites."""
    g = torch.Generator().manual_seed(3)
    for k in (3, 5):
        S, B, cin, cout = 9, 2, 5, 7
        occ = torch.rand(B, S, S, S, generator=g) < 0.3
        idx = occ.nonzero()
        idx = idx[torch.randperm(len(idx), generator=g)]
        feat = torch.randn(len(idx), cin, generator=g)
        w = torch.randn(cout, k, k, k, cin, generator=g)
        bias = torch.randn(cout, generator=g)
        nbr = oops.neighbor_table(idx[:, 1:].numpy(), idx[:, 0].numpy(), k)
        out = oops.subm_conv3d(feat, w, bias, nbr)
        dense = torch.zeros(B, cin, S, S, S)
        dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feat
        ref = F.conv3d(dense, w.permute(0, 4, 1, 2, 3), bias, padding=k // 2)
        ref = ref[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
        assert torch.allclose(out, ref, atol=1e-4, rtol=1e-4)


def test_segment_csr_definition():
    g = torch.Generator().manual_seed(4)
    src = torch.randn(20, 3, generator=g)
    ptr = [0, 3, 3, 10, 20]
    for red in ("sum", "mean", "max", "min"):
        out = oops.segment_csr(src, ptr, red)
        for i in range(4):
            seg = src[ptr[i]:ptr[i + 1]]
            if len(seg) == 0:
                exp = torch.zeros(3)
            else:
                exp = {"sum": seg.sum(0), "mean": seg.mean(0), "max": seg.max(0).values, "min": seg.min(0).values}[red]
            assert torch.allclose(out[i], exp, atol=1e-6)


def test_pool_partition_sizes():
    """All four curves are hierarchical: members of one z-order parent cell share code>>3."""
    torch.manual_seed(0)
    gc = torch.randint(0, 300, (4096, 3), dtype=torch.int32).numpy()
    code, order, inverse, depth = oser.serialize(gc, np.zeros(4096, np.int64), ORD, depth=9)
    cluster, indices, idx_ptr, head, ncode = oser.pool_partition(code, 1)
    assert len(head) == len(np.unique(code[0] >> 3)) == len(idx_ptr) - 1
    for k in range(4):
        assert np.array_equal(ncode[k][cluster], code[k] >> 3)


# ---- production widths (tests/golden/make_golden_prod.py) -----------------------------------------------------
def _prod():
    import importlib.util
    spec = importlib.util.spec_from_file_location("prod_inputs", os.path.join(os.path.dirname(__file__), "golden", "prod_inputs.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_prod_attention_matches_reference(golden_dir):
    """The oracle's SerializedAttention at the dec0 width (C=768, 16 heads of 48, K=1024, padded tail window)
    against the reference module's outputs and gradients."""
    mp = _prod()
    fx = load(golden_dir, "attention_prod.npz")
    gc, x, cot, sd = mp.att_inputs()
    n, K, H = mp.ATT["n"], mp.ATT["K"], mp.ATT["H"]
    for v in sd.values():
        v.requires_grad_(True)
    x.requires_grad_(True)
    lv = optv3.build_levels(gc.numpy(), np.array([n]), mp.ORD, ())[0]
    y = optv3.attention(x, sd, "", lv, mp.ATT["order_index"], H, K)
    (y * cot).sum().backward()
    rows = torch.from_numpy(fx["rows"])
    ref = torch.from_numpy(fx["y_rows"]).float()
    assert (1 - F.cosine_similarity(y.detach()[rows], ref, dim=1)).max() < 1e-6
    assert _rel(mp.proj(y), fx["y_proj"]) < 1e-5 and _rel(mp.proj(x.grad), fx["dx_proj"]) < 1e-5
    assert _rel(x.grad[rows], torch.from_numpy(fx["dx_rows"]).float()) < 1e-3          # fp16 storage
    for k, v in sd.items():
        assert _rel(mp.proj(v.grad), fx["grad_proj_" + k]) < 1e-4, k


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_prod_lang_ptv3_matches_reference(golden_dir, mode):
    """The oracle's full lang-pretrain PT-v3m1 (91.71 M parameters) on the 6,400-Gaussian room against the reference
    model: per-Gaussian cosine on the stored rows, projections of every row, input and parameter gradients."""
    mp = _prod()
    fx = load(golden_dir, "ptv3_lang_prod.npz")
    cfg = dict(optv3.DEFAULT_CFG)
    sd = optv3.init_state_dict(cfg, seed=5)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    gc, feat, cot = mp.lang_inputs()
    feat.requires_grad_(True)
    torch.manual_seed(mp.POOL_SEED)
    perms = [np.arange(4)] + [torch.randperm(4).numpy() for _ in cfg["stride"]]
    y = optv3.forward(sd, cfg, feat, gc.numpy(), np.array([len(gc)]), bn_training=(mode == "train"), perms=perms)
    (y * cot).sum().backward()
    rows = torch.from_numpy(fx["rows"])
    cosd = 1 - F.cosine_similarity(y.detach()[rows], torch.from_numpy(fx[f"{mode}_y_rows"]).float(), dim=1)
    print("oracle vs reference, lang model (%s): max cosine distance %.2e" % (mode, cosd.max()))
    assert cosd.max() < 1e-6
    assert _rel(mp.proj(y), fx[f"{mode}_y_proj"]) < 1e-4
    assert _rel(feat.grad, fx[f"{mode}_dfeat"]) < 2e-3
    for k in mp.GRAD_KEYS:
        assert _rel(mp.proj(sd[k].grad), fx[f"{mode}_grad_proj_{k}"]) < 3e-3, k


def _tf_sample(n, seed):
    g = np.random.RandomState(seed)
    coord = (g.rand(n, 3) * np.array([4.0, 3.0, 1.5])).astype(np.float32)
    return dict(coord=coord, color=(g.rand(n, 3) * 2 - 1).astype(np.float32), opacity=g.rand(n, 1).astype(np.float32),
                quat=g.randn(n, 4).astype(np.float32), scale=g.rand(n, 3).astype(np.float32),
                segment=g.randint(-1, 20, n).astype(np.int64), lang_feat=g.randn(n, 16).astype(np.float32),
                valid_feat_mask=(g.rand(n) < 0.9).astype(np.int64), name="scene%d" % seed)


def test_transform_restatements_match_the_reference_transforms(golden_dir):
    """oracle/transforms.py against what the reference's GridSample / SphereCrop / Collect / point_collate_fn produced
    (tests/golden/transforms.npz, generated by importing pointcept/datasets/transform.py and datasets/utils.py)."""
    from oracle import transforms as otf
    fx = load(golden_dir, "transforms.npz")
    # GridSample: occupied voxels, the point -> voxel partition (the reference numbers voxels by hash rank)
    d = _tf_sample(int(fx["gs_n"]), int(fx["gs_seed"]))
    gc, uniq, inv, cnt = otf.grid_sample_voxels(d["coord"], float(fx["gs_grid"]))
    assert np.array_equal(uniq.astype(np.int32), fx["gs_grid_coord_sorted"]) and len(uniq) == int(fx["gs_n_out"])
    ref_inv = fx["gs_inverse"]
    pairs = np.unique(np.stack([inv, ref_inv], 1), axis=0)
    assert len(pairs) == len(uniq)                                   # the two numberings are a bijection: same partition
    # SphereCrop
    d = _tf_sample(int(fx["sc_n"]), int(fx["sc_seed"]))
    pm = int(fx["sc_point_max"])
    idx = otf.sphere_crop_index(d["coord"], pm, len(d["coord"]) // 2)
    assert np.array_equal(d["coord"][idx], fx["sc_center_coord"]) and np.array_equal(d["segment"][idx], fx["sc_center_segment"])
    assert np.array_equal(d["lang_feat"][idx], fx["sc_center_lang"])
    idx = otf.sphere_crop_index(d["coord"], pm, int(fx["sc_random_center_index"]))
    assert np.array_equal(d["coord"][idx], fx["sc_random_coord"]) and np.array_equal(d["opacity"][idx], fx["sc_random_opacity"])
    idx = otf.sphere_crop_index(d["coord"], int(0.25 * len(d["coord"])), len(d["coord"]) // 2)
    assert np.array_equal(d["coord"][idx], fx["sc_rate_coord"])
    # Collect
    d = _tf_sample(int(fx["co_n"]), int(fx["co_seed"]))
    d["grid_coord"] = np.floor(d["coord"] / 0.02).astype(np.int64)
    out = otf.collect(d, ("coord", "grid_coord", "segment", "lang_feat", "valid_feat_mask", "name"), feat_keys=("color", "opacity", "quat", "scale"))
    assert sorted(out.keys()) == list(fx["co_keys"]) and np.array_equal(out["feat"], fx["co_feat"]) and np.array_equal(out["offset"], fx["co_offset"])
    # collate
    assert np.array_equal(otf.collate_offsets(fx["pc_sizes"]), fx["pc_offset"])
    assert np.array_equal(otf.collate_offsets(fx["pc_sizes"], mix=True), fx["pc_mix_offset"])
    assert np.array_equal(otf.collate_offsets(fx["pc_sizes"][:3], mix=True), fx["pc_mix_offset_odd"])
