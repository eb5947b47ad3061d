"""GPU parity: integer structure kernels vs the oracle / golden vectors.  Bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as oops
from oracle import ptv3 as optv3
from oracle import serialization as oser

pytestmark = pytest.mark.gpu
ORD = ("z", "z-trans", "hilbert", "hilbert-trans")


@pytest.fixture(scope="module")
def nv():
    from scenesplat_amd import native
    return native


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


@pytest.mark.parametrize("depth", [1, 2, 5, 8, 9, 13, 16])
def test_encode_matches_reference_golden(nv, golden_dir, depth):
    fx = np.load(os.path.join(golden_dir, "serialization.npz"))
    gc, b = fx[f"gc_d{depth}"], fx[f"b_d{depth}"]
    codes = nv.serialize_encode(dev(gc, torch.int32), dev(b, torch.int32), depth, ORD).cpu().numpy()
    for k, o in enumerate(ORD):
        assert np.array_equal(codes[k], fx[f"code_d{depth}_{o}"]), (depth, o)


@pytest.mark.parametrize("n,bits", [(1, 9), (255, 17), (2048, 24), (2049, 27), (102400, 24), (300000, 49), (1 << 20, 63)])
def test_argsort_stable_bit_exact(nv, n, bits):
    g = torch.Generator().manual_seed(n)
    hi = (1 << bits) - 1
    keys = torch.randint(0, min(hi, (1 << 62)) + 1, (3, n), generator=g, dtype=torch.int64)
    keys[1] = keys[1] % 7          # heavy ties: stability
    keys[2] = torch.sort(keys[2]).values
    order, inverse, skeys = nv.argsort_i64(keys.cuda(), bits)
    ref = np.stack([np.argsort(k, kind="stable") for k in keys.numpy()])
    assert np.array_equal(order.cpu().numpy(), ref)
    assert np.array_equal(skeys.cpu().numpy(), np.take_along_axis(keys.numpy(), ref, 1))
    inv = inverse.cpu().numpy()
    for k in range(3):
        assert np.array_equal(inv[k][ref[k]], np.arange(n))


def room(n_side, seed):
    h = max(2, n_side * 72 // 256)
    xs, ys = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij")
    floor = np.stack([xs.ravel(), ys.ravel(), np.zeros(n_side * n_side, int)], 1)
    yy, zz = np.meshgrid(np.arange(n_side), np.arange(1, h + 1), indexing="ij")
    wa = np.stack([np.zeros(yy.size, int), yy.ravel(), zz.ravel()], 1)
    wb = np.stack([np.full(yy.size, n_side - 1), yy.ravel(), zz.ravel()], 1)
    gc = np.concatenate([floor, wa, wb]).astype(np.int64)
    return gc[torch.randperm(len(gc), generator=torch.Generator().manual_seed(seed)).numpy()]


@pytest.mark.parametrize("shuffle", [False, True])
def test_plan_levels_match_oracle(nv, golden_dir, shuffle):
    from scenesplat_amd.plan import build_plan
    gc = room(64, 5)
    n = len(gc)
    offs = np.array([n // 4, n // 4 + 37, n])
    perms = [[2, 0, 3, 1], [1, 3, 0, 2], [3, 2, 1, 0], [0, 2, 1, 3]] if shuffle else None
    plan = build_plan(dev(gc), dev(offs), ORD, (2, 2, 2), perms)
    ref = optv3.build_levels(gc, offs, ORD, (2, 2, 2), perms)
    assert [l.n for l in plan.levels] == [l.n for l in ref]
    for lv, rl in zip(plan.levels, ref):
        assert lv.depth == rl.depth and lv.offsets[1:] == rl.offset.tolist()
        assert np.array_equal(lv.grid_coord.cpu().numpy(), rl.grid_coord)
        assert np.array_equal(lv.batch.cpu().numpy(), rl.batch)
        for j in range(4):
            assert np.array_equal(lv.code_row(j).cpu().numpy(), rl.code[j])
            assert np.array_equal(lv.order_row(j).cpu().numpy(), rl.order[j])
            assert np.array_equal(lv.inverse_row(j).cpu().numpy(), rl.inverse[j])
        if rl.cluster is not None:
            assert np.array_equal(lv.cluster.cpu().numpy(), rl.cluster)
            assert np.array_equal(lv.idx_ptr.cpu().numpy()[:lv.n + 1], rl.idx_ptr)
        assert not lv.has_duplicates
        for k in (3, 5):
            assert np.array_equal(lv.neighbors(k).cpu().numpy().T, oops.neighbor_table(rl.grid_coord, rl.batch, k))


def test_plan_matches_reference_point_serialization(nv, golden_dir):
    from scenesplat_amd.plan import build_plan
    fx = np.load(os.path.join(golden_dir, "serialization.npz"))
    plan = build_plan(dev(fx["room_gc"]), dev(fx["room_offset"]), ORD, ())
    lv = plan.levels[0]
    assert lv.depth == int(fx["room_depth"])
    assert np.array_equal(lv.codes.cpu().numpy(), fx["room_code"])
    assert np.array_equal(lv.order.cpu().numpy(), fx["room_order"])
    assert np.array_equal(lv.inverse.cpu().numpy(), fx["room_inverse"])


def test_window_index_matches_reference_padding(nv, golden_dir):
    from scenesplat_amd.plan import build_plan
    fx = np.load(os.path.join(golden_dir, "padding.npz"))
    for ci in range(int(fx["ncases"])):
        offs, K = fx[f"c{ci}_offset"], int(fx[f"c{ci}_K"])
        n = int(offs[-1])
        g = torch.Generator().manual_seed(ci)
        gc = torch.stack([torch.randperm(n, generator=g), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1)
        plan = build_plan(gc.cuda(), dev(offs), ("z", "hilbert"), ())
        lv = plan.levels[0]
        for j in range(2):
            w = lv.window(j, K)
            order = lv.order_row(j).cpu().numpy()
            inverse = lv.inverse_row(j).cpu().numpy()
            pad, unpad, cu = fx[f"c{ci}_pad"], fx[f"c{ci}_unpad"], fx[f"c{ci}_cu"]
            assert np.array_equal(w.win_start.cpu().numpy(), cu), ci
            gidx, sidx = w.gidx.cpu().numpy(), w.sidx.cpu().numpy()
            assert np.array_equal(gidx, order[pad]), ci                 # ptv3:184
            canon = unpad[inverse]                                      # ptv3:185: row -> its padded slot
            exp = np.full(len(pad), -1, np.int64); exp[canon] = np.arange(n)
            assert np.array_equal(np.where(sidx >= 0, sidx, -1), exp), ci
            borrowed = np.sort(-1 - sidx[sidx < 0])
            assert np.array_equal(borrowed, np.arange(len(pad) - n)), ci


def test_duplicate_voxels_detected_and_resolved_to_lowest_row(nv):
    from scenesplat_amd.plan import build_plan
    gc = np.array([[1, 1, 1], [2, 1, 1], [1, 1, 1], [3, 3, 3], [2, 1, 1]])
    plan = build_plan(dev(gc), dev(np.array([5])), ("z", "hilbert"), (2,))
    lv = plan.levels[0]
    assert lv.has_duplicates
    assert np.array_equal(lv.order_row(0).cpu().numpy(), np.argsort(oser.encode(gc, np.zeros(5, int), lv.depth, "z"), kind="stable"))
    assert np.array_equal(lv.neighbors(3).cpu().numpy().T, oops.neighbor_table(gc, np.zeros(5, int), 3))


def test_point_publishes_reference_shaped_keys_from_the_plan(nv, golden_dir):
    """Point.serialization / serialized_* / sparsify / padding() (reference structure.py:47-140, ptv3:114-170) against the
    reference's own outputs: the room fixture of serialization.npz and every case of padding.npz."""
    from scenesplat_amd.pointcept_api.structure import Point
    fx = np.load(os.path.join(golden_dir, "serialization.npz"))
    gc = dev(fx["room_gc"])
    p = Point(grid_coord=gc, offset=dev(fx["room_offset"]), feat=torch.zeros(len(gc), 4, device="cuda"))
    p.serialization(order=ORD)
    assert p.serialized_depth == int(fx["room_depth"])
    for key, ref in (("serialized_code", "room_code"), ("serialized_order", "room_order"), ("serialized_inverse", "room_inverse")):
        assert p[key].dtype == torch.int64 and np.array_equal(p[key].cpu().numpy(), fx[ref]), key
    sp = p.sparsify()
    assert p.sparse_shape == (fx["room_gc"].max(0) + 96).tolist() and sp.batch_size == len(fx["room_offset"])
    assert sp.indices.dtype == torch.int32 and np.array_equal(sp.indices[:, 1:].cpu().numpy(), fx["room_gc"])
    assert np.array_equal(sp.indices[:, 0].cpu().numpy(), oser.offset2batch(fx["room_offset"]))
    assert sp.replace_feature(p.feat + 1).features.shape == p.feat.shape
    # a model output carries the plan: the keys follow the level's CURRENT (shuffled) curve order
    torch.manual_seed(5)
    q = Point(grid_coord=gc, offset=dev(fx["room_offset"])).serialization(order=ORD, shuffle_orders=True)
    perm = q.plan.levels[0].curves
    assert sorted(perm) == [0, 1, 2, 3] and np.array_equal(q.serialized_code.cpu().numpy(), fx["room_code"][perm])
    assert np.array_equal(q.serialized_order.cpu().numpy(), fx["room_order"][perm])
    # padding(): the reference's (pad, unpad, cu_seqlens) for every case of the fixture
    fp = np.load(os.path.join(golden_dir, "padding.npz"))
    for ci in range(int(fp["ncases"])):
        offs, K = fp[f"c{ci}_offset"], int(fp[f"c{ci}_K"])
        n = int(offs[-1])
        g = torch.Generator().manual_seed(ci)
        gcc = torch.stack([torch.randperm(n, generator=g), torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)], 1)
        pt = Point(grid_coord=gcc.cuda(), offset=dev(offs)).serialization(order=("z", "hilbert"))
        for j in range(2):
            pad, unpad, cu = pt.padding(K, j)
            assert np.array_equal(pad.cpu().numpy(), fp[f"c{ci}_pad"]), ci
            assert np.array_equal(unpad.cpu().numpy(), fp[f"c{ci}_unpad"]), ci
            assert cu.dtype == torch.int32 and np.array_equal(cu.cpu().numpy(), fp[f"c{ci}_cu"]), ci
