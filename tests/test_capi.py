"""CPU: the C-ABI library builds/loads and exports every symbol include/scenesplat_hip.h declares
(no compute calls without a GPU); host-side pure logic."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "scenesplat_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from scenesplat_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
        assert s in _lib.PROTOTYPES, f"{s} has no ctypes prototype"
    for s in _lib.PROTOTYPES:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"
    assert lib.ss_version() >= 100


def test_ops_refuse_cpu_tensors():
    import torch
    from scenesplat_amd import native as nv
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nv.gather_rows(torch.zeros(4, 4), torch.zeros(4, dtype=torch.int32))
    from scenesplat_amd.pointcept_api import MODELS
    m = MODELS.build(dict(type="PT-v3m1", in_channels=4, enc_depths=(1, 1), enc_channels=(8, 16), enc_num_head=(1, 1),
                          enc_patch_size=(16, 16), stride=(2,), dec_depths=(1,), dec_channels=(8,), dec_num_head=(1,),
                          dec_patch_size=(16,)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(dict(feat=torch.zeros(8, 4), grid_coord=torch.zeros(8, 3, dtype=torch.long), offset=torch.tensor([8])))


def test_window_layout_matches_oracle_padding():
    import numpy as np
    from oracle import serialization as oser
    from scenesplat_amd.plan import window_layout
    for offs, K in [([10, 13, 19], 4), ([1500], 1024), ([2048, 2049], 1024), ([7, 71, 271], 64), ([100, 101], 128)]:
        counts = np.diff(np.array([0] + offs)).tolist()
        off_pad, win = window_layout(counts, K)
        pad, unpad, cu = oser.padding(offs, K)
        assert win == cu.tolist()
        assert off_pad[-1] == len(pad)


def test_state_dict_contract():
    """393 tensors / 91.71 M parameters, keys == the reference's."""
    from oracle import ptv3 as optv3
    from scenesplat_amd.pointcept_api import MODELS
    cfg = dict(optv3.DEFAULT_CFG)
    m = MODELS.build(dict(type="PT-v3m1", **cfg))
    sd = m.state_dict()
    ref = optv3.init_state_dict(cfg)   # key set pinned by a strict load into the reference (make_golden.py)
    assert len(sd) == 393 and set(sd) == set(ref)
    assert all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in sd)
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 91.71) < 0.01


def test_registry_build_contract():
    from scenesplat_amd.pointcept_api import LOSSES, MODELS
    assert {"PT-v3m1", "LangPretrainer"} <= set(MODELS.module_dict)
    assert {"CosineSimilarity", "L2Loss", "AggregatedContrastiveLoss"} <= set(LOSSES.module_dict)
    with pytest.raises(KeyError):
        MODELS.build(dict(type="nope"))
    with pytest.raises(TypeError, match="PointTransformerV3"):
        MODELS.build(dict(type="PT-v3m1", bogus=1))


def test_bfs_cluster_host_function_matches_oracle():
    """pointgroup bfs_cluster is a HOST function of the C-ABI (as in the reference): testable without a GPU."""
    import numpy as np
    import torch
    from oracle import pointops as opo
    from scenesplat_amd import pointops as po
    g = np.random.default_rng(0)
    xyz = g.random((300, 3), dtype=np.float32)
    ridx, rsl = opo.ballquery_batch_p(xyz, np.zeros(300, int), np.array([0, 300]), 0.15)
    lab = (xyz[:, 0] > 0.5).astype(np.int32)
    ci, co = po.bfs_cluster(torch.as_tensor(lab), torch.as_tensor(ridx), torch.as_tensor(rsl), 5)
    ri, ro = opo.bfs_cluster(lab, ridx, rsl, 5)
    assert np.array_equal(ci.numpy(), ri) and np.array_equal(co.numpy(), ro) and len(ro) > 2
