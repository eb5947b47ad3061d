"""Production-shape golden vectors from the REFERENCE's own Python (container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_prod.py            # attention_prod, ptv3_lang_prod
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_prod.py --full     # ptv3_lang_full (about 20 GB of memory)

Same import recipe as make_golden.py (four third-party packages stubbed; spconv / torch_scatter
arithmetic comes from the oracle and stays "parity unpinned").  What is new is the SHAPE: the
lang-pretrain widths the benchmark runs, so that the production kernels (MFMA attention at
d = 48 / K = 1024, the C = 768 conv / GEMM tiles, 22 blocks deep) are compared with reference
outputs and not only with each other:

  attention_prod.npz  SerializedAttention(C=768, H=16, K=1024) on 2,600 points: two full windows
                      and a tail window topped up with 472 borrowed points; y, dx and parameter
                      gradients of a seeded cotangent.
  ptv3_lang_prod.npz  the full lang-pretrain PT-v3m1 (91.71 M parameters) on a
                      6,400-Gaussian room (levels 6400/1600/400/100; dec0 = 6 full windows + a
                      padded tail), eval-BN and train-BN: features, input gradient, parameter
                      gradients.

Inputs and weights are NOT stored: the tests regenerate them from the same seeded CPU generators
(tests/golden/prod_inputs.py, shared with the tests).  Outputs are stored as a row subset in fp16 (rounding
adds ~1e-7 of cosine distance) plus an fp32 random projection of EVERY row / parameter gradient
(`proj`), so the fixtures stay a few MB.  Data only; no reference source.
"""
import importlib
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from prod_inputs import ATT, GRAD_KEYS, ORD, POOL_SEED, att_inputs, full_inputs, lang_inputs, proj, row_subset  # noqa: E402


def full_size(ptv3, optv3):
    """ptv3_lang_full.npz: the reference model's FORWARD on the benchmark's own workload shape (room-102400, levels
    102400/25600/6400/1600, 100 windows of 1024 at dec0), eval-BN and train-BN, under no_grad (the non-flash reference
    attention materialises 6.7 GB of scores per dec0 block; a backward pass would not fit this container's memory).
    Stored: 1,536 rows in fp16 + an fp32 projection and the norm of every row."""
    cfg = dict(optv3.DEFAULT_CFG)
    model = ptv3.PointTransformerV3(**cfg, drop_path=0.0, shuffle_orders=False, enable_flash=False,
                                    upcast_attention=False, upcast_softmax=False, enable_rpe=False)
    sd0 = optv3.init_state_dict(cfg, seed=5)
    gc, feat = full_inputs()
    n = len(gc)
    assert n == 102400
    rows = row_subset(n, 1536, seed=99)
    fx = {"rows": rows.numpy(), "n": np.int64(n)}
    for mode in ("eval", "train"):
        model.load_state_dict(sd0)
        model.train(mode == "train")
        torch.manual_seed(POOL_SEED)
        with torch.no_grad():
            y = model(dict(coord=gc.float() * 0.02, grid_coord=gc, feat=feat, offset=torch.tensor([n]))).feat
        fx[f"{mode}_y_rows"] = y[rows].half().numpy()
        fx[f"{mode}_y_proj"] = proj(y).numpy()
        fx[f"{mode}_y_norm"] = y.norm(dim=1).numpy()
        print(f"ptv3_lang_full[{mode}]: |y| mean %.3f" % float(y.norm(dim=1).mean()), flush=True)
        del y
    np.savez_compressed(os.path.join(HERE, "ptv3_lang_full.npz"), **fx)
    print("ptv3_lang_full.npz", os.path.getsize(os.path.join(HERE, "ptv3_lang_full.npz")) // 1024, "KiB")


def main():
    import make_golden as mg
    mg.install_stubs()
    if "--full" in sys.argv:
        ptv3_ = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m1_base")
        from oracle import ptv3 as optv3_
        return full_size(ptv3_, optv3_)
    ptv3 = importlib.import_module("pointcept.models.point_transformer_v3.point_transformer_v3m1_base")
    from pointcept.models.utils.structure import Point
    from oracle import ptv3 as optv3

    # ---- attention at the dec0 width ------------------------------------------------------------
    C, H, K, n = ATT["C"], ATT["H"], ATT["K"], ATT["n"]
    gc, x, cot, sd = att_inputs()
    att = ptv3.SerializedAttention(channels=C, num_heads=H, patch_size=K, order_index=ATT["order_index"], enable_flash=False,
                                   upcast_attention=False, upcast_softmax=False)
    att.load_state_dict(sd, strict=True)
    xr = x.clone().requires_grad_(True)
    pt = Point(grid_coord=gc, offset=torch.tensor([n]), feat=xr)
    pt.serialization(order=ORD)
    y = att(pt).feat
    (y * cot).sum().backward()
    rows = row_subset(n, 768)
    fx = {"rows": rows.numpy(), "y_rows": y.detach()[rows].half().numpy(), "dx_rows": xr.grad[rows].half().numpy(),
          "y_proj": proj(y).numpy(), "dx_proj": proj(xr.grad).numpy(),
          "y_norm": y.detach().norm(dim=1).numpy(), "dx_norm": xr.grad.norm(dim=1).numpy()}
    for k, p in att.named_parameters():
        fx["grad_proj_" + k] = proj(p.grad).numpy()
        fx["grad_norm_" + k] = np.float64(p.grad.double().norm())
    np.savez_compressed(os.path.join(HERE, "attention_prod.npz"), **fx)
    print("attention_prod: y norm mean %.3f" % float(y.norm(dim=1).mean()))

    # ---- the full lang-pretrain PT-v3m1 ---------------------------------------------------------
    cfg = dict(optv3.DEFAULT_CFG)
    model = ptv3.PointTransformerV3(**cfg, drop_path=0.0, shuffle_orders=False, enable_flash=False,
                                    upcast_attention=False, upcast_softmax=False, enable_rpe=False)
    sd0 = optv3.init_state_dict(cfg, seed=5)
    model.load_state_dict(sd0, strict=True)
    gc, feat, cot = lang_inputs()
    n = len(gc)
    assert n == 6400
    rows = row_subset(n, 1536)
    fx = {"rows": rows.numpy(), "n": np.int64(n)}
    params = dict(model.named_parameters())
    for mode in ("eval", "train"):
        model.load_state_dict(sd0)
        model.train(mode == "train")
        model.zero_grad()
        f = feat.clone().requires_grad_(True)
        torch.manual_seed(POOL_SEED)
        out = model(dict(coord=gc.float() * 0.02, grid_coord=gc, feat=f, offset=torch.tensor([n])))
        y = out.feat
        (y * cot).sum().backward()
        fx[f"{mode}_y_rows"] = y.detach()[rows].half().numpy()
        fx[f"{mode}_y_proj"] = proj(y).numpy()
        fx[f"{mode}_y_norm"] = y.detach().norm(dim=1).numpy()
        fx[f"{mode}_dfeat"] = f.grad.numpy()
        for k in GRAD_KEYS:
            fx[f"{mode}_grad_proj_{k}"] = proj(params[k].grad).numpy()
            fx[f"{mode}_grad_norm_{k}"] = np.float64(params[k].grad.double().norm())
        print(f"ptv3_lang_prod[{mode}]: |y| mean %.3f  max|y| %.2f" % (float(y.norm(dim=1).mean()), float(y.abs().max())))
    np.savez_compressed(os.path.join(HERE, "ptv3_lang_prod.npz"), **fx)
    for f_ in ("attention_prod.npz", "ptv3_lang_prod.npz"):
        print(f_, os.path.getsize(os.path.join(HERE, f_)) // 1024, "KiB")


if __name__ == "__main__":
    main()
