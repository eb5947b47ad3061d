"""Where does the bf16 bench configuration lose cosine?  Runs the full lang-pretrain PT-v3m1 on the 6,400-Gaussian
production fixture (tests/golden/ptv3_lang_prod.npz, reference outputs) under a ladder of precision settings and prints
the per-Gaussian cosine distance of each against the reference; with --taps also per-stage against the oracle."""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ptv3 as optv3  # noqa: E402  (checker only)
from scenesplat_amd import native as nv  # noqa: E402
from scenesplat_amd.pointcept_api import MODELS, RUNTIME  # noqa: E402

spec = importlib.util.spec_from_file_location("prod_inputs", os.path.join(ROOT, "tests", "golden", "prod_inputs.py"))
mp = importlib.util.module_from_spec(spec); spec.loader.exec_module(mp)
fx = np.load(os.path.join(ROOT, "tests", "golden", "ptv3_lang_prod.npz"))
cfg = dict(optv3.DEFAULT_CFG)
model = MODELS.build(dict(type="PT-v3m1", **cfg, drop_path=0.0, shuffle_orders=False)).cuda()
model.load_state_dict(optv3.init_state_dict(cfg, seed=5), strict=True)
gc, feat, cot = mp.lang_inputs()
rows = torch.from_numpy(fx["rows"])


def run(mode, autocast, **rt):
    old = dict(RUNTIME); RUNTIME.update(rt)
    try:
        model.train(mode == "train")
        torch.manual_seed(mp.POOL_SEED)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            y = model(dict(feat=feat.cuda(), grid_coord=gc.cuda(), offset=torch.tensor([len(gc)]).cuda())).feat
    finally:
        RUNTIME.clear(); RUNTIME.update(old)
    y = y.float().cpu()
    cd = 1 - F.cosine_similarity(y[rows].double(), torch.from_numpy(fx[f"{mode}_y_rows"]).double(), dim=1)
    return cd


ladder = [("fp32 everywhere (SIMT attn, fp32 conv)", False, dict(attn_impl=nv.ATTN_SIMT, conv_dtype=None)),
          ("bf16 conv only (no autocast)", False, dict(attn_impl=nv.ATTN_SIMT, conv_dtype=torch.bfloat16)),
          ("autocast, fp32 conv, SIMT attn", True, dict(attn_impl=nv.ATTN_SIMT, conv_dtype=None)),
          ("autocast, fp32 conv, MFMA attn", True, dict(attn_impl=nv.ATTN_MFMA, conv_dtype=None)),
          ("autocast, bf16x3 conv, MFMA attn", True, dict(attn_impl=nv.ATTN_MFMA, conv_dtype="bf16x3")),
          ("autocast, bf16 conv everywhere, MFMA attn", True, dict(attn_impl=nv.ATTN_MFMA, conv_dtype=torch.bfloat16, conv_split_max_channels=0)),
          ("bench: bf16 conv C>256, bf16x3 below", True, dict(attn_impl=nv.ATTN_MFMA, conv_dtype=torch.bfloat16, conv_split_max_channels=256)),
          ("bf16 conv C>512 only, bf16x3 below", True, dict(attn_impl=nv.ATTN_MFMA, conv_dtype=torch.bfloat16, conv_split_max_channels=512))]
if "--split-sweep" in sys.argv:
    ladder = [("bf16 conv, bf16x3 for C <= %d" % t, True, dict(attn_impl=nv.ATTN_MFMA, conv_dtype=torch.bfloat16, conv_split_max_channels=t))
              for t in (0, 32, 64, 128, 256)]
extra = [a for a in sys.argv[1:] if a.startswith("rt:")]
for e in extra:      # e.g. rt:conv_out_fp32=1
    k, v = e[3:].split("=")
    ladder.append(("bench + %s=%s" % (k, v), True, dict(attn_impl=nv.ATTN_MFMA, conv_dtype=torch.bfloat16, **{k: int(v)})))
for mode in ("eval", "train"):
    for name, ac, rt in ladder:
        cd = run(mode, ac, **rt)
        print("[%s] %-44s cosd mean %.2e  p99 %.2e  max %.2e" % (mode, name, cd.mean(), cd.quantile(0.99), cd.max()), flush=True)
