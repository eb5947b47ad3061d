"""Backward fixture at the benchmark's own size, from the PINNED ORACLE (container or any CPU box; about 2 minutes on 8 cores).

    python tests/golden/make_golden_full_bwd.py

ptv3_lang_full.npz holds the reference model's FORWARD on room-102400 only: the reference's non-flash attention
materialises 6.7 GB of scores per dec0 block, its backward does not fit this container (make_golden_prod.py).  The oracle
-- window-batched attention, pinned to the reference at these very widths (tests/test_oracle.py::test_prod_*: 2.4e-7 cosine,
gradients to 3e-3 against reference outputs) -- does forward + backward of the same chunk in about a minute, so the gradients
of the benchmarked configuration get a fixture at the benchmarked size:

  ptv3_lang_full_bwd.npz   eval-BN and train-BN: forward rows (cross-check against ptv3_lang_full.npz: the generator asserts
                           < 1e-6 cosine to the reference rows before it writes anything), dfeat (n, 11) in full, and an fp32
                           random projection + norm of 18 parameter gradients (prod_inputs.GRAD_KEYS), for the seeded cotangent
                           prod_inputs.full_cotangent().
Data only.  Inputs and weights are regenerated from seeded generators (prod_inputs.py)."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from prod_inputs import GRAD_KEYS, POOL_SEED, full_cotangent, full_inputs, proj  # noqa: E402
from oracle import ptv3 as optv3  # noqa: E402


def main():
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    ref = np.load(os.path.join(HERE, "ptv3_lang_full.npz"))
    rows = torch.from_numpy(ref["rows"])
    cfg = dict(optv3.DEFAULT_CFG)
    gc, feat0 = full_inputs()
    n = len(gc)
    cot = full_cotangent(n)
    fx = {"rows": ref["rows"], "n": np.int64(n)}
    for mode in ("eval", "train"):
        t0 = time.time()
        sd = optv3.init_state_dict(cfg, seed=5)
        for v in sd.values():
            if v.is_floating_point():
                v.requires_grad_(True)
        feat = feat0.clone().requires_grad_(True)
        torch.manual_seed(POOL_SEED)
        perms = [np.arange(4)] + [torch.randperm(4).numpy() for _ in cfg["stride"]]
        y = optv3.forward(sd, cfg, feat, gc.numpy(), np.array([n]), bn_training=(mode == "train"), perms=perms)
        cd = 1 - F.cosine_similarity(y.detach()[rows], torch.from_numpy(ref[f"{mode}_y_rows"]).float(), dim=1)
        print(f"[{mode}] oracle forward vs the reference rows of ptv3_lang_full.npz: max cosine distance %.2e (%.0f s)" % (cd.max(), time.time() - t0), flush=True)
        assert cd.max() < 1e-6, "the oracle left the reference: fixture NOT written"
        (y * cot).sum().backward()
        fx[f"{mode}_y_rows"] = y.detach()[rows].half().numpy()
        fx[f"{mode}_dfeat"] = feat.grad.numpy()
        for k in GRAD_KEYS:
            fx[f"{mode}_grad_proj_{k}"] = proj(sd[k].grad).numpy()
            fx[f"{mode}_grad_norm_{k}"] = np.float64(sd[k].grad.double().norm())
        print(f"[{mode}] backward done, |dfeat| %.3e (%.0f s)" % (float(feat.grad.norm()), time.time() - t0), flush=True)
        del y, sd
    out = os.path.join(HERE, "ptv3_lang_full_bwd.npz")
    np.savez_compressed(out, **fx)
    print("ptv3_lang_full_bwd.npz", os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
