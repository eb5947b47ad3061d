#!/usr/bin/env python
"""Where the tiny smoke model's bench-configuration cosine outliers come from: per-Gaussian max / mean for several runtimes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import ptv3 as optv3
from scenesplat_amd import native as nv
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.synthetic import room_chunk

import sys as _s
WIDE = len(_s.argv) > 1
cfg = dict(in_channels=11, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2),
           enc_depths=(1, 1, 1), enc_channels=(16, 32, 48), enc_num_head=(1, 2, 3), enc_patch_size=(64, 64, 16),
           dec_depths=(1, 1), dec_channels=(48, 32), dec_num_head=(1, 2), dec_patch_size=(64, 64))
if WIDE:
    w = int(_s.argv[1])
    cfg.update(enc_channels=(w // 4, w // 2, w), enc_num_head=(max(1, w // 64), max(1, w // 32), w // 16), dec_channels=(w, w // 2), dec_num_head=(w // 16, max(1, w // 32)))
data = room_chunk(n_side=32, seed=1, lang_dim=0)
sd = optv3.init_state_dict(cfg, seed=3)
model = MODELS.build(dict(type="PT-v3m1", **cfg, drop_path=0.0, shuffle_orders=False)).cuda().eval()
model.load_state_dict(sd, strict=True)
perms = [[0, 1, 2, 3], [2, 0, 3, 1], [1, 3, 0, 2]]
yo = optv3.forward(sd, cfg, data["feat"].clone(), data["grid_coord"].numpy(), data["offset"].numpy(), perms=perms).detach()
base = dict(RUNTIME)
for name, rt, amp in (("bench_runtime", bench_runtime(), True),
                      ("bench + split<=48", dict(bench_runtime(), conv_split_max_channels=48), True),
                      ("bench + split<=64", dict(bench_runtime(), conv_split_max_channels=64), True),
                      ("bench, attention SIMT fp32 math", dict(bench_runtime(), attn_impl=nv.ATTN_SIMT), True),
                      ("bench, old MFMA attention", dict(bench_runtime(), attn_headmajor=False), True),
                      ("autocast only (fp32 conv, SIMT attn)", dict(attn_impl=nv.ATTN_SIMT, conv_dtype=None), True),
                      ("fp32 everything", dict(attn_impl=nv.ATTN_SIMT, conv_dtype=None), False)):
    RUNTIME.clear(); RUNTIME.update(base); RUNTIME.update(rt)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = model(dict(feat=data["feat"].cuda(), grid_coord=data["grid_coord"].cuda(), offset=data["offset"].cuda()), perms=perms).feat
    cd = 1 - F.cosine_similarity(y.float().cpu(), yo, dim=1)
    print("%-40s mean %.2e  p99 %.2e  max %.2e   (rows above 1e-4: %d of %d)" % (name, cd.mean(), cd.quantile(0.99), cd.max(), int((cd > 1e-4).sum()), len(cd)))
