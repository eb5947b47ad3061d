"""Debug harness (GPU): which part of the split step breaks hipGraph capture?  variants: eager | onegraph | twograph"""
import faulthandler
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
from scenesplat_amd import native as nv
from scenesplat_amd.pointcept_api import MODELS, RUNTIME
from scenesplat_amd.pointcept_api.ptv3 import backward_tail
from scenesplat_amd.steady_state import SteadyStateStep
from scenesplat_amd.synthetic import room_chunk

TINY = dict(in_channels=11, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2),
            enc_depths=(1, 1, 1), enc_channels=(16, 32, 48), enc_num_head=(1, 2, 3), enc_patch_size=(64, 64, 16),
            dec_depths=(1, 1), dec_channels=(48, 32), dec_num_head=(1, 2), dec_patch_size=(64, 64))
variant = sys.argv[1]
RUNTIME.update(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA)
torch.manual_seed(11)
model = MODELS.build(dict(type="PT-v3m1", **TINY, drop_path=0.0, shuffle_orders=True)).cuda().train()
d = {k: v.cuda() for k, v in room_chunk(n_side=40, seed=3, lang_dim=0).items()}
n = d["feat"].shape[0]
box = {}


def fn(plan, t):
    cut = []
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=t["feat"], grid_coord=d["grid_coord"], offset=d["offset"], plan=plan, backward_cut=cut))
    torch.autograd.backward(out.feat, grad_tensors=t["cot"])
    box["cut"] = cut
    if variant == "onegraph":
        backward_tail(box.pop("cut"))
    return {"feat": out.feat}


def tail():
    backward_tail(box.pop("cut"))


steady = SteadyStateStep(fn, list(model.parameters()), warmup=1, tail=None if variant == "onegraph" else tail, between=lambda: None,
                         enabled=(variant != "eager"))
g = torch.Generator(device="cuda").manual_seed(1)
for it in range(5):
    feat = torch.randn(n, 11, device="cuda", generator=g)
    cot = torch.randn(n, 48, device="cuda", generator=g).to(torch.bfloat16)
    model.zero_grad(set_to_none=True)
    print(variant, "step", it, flush=True)
    steady(model.prepare_plan(d), dict(feat=feat, cot=cot))
    torch.cuda.synchronize()
    print(variant, "step", it, "ok: replays", steady.replays, "refused", steady.refused, "grads", sum(p.grad is not None for p in model.parameters()), flush=True)
