#!/usr/bin/env python
"""cProfile of the HOST side of eager training steps (room-102400, bench_runtime): where the Python enqueue time goes."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

RUNTIME.update(bench_runtime())
model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).cuda().train()
data = {k: v.cuda() for k, v in room_chunk(256, 0, lang_dim=0).items()}
cot = torch.randn(len(data["feat"]), 768, device="cuda").to(torch.bfloat16)


def step():
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"]))
    torch.autograd.backward(out.feat, grad_tensors=cot)


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print(s.getvalue()[:9000])
