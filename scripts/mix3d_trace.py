"""GPU (run under rocprofv3 --kernel-trace --stats): three eager training steps of the lang-pretrain PT-v3m1 under bench_runtime() on
a Mix3D-merged pair of chunks (duplicate voxels at level 0) -- the kernel list shows which conv kernels the duplicate regime runs
(no per-tap gather + torch.addmm launches)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

RUNTIME.update(bench_runtime())
torch.manual_seed(1)
model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).cuda().train()
a, b = room_chunk(256, 0, lang_dim=0), room_chunk(256, 1, lang_dim=0)
gc = torch.cat([a["grid_coord"], b["grid_coord"] + torch.tensor([3, 5, 0])]).cuda()
feat = torch.cat([a["feat"], b["feat"]]).cuda()
off = torch.tensor([len(gc)]).cuda()
for it in range(4):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o = model(dict(feat=feat, grid_coord=gc, offset=off))
    torch.autograd.backward(o.feat, grad_tensors=torch.ones_like(o.feat))
    torch.cuda.synchronize()
lv = o["plan"].levels[0]
dups = int((lv.neighbors(3)[13] != torch.arange(lv.n, device="cuda", dtype=torch.int32)).sum())
print("mix3d trace: %d Gaussians in one batch element, %d of them share their voxel with an earlier row; levels %s" % (lv.n, dups, [l.n for l in o["plan"].levels]))
