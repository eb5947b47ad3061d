"""GPU (run under rocprofv3 --kernel-trace --stats): the libs/pointops* replacements at the sizes bench.py's pointops lines use."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenesplat_amd import pointops as po
from scenesplat_amd.synthetic import room_chunk

small = room_chunk(256, 0, lang_dim=0)["coord"].cuda().contiguous()
big = room_chunk(800, 2, lang_dim=0)["coord"].cuda().contiguous()
for xyz in (small, big):
    n = xyz.shape[0]
    off = torch.tensor([n], dtype=torch.int32, device="cuda")
    for _ in range(3):
        po.knn_query(25, xyz, off, impl="grid")
n = small.shape[0]
off = torch.tensor([n], dtype=torch.int32, device="cuda")
po.knn_query(25, small, off, impl="brute")
g = torch.Generator(device="cuda").manual_seed(3)
nb = big.shape[0]
lab = torch.randint(0, 160, (nb,), device="cuda", generator=g).int()
val = torch.rand(nb, device="cuda", generator=g) < 0.9
for _ in range(3):
    po.neighbor_voting(big, lab, val, 25, -1, 160)
po.ball_query(16, 0.1, 0.0, small, off, impl="brute")
for _ in range(3):
    po.ball_query(16, 0.1, 0.0, small, off, impl="grid")
po.farthest_point_sampling(small, off, torch.tensor([n // 4], dtype=torch.int32, device="cuda"))
idx, _ = po.knn_query(16, small, off, impl="grid")
feat = torch.randn(n, 64, device="cuda", generator=g).requires_grad_(True)
for _ in range(3):
    torch.autograd.grad(po.grouping2(feat, idx), feat, torch.ones(n, 16, 64, device="cuda"))
torch.cuda.synchronize()
print("pointops trace done")
