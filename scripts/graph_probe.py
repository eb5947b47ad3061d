#!/usr/bin/env python
"""Diagnostic: can one fwd+bwd of the bench model be captured in a hipGraph (fixed plan), and what does a replay cost?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

RUNTIME.update(bench_runtime())
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).to(dev).train()
data = {k: v.to(dev) for k, v in room_chunk(n_side=256, seed=0, lang_dim=0).items()}
n = data["feat"].shape[0]
cot = torch.randn(n, LANG_PTV3["dec_channels"][0], device=dev).to(torch.bfloat16)
plan = model.prepare_plan(data)
torch.cuda.synchronize()

def fb():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"], plan=plan))
    torch.autograd.backward(out.feat, grad_tensors=cot)
    return out

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        fb()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()

def timeit(f, k=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        f()
    te = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, te / k * 1e3

def eager():
    model.zero_grad(set_to_none=True)
    fb()
ms, enq = timeit(eager)
print("eager, plan reused: %.2f ms/step (host enqueue %.2f)" % (ms, enq), flush=True)

g = torch.cuda.CUDAGraph()
model.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    out = fb()
torch.cuda.synchronize()
print("captured", flush=True)
ms, enq = timeit(g.replay)
print("graph replay: %.2f ms/step (host enqueue %.2f)" % (ms, enq), flush=True)
gn = sum(float(p.grad.float().norm()) ** 2 for p in model.parameters() if p.grad is not None) ** 0.5
print("grad norm after replay %.6g, out finite %s" % (gn, bool(torch.isfinite(out.feat.float()).all())))

# ---- what sits between two replays matters: (a) nothing, (b) a small kernel on the same stream, (c) a side-stream kernel + event
small_a, small_b = torch.zeros(1 << 20, device=dev), torch.ones(1 << 20, device=dev)
def replay_plus_kernel():
    small_a.copy_(small_b)
    g.replay()
ms, enq = timeit(replay_plus_kernel)
print("graph replay after a small same-stream kernel: %.2f ms/step (host enqueue %.2f)" % (ms, enq), flush=True)
side = torch.cuda.Stream()
def replay_plus_side():
    with torch.cuda.stream(side):
        small_a.copy_(small_b)
        ev = torch.cuda.Event(); ev.record(side)
    torch.cuda.current_stream().wait_event(ev)
    g.replay()
ms, enq = timeit(replay_plus_side)
print("graph replay after a side-stream kernel + event wait: %.2f ms/step (host enqueue %.2f)" % (ms, enq), flush=True)
def replay_plus_hostwork():
    model.zero_grad(set_to_none=True)
    g.replay()
ms, enq = timeit(replay_plus_hostwork)
print("graph replay after zero_grad(set_to_none): %.2f ms/step (host enqueue %.2f)" % (ms, enq), flush=True)
ms, enq = timeit(g.replay)
print("graph replay again: %.2f ms/step (host enqueue %.2f)" % (ms, enq), flush=True)
