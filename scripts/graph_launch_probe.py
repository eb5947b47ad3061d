#!/usr/bin/env python
"""Does hipGraphLaunch block the host while the previous launch of the SAME executable graph is still running?  (It decides whether
a replayed training step needs two alternating graph instances to keep the GPU queue fed.)"""
import time
import torch

a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)


def work(n=40):
    c = a
    for _ in range(n):
        c = torch.mm(c, b)
    return c


def capture():
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        work(2)
        g.capture_begin()
        out = work()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    return g, out


g1, o1 = capture()
g2, o2 = capture()
torch.cuda.synchronize()
t0 = time.perf_counter(); g1.replay(); torch.cuda.synchronize(); dur = time.perf_counter() - t0
print("one replay: %.2f ms of GPU work" % (dur * 1e3))
for name, seq in (("same graph twice", (g1, g1, g1, g1)), ("two graphs alternating", (g1, g2, g1, g2))):
    torch.cuda.synchronize()
    ts = []
    t_all = time.perf_counter()
    for g in seq:
        t0 = time.perf_counter(); g.replay(); ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    print("%-24s host ms per replay() call: %s | total wall %.2f ms" % (name, " ".join("%.2f" % t for t in ts), (time.perf_counter() - t_all) * 1e3))
# many-node graph: host cost of launching ~900 small kernels as a graph
x = torch.randn(1 << 20, device="cuda")
g3 = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x + 1
    g3.capture_begin()
    y = x
    for _ in range(900):
        y = y * 1.0001
    g3.capture_end()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); g3.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("900-node graph: replay() call %.2f ms on the host, %.2f ms until done" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
t0 = time.perf_counter(); g3.replay(); g3.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("900-node graph twice back to back: calls %.2f ms, done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
