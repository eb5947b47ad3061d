#!/usr/bin/env python
"""Headline benchmark: Gaussians/s of the PTv3 encoder forward+backward on synthetic
102,400-Gaussian chunks (BASELINE.json metric / configs[1]), bf16 autocast, one chunk per GPU
per step, DDP (RCCL) gradient all-reduce included when --gpus > 1.

A step = serialization/plan + forward + backward (seeded random cotangent) of the lang-pretrain
PT-v3m1 (91.71 M params) on one "room-102400" chunk already resident in HBM; data loading and the
optimizer are outside the metric.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-side", type=int, default=256, help="room side; 256 -> 102,400 Gaussians")
    ap.add_argument("--fixture", default="room", choices=["room", "uniform"],
                    help="room = the metric's workload (SURVEY 8d); uniform = the sparse stress fixture (not the metric)")
    ap.add_argument("--attn", default="auto", choices=["auto", "simt", "mfma"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse the N>1 path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--cpu-n-side", type=int, default=128, help="room side of the CPU-baseline sample (128 -> 25,600 Gaussians, ~10-20 s on 16 cores)")
    return ap.parse_args()


def event_time_ms(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def roofline_probes(model, data, impl):
    """Isolated launches of the kernels the north star prices, at the dec0 shapes of the workload, timed with HIP
    events on the launch stream: the dominant kernel (submanifold conv forward / dgrad on the LDS-DMA pipeline,
    MFMA-bound), the window attention forward (MFMA / VALU-bound) and the row gather (HBM-bound)."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.plan import build_plan
    plan = build_plan(data["grid_coord"], data["offset"], model.order, model.stride)
    lv = plan.levels[0]
    C, H, K = 768, 16, 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    pmc = {}
    try:
        # HBM-side bytes per launch at this shape, from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, gfx950 correction applied; scripts/gpu_pmc.sh)
        if lv.n == 102400:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
    except Exception:
        pass
    # --- dominant kernel: k_gemm8<true> (CPE conv of the dec0 blocks: 4 launches per step fwd + dgrad)
    nbr, perm = lv.neighbors(3), lv.conv_rowperm()
    x = torch.randn(lv.n, C, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(C, 27, C, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    pairs = int((nbr >= 0).sum().item())
    ms_c = event_time_ms(lambda: nv.subm_conv_fwd(x, w, None, nbr, perm), 5)
    flops_c = 2.0 * pairs * C * C
    conv = dict(bound="mfma", kernel="k_gemm8<true> subm conv fwd (dec0: n=%d, C=%d, 27 taps, %.2f pairs/site)" % (lv.n, C, pairs / lv.n),
                achieved=flops_c / (ms_c * 1e-3) / 1e12, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s",
                frac=flops_c / (ms_c * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
                traffic=pmc.get("k_gemm8<true>@dec0", {}).get("traffic_bytes_corrected"), traffic_unit="bytes/launch beyond L2 (PMC)",
                algorithmic_flops=flops_c, algorithmic_bytes=lv.n * C * 2 * 2 + 27 * C * C * 2, ms=ms_c)
    # --- window attention forward
    win = lv.window(0, K)
    qkv = torch.randn(lv.n, 3 * C, device="cuda", generator=g).to(torch.bfloat16)
    ms = event_time_ms(lambda: nv.window_attn_fwd(qkv, win, H, (C // H) ** -0.5, impl), 5)
    flops = sum(4.0 * L * L * (C // H) for L in [K] * win.num_windows) * H
    attn = dict(bound="mfma", kernel="k_attn_fwd_mfma<48> (dec0: %d windows x %d heads, K=%d, d=%d)" % (win.num_windows, H, K, C // H),
                achieved=flops / (ms * 1e-3) / 1e12, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s",
                frac=flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
                traffic=pmc.get("k_attn_fwd_mfma<48>@dec0", {}).get("traffic_bytes_corrected"), traffic_unit="bytes/launch (PMC)",
                algorithmic_flops=flops, ms=ms)
    idx = lv.order_row(0)
    out = torch.empty_like(x)
    ms2 = event_time_ms(lambda: nv.gather_rows(x, idx, out=out), 20)
    nbytes = lv.n * (2 * C * 2 + 4)
    hbm = dict(bound="hbm", kernel="gather_rows(%d x %d bf16)" % (lv.n, C), achieved=nbytes / (ms2 * 1e-3) / 1e9,
               peak=HBM_PEAK_GBS, unit="GB/s", frac=nbytes / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None, ms=ms2)
    return conv, attn, hbm


def cpu_baseline(n_side):
    """Oracle (pure-PyTorch fp32 CPU restatement of the reference path) fwd+bwd on a bounded
    sample of the same workload: the full lang-pretrain PT-v3m1 on a room of n_side."""
    from oracle import ptv3 as optv3
    from scenesplat_amd.synthetic import LANG_PTV3, room_chunk
    # the GPU box gives one GPU a 16-core share; os.cpu_count() reports the whole host and oversubscribes
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)), os.cpu_count() or 1)))
    cfg = {k: LANG_PTV3[k] for k in optv3.DEFAULT_CFG}
    sd = optv3.init_state_dict(cfg, seed=0)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    data = room_chunk(n_side=n_side, seed=0, lang_dim=0)
    n = len(data["feat"])
    cot = torch.randn(n, cfg["dec_channels"][0], generator=torch.Generator().manual_seed(1))
    t0 = time.time()
    y = optv3.forward(sd, cfg, data["feat"], data["grid_coord"].numpy(), data["offset"].numpy(), bn_training=True)
    (y * cot).sum().backward()
    dt = time.time() - t0
    return dict(value=n / dt, unit="Gaussians/s", cores=torch.get_num_threads(), kind="port",
                sample="1 fwd+bwd of the full lang-pretrain PT-v3m1 (fp32, oracle) on a %d-Gaussian room chunk, %.1f s" % (n, dt))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from scenesplat_amd import native as nv
    from scenesplat_amd.pointcept_api import MODELS, RUNTIME
    from scenesplat_amd.synthetic import LANG_PTV3, room_chunk

    impl = {"simt": nv.ATTN_SIMT, "mfma": nv.ATTN_MFMA}.get(args.attn)
    if impl is None:
        impl = nv.ATTN_MFMA if getattr(nv, "HAVE_MFMA_ATTN", False) else nv.ATTN_SIMT
    from scenesplat_amd.pointcept_api import bench_runtime
    RUNTIME.update(bench_runtime())          # the configuration tests/test_hip_prod.py pins to the reference
    RUNTIME["attn_impl"] = impl

    torch.manual_seed(1234 + rank)
    model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).to(dev).train()
    net = model
    if world > 1:
        # DDP as the reference builds it (engines/defaults.py:13-34): broadcast_buffers=False
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False,
                                                        gradient_as_bucket_view=True, bucket_cap_mb=100)
    if args.fixture == "uniform":
        from scenesplat_amd.synthetic import uniform_chunk
        data = {k: v.to(dev) for k, v in uniform_chunk(seed=rank).items()}
    else:
        data = {k: v.to(dev) for k, v in room_chunk(n_side=args.n_side, seed=rank, lang_dim=0).items()}
    n = data["feat"].shape[0]
    cot = torch.randn(n, LANG_PTV3["dec_channels"][0], device=dev, generator=torch.Generator(device=dev).manual_seed(7))

    # The integer plan (serialization, pooling partitions, window indices, rulebooks) of step k+1 is built on
    # a side stream while step k's backward runs, the way a data-loader prefetch would: its few
    # device->host round trips then never drain the float pipeline.  Every step still builds its own plan.
    side = torch.cuda.Stream()
    state = {"plan": model.prepare_plan(data, stream=side)}

    def step():
        net.zero_grad(set_to_none=True)
        plan, state["plan"] = state["plan"], None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(dict(feat=data["feat"], grid_coord=data["grid_coord"], offset=data["offset"], plan=plan))
        state["prev"] = plan
        # backward from the seeded random cotangent, fed directly as the output gradient (no loss kernels)
        torch.autograd.backward(out.feat, grad_tensors=cot.to(out.feat.dtype))
        # SS_BENCH_REUSE_PLAN=1 is a DIAGNOSTIC (host- vs GPU-bound?): it skips the per-step plan build and the line it
        # prints is not the metric
        state["plan"] = plan if os.environ.get("SS_BENCH_REUSE_PLAN") == "1" else model.prepare_plan(data, stream=side)

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    # at least 3 untimed steps before the timed region (allocator pools of both streams, hipBLASLt heuristics, lazy
    # code-object loads settle within the first three): the W warm-up steps asked for, topped up when W < 3
    for i in range(max(args.warmup, 3)):
        t_ = time.perf_counter(); step(); torch.cuda.synchronize()
        log("warmup step %d: %.1f ms" % (i, (time.perf_counter() - t_) * 1e3))
    torch.cuda.synchronize()
    if os.environ.get("SS_BENCH_GC", "freeze") == "freeze":
        # one full collection now, then keep the survivors out of later generation-2 scans: a step allocates ~10^5
        # Python objects (autograd nodes, ctypes wrappers) and an untimely full collection costs ~50 ms
        import gc
        gc.collect(); gc.freeze()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get("SS_BENCH_PER_STEP") == "1":      # DIAGNOSTIC: a device sync per step (perturbs the pipeline)
        for i in range(args.steps):
            t_ = time.perf_counter(); step(); torch.cuda.synchronize()
            ms_ = torch.cuda.memory_stats()
            log("step %d: %.1f ms  reserved %.1f GB  segments %d  device_allocs %d" % (
                i, (time.perf_counter() - t_) * 1e3, torch.cuda.memory_reserved() / 2**30, ms_.get("segment.all.current", 0),
                ms_.get("num_device_alloc", 0)))
    else:
        for _ in range(args.steps):
            step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        res = {
            "metric": "Gaussians/s encoder fwd+bwd, 102k-pt chunks", "value": world * n * args.steps / dt,
            "unit": "Gaussians/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic" if os.environ.get("SS_BENCH_REUSE_PLAN") != "1" else "synthetic, DIAGNOSTIC: plan reused (invalid as metric)",
            "config": {"workload": (("uniform-%d (stress fixture, NOT the metric workload):" if args.fixture == "uniform" else "room-%d:")
                                    + " PT-v3m1 lang-pretrain encoder (91.71M params, in=11, out=768) fwd+bwd, "
                                    "1 chunk of %d Gaussians per GPU per step, serialization included") % (n, n),
                       "gaussians_per_chunk": n, "chunks_per_gpu": 1,
                       "parallelism": "dp%d" % world, "attention_kernel": "mfma" if impl == nv.ATTN_MFMA else "simt"},
        }
        log("timed %d steps: %.1f ms/step" % (args.steps, dt / args.steps * 1e3))
        if world == 1:
            conv, attn, hbm = roofline_probes(model, data, impl)
            log("roofline probes done")
            res["roofline"] = conv
            res["roofline_attn"] = attn
            res["roofline_hbm"] = hbm
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args.cpu_n_side)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
