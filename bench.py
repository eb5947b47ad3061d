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
# before torch is imported and long before any GPU call: the host driver only supports dmabuf IPC (RCCL / tensor sharing across
# processes fails otherwise), and the HSA runtime reads the variable once, when it initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP maps streams onto 4 hardware queues by default.  With a process group initialised (RCCL's streams) the plan-building side
# stream came to share a queue with the compute stream: its device->host round trips then waited for the whole step and the host
# blocked 35 ms per step (one-rank rehearsal, round 3: 42.2 -> 39.5 ms/step with 8 queues).  Read once, when HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a five-line version banner to stdout when
# a process group comes up): file descriptor 1 is pointed at stderr for the whole run and the JSON line goes to the saved descriptor.
_REAL_STDOUT = os.dup(1)
os.dup2(2, 1)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def _page_in_gemm_code_objects():
    """hipBLASLt / rocBLAS load the code object of a GEMM shape class from disk the first time that class is used (lazy loading,
    ~350 MB of gfx950 files under torch/lib).  In the first process on a freshly booted box those reads come from a cold disk cache: the
    eager variable-size line (new GEMM shapes every step) read 131 ms per step there against 103-105 ms in every later process.  Reading
    the files once on a background thread pages them in before the lines that need them run; nothing is timed against it."""
    import glob
    import threading

    def run():
        lib = os.path.join(os.path.dirname(torch.__file__), "lib")
        for pat in ("hipblaslt/library/*gfx950*", "rocblas/library/*gfx950*"):
            for f in glob.glob(os.path.join(lib, pat)):
                try:
                    with open(f, "rb") as fh:
                        while fh.read(1 << 24):
                            pass
                except OSError:
                    pass
    threading.Thread(target=run, name="page-in-gemm-code-objects", daemon=True).start()



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
    ap.add_argument("--exchange", default="stage", choices=["stage", "ddp"],
                    help="N > 1 gradient averaging: one all-reduce per model stage (scenesplat_amd/grad_exchange.py) or torch DDP buckets")
    ap.add_argument("--cpu-n-side", type=int, default=256, help="room side of the CPU-baseline chunk (256 -> the metric's 102,400 Gaussians, about a minute on 16 cores)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay forward+backward as a hipGraph once the plan shape repeats (auto: single-rank runs)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline traffic = null)")
    ap.add_argument("--power", action="store_true", help="also run every matrix-bound probe back to back for 2.5 s with rocm-smi polled beside it (adds "
                    "`sustained` {ms, socket_power_w, sclk_mhz} to its roofline object; off by default: thousands of extra launches would "
                    "dominate the per-kernel averages of a rocprofv3 trace of this command)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-3 and fp32-conv secondary lines")
    return ap.parse_args()


def _probes_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("roofline_probes", os.path.join(ROOT, "scripts", "roofline_probes.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def pmc_traffic(log, timeout_s=240):
    """HBM-side bytes per launch of the probe kernels, MEASURED IN THIS RUN: two rocprofv3 child passes (FETCH_SIZE, then
    WRITE_SIZE: the TCC block cannot hold both, MI355X_MICROARCH.md "rocprofv3 PMC slots") over scripts/roofline_probes.py,
    counters only (--kernel-trace, no other trace domain).  gfx950 correction: FETCH_SIZE counts 64 B per 128-B request
    on wide coalesced reads -> doubled; WRITE_SIZE exact; both in KB.  Returns {kernel substring: bytes} or {} (then the
    bench line carries traffic = null rather than a stale number)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        log("pmc: rocprofv3 not on PATH, traffic = null")
        return {}
    out = {}
    tmp = tempfile.mkdtemp(prefix="ss_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "scripts", "roofline_probes.py")]
            t0 = time.time()
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            log("pmc pass %s: rc %d, %.0f s" % (counter, r.returncode, time.time() - t0))
            if r.returncode != 0:
                return {}
            acc = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter:
                        acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            for k, v in acc.items():
                # per launch: the probe program launches each kernel three times at one shape; max (not mean) so that a
                # smaller launch of the same kernel during the probes' set-up (plan building) cannot dilute it
                out.setdefault(k, 0.0)
                out[k] += mult * 1024.0 * max(v)
    except Exception as e:   # noqa: BLE001  (profiling is best effort; the timed numbers do not depend on it)
        log("pmc: %s: %s" % (type(e).__name__, e))
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def power_limited_peak(log):
    """Dense bf16 MFMA rate the chip SUSTAINS with toggling operands (scripts/ubench/mfma_peak.hip, random bf16 operands, ~0.3 s):
    the datasheet peak the `roofline` objects are priced against is reached only with constant operands; real data is
    power-limited well below it.  Informational, measured in this run; None when the probe binary is missing."""
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "ubench", "bin", "mfma_peak")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "1000000", "1"], capture_output=True, text=True, timeout=60).stdout
        vals = [(float(a), float(b)) for a, b in re.findall(r"([0-9.]+) TFLOP/s dense bf16 .*? shader clock ([0-9.]+) GHz", out)]
        if not vals:
            return None
        tf, ghz = vals[-1]
        log("dense bf16 MFMA with random operands: %.0f TFLOP/s sustained at %.2f GHz (datasheet peak 2500)" % (tf, ghz))
        return {"value": tf, "unit": "TFLOP/s", "shader_clock_ghz": ghz,
                "how": "scripts/ubench/mfma_peak.hip: register-resident v_mfma_f32_32x32x16_bf16, random operands alternating between two sets, "
                       "4 waves/SIMD, measured in this run"}
    except Exception as e:  # noqa: BLE001
        log("mfma_peak probe failed: %r" % (e,))
        return None


def roofline_lines(log, want_pmc, want_power=False):
    """roofline objects of the bench line: every probe timed with HIP events in THIS process, PMC traffic from child passes.  The
    matrix-bound probes are also run back to back for 2.5 s with rocm-smi polled beside them ("sustained": what the chip holds under
    that kernel alone -- these kernels run the package at its power cap)."""
    rp = _probes_module()
    probes = rp.build()
    traffic = pmc_traffic(log) if want_pmc else {}
    res = {}
    for p in probes:
        ms = rp.time_probe(p, iters=10 if p["bound"] == "hbm" else 5)
        tr = [v for k, v in traffic.items() if p["kernel"] in k]
        res[p["name"]] = rp.roofline_entry(p, ms, sum(tr) if tr else None)
        log("probe %-10s %.3f ms  %.0f %s (%.1f %% of peak)%s" % (p["name"], ms, res[p["name"]]["achieved"], res[p["name"]]["unit"],
                                                                  100 * res[p["name"]]["frac"],
                                                                  "  traffic %.0f MB" % (sum(tr) / 1e6) if tr else ""))
    if want_power:
        for p in probes:
            if p["bound"] != "mfma":
                continue
            try:
                sus = rp.sustained_probe(p)
            except Exception as e:  # noqa: BLE001 -- diagnostic only
                log("sustained probe %s skipped: %r" % (p["name"], e)); sus = None
            if sus:
                sus["achieved"] = p["flops"] / (sus["ms"] * 1e-3) / 1e12
                sus["frac"] = sus["achieved"] / res[p["name"]]["peak"]
                res[p["name"]]["sustained"] = sus
                log("sustained %-10s %.3f ms over %d launches = %.0f TFLOP/s (%.1f %%) at %.0f W, sclk %.0f MHz" % (
                    p["name"], sus["ms"], sus["launches"], sus["achieved"], 100 * sus["frac"], sus["socket_power_w"], sus["sclk_mhz"]))
    return res


def cpu_baseline(n_side, log):
    """Oracle (pure-PyTorch fp32 CPU restatement of the reference path, kind "port") fwd+bwd of the full lang-pretrain
    PT-v3m1 on one room chunk of the metric's size (n_side 256 -> 102,400 Gaussians; about a minute on 16 cores).  A
    heartbeat thread prints a progress line every 20 s: the GPU box's silence monitor kills a command that writes nothing
    for 7 minutes, and an earlier, slower oracle ran the full chunk in silence (gpurun_out/run3.log, run10.log: rc 137)."""
    import threading
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
    stop = threading.Event()
    stage = ["forward"]

    def beat():
        while not stop.wait(20.0):
            log("cpu baseline: %s, %.0f s elapsed (%d Gaussians, %d threads)" % (stage[0], time.time() - t0, n, torch.get_num_threads()))

    th = threading.Thread(target=beat, daemon=True)
    th.start()
    try:
        y = optv3.forward(sd, cfg, data["feat"], data["grid_coord"].numpy(), data["offset"].numpy(), bn_training=True)
        tf = time.time() - t0
        stage[0] = "backward (forward took %.0f s)" % tf
        (y * cot).sum().backward()
    finally:
        stop.set()
    dt = time.time() - t0
    return dict(value=n / dt, unit="Gaussians/s", cores=torch.get_num_threads(), kind="port",
                sample="1 fwd+bwd of the full lang-pretrain PT-v3m1 (fp32, oracle) on one %d-Gaussian room chunk, %.1f s" % (n, dt))


def secondary_lines(log, steps=3):
    """Secondary measurements next to the headline (never `value`): BASELINE config 3 -- LangPretrainer (PT-v3m1 + fused
    normalise/cosine/L2 head + contrastive loss) at B = 8 chunks x 102,400 Gaussians with 768-d targets -- and the headline
    workload with the conv in the REFERENCE's precision (fp32 operands, modules.py:64-75) instead of bf16."""
    import gc
    from scenesplat_amd import native as nv
    from scenesplat_amd.pointcept_api import MODELS, RUNTIME, bench_runtime
    from scenesplat_amd.synthetic import LANG_PTV3, room_chunk
    out = {}
    old = dict(RUNTIME)
    side = torch.cuda.Stream()

    def run(model, backbone, data, loss_mode, tag, n_steps):
        st = {"plan": backbone.prepare_plan(data, stream=side)}
        cot = None if loss_mode else torch.randn(data["feat"].shape[0], 768, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))

        def step():
            model.zero_grad(set_to_none=True)
            plan, st["plan"] = st["plan"], None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                o = model(dict(data, plan=plan))
            if loss_mode:
                o["loss"].backward()
            else:
                torch.autograd.backward(o.feat, grad_tensors=cot.to(o.feat.dtype))
            st["plan"] = backbone.prepare_plan(data, stream=side)

        for i in range(2):
            t_ = time.perf_counter(); step(); torch.cuda.synchronize()
            log("%s warm-up %d: %.1f ms" % (tag, i, (time.perf_counter() - t_) * 1e3))
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_steps

    try:
        RUNTIME.update(bench_runtime())
        crit = [dict(type="CosineSimilarity", reduction="mean", loss_weight=1.0), dict(type="L2Loss", reduction="mean", loss_weight=1.0),
                dict(type="AggregatedContrastiveLoss", temperature=0.2, reduction="mean", loss_weight=0.02, schedule="last_75")]
        torch.manual_seed(1)
        model = MODELS.build(dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **LANG_PTV3), criteria=crit)).cuda().train()
        data = {k: v.cuda() for k, v in room_chunk(256, 0, lang_dim=768, batch=8).items()}
        data["epoch_progress"] = 0.5
        n = data["feat"].shape[0]
        torch.cuda.reset_peak_memory_stats()
        dt = run(model, model.backbone, data, True, "config 3 (B=8 LangPretrainer)", steps)
        out["config3_lang_pretrainer_b8"] = dict(metric="Gaussians/s LangPretrainer fwd+bwd incl. distillation head, 8 x 102,400-Gaussian chunks, 768-d targets, 1 GPU",
                                                 value=n / dt, unit="Gaussians/s", ms_per_step=dt * 1e3, steps=steps,
                                                 peak_mem_GiB=torch.cuda.max_memory_allocated() / 2**30, dtype="bf16")
        del model, data
        gc.collect(); torch.cuda.empty_cache()
        RUNTIME["conv_dtype"] = None          # the reference's conv precision: fp32 operands (per-tap gather + fp32 GEMM)
        torch.manual_seed(1)
        model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).cuda().train()
        data = {k: v.cuda() for k, v in room_chunk(256, 0, lang_dim=0).items()}
        dt = run(model, model, data, False, "conv fp32 (reference semantics)", steps)
        out["conv_fp32_reference_precision"] = dict(metric="Gaussians/s encoder fwd+bwd, 102k-pt chunk, submanifold conv with fp32 operands (reference semantics), other GEMMs bf16",
                                                    value=data["feat"].shape[0] / dt, unit="Gaussians/s", ms_per_step=dt * 1e3, steps=steps,
                                                    dtype="bf16 autocast + fp32 conv")
        del model
        gc.collect(); torch.cuda.empty_cache()
        RUNTIME["conv_dtype"] = "bf16x3"      # the same precision on the MFMA kernels (hi/lo-split operands, 3x the conv FLOPs)
        torch.manual_seed(1)
        model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).cuda().train()
        dt = run(model, model, data, False, "conv bf16x3 (reference precision on MFMA)", steps)
        out["conv_bf16x3_reference_precision_mfma"] = dict(metric="Gaussians/s encoder fwd+bwd, 102k-pt chunk, submanifold conv on hi/lo-split bf16 operands (near-fp32 result), other GEMMs bf16",
                                                           value=data["feat"].shape[0] / dt, unit="Gaussians/s", ms_per_step=dt * 1e3, steps=steps,
                                                           dtype="bf16 autocast + bf16x3 conv")
        del model, data
        gc.collect(); torch.cuda.empty_cache()
        # ---- variable-size training (round 3): what real batches look like.  The reference crops every sample to at most 192,000
        # Gaussians (SphereCrop, configs/concat_dataset/lang-pretrain-...-contrastive.py:271) and runs 2 chunks per GPU
        # (submit/...-nccl.sh:47-51), so no two steps share a plan shape: every step runs EAGERLY (no hipGraph replay), with a new
        # plan.  N per chunk drawn per step from 60,000 .. 192,000 (seeded); B = 2.
        RUNTIME.update(bench_runtime())
        torch.manual_seed(1)
        model = MODELS.build(dict(type="PT-v3m1", **LANG_PTV3)).cuda().train()
        import random as _random
        rng = _random.Random(5)
        big = room_chunk(352, 3, lang_dim=0)                     # 352^2 + 2 * 352 * 99 = 193,600 unique voxels to crop from
        gcb, fb = big["grid_coord"], big["feat"]

        def batch_of(sizes, merge=False):
            parts, off = [], []
            for j, m in enumerate(sizes):
                sel = torch.randperm(len(gcb), generator=torch.Generator().manual_seed(100 * j + m))[:m]
                parts.append((gcb[sel], fb[sel])); off.append(m)
            # merge: Mix3D (datasets/utils.py:43-47, mix_prob = 0.8 in every language config) -- the two samples become ONE batch
            # element; both are crops of the same room, so ~(m / 193,600)^2 of its voxels hold two Gaussians (duplicate voxels)
            return dict(grid_coord=torch.cat([p[0] for p in parts]).cuda(), feat=torch.cat([p[1] for p in parts]).cuda(),
                        offset=(torch.tensor([sum(off)]) if merge else torch.tensor(off).cumsum(0)).cuda())
        # two untimed steps at the SphereCrop cap (2 x 192,000) first: the caching allocator's pools are then as large as any later
        # batch needs -- a run that has seen its largest batch, as any training run has after a few hundred steps.  (Without this the
        # timed steps paid hipMalloc / hipFree storms whenever a batch outgrew the warm-up ones: 93-160 ms per step across boxes.)
        sizes = [[192000, 192000]] * 2 + [[rng.randrange(60000, 192001) for _ in range(2)] for _ in range(4 + steps)]

        def eager_line(merge):
            batches = [batch_of(sz, merge) for sz in sizes]
            n_warm = len(batches) - steps              # 2 steps at the cap + 4 of random size, all untimed; then `steps` new shapes, timed
            tot, t0, dups = 0, 0.0, 0
            for i, b in enumerate(batches):
                if i == n_warm:
                    # (as the primary line does: one full collection, then no generation-2 scan inside the three timed steps -- an eager
                    # step allocates ~10^5 Python objects and an untimely full collection costs ~50 ms)
                    gc.collect(); gc.disable()
                    torch.cuda.synchronize(); t0 = time.perf_counter(); tot = 0
                model.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    o = model(dict(b))
                torch.autograd.backward(o.feat, grad_tensors=torch.ones_like(o.feat))
                tot += b["feat"].shape[0]
                if merge and not o["plan"].levels[0].has_duplicates:
                    raise RuntimeError("Mix3D line: a merged batch without duplicate voxels")
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0
            gc.enable()
            if merge:      # Gaussians that share their voxel with an earlier row, last batch (one host read, after the timed region)
                lv0 = o["plan"].levels[0]
                dups = int((lv0.neighbors(3)[13] != torch.arange(lv0.n, device="cuda", dtype=torch.int32)).sum())
            del batches
            return tot, dt_, dups

        # Two passes over the same size sequence.  The FIRST pass of a fresh process also loads, from disk, the hipBLASLt / rocBLAS
        # code objects of every GEMM shape class these sizes touch (a cold box: 145 ms/step against 101 ms warm, round 4); a
        # training run pays that once in its first few hundred steps.  `value` is the second pass -- the steady state of a run --,
        # the first is reported beside it.  (The Mix3D line below runs the same sizes after both, i.e. warm as well.)
        tot0, dt0, _ = eager_line(False)
        tot, dt, _ = eager_line(False)
        out["variable_size_training_b2"] = dict(metric="Gaussians/s encoder fwd+bwd, EAGER launches, 2 chunks per step of 60,000-192,000 Gaussians each (a new plan shape every step), 1 GPU",
                                                value=tot / dt, unit="Gaussians/s", ms_per_step=dt / steps * 1e3, steps=steps,
                                                gaussians_per_step=tot / steps, first_pass_ms_per_step=dt0 / steps * 1e3,
                                                first_pass_value=tot0 / dt0, dtype="bf16")
        gc.collect(); torch.cuda.empty_cache()
        # ---- the Mix3D regime (round 4): the SAME batches with the two samples merged into one batch element, as point_collate_fn
        # does in 80 % of the reference's training steps -- duplicate voxels at level 0, handled inside the MFMA conv paths
        tot0, dt0, _ = eager_line(True)            # (first pass: the merged batches pool to level sizes -- GEMM shape classes -- of their own)
        tot, dt, dups = eager_line(True)
        out["mix3d_training_b2"] = dict(metric="Gaussians/s encoder fwd+bwd, EAGER launches, the variable-size batches with both samples Mix3D-merged into one batch element (duplicate voxels at level 0), 1 GPU",
                                        value=tot / dt, unit="Gaussians/s", ms_per_step=dt / steps * 1e3, steps=steps,
                                        gaussians_per_step=tot / steps, duplicate_rows_last_batch=dups,
                                        first_pass_ms_per_step=dt0 / steps * 1e3, first_pass_value=tot0 / dt0,
                                        vs_variable_size_training_b2=(tot / dt) / out["variable_size_training_b2"]["value"], dtype="bf16")
        # ---- the stress fixture of SURVEY 8d ("uniform-102400": unique voxels uniform in 300 x 300 x 150, ~1-2 neighbours per site,
        # pooled levels that barely shrink: 102,400 / 99,735 / 81,633 / 26,646): real scenes sit between it and the room
        from scenesplat_amd.synthetic import uniform_chunk
        udata = {k: v.cuda() for k, v in uniform_chunk(seed=0).items()}
        dt = run(model, model, udata, False, "uniform-102400 stress fixture", steps)
        out["uniform_102400_stress_fixture"] = dict(metric="Gaussians/s encoder fwd+bwd, uniform-102400 stress fixture (NOT the metric workload), eager launches, 1 GPU",
                                                    value=udata["feat"].shape[0] / dt, unit="Gaussians/s", ms_per_step=dt * 1e3, steps=steps, dtype="bf16")
        del model, udata
        gc.collect(); torch.cuda.empty_cache()
        # ---- BASELINE config 5 end to end: 1,000,000-Gaussian region -> LangPretrainer.eval()(input, chunk_size=600000) (the call
        # form of engines/test.py:329-351 / evaluator.py:762) -> 160-label feature x text scan
        from scenesplat_amd import native as nv5
        torch.manual_seed(1)
        model = MODELS.build(dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **LANG_PTV3), criteria=[])).cuda().eval()
        data = {k: v.cuda() for k, v in room_chunk(800, 2, lang_dim=0).items()}
        text = torch.nn.functional.normalize(torch.randn(160, 768, device="cuda"), dim=1).to(torch.bfloat16)
        torch.cuda.reset_peak_memory_stats()
        times = []
        for i in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                feat = model(dict(data), chunk_size=600000)["point_feat"]["feat"]
            mp_, am_ = nv5.feat_text_scan(feat.to(torch.bfloat16).contiguous(), text)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
            del feat
        n5 = data["feat"].shape[0]
        out["config5_open_vocab_inference_1m"] = dict(metric="Gaussians/s open-vocabulary inference end to end: 1,000,000-Gaussian region, chunk_size=600000 encoder forward (2 chunks) + 160-label scan, 1 GPU",
                                                      value=n5 / min(times[1:]), unit="Gaussians/s", ms_per_region=min(times[1:]) * 1e3,
                                                      peak_mem_GiB=torch.cuda.max_memory_allocated() / 2**30, dtype="bf16")
        del model, data
        gc.collect(); torch.cuda.empty_cache()
    except Exception as e:   # noqa: BLE001  (secondary lines must never take the headline down)
        log("secondary: %s: %s" % (type(e).__name__, e))
        out["error"] = "%s: %s" % (type(e).__name__, e)
    finally:
        RUNTIME.clear(); RUNTIME.update(old)
    return out


def pointops_lines(log):
    """Secondary lines for the libs/pointops* replacements (round 4): queries/s and achieved GB/s of the kernels behind
    knn_query / ball_query / farthest_point_sampling / grouping and the evaluator's neighbour voting (evaluator.py:697-739), on room
    clouds of 102,400 and 1,000,000 Gaussians.  Timed with CUDA events on the current stream, 3 repetitions, best."""
    from scenesplat_amd import pointops as po
    from scenesplat_amd.synthetic import room_chunk
    out = {}

    def best_ms(fn, reps=3):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return min(ts)
    try:
        small = room_chunk(256, 0, lang_dim=0)["coord"].cuda().contiguous()
        big = room_chunk(800, 2, lang_dim=0)["coord"].cuda().contiguous()
        for name, xyz in (("102400", small), ("1000000", big)):
            n = xyz.shape[0]
            off = torch.tensor([n], dtype=torch.int32, device="cuda")
            ms = best_ms(lambda: po.knn_query(25, xyz, off, impl="grid"))
            out["knn_k25_grid_%s" % name] = dict(ms=ms, queries_per_s=n / ms * 1e3, what="exact kNN, k = 25, self query, hash-grid ring search incl. grid build (csrc/knn_grid.hip)")
            log("pointops: knn k=25 grid, n=%s: %.2f ms (%.1f M queries/s)" % (name, ms, n / ms / 1e3))
        n = small.shape[0]
        off = torch.tensor([n], dtype=torch.int32, device="cuda")
        ms = best_ms(lambda: po.knn_query(25, small, off, impl="brute"), reps=2)
        out["knn_k25_brute_102400"] = dict(ms=ms, queries_per_s=n / ms * 1e3, what="the reference's algorithm (O(m n) scan, LDS-tiled): %.1f G distance evaluations/s" % (n * n / ms / 1e6))
        log("pointops: knn k=25 brute force, n=102400: %.2f ms" % ms)
        g = torch.Generator(device="cuda").manual_seed(3)
        nb = big.shape[0]
        lab = torch.randint(0, 160, (nb,), device="cuda", generator=g).int()
        val = torch.rand(nb, device="cuda", generator=g) < 0.9
        ms = best_ms(lambda: po.neighbor_voting(big, lab, val, 25, -1, 160))
        out["neighbor_voting_1m_k25"] = dict(ms=ms, gaussians_per_s=nb / ms * 1e3, what="evaluator.py:697-739: kNN (k = 25) among the 90 % valid Gaussians for all 1,000,000 + majority vote (reference: CPU cKDTree + numba)")
        log("pointops: neighbor voting, 1,000,000 Gaussians, k=25: %.2f ms" % ms)
        ms = best_ms(lambda: po.ball_query(16, 0.1, 0.0, small, off, impl="brute"), reps=2)
        out["ball_query_16_r0.1_brute_102400"] = dict(ms=ms, queries_per_s=n / ms * 1e3, what="ball_query nsample 16, radius 0.1 m: the reference's algorithm (one thread per query scanning its batch element, 16 KiB of candidate scratch each)")
        ms = best_ms(lambda: po.ball_query(16, 0.1, 0.0, small, off, impl="grid"))
        out["ball_query_16_r0.1_102400"] = dict(ms=ms, queries_per_s=n / ms * 1e3, what="the same result on the hash grid (one wave per query, candidates sorted in LDS), incl. grid build")
        noff = torch.tensor([n // 4], dtype=torch.int32, device="cuda")
        ms = best_ms(lambda: po.farthest_point_sampling(small, off, noff), reps=1)
        out["fps_102400_to_25600"] = dict(ms=ms, samples_per_s=(n // 4) / ms * 1e3, what="farthest point sampling 102,400 -> 25,600 (one workgroup per batch element, sequential by definition)")
        idx, _ = po.knn_query(16, small, off, impl="grid")
        feat = torch.randn(n, 64, device="cuda", generator=g).requires_grad_(True)
        grp = po.grouping2(feat, idx)
        go = torch.randn_like(grp)
        ms_f = best_ms(lambda: po.grouping2(feat, idx))
        ms_b = best_ms(lambda: torch.autograd.grad(po.grouping2(feat, idx), feat, go)) - ms_f
        by = n * 16 * 64 * 4
        out["grouping_16x64_102400"] = dict(fwd_ms=ms_f, fwd_GBps=(by + n * 64 * 4) / ms_f / 1e6, bwd_ms=ms_b, bwd_GBps=(by + n * 64 * 4) / max(ms_b, 1e-3) / 1e6,
                                            what="grouping (gather of 16 neighbour rows of 64 fp32 channels) forward / backward (fp32 atomics)")
        log("pointops: ball grid %.2f ms (brute %.2f), fps %.1f ms, grouping fwd %.3f / bwd %.3f ms" % (
            out["ball_query_16_r0.1_102400"]["ms"], out["ball_query_16_r0.1_brute_102400"]["ms"], out["fps_102400_to_25600"]["ms"], ms_f, ms_b))
    except Exception as e:  # noqa: BLE001  (secondary lines must never take the headline down)
        log("pointops lines: %s: %s" % (type(e).__name__, e))
        out["error"] = "%s: %s" % (type(e).__name__, e)
    return out


def main():
    args = parse()
    if os.environ.get("SS_BENCH_PAGE_IN", "1") != "0":
        _page_in_gemm_code_objects()
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
    # SS_BENCH_FORCE_DDP=1 (DIAGNOSTIC, single rank): a one-rank RCCL process group and the DDP wrapper of the N > 1 path -- bucket hooks,
    # bucket views and a (trivial) all-reduce per bucket run as they do with more ranks; the line it prints is not the metric
    force_ddp = world == 1 and os.environ.get("SS_BENCH_FORCE_DDP") in ("1", "stage", "ddp")
    if force_ddp and os.environ.get("SS_BENCH_FORCE_DDP") in ("stage", "ddp"):
        args.exchange = os.environ["SS_BENCH_FORCE_DDP"]
    elif force_ddp:
        args.exchange = "ddp"
    if force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
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
    exchange = None
    if (world > 1 or force_ddp) and args.exchange == "ddp":
        # DDP as the reference builds it (engines/defaults.py:13-34): broadcast_buffers=False
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False,
                                                        gradient_as_bucket_view=True, bucket_cap_mb=100)
    elif world > 1 or force_ddp:
        # round 3 (default): the same gradient average as one all-reduce per model STAGE, issued when the stage's gradients become
        # final in the backward pass (dec0 first: 52 % of the bytes, overlapped with the rest of the backward); no wrapper, no
        # per-parameter bucket copies (DDP: +2.3 ms of copies per step on one rank, round 2)
        from scenesplat_amd.grad_exchange import StageGradExchange
        for t_ in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t_.data, src=0)
        # no host callbacks inside the backward: the step stays a hipGraph replay on every rank.  Round 4, the SPLIT form: two
        # graphs cut where the backward leaves dec0 (52 % of the gradient bytes are final there); dec0's slice is all-reduced on the
        # process group's stream while the second graph runs the rest of the backward, the other slice after it.  --graph off runs the
        # same two-call backward and the same two collectives with eager launches (SS_BENCH_EXCHANGE_HOOKS=1: the round-3 hook form)
        exchange = StageGradExchange(model, force=force_ddp, hooks=(args.graph == "off" and os.environ.get("SS_BENCH_EXCHANGE_HOOKS") == "1"))
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
    # SS_BENCH_PLAN_AHEAD=0 is a DIAGNOSTIC: build each plan between two steps on the step loop's own thread (round 2's form)
    plan_ahead = None
    if os.environ.get("SS_BENCH_PLAN_AHEAD", "1") != "0" and os.environ.get("SS_BENCH_REUSE_PLAN") != "1":
        from scenesplat_amd.plan import PlanAhead
        plan_ahead = PlanAhead(lambda: model.prepare_plan(data, stream=side), depth=2)
    state = {"plan": plan_ahead.get() if plan_ahead is not None else model.prepare_plan(data, stream=side)}

    cot16 = cot.to(torch.bfloat16)

    split = exchange is not None and not exchange._hooks and os.environ.get("SS_BENCH_EXCHANGE_SPLIT", "1") != "0"
    cutbox = {}
    from scenesplat_amd.pointcept_api.ptv3 import backward_tail

    def fwd_bwd(plan, t):
        cut = [] if split else None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(dict(feat=t["feat"], grid_coord=data["grid_coord"], offset=data["offset"], plan=plan,
                           **({"backward_cut": cut} if split else {})))
        # backward from the seeded random cotangent, fed directly as the output gradient (no loss kernels)
        torch.autograd.backward(out.feat, grad_tensors=t["cot"])     # split: down to the inputs of dec0 only (the model's backward cut)
        if split:
            exchange.pack("early")         # dec0's gradients are final: into their slice (captured with graph 1)
            cutbox["cut"] = cut
        elif exchange is not None:
            if exchange._hooks:
                exchange.finish()          # hook form: the stage all-reduces were issued by hooks during the backward
            else:
                exchange.pack()            # round-3 packed form: one all-reduce over the whole model follows the replay (step())
        return {"feat": out.feat}

    def fwd_bwd_tail():
        backward_tail(cutbox.pop("cut"))   # the rest of the backward (graph 2), while dec0's all-reduce runs on RCCL's stream
        exchange.pack("late")

    # Steady state (scenesplat_amd/steady_state.py): every chunk of the room has the same plan SHAPE, so after two eager
    # steps the ~1,100 launches of forward + backward are captured in a hipGraph and later steps replay it -- the plan is
    # still built anew every step (on the side stream) and copied into the captured plan's tensors; every kernel still
    # runs.  With more than one rank the step stays eager: DDP's bucket hooks are host callbacks.
    # N > 1 (round 3): with the stage exchange in its packed form the step is replayed as a graph on every rank too
    use_graph = (args.graph == "on" or (args.graph == "auto" and (world == 1 or args.exchange == "stage"))) and not (force_ddp and args.exchange == "ddp") \
        and not (world > 1 and args.exchange == "ddp")
    from scenesplat_amd.steady_state import CaptureInvalidated, SteadyStateStep
    steady = SteadyStateStep(fwd_bwd, list(model.parameters()), warmup=1, enabled=use_graph,
                             tail=fwd_bwd_tail if split else None, between=(lambda: exchange.reduce_begin("early")) if split else None)

    if os.environ.get("SS_BENCH_INJECT_CAPTURE_INVALIDATED") == "1" and os.environ.get("SS_BENCH_CHILD") != "1":
        # TEST HOOK (tests/test_hip_bench_paths.py): the first capture attempt reports an invalidated capture without touching the
        # stream, so that the handler below -- fresh child process with eager launches, its JSON line forwarded -- is exercised
        def _injected(plan_, inputs_):
            raise CaptureInvalidated("injected by SS_BENCH_INJECT_CAPTURE_INVALIDATED")
        steady._capture = _injected

    seg = {"zero": 0.0, "steady": 0.0, "reduce": 0.0, "plan": 0.0} if os.environ.get("SS_BENCH_HOST_SEGMENTS") else None

    def step():
        t_a = time.perf_counter()
        net.zero_grad(set_to_none=True)
        plan, state["plan"] = state["plan"], None
        t_b = time.perf_counter()
        try:
            steady(plan, {"feat": data["feat"], "cot": cot16})
        except CaptureInvalidated as e:
            # An invalidated capture cannot be ended, and this thread still owns it: the allocator keeps the capture's pool
            # registered, device-wide synchronisation and empty_cache() are refused from here on.  This PROCESS is done.  One rank:
            # a FRESH child process repeats the run with --graph off and its JSON line becomes ours (started as a child, never an
            # exec of a process that has initialised the GPU).  More ranks: exit non-zero at once -- torchrun ends the other ranks;
            # a hang is the one outcome that must not happen.
            log("hipGraph capture invalidated: %s" % e)
            if world == 1 and not force_ddp and os.environ.get("SS_BENCH_CHILD") != "1":
                import subprocess
                argv = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + ["--graph", "off"]
                log("re-running in a fresh child process with eager launches: %s" % " ".join(argv[1:]))
                r = subprocess.run(argv, env=dict(os.environ, SS_BENCH_CHILD="1"), stdout=subprocess.PIPE)
                os.write(_REAL_STDOUT, r.stdout)
                os._exit(r.returncode)
            os._exit(3)
        t_c = time.perf_counter()
        if split:
            exchange.reduce_begin("late"); exchange.reduce_end()
        elif exchange is not None and not exchange._hooks:
            exchange.reduce()
        t_d = time.perf_counter()
        # SS_BENCH_REUSE_PLAN=1 is a DIAGNOSTIC (host- vs GPU-bound?): it skips the per-step plan build and the line it
        # prints is not the metric
        if plan_ahead is not None:
            state["plan"] = plan_ahead.get()       # built by the plan thread while earlier steps ran (scenesplat_amd/plan.py:PlanAhead)
        else:
            state["plan"] = plan if os.environ.get("SS_BENCH_REUSE_PLAN") == "1" else model.prepare_plan(data, stream=side)
        if seg is not None:      # DIAGNOSTIC: where the host spends a step (ms, summed over all steps incl. warm-up)
            seg["zero"] += t_b - t_a; seg["steady"] += t_c - t_b; seg["reduce"] += t_d - t_c; seg["plan"] += time.perf_counter() - t_d

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    # at least 3 untimed steps before the timed region (allocator pools of both streams, hipBLASLt heuristics, lazy
    # code-object loads settle within the first three): the W warm-up steps asked for, topped up when W < 3
    for i in range(max(args.warmup, 3)):
        t_ = time.perf_counter(); step(); torch.cuda.synchronize()
        log("warmup step %d: %.1f ms" % (i, (time.perf_counter() - t_) * 1e3))
    # the steady-state path needs an eager step, a sync-checked eager step and the capturing step before it replays: all
    # of them belong to the warm-up, never to the timed region
    extra = 0
    while use_graph and steady.replays == 0 and steady.refused is None and extra < 4:
        t_ = time.perf_counter(); step(); torch.cuda.synchronize(); extra += 1
        log("extra warmup step (graph capture): %.1f ms" % ((time.perf_counter() - t_) * 1e3))
    torch.cuda.synchronize()
    if os.environ.get("SS_BENCH_GC", "freeze") == "freeze":
        # one full collection now, then keep the survivors out of later generation-2 scans: a step allocates ~10^5
        # Python objects (autograd nodes, ctypes wrappers) and an untimely full collection costs ~50 ms
        import gc
        gc.collect(); gc.freeze()
    # --graph auto on one rank: replay and eager launches run the same kernels, and since round 3 the host enqueues a step (27 ms)
    # faster than the GPU runs it, so neither is host-bound; which one is faster depends on the process (the replayed step re-casts
    # the weight shadows and copies the plan in, 0.3-0.4 ms; its seam is 0.3-1 ms; a cold process replays 2-3 ms slower).  Untimed
    # calibration: a few steps each way, the faster one runs the timed region (and is named in the JSON line).
    calib = None
    if use_graph and args.graph == "auto" and world == 1 and exchange is None and steady.replays > 0 and steady.refused is None \
            and os.environ.get("SS_BENCH_CALIBRATE", "1") != "0":
        calib = {}
        for mode in ("replay", "eager", "replay", "eager"):
            steady.enabled = mode == "replay"
            step(); torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            calib[mode] = min(calib.get(mode, 1e9), (time.perf_counter() - t_) / 4 * 1e3)
        steady.enabled = calib["replay"] <= calib["eager"]
        log("calibration: replay %.2f ms/step, eager %.2f ms/step -> %s" % (calib["replay"], calib["eager"], "replay" if steady.enabled else "eager launches"))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if use_graph:
        log("steady state: %d replays, %d eager steps so far%s" % (steady.replays, steady.eager_steps,
                                                                     "; capture REFUSED: " + steady.refused if steady.refused else ""))
    replays_before = steady.replays
    t0 = time.perf_counter()
    t_enq = 0.0
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
        t_enq = time.perf_counter() - t0       # host time to ENQUEUE the timed steps (diagnostic: host- vs GPU-bound)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if plan_ahead is not None:
        plan_ahead.close()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0 and seg is not None:
        log("host segments over all steps (s): " + ", ".join("%s %.3f" % kv for kv in seg.items()))
    if rank == 0 and exchange is not None and exchange.prof is not None:
        log("exchange host time per step: launches %.2f ms, finish %.2f ms (%d steps)" % (
            1e3 * exchange.prof["launch_s"] / max(1, exchange.prof["steps"]), 1e3 * exchange.prof["finish_s"] / max(1, exchange.prof["steps"]), exchange.prof["steps"]))
    if rank == 0:
        res = {
            "metric": "Gaussians/s encoder fwd+bwd, 102k-pt chunks", "value": world * n * args.steps / dt,
            "unit": "Gaussians/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": ("synthetic, DIAGNOSTIC: one-rank DDP rehearsal (invalid as metric)" if force_ddp else
                                     "synthetic" if os.environ.get("SS_BENCH_REUSE_PLAN") != "1" else "synthetic, DIAGNOSTIC: plan reused (invalid as metric)"),
            "config": {"workload": (("uniform-%d (stress fixture, NOT the metric workload):" if args.fixture == "uniform" else "room-%d:")
                                    + " PT-v3m1 lang-pretrain encoder (91.71M params, in=11, out=768) fwd+bwd, "
                                    "1 chunk of %d Gaussians per GPU per step, serialization included") % (n, n),
                       "gaussians_per_chunk": n, "chunks_per_gpu": 1,
                       "parallelism": "dp%d" % world,
                       "gradient_exchange": ((args.exchange + (" (split: dec0 slice all-reduced under the rest of the backward)" if split else ""))
                                             if (world > 1 or force_ddp) else "none"), "attention_kernel": "mfma" if impl == nv.ATTN_MFMA else "simt",
                       "execution": ((("hipGraph replay of forward+backward (%d of the %d timed steps; a plan built for every step, by the plan thread, and copied in)"
                                       % (steady.replays - replays_before, args.steps)) if steady.replays > replays_before else "eager launches")
                                     + ("" if calib is None else "; chosen by an untimed calibration in this process: replay %.2f vs eager %.2f ms/step" % (calib["replay"], calib["eager"])))},
        }
        log("timed %d steps: %.1f ms/step (host enqueue %.1f ms/step)" % (args.steps, dt / args.steps * 1e3, t_enq / args.steps * 1e3))
        if world == 1:
            import gc
            state.clear()
            net = model = None                                 # (cells of step(): releases the model and its plan)
            gc.collect(); torch.cuda.empty_cache()
            rl = roofline_lines(log, not args.no_pmc, args.power)
            plp = power_limited_peak(log)
            for ent in rl.values():
                if ent.get("bound") == "mfma" and plp:
                    ent["power_limited_peak"] = plp["value"]
                    ent["frac_of_power_limited_peak"] = ent["achieved"] / plp["value"]
            res["power_limited_mfma_peak"] = plp
            res["roofline"] = rl["conv_fwd"]                  # the dominant kernel of the step
            res["roofline_conv_wgrad"] = rl["conv_wgrad"]
            res["roofline_attn"] = rl["attn_fwd"]             # the kernel the north star names
            res["roofline_attn_bwd"] = rl["attn_bwd"]
            # said plainly in the line itself: north_star asks for >= 50 % of the MFMA peak on window attention
            for key_, tgt in (("roofline_attn", 0.50), ("roofline_attn_bwd", 0.50)):
                f_ = res[key_]["frac"]
                res[key_]["note"] = ("north-star target: >= %.2f of the dense bf16 MFMA peak on window attention -- %s (%.2f)"
                                     % (tgt, "met" if f_ >= tgt else "NOT met", f_))
            res["roofline_hbm"] = rl["unpool_fwd"]            # gather / scatter kernels THE STEP runs (dec0 unpooling seam): forward ...
            res["roofline_hbm_bwd"] = rl["unpool_bwd"]        # ... and backward (the grid-pool scatter the north star names)
            res["roofline_hbm_gather"] = rl["gather_hbm"]     # the row-gather kernel on a working set far beyond the Infinity Cache
            res["roofline_scan"] = rl["scan"]
            del rl
            gc.collect(); torch.cuda.empty_cache()
            if not args.no_secondary:
                res["secondary"] = secondary_lines(log)
                res["secondary"]["pointops"] = pointops_lines(log)
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args.cpu_n_side, log)
        sys.stdout.flush()
        os.write(_REAL_STDOUT, (json.dumps(res) + "\n").encode())
    if world > 1 or force_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
