"""GPU: fused distillation head, LangPretrainer._chunked_forward, bf16 shadow freshness, BASELINE config 3
(B = 8 chunks x 102,400 Gaussians with 768-d targets), and the data-parallel path on the real HIP model
(2 ranks sharing cuda:0 over gloo)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import losses as olosses

pytestmark = pytest.mark.gpu

TINY = dict(in_channels=11, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2),
            enc_depths=(1, 1, 1), enc_channels=(16, 32, 48), enc_num_head=(1, 2, 3), enc_patch_size=(64, 64, 16),
            dec_depths=(1, 1), dec_channels=(48, 32), dec_num_head=(1, 2), dec_patch_size=(64, 64))
CRIT = [dict(type="CosineSimilarity", reduction="mean", loss_weight=1.0),
        dict(type="L2Loss", reduction="mean", loss_weight=1.0),
        dict(type="AggregatedContrastiveLoss", temperature=0.2, reduction="mean", loss_weight=0.02, schedule="last_75")]


class _Runtime:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        from scenesplat_amd.pointcept_api import RUNTIME
        self.old = dict(RUNTIME); RUNTIME.update(self.kw)

    def __exit__(self, *a):
        from scenesplat_amd.pointcept_api import RUNTIME
        RUNTIME.clear(); RUNTIME.update(self.old)


# ---- fused head ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [48, 768])
def test_fused_head_matches_oracle(C):
    """normalize + cosine + L2 in one pass (csrc/head.hip) against F.normalize + the oracle's losses, values and the
    gradient w.r.t. the un-normalised features, with an extra gradient arriving at the unit features."""
    from scenesplat_amd import functional as SF
    g = torch.Generator().manual_seed(C)
    n = 3001
    feat = torch.randn(n, C, generator=g) * torch.rand(n, 1, generator=g) * 5
    tgt = F.normalize(torch.randn(n, C, generator=g), dim=1)
    tgt[5] = 0                                                     # zero-norm target row (eps path of the cosine)
    mask = torch.rand(n, generator=g) < 0.85
    wextra = torch.randn(n, C, generator=g) * 1e-3
    fo = feat.clone().requires_grad_(True)
    po = F.normalize(fo, p=2, dim=1)
    lo = 0.7 * olosses.cosine_similarity_loss(po, tgt, mask) + 1.3 * olosses.l2_loss(po, tgt, mask) + (po * wextra).sum()
    lo.backward()
    fg = feat.cuda().requires_grad_(True)
    p, sums = SF.lang_head(fg, tgt.cuda(), mask.cuda(), True)
    loss = 0.7 * sums[0] / sums[2] + 1.3 * sums[1] / sums[2] + (p * wextra.cuda()).sum()
    loss.backward()
    SF.lang_head_release()
    assert int(sums[2].item()) == int(mask.sum())
    assert torch.allclose(p.detach().cpu(), po.detach(), atol=2e-7, rtol=1e-6)
    assert abs(loss.item() - lo.item()) < 3e-6 * abs(lo.item())
    rel = (fg.grad.cpu() - fo.grad).norm() / fo.grad.norm()
    print("fused head C=%d: loss %.6f vs %.6f, grad rel err %.2e" % (C, loss.item(), lo.item(), rel))
    assert rel < 1e-5
    # bf16 features in (what a bf16 backbone would hand over): same math on the rounded values
    fb = feat.to(torch.bfloat16).cuda().requires_grad_(True)
    pb, sb = SF.lang_head(fb, tgt.cuda(), mask.cuda(), True)
    (sb[0] / sb[2]).backward()
    SF.lang_head_release()
    fo2 = feat.to(torch.bfloat16).float().requires_grad_(True)
    l2 = olosses.cosine_similarity_loss(F.normalize(fo2, dim=1), tgt, mask)
    l2.backward()
    assert abs((sb[0] / sb[2]).item() - l2.item()) < 1e-5
    assert (fb.grad.float().cpu() - fo2.grad).norm() / fo2.grad.norm() < 6e-3          # bf16 gradient storage


def test_criteria_share_one_head_pass_and_match_reference(golden_dir):
    """LangPretrainer's criteria fan-out: CosineSimilarity and L2Loss find the fused pass by identity (no second read),
    and the standalone call form (pred already unit rows) reproduces the reference's golden losses."""
    from scenesplat_amd import functional as SF
    from scenesplat_amd.pointcept_api import build_criteria
    fx = np.load(os.path.join(golden_dir, "losses.npz"))
    pred0, tgt = torch.from_numpy(fx["pred"]).cuda(), torch.from_numpy(fx["tgt"]).cuda()
    mask, seg = torch.from_numpy(fx["mask"]).cuda(), torch.from_numpy(fx["seg"]).cuda()
    crit = build_criteria(CRIT)
    calls = []
    orig = SF._LangHead.apply
    SF._LangHead.apply = staticmethod(lambda *a: (calls.append(a[3]), orig(*a))[1])
    try:
        raw = (pred0 * 3.0).requires_grad_(True)                       # un-normalised features with the same directions
        p, _ = SF.lang_head(raw, tgt, mask, True)
        loss = crit(p, tgt, valid_feat_mask=mask, segment=seg, epoch_progress=0.1)
        SF.lang_head_release()
        assert calls == [True]                                         # one fused pass; neither loss made its own
        loss.backward()
        assert abs(loss.item() - float(fx["loss_ep0.1"])) < 3e-6 * abs(float(fx["loss_ep0.1"]))
        calls.clear()
        pr = pred0.clone().requires_grad_(True)
        loss2 = crit(pr, tgt, valid_feat_mask=mask, segment=seg, epoch_progress=0.1)
        assert calls == [False]                                        # standalone: ONE shared un-normalised pass
        loss2.backward()
        assert abs(loss2.item() - float(fx["loss_ep0.1"])) < 2e-6 * abs(float(fx["loss_ep0.1"])) + 1e-6
        assert torch.allclose(pr.grad.cpu(), torch.from_numpy(fx["dpred_ep0.1"]), atol=1e-7, rtol=1e-4)
    finally:
        SF._LangHead.apply = orig
        SF.lang_head_release()


# ---- LangPretrainer._chunked_forward ----------------------------------------------------------------------------
def _tiny_lang(drop_path=0.0):
    from scenesplat_amd.pointcept_api import MODELS
    torch.manual_seed(5)
    return MODELS.build(dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **TINY, drop_path=drop_path, shuffle_orders=False),
                             criteria=CRIT)).cuda()


def _tiny_input(n_side=40, seed=0):
    from scenesplat_amd.synthetic import room_chunk
    d = {k: v.cuda() for k, v in room_chunk(n_side=n_side, seed=seed, lang_dim=48, num_classes=4).items()}
    d["epoch_progress"] = 0.6
    return d


def test_chunked_forward_eval_equals_per_chunk_forwards():
    """models/default.py:115-176: chunk_size < N in eval -> the concatenation of independent per-chunk forwards
    (each chunk re-serialised on its own), unit rows."""
    model = _tiny_lang().eval()
    d = _tiny_input()
    n = d["coord"].shape[0]
    cs = 1000
    assert n > 2 * cs
    with torch.no_grad():
        torch.manual_seed(3)
        full = model(d, chunk_size=cs)["point_feat"]["feat"]
        torch.manual_seed(3)
        parts = []
        for s in range(0, n, cs):
            e = min(s + cs, n)
            sub = {k: v[s:e] for k, v in d.items() if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n}
            sub["offset"] = torch.tensor([e - s], device="cuda"); sub["epoch_progress"] = 0.6
            parts.append(model(sub)["point_feat"]["feat"])
        whole = model(d, chunk_size=600000)["point_feat"]["feat"]        # evaluator.py:762 call form, N < chunk_size
    assert full.shape == (n, 48) and torch.equal(full, torch.cat(parts))
    assert torch.allclose(full.norm(dim=1), torch.ones(n, device="cuda"), atol=1e-5)
    assert not torch.allclose(full, whole, atol=1e-3)                    # chunking changes the context, as in the reference


def test_chunked_forward_training_bf16_runs_several_forwards_before_one_backward():
    """Training with chunk_size < N under bf16 autocast: one backbone forward per chunk, ONE backward over the mean of
    the chunk losses.  The bf16 weight shadows must not be overwritten between those forwards (the autograd graph of
    the first chunk saved them), and the loss must equal the mean of the per-chunk losses."""
    from scenesplat_amd import native as nv
    model = _tiny_lang().train()
    d = _tiny_input()
    n = d["coord"].shape[0]
    cs = 1200
    with _Runtime(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA):
        torch.manual_seed(4)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(d, chunk_size=cs)
        out["loss"].backward()
        g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
        assert all(torch.isfinite(g).all() for g in g1.values())
        model.zero_grad()
        torch.manual_seed(4)
        losses = []
        with torch.autocast("cuda", dtype=torch.bfloat16):
            for s in range(0, n, cs):
                e = min(s + cs, n)
                sub = {k: v[s:e] for k, v in d.items() if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n}
                sub["offset"] = torch.tensor([e - s], device="cuda"); sub["epoch_progress"] = None
                losses.append(model(sub)["loss"])
        ref = torch.stack(losses).mean()
    # not bit-equal by construction: the second pass starts from the BatchNorm running means the first pass left behind, and those
    # condition the one-pass batch variance (sums of x - running_mean); measured 2e-6 .. 2e-5 relative depending on the reduction
    # tree of the statistics.  A stale or overwritten weight shadow shows up at 1e-2.
    assert abs(out["loss"].item() - ref.item()) < 1e-4 * abs(ref.item()) + 1e-6


def test_bf16_shadows_follow_the_parameters():
    """autocast train step -> optimizer.step -> eval forward WITHOUT autocast (the evaluator's call form) with the conv
    in bf16: must use the updated conv weights (equal to a fresh model holding the same state dict), also with the
    refresh switched off."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.pointcept_api import MODELS
    d = _tiny_input(32, 1)
    inp = dict(feat=d["feat"], grid_coord=d["grid_coord"], offset=d["offset"])
    for shadows in (True, False):
        with _Runtime(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_SIMT, param_shadows=shadows):
            torch.manual_seed(0)
            model = MODELS.build(dict(type="PT-v3m1", **TINY, drop_path=0.0, shuffle_orders=False)).cuda().train()
            opt = torch.optim.SGD(model.parameters(), lr=0.5)
            for _ in range(2):
                torch.manual_seed(1)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = model(dict(inp)).feat
                opt.zero_grad(); y.float().square().mean().backward(); opt.step()
            model.eval()
            fresh = MODELS.build(dict(type="PT-v3m1", **TINY, drop_path=0.0, shuffle_orders=False)).cuda().eval()
            fresh.load_state_dict(model.state_dict())
            with torch.no_grad():
                torch.manual_seed(2); a = model(dict(inp)).feat
                torch.manual_seed(2); b = fresh(dict(inp)).feat
            assert torch.equal(a, b), (shadows, (a - b).abs().max())


def test_shadows_recorded_inside_a_capture_serve_that_capture_and_nothing_else(monkeypatch):
    """refresh_shadows inside a hipGraph capture only RECORDS the casts: the shadows must serve the layers of that capture (no
    per-use re-cast kernels in the graph -- 188 small copies and 1.1 ms per replayed step when this was wrong), and must read as
    stale to eager code afterwards (a refused capture leaves weights one optimizer step old in them)."""
    from scenesplat_amd import functional as SF, native as nv
    # the multi-tensor copy is capturable as it stands; the one-launch group cast uploads its descriptor table through the
    # steady-state capture pool (steady_state.py), which this bare capture does not set up
    monkeypatch.setattr(SF, "SHADOW_GROUP_CAST", False)
    lin = torch.nn.Linear(64, 64).cuda()
    src, dst = SF.register_shadows([lin.weight, lin.bias])
    SF.refresh_shadows(src, dst)
    assert SF.bf16_of(lin.weight) is dst[0] and nv.stream_capture_id() == 0
    with torch.no_grad():
        lin.weight.add_(1.0)
    assert SF.bf16_of(lin.weight) is not dst[0]                       # stale after an in-place update
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin(capture_error_mode="thread_local")
        ids = nv.stream_capture_id()
        SF.refresh_shadows(src, dst)
        inside = (SF.bf16_of(lin.weight) is dst[0], SF.bf16_of(lin.bias) is dst[1])
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    assert ids > 0 and inside == (True, True)
    assert SF.bf16_of(lin.weight) is not dst[0]                       # recorded, not run: stale for eager code
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(dst[0], lin.weight.detach().to(torch.bfloat16))
    SF.refresh_shadows(src, dst)                                      # eager refresh makes them current again
    assert SF.bf16_of(lin.weight) is dst[0]


def test_plan_ahead_thread_hands_over_plans_equal_to_direct_builds():
    """PlanAhead (scenesplat_amd/plan.py): plans built on a host thread and a side stream, consumed by the step loop: same
    signature and tensors as a direct build, usable by the model, failures surface in get(), close() ends the thread."""
    from scenesplat_amd.plan import PlanAhead
    from scenesplat_amd.pointcept_api import MODELS
    d = _tiny_input(32, 1)
    inp = dict(feat=d["feat"], grid_coord=d["grid_coord"], offset=d["offset"])
    with _Runtime(conv_dtype=torch.float32, attn_impl=0, param_shadows=False):
        torch.manual_seed(0)
        model = MODELS.build(dict(type="PT-v3m1", **TINY, drop_path=0.0, shuffle_orders=False)).cuda().eval()
        side = torch.cuda.Stream()
        perms = model.draw_perms()            # pooled levels shuffle their curve order per plan (ptv3:371-444): pin it
        ahead = PlanAhead(lambda: model.prepare_plan(inp, perms=perms, stream=side), depth=2)
        direct = model.prepare_plan(inp, perms=perms)
        with torch.no_grad():
            ref = model(dict(inp, plan=direct)).feat
            for _ in range(3):
                plan = ahead.get()
                assert plan.signature() == direct.signature()
                torch.cuda.current_stream().wait_event(plan.ready_event)
                for (ra, ta), (rb, tb) in zip(plan._slots(), direct._slots()):
                    assert ra == rb and ta.shape == tb.shape
                    if ra[1] in ("codes", "order", "inverse", "grid_coord", "batch"):      # (others carry unwritten tails)
                        assert torch.equal(ta, tb), ra
                out = model(dict(inp, plan=plan)).feat
                assert torch.equal(out, ref)
        ahead.close()
        assert not ahead._t.is_alive()

        # torch's sync detector is process-wide: a build that runs while another thread has it armed (steady_state's checked eager
        # step) raises inside the plan thread -- which must wait the window out, not die
        calls = []

        def build():
            calls.append(1)
            return model.prepare_plan(inp, perms=perms, stream=side)
        torch.cuda.set_sync_debug_mode("error")
        try:
            armed = PlanAhead(build, depth=1)
            import time
            time.sleep(0.2)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        plan = armed.get()
        assert plan.signature() == direct.signature() and len(calls) >= 2          # the first build(s) hit the armed window
        armed.close()

        def boom():
            raise ValueError("no batch")
        bad = PlanAhead(boom, depth=1)
        with pytest.raises(RuntimeError, match="plan build thread failed"):
            bad.get()
        bad.close()


# ---- BASELINE config 3 -------------------------------------------------------------------------------------------
def test_config3_lang_pretrainer_b8_x_102400_one_step():
    """ScanNet vision-language pretrain shape: LangPretrainer (PT-v3m1 lang config + 3 criteria), batch = 8 chunks of
    102,400 Gaussians with 768-d targets, one bf16 training step on one GPU: finite loss, every parameter gets a
    finite gradient, peak memory inside one MI355X's 288 GB."""
    from scenesplat_amd.pointcept_api import MODELS, bench_runtime
    from scenesplat_amd.synthetic import LANG_PTV3, room_chunk
    torch.cuda.reset_peak_memory_stats()
    with _Runtime(**bench_runtime()):
        torch.manual_seed(0)
        model = MODELS.build(dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **LANG_PTV3), criteria=CRIT)).cuda().train()
        data = {k: v.cuda() for k, v in room_chunk(n_side=256, seed=0, lang_dim=768, batch=8).items()}
        data["epoch_progress"] = 0.5
        assert data["feat"].shape[0] == 8 * 102400 and data["lang_feat"].shape == (819200, 768)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(data)["loss"]
        loss.backward()
        torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2**30
    print("config 3: loss %.4f, peak memory %.1f GiB" % (loss.item(), peak))
    assert torch.isfinite(loss) and 0.5 < loss.item() < 4.0
    assert peak < 268.0                                             # 288 GB = 268 GiB
    bad = [k for k, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not bad, bad[:5]


@pytest.mark.parametrize("mode", ["fp32", "bench"])
def test_config3_b8_eval_forward_equals_the_eight_stand_alone_forwards(mode):
    """Config 3's batch shape: the eval-BN forward of B = 8 chunks x 102,400 Gaussians equals, row for row, the eight stand-alone
    forwards of its chunks (same weights, same curve permutations) -- the property that proves the batch bits of the codes, the
    per-element window boundaries and the per-element rulebooks at full size (serialization/default.py:21-23, ptv3:141-164).
    fp32 mode (per-tap fp32 conv, fp32-math attention): equal to fp32 summation noise.  bench mode (bf16 autocast, MFMA kernels):
    equal to the run-to-run noise of the bf16 pipeline (tile boundaries move with the batch; fp32 atomics at the small levels)."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.pointcept_api import MODELS, bench_runtime
    from scenesplat_amd.synthetic import LANG_PTV3, room_chunk
    rt = bench_runtime() if mode == "bench" else dict(attn_impl=nv.ATTN_SIMT, conv_dtype=None)
    with _Runtime(**rt):
        torch.manual_seed(0)
        model = MODELS.build(dict(type="PT-v3m1", **dict(LANG_PTV3, shuffle_orders=False))).cuda().eval()
        data = room_chunk(n_side=256, seed=0, lang_dim=0, batch=8)
        offs = [0] + data["offset"].tolist()
        assert offs[-1] == 8 * 102400
        perms = [list(range(4)), [2, 0, 3, 1], [1, 3, 0, 2], [3, 2, 1, 0]]

        def fwd(feat, gc, offset):
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bench")):
                out = model(dict(feat=feat.cuda(), grid_coord=gc.cuda(), offset=offset.cuda()), perms=perms)
            return out.feat.float(), out["plan"]
        y8, plan = fwd(data["feat"], data["grid_coord"], data["offset"])
        assert plan.levels[0].offsets == offs and plan.levels[0].window(0, 1024).num_windows == 800
        worst = (0.0, 0.0)
        for i in range(8):
            a, b = offs[i], offs[i + 1]
            yi, _ = fwd(data["feat"][a:b], data["grid_coord"][a:b], torch.tensor([b - a]))
            cd = float((1 - F.cosine_similarity(y8[a:b].double(), yi.double(), dim=1)).max())
            rel = float((y8[a:b] - yi).norm() / yi.norm())
            worst = (max(worst[0], cd), max(worst[1], rel))
        print("config 3 [%s]: B = 8 eval forward vs the 8 stand-alone forwards: max per-row cosine distance %.2e, worst chunk rel %.2e" % ((mode,) + worst))
        if mode == "fp32":
            assert worst[0] < 1e-9 and worst[1] < 1e-5, worst
        else:
            assert worst[0] < 2e-5 and worst[1] < 5e-3, worst


# ---- data parallel on the real model ---------------------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from scenesplat_amd import native as nv
    from scenesplat_amd.pointcept_api import MODELS, RUNTIME, engine
    from scenesplat_amd.synthetic import room_chunk
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    RUNTIME.update(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA)
    torch.manual_seed(0)
    model = MODELS.build(dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **TINY, drop_path=0.1, shuffle_orders=True),
                              criteria=CRIT)).cuda()
    ddp = engine.create_ddp_model(model, broadcast_buffers=False)
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    opt = torch.optim.AdamW(ddp.parameters(), lr=1e-3)
    ddp.train()
    losses = []
    for step in range(2):
        d = {k: v.cuda() for k, v in room_chunk(n_side=32, seed=10 * rank + step, lang_dim=48, num_classes=4).items()}
        d["epoch_progress"] = 0.6
        torch.manual_seed(100 + rank + step)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = ddp(d)["loss"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    sd = {k: v.detach().float().cpu().numpy().copy() for k, v in model.state_dict().items() if v.is_floating_point() and "running_" not in k}
    q.put((rank, sd, losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_on_the_hip_model_end_with_identical_weights():
    """engines/defaults.py:13-34 on the real HIP LangPretrainer: 2 processes share cuda:0 (gloo; RCCL needs one GPU per
    rank), different data per rank, 2 optimizer steps.  DDP's bucket hooks must see the gradients of the custom
    autograd Functions (incl. the fp32 weight-gradient arena): both ranks end with IDENTICAL weights."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda r: r[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][2] != res[1][2]                                   # different shards -> different losses
    for k in res[0][1]:
        assert np.array_equal(res[0][1][k], res[1][1][k]), k
    assert all(np.isfinite(v).all() for v in res[0][1].values())


def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    from scenesplat_amd import native as nv
    from scenesplat_amd.pointcept_api import MODELS, RUNTIME
    from scenesplat_amd.synthetic import room_chunk
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    RUNTIME.update(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA)
    out = {}
    for wrapped in (False, True):
        torch.manual_seed(0)
        model = MODELS.build(dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **TINY, drop_path=0.0, shuffle_orders=False),
                                  criteria=CRIT[:2])).cuda().train()
        # exactly bench.py's wrapper (engines/defaults.py:13-34 with the bucket view / bucket size of the bench)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False, gradient_as_bucket_view=True,
                                                        bucket_cap_mb=100) if wrapped else model
        opt = torch.optim.SGD(net.parameters(), lr=1e-2)
        for step in range(2):
            d = {k: v.cuda() for k, v in room_chunk(n_side=32, seed=step, lang_dim=48, num_classes=4).items()}
            d["epoch_progress"] = 0.1
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = net(d)["loss"]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        out[wrapped] = torch.cat([p.detach().float().flatten() for p in model.parameters()]).cpu().numpy()
    # the packed stage exchange (grad_exchange.StageGradExchange._reduce_whole) asks RCCL for ReduceOp.AVG: the op must exist on this
    # stack and average a large fp32 buffer exactly on one rank; then the exchange itself in its packed form (force: one-rank rehearsal)
    from scenesplat_amd.grad_exchange import StageGradExchange
    x = torch.randn(1 << 22, device="cuda"); y = x.clone()
    dist.all_reduce(y, op=dist.ReduceOp.AVG)
    avg_ok = bool(torch.equal(x, y))
    ex = StageGradExchange(model, force=True, hooks=False)
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(d)["loss"]
    loss.backward()
    ref = {n_: p.grad.clone() for n_, p in model.named_parameters() if p.grad is not None}
    ex.pack(); ex.reduce()
    torch.cuda.synchronize()
    packed_ok = all(torch.equal(p.grad, ref[n_]) for n_, p in model.named_parameters() if n_ in ref) and len(ex.whole) == 1
    q.put((out[False], out[True], dist.get_backend(), avg_ok, packed_ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_over_rccl_single_rank_matches_the_unwrapped_model():
    """The N > 1 path of bench.py on RCCL itself, as far as one GPU allows: a one-rank "nccl" process group, the HIP LangPretrainer
    under DistributedDataParallel with bench.py's arguments (bucket views, 100 MB buckets), two optimizer steps -- RCCL's communicator,
    the bucket hooks on the custom autograd Functions and the fp32 gradient arena all run; the weights equal those of the unwrapped
    model (a one-rank all-reduce is the identity; bf16 noise apart)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    plain, ddp, backend, avg_ok, packed_ok = q.get(timeout=500)
    p.join(60)
    assert p.exitcode == 0 and backend == "nccl"
    assert avg_ok and packed_ok            # RCCL implements ReduceOp.AVG; pack() + reduce() leave every gradient in its slot, unchanged on one rank
    assert np.isfinite(ddp).all()
    assert np.linalg.norm(ddp - plain) <= 2e-2 * np.linalg.norm(plain)


def _stage_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from scenesplat_amd import native as nv
    from scenesplat_amd.grad_exchange import StageGradExchange
    from scenesplat_amd.pointcept_api import MODELS, RUNTIME
    from scenesplat_amd.synthetic import room_chunk
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    RUNTIME.update(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA)
    torch.manual_seed(0)
    model = MODELS.build(dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **TINY, drop_path=0.1, shuffle_orders=True),
                              criteria=CRIT)).cuda().train()
    ex = StageGradExchange(model)
    assert len(ex.stages) >= 5 and sum(len(st["params"]) for st in ex.stages.values()) == sum(p.requires_grad for p in model.parameters())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    losses = []
    for step in range(2):
        d = {k: v.cuda() for k, v in room_chunk(n_side=32, seed=10 * rank + step, lang_dim=48, num_classes=4).items()}
        d["epoch_progress"] = 0.6
        torch.manual_seed(100 + rank + step)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(d)["loss"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        ex.finish()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    sd = {k: v.detach().float().cpu().numpy().copy() for k, v in model.state_dict().items() if v.is_floating_point() and "running_" not in k}
    q.put((rank, sd, losses))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_stage_grad_exchange_two_ranks_on_the_hip_model_equal_the_ddp_weights():
    """The round-3 gradient exchange (one all-reduce per model stage from post-accumulate hooks, scenesplat_amd/grad_exchange.py) on
    the real HIP LangPretrainer: 2 processes share cuda:0 over gloo, different data per rank, 2 AdamW steps under bf16 autocast.
    Both ranks end with IDENTICAL weights, and those weights are the ones torch DDP produces from the same seeds (the gradient
    of every custom autograd Function -- grouped stage weight gradients, the fp32 arena -- reaches its stage buffer complete)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = {}
    for name, worker in (("stage", _stage_worker), ("ddp", _ddp_worker)):
        world, port = 2, _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        [p.start() for p in procs]
        res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda r: r[0])
        [p.join(60) for p in procs]
        assert all(p.exitcode == 0 for p in procs)
        for k in res[0][1]:
            assert np.array_equal(res[0][1][k], res[1][1][k]), (name, k)
        out[name] = res[0][1]
    num = sum(float(((out["stage"][k] - out["ddp"][k]) ** 2).sum()) for k in out["ddp"]) ** 0.5
    den = sum(float((out["ddp"][k] ** 2).sum()) for k in out["ddp"]) ** 0.5
    print("stage exchange vs DDP after 2 AdamW steps: relative weight difference %.2e" % (num / den))
    assert num <= 2e-3 * den          # fp32 atomics order inside the step; a missing or double-counted gradient is O(lr) = 1e-3 per element


# ---- GPU data fast path: SphereCrop / Collect / collate -----------------------------------------
def test_gpu_sphere_crop_collect_collate_against_reference_semantics():
    """pointcept/datasets/transform.py:1420-1548 (SphereCrop random / center), :320-352 (Collect) and
    datasets/utils.py:8-48 (point_collate_fn + Mix3D) restated in numpy and compared with the device versions."""
    import random
    from scenesplat_amd import gpu_transforms as T
    g = torch.Generator().manual_seed(0)
    n = 5000
    d = dict(coord=torch.rand(n, 3, generator=g) * 4, color=torch.rand(n, 3, generator=g), opacity=torch.rand(n, 1, generator=g),
             quat=torch.randn(n, 4, generator=g), scale=torch.rand(n, 3, generator=g), lang_feat=torch.randn(n, 768, generator=g),
             valid_feat_mask=torch.rand(n, generator=g) < 0.9, segment=torch.randint(-1, 20, (n,), generator=g), name="scene0")
    dg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
    out = T.sphere_crop(dg, point_max=1500, mode="center")
    cg = dg["coord"]
    # distances evaluated with the same fp32 arithmetic as the device (a different summation order could swap near-ties)
    dist2 = lambda ci: (cg - cg[ci]).square().sum(1).float().cpu().numpy()
    idx = np.argsort(dist2(n // 2), kind="stable")[:1500]                                 # transform.py:1513-1517
    for k in ("coord", "color", "opacity", "quat", "scale", "lang_feat", "valid_feat_mask", "segment"):
        assert torch.equal(out[k].cpu(), d[k][torch.from_numpy(idx)]), k
    assert out["name"] == "scene0"
    assert T.sphere_crop(dg, point_max=6000)["coord"] is dg["coord"]                      # small samples pass through
    out = T.sphere_crop(dg, sample_rate=0.25, mode="random", generator=torch.Generator().manual_seed(5))
    ci = int(torch.randint(0, n, (1,), generator=torch.Generator().manual_seed(5)))
    idx = np.argsort(dist2(ci), kind="stable")[:n // 4]
    assert torch.equal(out["coord"].cpu(), d["coord"][torch.from_numpy(idx)])
    # Collect: keys + offset + feat = cat(color, opacity, quat, scale) (configs/...contrastive.py:93)
    col = T.collect(out, keys=("coord", "segment", "lang_feat", "valid_feat_mask"), feat_keys=("color", "opacity", "quat", "scale"))
    assert set(col) == {"coord", "segment", "lang_feat", "valid_feat_mask", "offset", "feat"}
    assert col["feat"].shape == (n // 4, 11) and int(col["offset"][0]) == n // 4
    assert torch.equal(col["feat"], torch.cat([out[k].float() for k in ("color", "opacity", "quat", "scale")], 1))
    # collate 4 samples; Mix3D keeps every second offset (utils.py:44-47)
    samples = []
    for i, m in enumerate((300, 500, 200, 400)):
        samples.append(dict(coord=out["coord"][:m], feat=col["feat"][:m], offset=torch.tensor([m], device="cuda"), name="s%d" % i))
    b = T.point_collate(samples, mix_prob=0.0)
    assert b["offset"].tolist() == [300, 800, 1000, 1400] and b["coord"].shape[0] == 1400 and b["name"] == ["s0", "s1", "s2", "s3"]
    b = T.point_collate(samples, mix_prob=1.0, rng=random.Random(0))
    assert b["offset"].tolist() == [800, 1400]
    # and the collated Mix3D batch runs through the planner (duplicate voxels tolerated)
    from scenesplat_amd.plan import build_plan
    gc = torch.floor(b["coord"] / 0.02).int()
    plan = build_plan(gc - gc.amin(0, keepdim=True), b["offset"], ("z", "hilbert"), (2,))
    assert plan.levels[0].n == 1400


# ---- grouped Linear weight gradients ------------------------------------------------------------------------------------
def test_linear_wgrad_group_matches_fp32_reference():
    """ss_linear_wgrad_group: one launch for many (m, k, n) problems, ragged shapes, with and without bias sums."""
    from scenesplat_amd import native as nv
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(1600, 256, 768), (1600, 256, 256), (1600, 256, 1024), (1600, 1024, 256), (6400, 128, 384), (2047, 64, 192),
              (25600, 512, 512), (1024, 32, 96)]
    items, refs = [], []
    for i, (m, k, n) in enumerate(shapes):
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        dy = torch.randn(m, n, device="cuda", generator=g).to(torch.bfloat16)
        dw = torch.zeros(n, k, device="cuda")
        db = torch.zeros(n, device="cuda") if i % 2 == 0 else None
        items.append((x, dy, dw, db))
        refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
    nv.linear_wgrad_group(items)
    torch.cuda.synchronize()
    for (x, dy, dw, db), (rw, rb) in zip(items, refs):
        assert (dw - rw).norm() <= 1e-5 * rw.norm(), (x.shape, dy.shape)
        if db is not None:
            assert (db - rb).norm() <= 1e-5 * rb.norm()


def test_grouped_stage_wgrads_equal_per_layer_launches(monkeypatch):
    """The stage identity node (functional._StageParams): parameter gradients of the tiny LangPretrainer with the Linear
    weight gradients queued per stage and launched as groups equal those of one launch per layer."""
    from scenesplat_amd import functional as SF
    from scenesplat_amd import native as nv
    d = _tiny_input(40, 2)
    grads = {}
    for mode, cap in (("grouped", 32768), ("byte_capped", 32768), ("single", 0)):
        monkeypatch.setattr(SF, "WGRAD_GROUP_MAX_ROWS", cap)
        # byte_capped: the stage launches what it holds whenever the queued operands exceed the cap (here: nearly every layer)
        monkeypatch.setattr(SF, "WGRAD_GROUP_MAX_BYTES", 200_000 if mode == "byte_capped" else 8 << 30)
        model = _tiny_lang().train()
        calls = []
        orig = nv.linear_wgrad_group
        monkeypatch.setattr(nv, "linear_wgrad_group", lambda items: (calls.append(len(items)), orig(items))[1])
        with _Runtime(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA):
            torch.manual_seed(9)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = model(dict(d))["loss"]
            loss.backward()
        monkeypatch.setattr(nv, "linear_wgrad_group", orig)
        grads[mode] = {k: p.grad.clone() for k, p in model.named_parameters()}
        if mode == "grouped":
            assert sum(calls) >= 10 and len(calls) <= 5, calls         # a handful of launches carry the stage's Linears
            n_items = sum(calls)
        elif mode == "byte_capped":
            assert sum(calls) == n_items and len([c for c in calls if c]) > 5, calls     # the same layers in more, smaller groups
        else:
            assert not calls
    for k in grads["single"]:
        for mode in ("grouped", "byte_capped"):
            a, b = grads[mode][k], grads["single"][k]
            assert a.dtype == b.dtype and (a - b).norm() <= 2e-3 * b.norm() + 1e-7, (mode, k, (a - b).norm() / b.norm())


# ---- steady-state hipGraph replay (scenesplat_amd/steady_state.py) --------------------------------------------
def _steady_setup(drop_path=0.0):
    from scenesplat_amd.pointcept_api import MODELS
    from scenesplat_amd.synthetic import room_chunk
    torch.manual_seed(11)
    model = MODELS.build(dict(type="PT-v3m1", **TINY, drop_path=drop_path, shuffle_orders=True)).cuda().train()
    d = {k: v.cuda() for k, v in room_chunk(n_side=40, seed=3, lang_dim=0).items()}
    n = d["feat"].shape[0]

    def fn(plan, t):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(dict(feat=t["feat"], grid_coord=d["grid_coord"], offset=d["offset"], plan=plan))
        torch.autograd.backward(out.feat, grad_tensors=t["cot"])
        return {"feat": out.feat}
    return model, d, n, fn


def test_steady_state_replay_equals_the_eager_step():
    """Captured fwd+bwd replayed with OTHER features, OTHER curve permutations (shuffle_orders) and after optimizer steps
    equals the eager step on the same plan: output rows and every parameter gradient (fp32 atomics: 2e-3 relative)."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.steady_state import SteadyStateStep
    with _Runtime(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA):
        model, d, n, fn = _steady_setup()
        params = list(model.parameters())
        steady = SteadyStateStep(fn, params, warmup=1)
        g = torch.Generator(device="cuda").manual_seed(1)
        for it in range(5):
            feat = torch.randn(n, 11, device="cuda", generator=g)
            cot = torch.randn(n, TINY["dec_channels"][0], device="cuda", generator=g).to(torch.bfloat16)
            perms = model.draw_perms()
            model.zero_grad(set_to_none=True)
            ref = fn(model.prepare_plan(d, perms=perms), dict(feat=feat, cot=cot))["feat"].detach().float().clone()
            rg = {k: p.grad.clone() for k, p in model.named_parameters()}
            model.zero_grad(set_to_none=True)
            out = steady(model.prepare_plan(d, perms=perms), dict(feat=feat, cot=cot))["feat"].float()
            assert steady.refused is None, steady.refused
            # two EAGER runs of this bf16 step already differ by ~3e-3 (fp32 atomics in the statistics, amplified through
            # the bf16 roundings of a dozen layers); a stale index table or stale weights are O(0.1 - 1)
            assert (out - ref).norm() <= 1e-2 * ref.norm(), (it, (out - ref).norm() / ref.norm())
            errs = {}
            for k, p in model.named_parameters():
                assert p.grad is not None, k
                errs[k] = (float((p.grad - rg[k]).norm()), float(rg[k].norm()))
            num, den = sum(e * e for e, _ in errs.values()) ** 0.5, sum(r * r for _, r in errs.values()) ** 0.5
            assert num <= 2e-2 * den, (it, num / den)
            # tensor by tensor only where the gradient is not a cancelling sum of the random cotangent (bias sums of the
            # first blocks differ by tens of percent between two EAGER runs): every tensor holding >= 2 % of the norm
            for k, (e, r) in errs.items():
                assert r < 0.02 * den or e <= 0.1 * r, (it, k, e / r)
            with torch.no_grad():            # the next replay must see these weights (shadows are re-cast inside the graph)
                for p in params:
                    p.mul_(1.0 + 0.1 * (it % 2 * 2 - 1))
        assert steady.eager_steps == 2 and steady.replays == 3          # warm-up step, sync-checked step, then the graph


def test_steady_state_falls_back_to_eager_on_other_shapes_and_refused_captures():
    from scenesplat_amd import native as nv
    from scenesplat_amd.steady_state import SteadyStateStep
    from scenesplat_amd.synthetic import room_chunk
    with _Runtime(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA):
        model, d, n, fn = _steady_setup()
        steady = SteadyStateStep(fn, model.parameters(), warmup=0)
        cot = torch.randn(n, TINY["dec_channels"][0], device="cuda").to(torch.bfloat16)
        steady(model.prepare_plan(d), dict(feat=d["feat"], cot=cot))          # the sync-checked eager step
        steady(model.prepare_plan(d), dict(feat=d["feat"], cot=cot))
        assert steady.replays == 1 and steady.eager_steps == 1
        # another geometry -> another signature: its first step (warmup=0) is captured too, the first graph stays usable
        d2 = {k: v.cuda() for k, v in room_chunk(n_side=36, seed=4, lang_dim=0).items()}
        n2 = d2["feat"].shape[0]

        def fn2(plan, t):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(dict(feat=t["feat"], grid_coord=d2["grid_coord"], offset=d2["offset"], plan=plan))
            torch.autograd.backward(out.feat, grad_tensors=t["cot"])
            return {"feat": out.feat}
        assert model.prepare_plan(d2).signature() != model.prepare_plan(d).signature()
        # a step that reads a value on the host cannot be captured: the exception is kept, the step runs eagerly
        def syncing(plan, t):
            r = fn2(plan, t)
            float(r["feat"].detach().float().sum())
            return r
        bad = SteadyStateStep(syncing, model.parameters(), warmup=0)
        cot2 = torch.randn(n2, TINY["dec_channels"][0], device="cuda").to(torch.bfloat16)
        out = bad(model.prepare_plan(d2), dict(feat=d2["feat"], cot=cot2))["feat"]
        assert bad.refused is not None and bad.replays == 0, bad.refused      # refused by the sync-checked eager step
        assert torch.isfinite(out.float()).all()
        out = bad(model.prepare_plan(d2), dict(feat=d2["feat"], cot=cot2))["feat"]        # and keeps working
        assert bad.replays == 0 and torch.isfinite(out.float()).all()
        out = steady(model.prepare_plan(d), dict(feat=d["feat"], cot=cot))["feat"]        # the first graph is intact
        assert steady.replays == 2 and torch.isfinite(out.float()).all()


def test_steady_state_error_inside_a_capture_falls_back_to_eager_with_fresh_weights():
    """A step that raises ONLY while its stream captures (round-2 advisor finding): the capture is still healthy, so it is
    ended, that signature is refused (others are not), and the eager fallback runs on CURRENT weights -- an optimizer step taken
    before the failing capture must not be hidden by bf16 shadows stamped during the capture."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.steady_state import SteadyStateStep
    with _Runtime(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA):
        model, d, n, fn = _steady_setup()
        cot = torch.randn(n, TINY["dec_channels"][0], device="cuda").to(torch.bfloat16)

        def failing(plan, t):
            r = fn(plan, t)
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("injected failure inside the capture")
            return r
        steady = SteadyStateStep(failing, model.parameters(), warmup=0)
        perms = model.draw_perms()                                             # one curve permutation for every call below
        steady(model.prepare_plan(d, perms=perms), dict(feat=d["feat"], cot=cot))          # the sync-checked eager step
        with torch.no_grad():                                                  # "optimizer step" before the capture is attempted
            for p in model.parameters():
                p.mul_(1.05)
        out = steady(model.prepare_plan(d, perms=perms), dict(feat=d["feat"], cot=cot))["feat"].float()
        assert steady.refused is not None and "injected" in steady.refused and steady.poisoned is None and steady.replays == 0
        assert nv.stream_capture_status() == 0
        # the fallback equals a plain eager step on the current weights
        for p in model.parameters():
            p.grad = None
        ref = fn(model.prepare_plan(d, perms=perms), dict(feat=d["feat"], cot=cot))["feat"].detach().float()
        # two eager runs of this bf16 step differ by ~3e-3 (fp32 atomics); weights one step stale are O(0.1)
        assert float((out - ref).norm() / ref.norm()) < 1e-2
        out2 = steady(model.prepare_plan(d, perms=perms), dict(feat=d["feat"], cot=cot))["feat"].float()      # refused signature: eager from now on
        assert steady.replays == 0 and torch.isfinite(out2).all()


def test_steady_state_replays_draw_fresh_droppath_masks():
    """DropPath inside a replayed graph: the Philox offset of the default generator advances per replay (torch registers
    the generator with the capture), so two replays on the SAME inputs differ -- and equal inputs with drop_path = 0 do not."""
    from scenesplat_amd import native as nv
    from scenesplat_amd.steady_state import SteadyStateStep
    with _Runtime(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA):
        for dp, differs in ((0.3, True), (0.0, False)):
            model, d, n, fn = _steady_setup(drop_path=dp)
            steady = SteadyStateStep(fn, model.parameters(), warmup=0)
            cot = torch.randn(n, TINY["dec_channels"][0], device="cuda").to(torch.bfloat16)
            perms = model.draw_perms()
            outs = []
            for _ in range(4):
                outs.append(steady(model.prepare_plan(d, perms=perms), dict(feat=d["feat"], cot=cot))["feat"].float().clone())
            assert steady.replays == 3 and steady.refused is None, steady.refused
            rel = float((outs[2] - outs[3]).norm() / outs[3].norm())
            assert (rel > 0.05) if differs else (rel < 1e-2), (dp, rel)


def test_transposed_weight_copies_feed_the_linear_dgrad():
    """ss_transpose16_group against torch, and nn.Linear's input gradient through the (in, out) copy (hipBLASLt NT form) against the plain
    NN form, including a refresh after the weights changed."""
    from scenesplat_amd import functional as SF
    from scenesplat_amd import native as nv
    g = torch.Generator(device="cuda").manual_seed(3)
    mats = [torch.randn(r, c, device="cuda", generator=g).to(torch.bfloat16) for r, c in ((768, 2304), (65, 130), (1, 64), (512, 512))]
    outs = [torch.empty(m.shape[1], m.shape[0], dtype=torch.bfloat16, device="cuda") for m in mats]
    nv.transpose16_group(list(zip(mats, outs)))
    for m, o in zip(mats, outs):
        assert torch.equal(o, m.t().contiguous())
    lin = torch.nn.Linear(256, 512).cuda()
    x = torch.randn(4096, 256, device="cuda", generator=g)
    cot = torch.randn(4096, 512, device="cuda", generator=g).to(torch.bfloat16)

    def run():
        xin = x.clone().requires_grad_(True)
        lin.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = SF.linear(xin, lin.weight, lin.bias)
        y.backward(cot)
        return y.detach().float(), xin.grad.float(), lin.weight.grad.float().clone()
    src, dst = SF.register_shadows([lin.weight, lin.bias])
    SF.refresh_shadows(src, dst)
    y0, dx0, dw0 = run()                                       # NN form
    assert SF.bf16_t_of(lin.weight) is None
    SF.register_transposed([lin.weight])
    SF.refresh_shadows(src, dst)
    wt = SF.bf16_t_of(lin.weight)
    assert wt is not None and torch.equal(wt, lin.weight.detach().to(torch.bfloat16).t().contiguous())
    y1, dx1, dw1 = run()                                       # NT form
    assert torch.equal(y0, y1) and (dx1 - dx0).norm() <= 1e-3 * dx0.norm() and (dw1 - dw0).norm() <= 2e-3 * dw0.norm()
    with torch.no_grad():
        lin.weight.mul_(1.5)
    assert SF.bf16_t_of(lin.weight) is None                    # stale until refreshed: the caller falls back to the NN form
    SF.refresh_shadows(src, dst)
    assert torch.equal(SF.bf16_t_of(lin.weight), lin.weight.detach().to(torch.bfloat16).t().contiguous())
    _, dx2, _ = run()
    assert (dx2 - 1.5 * dx0).norm() <= 1e-2 * (1.5 * dx0).norm()


def test_grouped_weight_mirror_equals_per_weight_launches():
    from scenesplat_amd import native as nv
    g = torch.Generator(device="cuda").manual_seed(5)
    ws = [torch.randn(co, t, ci, device="cuda", generator=g).to(torch.bfloat16) for co, t, ci in ((768, 27, 768), (32, 125, 16), (48, 27, 40), (8, 1, 8))]
    outs = [torch.empty(w.shape[2], w.shape[1], w.shape[0], dtype=torch.bfloat16, device="cuda") for w in ws]
    nv.subm_weight_mirror_group(list(zip(ws, outs)))
    for w, o in zip(ws, outs):
        assert torch.equal(o, nv.subm_weight_mirror(w))
        assert torch.equal(o, w.flip(1).permute(2, 1, 0).contiguous())


def test_trainer_steady_state_option_replays_and_matches_the_eager_trainer():
    """cfg["steady_state"] = True: the Trainer replays forward + backward of repeated batch shapes as a hipGraph and ends, after the
    same batches, with the weights of the plain Trainer (bf16 noise apart); the aggregated contrastive loss (random half split, class sums by
    sort + segment sums, no host reads) is captured with the rest."""
    import tempfile
    from scenesplat_amd import native as nv
    from scenesplat_amd.pointcept_api import engine
    from scenesplat_amd.synthetic import room_chunk

    def loader(k=6):
        base = room_chunk(n_side=40, seed=3, lang_dim=48, num_classes=4)
        g = torch.Generator().manual_seed(0)
        out = []
        for _ in range(k):
            d = {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in base.items()}
            d["feat"] = torch.randn(d["feat"].shape, generator=g)
            d["lang_feat"] = F.normalize(torch.randn(d["lang_feat"].shape, generator=g), dim=1)
            out.append(d)
        return out

    def cfg(tmp, crit, steady):
        return dict(model=dict(type="LangPretrainer", backbone=dict(type="PT-v3m1", **TINY, drop_path=0.0, shuffle_orders=False), criteria=crit),
                    device="cuda", eval_epoch=1, save_path=tmp, enable_amp=True, clip_grad=1.0, steady_state=steady,
                    optimizer=dict(type="AdamW", lr=2e-3, weight_decay=0.05),
                    scheduler=dict(type="OneCycleLR", max_lr=2e-3, pct_start=0.3, anneal_strategy="cos", div_factor=10.0, final_div_factor=100.0),
                    hooks=[])
    with _Runtime(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA), tempfile.TemporaryDirectory() as tmp:
        weights = {}
        for steady in (False, True):
            torch.manual_seed(21)
            tr = engine.Trainer(cfg(tmp, CRIT[:2], steady), train_loader=loader())
            tr.train()
            weights[steady] = torch.cat([p.detach().float().flatten() for p in tr.model.parameters()])
            if steady:
                assert tr._steady is not None and tr._steady.refused is None, tr._steady.refused
                assert tr._steady.replays == 4 and tr._steady.eager_steps == 2          # warm-up, checked step, then the graph
        rel = float((weights[True] - weights[False]).norm() / weights[False].norm())
        assert rel < 2e-2, rel
        # contrastive loss with its gate open (epoch_progress is 0 in the only epoch: a schedule that is always on)
        crit = CRIT[:2] + [dict(type="AggregatedContrastiveLoss", temperature=0.2, reduction="mean", loss_weight=0.02, schedule="all")]
        torch.manual_seed(21)
        tr = engine.Trainer(cfg(tmp, crit, True), train_loader=loader(4))
        tr.train()
        # (the segmented-sum form of the contrastive loss reads nothing on the host either: the whole criteria stack is replayed)
        assert tr._steady.refused is None and tr._steady.replays == 2, (tr._steady.refused, tr._steady.replays)
        assert all(torch.isfinite(p).all() for p in tr.model.parameters())
