"""GPU: the SPLIT data-parallel step of round 4 -- PointTransformerV3's backward cut, the two-graph SteadyStateStep and the
early / late slices of StageGradExchange (reference behaviour: DDP's bucketed all-reduce overlapped with backward,
pointcept/engines/defaults.py:13-34, engines/train.py:208,223).

  * the two-call backward equals the one-call backward on the real HIP model, eagerly and replayed as two hipGraphs;
  * two ranks sharing cuda:0 over gloo, the step replayed as two graphs with the dec0 slice all-reduced between them, end with
    identical weights that equal the hook-form exchange's;
  * one rank over RCCL ("nccl"): the asynchronous ReduceOp.AVG all-reduces on the process group's own stream beside the second
    graph leave every gradient where an unsplit backward puts it."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TINY = dict(in_channels=11, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2),
            enc_depths=(1, 1, 1), enc_channels=(16, 32, 48), enc_num_head=(1, 2, 3), enc_patch_size=(64, 64, 16),
            dec_depths=(1, 1), dec_channels=(48, 32), dec_num_head=(1, 2), dec_patch_size=(64, 64))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _split_fns(model, d, exchange=None, counter=None):
    """(fn, tail, between) of a split step on `model`: what bench.py builds for N > 1."""
    from scenesplat_amd.pointcept_api.ptv3 import backward_tail
    box = {}

    def fn(plan, t):
        cut = []
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(dict(feat=t["feat"], grid_coord=d["grid_coord"], offset=d["offset"], plan=plan, backward_cut=cut))
        torch.autograd.backward(out.feat, grad_tensors=t["cot"])
        assert len(cut) == 2                                  # x from dec1 and the level-0 skip
        if exchange is not None:
            exchange.pack("early")
        box["cut"] = cut
        return {"feat": out.feat}

    def tail():
        backward_tail(box.pop("cut"))
        if exchange is not None:
            exchange.pack("late")

    def between():
        if counter is not None:
            counter.append(1)
        if exchange is not None:
            exchange.reduce_begin("early")
    return fn, tail, between


def _setup(drop_path=0.0, seed=11):
    from scenesplat_amd.pointcept_api import MODELS
    from scenesplat_amd.synthetic import room_chunk
    torch.manual_seed(seed)
    model = MODELS.build(dict(type="PT-v3m1", **TINY, drop_path=drop_path, shuffle_orders=True)).cuda().train()
    d = {k: v.cuda() for k, v in room_chunk(n_side=40, seed=3, lang_dim=0).items()}
    return model, d, d["feat"].shape[0]


def _plain_step(model, d, feat, cot, perms):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(dict(feat=feat, grid_coord=d["grid_coord"], offset=d["offset"], plan=model.prepare_plan(d, perms=perms)))
    torch.autograd.backward(out.feat, grad_tensors=cot)
    return out.feat.detach().float().clone()


def _grad_err(model, ref):
    num = sum(float((p.grad - ref[k]).norm()) ** 2 for k, p in model.named_parameters()) ** 0.5
    den = sum(float(ref[k].norm()) ** 2 for k in ref) ** 0.5
    return num / den


def test_backward_cut_two_calls_equal_one_call_eager_and_as_two_graphs():
    from scenesplat_amd import native as nv
    from scenesplat_amd.pointcept_api import RUNTIME
    from scenesplat_amd.steady_state import SteadyStateStep
    old = dict(RUNTIME)
    RUNTIME.update(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA)
    try:
        model, d, n = _setup()
        calls = []
        fn, tail, between = _split_fns(model, d, counter=calls)
        steady = SteadyStateStep(fn, list(model.parameters()), warmup=1, tail=tail, between=between)
        g = torch.Generator(device="cuda").manual_seed(1)
        for it in range(5):
            feat = torch.randn(n, 11, device="cuda", generator=g)
            cot = torch.randn(n, TINY["dec_channels"][0], device="cuda", generator=g).to(torch.bfloat16)
            perms = model.draw_perms()
            # reference: the ordinary one-call backward, no cut (nothing of its autograd graph may outlive this block: a captured
            # step must not meet AccumulateGrad nodes of another stream, steady_state.py)
            model.zero_grad(set_to_none=True)
            ref_y = _plain_step(model, d, feat, cot, perms)
            ref = {k: p.grad.clone() for k, p in model.named_parameters()}
            assert all(p.grad is not None for p in model.parameters())
            model.zero_grad(set_to_none=True)
            y = steady(model.prepare_plan(d, perms=perms), dict(feat=feat, cot=cot))["feat"].float()
            assert steady.refused is None, steady.refused
            assert all(p.grad is not None for p in model.parameters())
            assert (y - ref_y).norm() <= 1e-2 * ref_y.norm()
            err = _grad_err(model, ref)
            assert err <= 2e-2, (it, err)                       # two eager runs of this bf16 step differ by ~3e-3 (fp32 atomics)
        assert steady.eager_steps == 2 and steady.replays == 3   # warm-up, sync-checked step, then two graphs per step
        assert len(calls) == 5                                    # between() ran once per step, eager or replayed, never inside a capture
    finally:
        RUNTIME.clear(); RUNTIME.update(old)


def _guard(fn):
    """A worker that dies silently leaves the parent waiting for its queue entry until the test's timeout: report instead."""
    import functools
    import traceback

    @functools.wraps(fn)
    def run(*a):
        try:
            fn(*a)
        except BaseException:  # noqa: BLE001
            q = [x for x in a if hasattr(x, "put")][0]
            q.put(("ERROR", traceback.format_exc()))
            raise
    return run


def _split_worker(rank, world, port, q, mode):
    _guard(_split_worker_body)(rank, world, port, q, mode)


def _split_worker_body(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from scenesplat_amd import native as nv
    from scenesplat_amd.grad_exchange import StageGradExchange
    from scenesplat_amd.pointcept_api import RUNTIME
    from scenesplat_amd.steady_state import SteadyStateStep
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    RUNTIME.update(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA)
    model, d, n = _setup(seed=0)                                # same weights on both ranks
    ex = StageGradExchange(model, hooks=(mode == "hooks"))
    (ent,) = ex.whole.values()
    assert ex.stages["dec.dec0"]["flat"].data_ptr() == ent[1].data_ptr() and 0 < ent[2] < ent[1].numel()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)       # (update size independent of the scale of the random cotangent)
    fn, tail, between = _split_fns(model, d, exchange=None if mode == "hooks" else ex)
    steady = SteadyStateStep(fn, list(model.parameters()), warmup=1, tail=tail, between=between, enabled=(mode == "graph"))
    g = torch.Generator(device="cuda").manual_seed(100 + rank)     # different data per rank
    torch.manual_seed(5)                                        # the same curve permutations on both ranks and in every mode
    for it in range(5):
        feat = torch.randn(n, 11, device="cuda", generator=g)
        cot = torch.randn(n, TINY["dec_channels"][0], device="cuda", generator=g).to(torch.bfloat16)
        plan = model.prepare_plan(d, perms=model.draw_perms())
        opt.zero_grad(set_to_none=True)
        if mode == "hooks":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(dict(feat=feat, grid_coord=d["grid_coord"], offset=d["offset"], plan=plan))
            torch.autograd.backward(out.feat, grad_tensors=cot)
            del out
            ex.finish()
        else:
            steady(plan, dict(feat=feat, cot=cot))
            bad = [k for k, p in model.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
            assert not bad, ("non-finite gradient after the step", mode, it, steady.replays, bad[:4])
            ex.reduce_begin("late"); ex.reduce_end()
            bad = [k for k, p in model.named_parameters() if not bool(torch.isfinite(p.grad).all())]
            assert not bad, ("non-finite gradient after the exchange", mode, it, steady.replays, bad[:4])
        for st in ex.stages.values():
            for p, v in zip(st["params"], st["views"]):
                assert p.grad.data_ptr() == v.data_ptr()
        opt.step()
    torch.cuda.synchronize()
    if mode == "graph":
        assert steady.refused is None and steady.replays == 3, (steady.refused, steady.replays)
    sd = {k: v.detach().float().cpu().numpy().copy() for k, v in model.state_dict().items() if v.is_floating_point() and "running_" not in k}
    q.put((rank, sd))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_split_exchange_two_ranks_replayed_as_two_graphs_equal_the_hook_form():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = {}
    for mode in ("hooks", "eager", "graph"):
        world, port = 2, _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=_split_worker, args=(r, world, port, q, mode)) for r in range(world)]
        [p.start() for p in procs]
        res = [q.get(timeout=600) for _ in range(world)]
        assert not any(r[0] == "ERROR" for r in res), [r[1] for r in res if r[0] == "ERROR"][0]
        res = sorted(res, key=lambda r: r[0])
        [p.join(60) for p in procs]
        assert all(p.exitcode == 0 for p in procs), mode
        for k in res[0][1]:
            assert np.array_equal(res[0][1][k], res[1][1][k]), (mode, k)       # the ranks agree bit for bit
        out[mode] = res[0][1]
    for mode in ("graph", "eager"):
        num = sum(float(((out[mode][k] - out["hooks"][k]) ** 2).sum()) for k in out["hooks"]) ** 0.5
        den = sum(float((out["hooks"][k] ** 2).sum()) for k in out["hooks"]) ** 0.5
        print("split exchange (%s) vs hook form after 5 AdamW steps: relative weight difference %.2e" % (mode, num / den))
        assert num <= 2e-3 * den, (mode, num / den)


def _rccl_split_worker(port, q):
    _guard(_rccl_split_worker_body)(port, q)


def _rccl_split_worker_body(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    from scenesplat_amd import native as nv
    from scenesplat_amd.grad_exchange import StageGradExchange
    from scenesplat_amd.pointcept_api import RUNTIME
    from scenesplat_amd.steady_state import SteadyStateStep
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    RUNTIME.update(conv_dtype=torch.bfloat16, attn_impl=nv.ATTN_MFMA)
    model, d, n = _setup(seed=0)
    ex = StageGradExchange(model, force=True, hooks=False)
    fn, tail, between = _split_fns(model, d, exchange=ex)
    steady = SteadyStateStep(fn, list(model.parameters()), warmup=1, tail=tail, between=between)
    g = torch.Generator(device="cuda").manual_seed(3)
    errs = []
    for it in range(5):
        feat = torch.randn(n, 11, device="cuda", generator=g)
        cot = torch.randn(n, TINY["dec_channels"][0], device="cuda", generator=g).to(torch.bfloat16)
        perms = model.draw_perms()
        model.zero_grad(set_to_none=True)
        _plain_step(model, d, feat, cot, perms)
        ref = {k: p.grad.clone() for k, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
        steady(model.prepare_plan(d, perms=perms), dict(feat=feat, cot=cot))
        ex.reduce_begin("late"); ex.reduce_end()
        torch.cuda.synchronize()
        errs.append(_grad_err(model, ref))
    q.put((dist.get_backend(), errs, steady.replays, steady.refused, bool(ex._avg_ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_split_exchange_over_rccl_single_rank_leaves_the_gradients_of_an_unsplit_backward():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_split_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=500)
    assert got[0] != "ERROR", got[1]
    backend, errs, replays, refused, avg_ok = got
    p.join(60)
    assert p.exitcode == 0 and backend == "nccl" and refused is None and replays == 3 and avg_ok
    print("split exchange over one-rank RCCL: gradient error vs the unsplit backward per step: " + " ".join("%.1e" % e for e in errs))
    assert max(errs) <= 2e-2
