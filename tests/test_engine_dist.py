"""CPU: trainer/hook protocol, optimizer param groups, checkpoint round trip, and the N>1 data-parallel
path rehearsed with world_size-2 gloo processes (the GPU path differs only in backend = RCCL)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from scenesplat_amd.pointcept_api import MODELS, engine
from scenesplat_amd.pointcept_api.engine import HookBase


class _Toy(nn.Module):
    """Stand-in model with the LangPretrainer output contract ({'loss': 0-d}); 'block' in a name."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Linear(4, 8)
        self.block0 = nn.Linear(8, 1)

    def forward(self, d):
        return dict(loss=self.block0(torch.tanh(self.stem(d["feat"]))).pow(2).mean())


if "ToyLang" not in MODELS.module_dict:
    MODELS.register_module("ToyLang", module=_Toy)


def _cfg(tmp):
    return dict(model=dict(type="ToyLang"), device="cpu", eval_epoch=2, save_path=tmp, enable_amp=False, clip_grad=1.0,
                optimizer=dict(type="AdamW", lr=6e-3, weight_decay=0.05),
                param_dicts=[dict(keyword="block", lr=6e-4)],
                scheduler=dict(type="OneCycleLR", max_lr=[6e-3, 6e-4], pct_start=0.05, anneal_strategy="cos",
                               div_factor=10.0, final_div_factor=1000.0),
                hooks=[dict(type="CheckpointLoader"), dict(type="IterationTimer", warmup_iter=1),
                       dict(type="InformationWriter", interval=1), dict(type="CheckpointSaver", save_freq=None)])


def _loader(n=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [dict(feat=torch.randn(16, 4, generator=g)) for _ in range(n)]


def test_trainer_hook_protocol_and_checkpoint_roundtrip():
    calls = []

    class Spy(HookBase):
        def before_train(self): calls.append("before_train")
        def before_epoch(self): calls.append("before_epoch")
        def before_step(self): calls.append("before_step")
        def after_step(self): calls.append("after_step"); assert "model_output_dict" in self.trainer.comm_info
        def after_epoch(self): calls.append("after_epoch"); self.trainer.comm_info["current_metric_value"] = float(self.trainer.epoch)
        def after_train(self): calls.append("after_train")

    with tempfile.TemporaryDirectory() as tmp:
        cfg = _cfg(tmp)
        cfg["hooks"] = [Spy()] + cfg["hooks"]
        tr = engine.TRAINERS.build(dict(type="DefaultTrainer", cfg=cfg, train_loader=_loader()))
        # param groups: names containing "block" get the lower lr (utils/optimizer.py:13-48)
        assert len(tr.optimizer.param_groups) == 2 and len(tr.optimizer.param_groups[1]["params"]) == 2
        tr.train()
        assert calls[0] == "before_train" and calls[-1] == "after_train"
        assert calls.count("before_step") == calls.count("after_step") == 10 and calls.count("after_epoch") == 2
        last = os.path.join(tmp, "model", "model_last.pth")
        assert os.path.isfile(last) and os.path.isfile(os.path.join(tmp, "model", "model_best.pth"))
        ck = torch.load(last, weights_only=True)
        assert set(ck) == {"epoch", "state_dict", "optimizer", "scheduler", "scaler", "best_metric_value"} and ck["epoch"] == 2
        # resume: weights with a DDP 'module.' prefix are stripped; epoch restored
        ck["state_dict"] = {"module." + k: v for k, v in ck["state_dict"].items()}
        torch.save(ck, last)
        cfg2 = _cfg(tmp); cfg2.update(weight=last, resume=True, eval_epoch=3)
        tr2 = engine.Trainer(cfg2, train_loader=_loader())
        tr2.before_train()
        assert tr2.start_epoch == 2
        for k, v in tr.model.state_dict().items():
            assert torch.equal(tr2.model.state_dict()[k], v)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    engine.init_distributed("gloo")
    assert engine.get_world_size() == world and engine.get_rank() == rank
    torch.manual_seed(0)                      # same init on every rank, as DDP broadcast would enforce
    cfg = _cfg(os.path.join(tmp, f"r{rank}"))
    cfg["hooks"] = []
    tr = engine.Trainer(cfg, train_loader=_loader(3, seed=100 + rank))   # rank-specific shard of the data
    assert isinstance(tr.model, nn.parallel.DistributedDataParallel)
    tr.train()
    sd = {k: v.numpy().copy() for k, v in tr.model.module.state_dict().items()}   # by value: the worker exits before the parent reads
    red = engine.reduce_dict({"x": torch.tensor(float(rank + 1))})
    # the throughput reduction bench.py uses: max over ranks
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, sd, float(red["x"]), float(t)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_ddp_world_size_2_gloo_matches_single_process_average():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_worker, args=(r, world, port, tmp, q)) for r in range(world)]
        [p.start() for p in procs]
        res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda r: r[0])
        [p.join(30) for p in procs]
        assert all(p.exitcode == 0 for p in procs)
        # ranks end with identical weights (gradients were averaged every step)
        for k in res[0][1]:
            assert torch.allclose(torch.as_tensor(res[0][1][k]), torch.as_tensor(res[1][1][k]), atol=1e-7), k
        assert res[0][2] == res[1][2] == 1.5 and res[0][3] == res[1][3] == 0.2
        # and equal to one process averaging the two shards' gradients by hand
        torch.manual_seed(0)
        cfg = _cfg(os.path.join(tmp, "single")); cfg["hooks"] = []
        tr = engine.Trainer(cfg, train_loader=_loader(3))
        la, lb = _loader(3, seed=100), _loader(3, seed=101)
        tr.model.train()
        for _ in range(tr.max_epoch):
            for a, b in zip(la, lb):
                tr.optimizer.zero_grad(set_to_none=True)
                ((tr.model(a)["loss"] + tr.model(b)["loss"]) / 2).backward()
                torch.nn.utils.clip_grad_norm_(tr.model.parameters(), 1.0)
                tr.optimizer.step(); tr.scheduler.step()
        for k, v in tr.model.state_dict().items():
            assert torch.allclose(torch.as_tensor(res[0][1][k]), v, atol=1e-6), k


def test_steady_state_step_disabled_is_the_plain_callable():
    """SteadyStateStep with enabled=False (what bench.py uses under DDP): every call runs fn eagerly and hands back detached
    outputs; nothing touches the GPU."""
    from scenesplat_amd.steady_state import SteadyStateStep
    w = torch.nn.Parameter(torch.ones(3))
    calls = []

    def fn(plan, t):
        calls.append(plan)
        y = (w * t["x"]).sum()
        y.backward()
        return {"y": y}
    step = SteadyStateStep(fn, [w], enabled=False)
    out = step("plan-a", {"x": torch.arange(3.0)})
    assert calls == ["plan-a"] and step.eager_steps == 1 and step.replays == 0
    assert out["y"].grad_fn is None and float(out["y"]) == 3.0
    assert torch.equal(w.grad, torch.arange(3.0))


# ---- stage-wise gradient exchange (scenesplat_amd/grad_exchange.py) -------------------------------------------------------------
class _Staged(nn.Module):
    """Parameter names in PT-v3m1's stage naming, one parameter that takes no part in the loss, one shared by two paths."""

    def __init__(self):
        super().__init__()
        self.embedding = nn.Linear(4, 8)
        self.enc = nn.ModuleDict(dict(enc0=nn.Linear(8, 8), enc1=nn.ModuleDict(dict(down=nn.Linear(8, 8), block0=nn.Linear(8, 8)))))
        self.dec = nn.ModuleDict(dict(dec0=nn.Linear(8, 1)))
        self.unused = nn.Linear(3, 3)

    def forward(self, d):
        h = torch.tanh(self.embedding(d["feat"]))
        h = torch.tanh(self.enc["enc0"](h)) + h
        if d.get("skip_enc1"):                 # a data-dependent branch: enc1 receives no gradient this step (on this rank)
            h = self.enc["enc0"](h)
        else:
            h = self.enc["enc1"]["block0"](torch.tanh(self.enc["enc1"]["down"](h))) + self.enc["enc0"](h)     # enc0 used twice
        cut = d.get("backward_cut")
        if cut is not None:                    # the backward cut of PointTransformerV3.forward, in miniature
            leaf = h.detach().requires_grad_(True)
            cut.append((h, leaf)); h = leaf
        return dict(loss=self.dec["dec0"](h).pow(2).mean())


def _stage_worker(rank, world, port, q, hooks=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    engine.init_distributed("gloo")
    from scenesplat_amd.grad_exchange import StageGradExchange, default_stage_of
    torch.manual_seed(0)
    model = _Staged()
    ex = StageGradExchange(model, hooks=(hooks is True))
    assert set(ex.stages) == {"embedding", "enc.enc0", "enc.enc1", "dec.dec0", "other"}
    assert default_stage_of("module.backbone.dec.dec0.block1.mlp.0.fc1.weight") == "dec.dec0"
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    from scenesplat_amd.pointcept_api.ptv3 import backward_tail
    assert [st["label"] for st in ex._sequence] == ["other", "dec.dec0", "enc.enc1", "enc.enc0", "embedding"]     # fixed launch order
    # the early stage sits at the front of the allocation: two contiguous slices
    (ent,) = ex.whole.values()
    assert ex.stages["dec.dec0"]["flat"].data_ptr() == ent[1].data_ptr() and 0 < ent[2] < ent[1].numel()
    for it, d in enumerate(_loader(3, seed=200 + rank)):
        if it == 1:
            opt.zero_grad(set_to_none=False)       # gradients stay the stage-buffer views: accumulated in place, no pack copy
        else:
            opt.zero_grad(set_to_none=True)
        if it == 2 and rank == 1:
            d = dict(d, skip_enc1=True)            # enc1 takes no part on THIS rank only: the collective sequence must not change
        if hooks == "split":
            cut = []
            model(dict(d, backward_cut=cut))["loss"].backward()       # down to the cut: dec0's gradients are final
            ex.pack("early"); ex.reduce_begin("early")
            backward_tail(cut)
            ex.pack("late"); ex.reduce_begin("late"); ex.reduce_end()
        else:
            model(d)["loss"].backward()
            if hooks:
                ex.finish()
            else:
                ex.pack(); ex.reduce()             # the packed form of graph-replayed steps: pack inside the step, all-reduce after it
        assert model.unused.weight.grad is not None and float(model.unused.weight.grad.abs().sum()) == 0.0
        for st in ex.stages.values():
            for p, v in zip(st["params"], st["views"]):
                assert p.grad.data_ptr() == v.data_ptr()
        opt.step()
    q.put((rank, {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("hooks", [True, False, "split"])
def test_stage_grad_exchange_world_size_2_gloo_matches_hand_average(hooks):
    """One all-reduce per model stage -- launched from post-accumulate hooks during the backward in a FIXED order, or (hooks=False)
    packed at the end of the step and reduced after it, or ("split", round 4) the two-call backward with the early slice (dec0)
    all-reduced while the rest of the backward runs: both ranks end with the weights of a single process that averages the two
    shards' gradients by hand (what DDP computes, engines/defaults.py:13-34) -- including a step in which one stage receives no
    gradient on ONE rank only."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stage_worker, args=(r, world, port, q, hooks)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda r: r[0])
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for k in res[0][1]:
        assert torch.allclose(torch.as_tensor(res[0][1][k]), torch.as_tensor(res[1][1][k]), atol=1e-7), k
    torch.manual_seed(0)
    model = _Staged()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for it, (a, b) in enumerate(zip(_loader(3, seed=200), _loader(3, seed=201))):
        opt.zero_grad(set_to_none=True)
        ((model(a)["loss"] + model(dict(b, skip_enc1=(it == 2)))["loss"]) / 2).backward()
        opt.step()
    for k, v in model.state_dict().items():
        assert torch.allclose(torch.as_tensor(res[0][1][k]), v, atol=1e-6), k


def _stage_trainer_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    engine.init_distributed("gloo")
    torch.manual_seed(rank)                   # DIFFERENT init per rank: the trainer must broadcast rank 0's weights
    cfg = _cfg(os.path.join(tmp, f"r{rank}"))
    cfg["hooks"] = []; cfg["grad_exchange"] = "stage"
    tr = engine.Trainer(cfg, train_loader=_loader(3, seed=100 + rank))
    assert not isinstance(tr.model, nn.parallel.DistributedDataParallel) and tr.grad_exchange is not None
    tr.train()
    q.put((rank, {k: v.numpy().copy() for k, v in tr.model.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_trainer_with_stage_grad_exchange_equals_the_ddp_trainer():
    """cfg["grad_exchange"] = "stage": same weights as the DDP trainer of test_ddp_world_size_2_gloo_... (hand average)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_stage_trainer_worker, args=(r, world, port, tmp, q)) for r in range(world)]
        [p.start() for p in procs]
        res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda r: r[0])
        [p.join(30) for p in procs]
        assert all(p.exitcode == 0 for p in procs)
        for k in res[0][1]:
            assert torch.allclose(torch.as_tensor(res[0][1][k]), torch.as_tensor(res[1][1][k]), atol=1e-7), k
        torch.manual_seed(0)
        cfg = _cfg(os.path.join(tmp, "single")); cfg["hooks"] = []
        tr = engine.Trainer(cfg, train_loader=_loader(3))
        la, lb = _loader(3, seed=100), _loader(3, seed=101)
        tr.model.train()
        for _ in range(tr.max_epoch):
            for a, b in zip(la, lb):
                tr.optimizer.zero_grad(set_to_none=True)
                ((tr.model(a)["loss"] + tr.model(b)["loss"]) / 2).backward()
                torch.nn.utils.clip_grad_norm_(tr.model.parameters(), 1.0)
                tr.optimizer.step(); tr.scheduler.step()
        for k, v in tr.model.state_dict().items():
            assert torch.allclose(torch.as_tensor(res[0][1][k]), v, atol=1e-6), k
