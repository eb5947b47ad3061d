"""Trainer / hook / distributed boundary of the hot path (reference: pointcept/engines/train.py:35-343,
engines/defaults.py:13-34, engines/hooks/default.py:1-27, engines/hooks/misc.py:33-300,
utils/optimizer.py:13-48, utils/scheduler.py:100-134, utils/comm.py).

Same callback protocol and trainer attributes the reference hooks rely on (cfg, model, optimizer,
scheduler, scaler, train_loader, epoch/start_epoch/max_epoch, best_metric_value, comm_info{iter,
input_dict, model_output_dict, iter_info, current_metric_value, ...}).  Differences, all on the
MI355X side of the boundary: AMP is bf16 autocast (no GradScaler: `scaler` is None and checkpoints
carry `scaler: None`); the process group is RCCL ("nccl" under ROCm) for GPU runs and gloo for CPU
rehearsal; one process per GPU launched by torchrun (RANK/LOCAL_RANK/WORLD_SIZE) instead of
mp.spawn; InformationWriter syncs the loss every `interval` steps, not every step.
"""
import os
import time
import weakref
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn

from .registry import HOOKS, MODELS, TRAINERS, Registry

OPTIMIZERS = Registry("optimizers")
OPTIMIZERS.register_module(module=torch.optim.SGD, name="SGD")
OPTIMIZERS.register_module(module=torch.optim.Adam, name="Adam")
OPTIMIZERS.register_module(module=torch.optim.AdamW, name="AdamW")
SCHEDULERS = Registry("schedulers")


# ---- comm (utils/comm.py) ----------------------------------------------------------------------------
def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def init_distributed(backend=None):
    """One process per GPU under torchrun.  backend None: "nccl" (= RCCL over xGMI on ROCm) when a GPU
    is present, else gloo."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or (dist.is_available() and dist.is_initialized()):
        return get_rank(), get_world_size()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
    return get_rank(), get_world_size()


def reduce_dict(input_dict, average=True):
    """utils/comm.py:171-198"""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().float() for k in names])
        dist.all_reduce(values)
        if average:
            values /= world
        return {k: v for k, v in zip(names, values)}


def create_ddp_model(model, *, fp16_compression=False, **kwargs):
    """engines/defaults.py:13-34: plain DDP with the caller's kwargs; identity at world size 1.
    bucket_cap_mb defaults to 100 MB on GPU: the gradient profile is decoder-heavy (dec0 = 52 % of the
    366.8 MB) and xGMI rings are per-link bound, so fewer, larger buckets."""
    if get_world_size() == 1:
        return model
    if next(model.parameters()).is_cuda:
        kwargs.setdefault("device_ids", [torch.cuda.current_device()])
        kwargs.setdefault("bucket_cap_mb", 100)
        kwargs.setdefault("gradient_as_bucket_view", True)
    ddp = nn.parallel.DistributedDataParallel(model, **kwargs)
    if fp16_compression:
        from torch.distributed.algorithms.ddp_comm_hooks import default as comm_hooks
        ddp.register_comm_hook(state=None, hook=comm_hooks.bf16_compress_hook)
    return ddp


# ---- optimizer / scheduler (utils/optimizer.py:13-48, utils/scheduler.py:100-134) -----------------------
def build_optimizer(cfg, model, param_dicts=None):
    cfg = dict(cfg)
    if param_dicts is None:
        cfg["params"] = model.parameters()
    else:
        groups = [dict(params=[], lr=cfg["lr"])]
        for pd in param_dicts:
            g = dict(params=[])
            for k in ("lr", "momentum", "weight_decay"):
                if k in pd:
                    g[k] = pd[k]
            groups.append(g)
        for n, p in model.named_parameters():
            for i, pd in enumerate(param_dicts):
                if pd["keyword"] in n:
                    groups[i + 1]["params"].append(p)
                    break
            else:
                groups[0]["params"].append(p)
        cfg["params"] = groups
    return OPTIMIZERS.build(cfg=cfg)


@SCHEDULERS.register_module()
class OneCycleLR(torch.optim.lr_scheduler.OneCycleLR):
    def __init__(self, optimizer, max_lr, total_steps=None, pct_start=0.3, anneal_strategy="cos", cycle_momentum=True,
                 base_momentum=0.85, max_momentum=0.95, div_factor=25.0, final_div_factor=1e4, three_phase=False,
                 last_epoch=-1):
        super().__init__(optimizer=optimizer, max_lr=max_lr, total_steps=total_steps, pct_start=pct_start,
                         anneal_strategy=anneal_strategy, cycle_momentum=cycle_momentum, base_momentum=base_momentum,
                         max_momentum=max_momentum, div_factor=div_factor, final_div_factor=final_div_factor,
                         three_phase=three_phase, last_epoch=last_epoch)


def build_scheduler(cfg, optimizer):
    cfg = dict(cfg)
    cfg["optimizer"] = optimizer
    return SCHEDULERS.build(cfg=cfg)


# ---- hooks (engines/hooks/default.py, misc.py) -----------------------------------------------------------
class HookBase:
    trainer = None

    def before_train(self): pass
    def before_epoch(self): pass
    def before_step(self): pass
    def after_step(self): pass
    def after_epoch(self): pass
    def after_train(self): pass
    def before_eval(self): pass


@HOOKS.register_module()
class IterationTimer(HookBase):
    """hooks/misc.py:33-75: data_time / batch_time into comm_info['iter_info']."""

    def __init__(self, warmup_iter=1):
        self._warmup_iter, self._t0, self._iter = warmup_iter, time.perf_counter(), 0

    def before_epoch(self):
        self._t0 = time.perf_counter()

    def before_step(self):
        self.trainer.comm_info["data_time"] = time.perf_counter() - self._t0

    def after_step(self):
        now = time.perf_counter()
        self.trainer.comm_info["batch_time"] = now - self._t0
        self._t0 = now
        self._iter += 1
        ci = self.trainer.comm_info
        ci["iter_info"] = "Train: [{}/{}][{}/{}] Data {:.3f} Batch {:.3f} ".format(
            self.trainer.epoch + 1, self.trainer.max_epoch, ci["iter"] + 1, ci.get("iter_per_epoch", 0),
            ci["data_time"], ci["batch_time"])


@HOOKS.register_module()
class InformationWriter(HookBase):
    """hooks/misc.py:79-143.  The reference calls .item() on every output every step (a device sync);
    here the sync happens every `interval` steps."""

    def __init__(self, interval=10):
        self.interval, self.history = interval, []

    def after_step(self):
        ci = self.trainer.comm_info
        if (ci["iter"] + 1) % self.interval:
            return
        out = ci.get("model_output_dict", {})
        vals = {k: float(v.detach().float().item()) for k, v in out.items() if torch.is_tensor(v) and v.dim() == 0}
        lr = self.trainer.optimizer.state_dict()["param_groups"][0]["lr"] if self.trainer.optimizer else 0.0
        self.history.append(dict(iter=ci["iter"], lr=lr, **vals))
        if is_main_process() and getattr(self.trainer, "logger", None):
            self.trainer.logger(ci.get("iter_info", "") + " ".join(f"{k}: {v:.4f}" for k, v in vals.items()) + f" Lr: {lr:.5f}")


@HOOKS.register_module()
class CheckpointSaver(HookBase):
    """hooks/misc.py:147-204: rank-0 atomic save of {epoch, state_dict, optimizer, scheduler, scaler,
    best_metric_value} to model/model_last.pth (+ model_best.pth when the metric improved)."""

    def __init__(self, save_freq=None):
        self.save_freq = save_freq

    def after_epoch(self):
        tr = self.trainer
        if is_main_process():
            is_best = False
            cur = tr.comm_info.get("current_metric_value")
            if cur is not None and cur > tr.best_metric_value:
                tr.best_metric_value, is_best = cur, True
            d = os.path.join(tr.cfg["save_path"], "model")
            os.makedirs(d, exist_ok=True)
            fn = os.path.join(d, "model_last.pth")
            torch.save(dict(epoch=tr.epoch + 1, state_dict=tr.model.state_dict(),
                            optimizer=tr.optimizer.state_dict() if tr.optimizer else None,
                            scheduler=tr.scheduler.state_dict() if tr.scheduler else None,
                            scaler=None, best_metric_value=tr.best_metric_value), fn + ".tmp")
            os.replace(fn + ".tmp", fn)
            if is_best:
                import shutil
                shutil.copyfile(fn, os.path.join(d, "model_best.pth"))
            if self.save_freq and (tr.epoch + 1) % self.save_freq == 0:
                import shutil
                shutil.copyfile(fn, os.path.join(d, f"epoch_{tr.epoch + 1}.pth"))
        synchronize()


@HOOKS.register_module()
class CheckpointLoader(HookBase):
    """hooks/misc.py:208-300: add/strip the DDP `module.` prefix, optional keyword replacement, drop keys
    missing from the model or with a shape mismatch, strict=False load; restore trainer state on resume."""

    def __init__(self, keywords="", replacement=None, strict=False):
        self.keywords, self.replacement, self.strict = keywords, replacement if replacement is not None else keywords, strict

    def before_train(self):
        tr = self.trainer
        weight = tr.cfg.get("weight")
        if not weight or not os.path.isfile(weight):
            return
        # released / third-party checkpoints come through here: never unpickle arbitrary objects (the engine's own
        # checkpoint -- state_dict + AdamW + OneCycleLR state + scalars -- loads under weights_only=True)
        ckpt = torch.load(weight, map_location="cpu", weights_only=True)
        model_sd = tr.model.state_dict()
        wrapped = isinstance(tr.model, nn.parallel.DistributedDataParallel)
        new = OrderedDict()
        for k, v in ckpt["state_dict"].items():
            if k.startswith("module.") and not wrapped:
                k = k[7:]
            elif not k.startswith("module.") and wrapped:
                k = "module." + k
            if self.keywords in k:
                k = k.replace(self.keywords, self.replacement, 1)
            if k in model_sd and model_sd[k].shape == v.shape:
                new[k] = v
        tr.comm_info["checkpoint_load"] = tr.model.load_state_dict(new, strict=self.strict)
        if tr.cfg.get("resume"):
            tr.start_epoch = ckpt["epoch"]
            tr.best_metric_value = ckpt["best_metric_value"]
            if tr.optimizer and ckpt.get("optimizer"):
                tr.optimizer.load_state_dict(ckpt["optimizer"])
            if tr.scheduler and ckpt.get("scheduler"):
                tr.scheduler.load_state_dict(ckpt["scheduler"])


def build_hooks(cfg):
    return [HOOKS.build(c) if isinstance(c, dict) else c for c in cfg]


# ---- trainers (engines/train.py:35-343) ----------------------------------------------------------------------
class TrainerBase:
    def __init__(self):
        self.hooks, self.epoch, self.start_epoch, self.max_epoch, self.max_iter = [], 0, 0, 0, 0
        self.comm_info = dict()
        self.data_iterator = enumerate([])

    def register_hooks(self, hooks):
        hooks = build_hooks(hooks)
        for h in hooks:
            assert isinstance(h, HookBase)
            h.trainer = weakref.proxy(self)
        self.hooks.extend(hooks)

    def train(self):
        self.before_train()
        for self.epoch in range(self.start_epoch, self.max_epoch):
            self.before_epoch()
            for self.comm_info["iter"], self.comm_info["input_dict"] in self.data_iterator:
                self.before_step(); self.run_step(); self.after_step()
            self.after_epoch()
        self.after_train()

    def before_eval(self): [h.before_eval() for h in self.hooks]
    def before_train(self): [h.before_train() for h in self.hooks]
    def before_epoch(self): [h.before_epoch() for h in self.hooks]
    def before_step(self): [h.before_step() for h in self.hooks]
    def after_step(self): [h.after_step() for h in self.hooks]
    def after_epoch(self): [h.after_epoch() for h in self.hooks]
    def after_train(self): [h.after_train() for h in self.hooks]

    def run_step(self):
        raise NotImplementedError


@TRAINERS.register_module("DefaultTrainer")
class Trainer(TrainerBase):
    """cfg: mapping with model / optimizer / scheduler / param_dicts / hooks / enable_amp / clip_grad /
    eval_epoch / save_path / find_unused_parameters [/ weight / resume / device].  The train loader is any
    sized iterable of input dicts (the reference builds a DataLoader from cfg.data; data loading is outside
    the hot path, so it is injected: `train_loader=`)."""

    def __init__(self, cfg, train_loader=None, logger=None):
        super().__init__()
        self.cfg = cfg
        self.max_epoch = cfg.get("eval_epoch", 1)
        self.best_metric_value = -float("inf")
        self.logger = logger
        self.device = torch.device(cfg.get("device", "cuda"))
        self.grad_exchange = None
        self.model = self.build_model()
        self.train_loader, self.val_loader = train_loader, None
        self.optimizer = build_optimizer(cfg["optimizer"], self.model, cfg.get("param_dicts"))
        sched = dict(cfg["scheduler"])
        sched.setdefault("total_steps", max(1, len(train_loader)) * self.max_epoch)
        self.scheduler = build_scheduler(sched, self.optimizer)
        self.scaler = None   # bf16 autocast needs no loss scaling
        self._steady, self._steady_host = None, {}
        self.comm_info["iter_per_epoch"] = len(train_loader)
        self.register_hooks(cfg.get("hooks", []))

    def build_model(self):
        model = MODELS.build(self.cfg["model"])
        if self.cfg.get("sync_bn"):
            model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
        model = model.to(self.device)
        if self.cfg.get("grad_exchange") == "stage" and get_world_size() > 1:
            # round 3: the gradient average of DDP (engines/defaults.py:13-34) as one all-reduce per model STAGE, launched when the
            # stage's gradients become final in the backward pass (scenesplat_amd/grad_exchange.py); no wrapper, no bucket copies.
            # Same initial weights on every rank, as DDP's constructor broadcast enforces:
            # cfg.find_unused_parameters needs no counterpart here: the stage all-reduces are issued in a FIXED order whatever
            # gradients arrive (grad_exchange.py), and a parameter that took no part in a step is reduced as zeros.
            from ..grad_exchange import StageGradExchange
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0)
            self.grad_exchange = StageGradExchange(model)
            return model
        return create_ddp_model(model, broadcast_buffers=False,
                                find_unused_parameters=self.cfg.get("find_unused_parameters", False))

    def train(self):
        self.before_train()
        for self.epoch in range(self.start_epoch, self.max_epoch):
            sampler = getattr(self.train_loader, "sampler", None)
            if get_world_size() > 1 and hasattr(sampler, "set_epoch"):
                sampler.set_epoch(self.epoch)
            self.model.train()
            self.data_iterator = enumerate(self.train_loader)
            self.before_epoch()
            for self.comm_info["iter"], self.comm_info["input_dict"] in self.data_iterator:
                self.before_step(); self.run_step(); self.after_step()
                if self.comm_info["iter"] == 2 and self.epoch == self.start_epoch and self.cfg.get("gc_freeze", True):
                    # a step allocates ~10^5 Python objects (autograd nodes, ctypes wrappers); an untimely full
                    # collection over the long-lived model objects stalls the enqueue thread for ~50 ms
                    import gc
                    gc.collect(); gc.freeze()
            self.after_epoch()
        self.after_train()

    # -- optional steady-state path (cfg["steady_state"] = True; single rank, CUDA) ---------------------------------------
    # Forward + backward of batches whose integer plan has a shape seen before are replayed as ONE hipGraph launch
    # (scenesplat_amd/steady_state.py); optimizer, scheduler, gradient clipping and the hooks stay where they are.  Batches of
    # any other shape, and models whose step reads values on the host, run eagerly through the same object (LangPretrainer with all three
    # criteria is capturable: the segmented-sum contrastive loss has no host read).  The reference has no counterpart
    # (pointcept/engines/train.py:142-196 launches every kernel of every step from Python).
    def _steady_run_step(self, inp, amp):
        from ..steady_state import SteadyStateStep
        model = self.model
        backbone = getattr(model, "backbone", model)
        if self._steady is None:
            def fwd_bwd(plan, tensors):
                d = dict(self._steady_host); d.update(tensors); d["plan"] = plan
                with torch.autocast(self.device.type, dtype=torch.bfloat16, enabled=amp):
                    out = model(d)
                out["loss"].backward()
                return {k: v for k, v in out.items() if torch.is_tensor(v)}
            self._steady = SteadyStateStep(fwd_bwd, [p for p in model.parameters() if p.requires_grad], warmup=1)
        tensors = {k: v for k, v in inp.items() if isinstance(v, torch.Tensor) and v.is_cuda}
        self._steady_host = {k: v for k, v in inp.items() if k not in tensors}
        # the key holds the host DECISIONS of the step: the model's own summary when it has one (LangPretrainer.steady_key: the
        # schedule gates, not the epoch_progress float, which would re-capture every epoch), every other host scalar and
        # list / tuple of scalars verbatim
        def _hashable(v):
            if isinstance(v, (int, float, bool, str, type(None))):
                return repr(v)
            if isinstance(v, (list, tuple)) and all(isinstance(e, (int, float, bool, str, type(None))) for e in v):
                return repr(tuple(v))
            return None
        own = model.steady_key(self._steady_host) if hasattr(model, "steady_key") else None
        skip = {"epoch_progress"} if own is not None else set()
        key = (own,) + tuple(sorted((k, _hashable(v)) for k, v in self._steady_host.items() if k not in skip and _hashable(v) is not None))
        plan = backbone.prepare_plan(inp)
        self.optimizer.zero_grad(set_to_none=True)
        return self._steady(plan, tensors, key=(key, bool(model.training)))

    def run_step(self):
        inp = self.comm_info["input_dict"]
        for k, v in inp.items():
            if isinstance(v, torch.Tensor):
                inp[k] = v.to(self.device, non_blocking=True)
        amp = bool(self.cfg.get("enable_amp")) and self.device.type == "cuda"
        if (self.cfg.get("steady_state") and self.device.type == "cuda" and get_world_size() == 1
                and hasattr(getattr(self.model, "backbone", self.model), "prepare_plan") and "grid_coord" in inp):
            inp["epoch_progress"] = self.epoch / self.max_epoch
            out = self._steady_run_step(inp, amp)
            if self.cfg.get("clip_grad") is not None:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.cfg["clip_grad"])
            self.optimizer.step()
            self.scheduler.step()
            self.comm_info["model_output_dict"] = out
            return
        with torch.autocast(self.device.type, dtype=torch.bfloat16, enabled=amp):
            inp["epoch_progress"] = self.epoch / self.max_epoch
            out = self.model(inp)
            loss = out["loss"]
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self.grad_exchange is not None:
            self.grad_exchange.finish()
        if self.cfg.get("clip_grad") is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.cfg["clip_grad"])
        self.optimizer.step()
        self.scheduler.step()
        self.comm_info["model_output_dict"] = out


# ---- multi-dataset training (engines/train.py:346-374, datasets/dataloader.py:23-102) ---------------------------------------
class MultiDatasetLoader:
    """The batch schedule of the reference's MultiDatasetDataloader over already-built per-dataset loaders (data loading itself is
    outside the hot path and injected, as for Trainer): every batch comes from ONE dataset; a round takes ratios[i] batches from
    loader i in turn; the FIRST loader defines the epoch (the iteration ends when it is exhausted), the others restart when they
    run out.  len() = the number of batches of one such epoch."""

    def __init__(self, dataloaders, ratios):
        self.dataloaders, self.ratios = list(dataloaders), [int(r) for r in ratios]
        if not self.dataloaders or len(self.dataloaders) != len(self.ratios) or min(self.ratios) < 1:
            raise ValueError("MultiDatasetLoader: one positive ratio per loader")
        self.sampler = self            # Trainer.train calls train_loader.sampler.set_epoch under DDP (dataloader.py:115-121)

    def set_epoch(self, epoch):
        for dl in self.dataloaders:
            s = getattr(dl, "sampler", None)
            if hasattr(s, "set_epoch"):
                s.set_epoch(epoch)

    def __iter__(self):
        import itertools
        slots = [i for i, r in enumerate(self.ratios) for _ in range(r)]      # one round: r_i consecutive slots for loader i
        streams = [iter(dl) for dl in self.dataloaders]
        done = object()
        for i in itertools.cycle(slots):
            batch = next(streams[i], done)
            if batch is done:
                if i == 0:                       # the main dataset defines the epoch
                    return
                streams[i] = iter(self.dataloaders[i])    # the others start over
                batch = next(streams[i])
            yield batch

    def __len__(self):
        full, rem = divmod(len(self.dataloaders[0]), self.ratios[0])
        return full * sum(self.ratios) + rem


@TRAINERS.register_module("MultiDatasetTrainer")
class MultiDatasetTrainer(Trainer):
    """The trainer type of the concat-dataset language configs (configs/concat_dataset/lang-pretrain-...-contrastive.py:111-112;
    engines/train.py:346-374): batches follow MultiDatasetLoader's schedule, `iter_per_epoch` is the length of that schedule and the
    scheduler's total_steps = iter_per_epoch x (max_epoch - start_epoch).  train_loader: a MultiDatasetLoader, or a list of
    (loader, ratio) pairs (ratio = the dataset's `loop` in the reference's config)."""

    def __init__(self, cfg, train_loader=None, logger=None):
        if not isinstance(train_loader, MultiDatasetLoader):
            pairs = list(train_loader or [])
            train_loader = MultiDatasetLoader([p[0] for p in pairs], [p[1] for p in pairs])
        cfg = dict(cfg)
        sched = dict(cfg["scheduler"])
        sched["total_steps"] = max(1, len(train_loader)) * (cfg.get("eval_epoch", 1) - 0)     # start_epoch is 0 before any resume
        cfg["scheduler"] = sched
        super().__init__(cfg, train_loader=train_loader, logger=logger)
        self.comm_info["iter_per_epoch"] = len(train_loader)
