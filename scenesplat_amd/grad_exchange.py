"""Data-parallel gradient exchange by STAGE SLICES (replaces DistributedDataParallel's bucket machinery on this path).

Reference behaviour (pointcept/engines/defaults.py:13-34 -> torch DDP, engines/train.py:190-196): every rank runs the same
model on its own chunks and the parameter gradients are averaged over the ranks before the optimizer step, overlapped with the
backward pass.  DDP implements that with ~25 MB buckets filled by one copy kernel per parameter (393 parameters here: +2.3 ms
of copies per step on ONE rank, measured in round 2) and host-side bucket hooks.

This file does the same averaging with what the PTv3 backward already knows: gradients become final STAGE BY STAGE, in the
reverse order of the forward (dec0 first -- and dec0 holds 52 % of the 366.8 MB).  Each stage owns ONE flat fp32 buffer with a
fixed slot per parameter; when the last gradient of a stage has been accumulated (a post-accumulate-grad hook per parameter,
counted per stage) the stage's gradients are packed into its buffer with ONE multi-tensor copy and ONE all-reduce is issued on
the process group's own stream -- RCCL (backend "nccl") over xGMI on the GPU box, gloo in the CPU tests -- while the backward
of the earlier stages keeps running.  finish() waits for the collectives, divides by the world size and leaves every
`param.grad` a VIEW of its stage buffer (no copy back).  9 large collectives and 9 copies per step instead of 4 buckets + 393
copies; no host callback per parameter beyond a counter.

    ex = StageGradExchange(model)            # after the model is on its device, process group initialised
    loss.backward(); ex.finish()             # every step; then optimizer.step()

hooks=False ("packed" mode, for steps replayed as a hipGraph -- steady_state.py -- where no host callback can run inside the
backward): the step itself ends with ex.pack() (one multi-tensor copy per stage, captured with the step), and after the replay
ex.reduce() issues ONE all-reduce over the whole model (the stage buffers are consecutive slices of one allocation; ReduceOp.AVG where
the backend has it) and finishes.  Nothing overlaps the backward then (about 2-3 ms of exposed all-reduce at
8 ranks for the 367 MB of this model), but the host enqueues ~170 calls per step instead of ~1,300: measured on one rank, the
eager step under either DDP or the hook form is HOST-bound (44-47 ms/step of enqueue against 39 ms of GPU work), so the packed form
is what bench.py uses for N > 1.

World size 1 (or no process group): nothing is installed and every method is a no-op.
"""
import os
import re
import time

import torch
import torch.distributed as dist


def default_stage_of(name):
    """Parameter name -> stage label, in PT-v3m1's module naming (embedding, enc.encK[.down], dec.decK[.up]): what becomes final
    together in the backward pass.  Anything else falls into one trailing stage."""
    m = re.match(r"(?:module\.|backbone\.)*((?:enc|dec)\.(?:enc|dec)\d+)", name)
    if m:
        return m.group(1)
    m = re.match(r"(?:module\.|backbone\.)*(embedding)", name)
    return m.group(1) if m else "other"


class StageGradExchange:
    def __init__(self, model, process_group=None, stage_of=default_stage_of, average=True, force=False, hooks=True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.average = average
        self.stages = {}              # label -> dict(params, views, flat, pending, handle)
        self.prof = {"hooks": 0, "launch_s": 0.0, "finish_s": 0.0, "steps": 0} if os.environ.get("SS_EXCHANGE_PROFILE") else None
        self._order = []
        self._hooks = []
        self._avg_ok = None           # does the backend implement ReduceOp.AVG?  (probed by the first packed reduce)
        self.whole = {}
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())     # force: one-rank rehearsal
        if not self.active:
            return
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            st = self.stages.setdefault(stage_of(name), dict(params=[], names=[]))
            st["params"].append(p); st["names"].append(name)
        plans = {}
        for label, st in self.stages.items():
            dev, dt = st["params"][0].device, st["params"][0].dtype
            if any(p.device != dev or p.dtype != dt for p in st["params"]):
                raise RuntimeError(f"stage {label}: parameters on several devices / dtypes")
            offs, tot = [], 0
            for p in st["params"]:
                offs.append(tot)
                tot += (p.numel() + 63) // 64 * 64          # slots start on 256-byte boundaries
            plans[label] = (dev, dt, offs, tot)
        # the stage buffers are consecutive slices of ONE allocation per (device, dtype): the packed form reduces the whole model
        # with one collective, the hook form one slice per stage
        self.whole = {}
        for label, (dev, dt, offs, tot) in plans.items():
            self.whole.setdefault((dev, dt), [0, None])[0] += tot
        for key, ent in self.whole.items():
            ent[1] = torch.zeros(ent[0], dtype=key[1], device=key[0]); ent[0] = 0
        for label, st in self.stages.items():
            dev, dt, offs, tot = plans[label]
            ent = self.whole[(dev, dt)]
            st["flat"] = ent[1][ent[0]:ent[0] + tot]; ent[0] += tot
            st["views"] = [st["flat"][o:o + p.numel()].view_as(p) for o, p in zip(offs, st["params"])]
            st["count"], st["handle"] = 0, None
            if hooks:
                for p in st["params"]:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(st)))

    def _make_hook(self, st):
        def hook(_param):
            st["count"] += 1
            if st["count"] == len(st["params"]):
                if self.prof is not None:
                    t0 = time.perf_counter(); self._launch(st); self.prof["launch_s"] += time.perf_counter() - t0
                else:
                    self._launch(st)
        return hook

    def _launch(self, st):
        """Every gradient of the stage is final: pack (skipping those that already live in the stage buffer) and all-reduce."""
        self._pack(st)
        st["handle"] = dist.all_reduce(st["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._order.append(st)

    def _pack(self, st):
        src, dst = [], []
        for p, v in zip(st["params"], st["views"]):
            g = p.grad
            if g is None:
                v.zero_()                                  # a parameter that took no part in this step
            elif g.data_ptr() != v.data_ptr():
                src.append(g); dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)

    def pack(self):
        """Packed mode, inside the (captured) step after backward(): every stage's gradients into its buffer."""
        if not self.active:
            return
        for st in self.stages.values():
            self._pack(st)

    def reduce(self):
        """Packed mode, after the step (outside any capture): all-reduce every stage buffer, average, point param.grad at the slots."""
        if not self.active:
            return
        # stream-synchronous collectives (async_op=False): with RCCL the call returns once the all-reduce is enqueued and the compute
        # stream is made to wait for it -- nothing overlaps here anyway, and no Work handle has to be waited for on the host
        t0 = time.perf_counter()
        for ent in self.whole.values():
            self._reduce_whole(ent[1])
        for st in self.stages.values():
            for p, v in zip(st["params"], st["views"]):
                p.grad = v
            st["count"], st["handle"] = 0, None
        if self.prof is not None:
            self.prof["finish_s"] += time.perf_counter() - t0; self.prof["steps"] += 1

    def _reduce_whole(self, buf):
        """One collective over every stage buffer of a (device, dtype): the mean when the backend has it (RCCL: ReduceOp.AVG, no
        extra pass over the 367 MB), else sum + one division (gloo)."""
        if self.average and self.world > 1 and self._avg_ok is not False:
            try:
                dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
                self._avg_ok = True
                return
            except (RuntimeError, ValueError, NotImplementedError):
                if self._avg_ok:            # it worked before: a real failure, not a missing feature
                    raise
                self._avg_ok = False
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        if self.average and self.world > 1:
            buf.div_(self.world)

    def finish(self):
        """Call after backward(): launches the stages whose count never completed (parameters without a gradient this step), waits
        for every collective, averages, and points each param.grad at its slot."""
        if not self.active:
            return
        t0 = time.perf_counter()
        for st in self.stages.values():
            if st["handle"] is None:
                self._launch(st)
        for st in self._order:
            st["handle"].wait()
            if self.average and self.world > 1:
                st["flat"].div_(self.world)
            for p, v in zip(st["params"], st["views"]):
                p.grad = v
            st["count"], st["handle"] = 0, None
        self._order = []
        if self.prof is not None:
            self.prof["finish_s"] += time.perf_counter() - t0; self.prof["steps"] += 1

    def zero_grad(self):
        """set_to_none for every managed parameter (the slots are overwritten by the next pack)."""
        for st in self.stages.values():
            for p in st["params"]:
                p.grad = None
            st["count"] = 0

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def describe(self):
        return {k: dict(params=len(st["params"]), mbytes=st["flat"].numel() * st["flat"].element_size() / 1e6) for k, st in self.stages.items()}
