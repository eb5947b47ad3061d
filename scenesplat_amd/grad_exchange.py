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

The collectives of a step are issued in a FIXED order on every rank -- the stages in reverse registration order, i.e. the order
in which PT-v3m1's backward finishes them -- whatever order the hooks fire in: a stage is launched only once every stage before it
in that order has been launched, the rest in finish().  A parameter that receives no gradient on some ranks only (a data-dependent
branch) therefore cannot reorder or resize the sequence of all-reduces (its slot is reduced as zeros).

hooks=False ("packed" mode, for steps replayed as a hipGraph -- steady_state.py -- where no host callback can run inside the
backward).  Round 4, the SPLIT form bench.py runs for N > 1: the stages named in `early` (dec.dec0: 52 % of the 366.8 MB, final
first) sit at the front of the one allocation; the step is two graphs (forward + the backward of the early stages, then the rest
of the backward -- PointTransformerV3's backward cut) and

    ex.pack("early")  [captured in graph 1]   ex.reduce_begin("early")  [between the replays: one asynchronous all-reduce of the
    early slice on the process group's stream, overlapped with graph 2]   ex.pack("late")  [captured in graph 2]
    ex.reduce_begin("late"); ex.reduce_end()  [after graph 2]

ReduceOp.AVG where the backend has it (RCCL), else sum + one division (gloo).  ex.pack(); ex.reduce() is the round-3 form: one
all-reduce over the whole model after the step, nothing overlapped.  The host enqueues ~170 calls per step either way instead of
~1,300: measured on one rank, the eager step under DDP or the hook form is HOST-bound (44-47 ms/step of enqueue against 39 ms of
GPU work).

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
    def __init__(self, model, process_group=None, stage_of=default_stage_of, average=True, force=False, hooks=True,
                 early=("dec.dec0",)):
        self.group = process_group
        self.early = tuple(early)
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.average = average
        self.stages = {}              # label -> dict(params, views, flat, pending, handle)
        self.prof = {"hooks": 0, "launch_s": 0.0, "finish_s": 0.0, "steps": 0} if os.environ.get("SS_EXCHANGE_PROFILE") else None
        self._order = []
        self._hooks = []
        self._sequence, self._cursor = [], 0          # hook form: the fixed launch order and how far this step has got in it
        self._pending = []                            # split form: (work handle | None, buffer, needs division)
        # ReduceOp.AVG only where the backend is known to implement it for device tensors (RCCL / NCCL); gloo accepts the enum for
        # CUDA tensors without raising and hands back garbage (NaN: tests/test_hip_split_exchange.py), so it is never probed there
        self._avg_ok = None
        if dist.is_available() and dist.is_initialized():
            self._avg_ok = True if dist.get_backend(process_group) == "nccl" else False
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
            self.whole.setdefault((dev, dt), [0, None, 0])[0] += tot
        for key, ent in self.whole.items():
            ent[1] = torch.zeros(ent[0], dtype=key[1], device=key[0]); ent[0] = 0
        # the EARLY stages (final first in the backward) take the front of each allocation: two contiguous slices per (device, dtype)
        labels = [lb for lb in self.stages if lb in self.early] + [lb for lb in self.stages if lb not in self.early]
        for label in labels:
            st = self.stages[label]
            dev, dt, offs, tot = plans[label]
            ent = self.whole[(dev, dt)]
            st["flat"] = ent[1][ent[0]:ent[0] + tot]; ent[0] += tot
            if label in self.early:
                ent[2] = ent[0]                        # end of the early slice
            st["views"] = [st["flat"][o:o + p.numel()].view_as(p) for o, p in zip(offs, st["params"])]
            st["count"], st["handle"], st["ready"], st["label"] = 0, None, False, label
            if hooks:
                for p in st["params"]:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(st)))
        # fixed launch order of the hook form: reverse registration order = the order PT-v3m1's backward finishes its stages in
        self._sequence = [self.stages[lb] for lb in reversed(list(self.stages))]

    def _make_hook(self, st):
        def hook(_param):
            st["count"] += 1
            if st["count"] == len(st["params"]):
                st["ready"] = True
                t0 = time.perf_counter()
                # launch in the fixed sequence only: this stage and any later one that was waiting for it
                while self._cursor < len(self._sequence) and self._sequence[self._cursor]["ready"]:
                    self._launch(self._sequence[self._cursor]); self._cursor += 1
                if self.prof is not None:
                    self.prof["launch_s"] += time.perf_counter() - t0
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

    def pack(self, which="all"):
        """Packed mode, inside the (captured) step after (a part of) backward(): the gradients of every stage ("all"), of the early
        stages or of the late ones into their buffers."""
        if not self.active:
            return
        for label, st in self.stages.items():
            if which == "all" or (which == "early") == (label in self.early):
                self._pack(st)

    def reduce_begin(self, which):
        """Split form, outside any capture: ONE asynchronous all-reduce of the early or the late slice of every allocation, on the
        process group's own stream (RCCL: behind an event of the current stream, so it starts when the packed gradients are
        final and runs beside whatever the current stream is given next)."""
        if not self.active:
            return
        t0 = time.perf_counter()
        for ent in self.whole.values():
            buf = ent[1][:ent[2]] if which == "early" else ent[1][ent[2]:]
            if buf.numel() == 0:
                continue
            h, div = None, False
            if self.average and self.world > 1 and self._avg_ok is not False:
                try:
                    h = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                    self._avg_ok = True
                except (RuntimeError, ValueError, NotImplementedError):
                    if self._avg_ok:            # it worked before: a real failure, not a missing feature
                        raise
                    self._avg_ok = False
            if h is None:
                h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                div = self.average and self.world > 1
            self._pending.append((h, buf, div))
        if self.prof is not None:
            self.prof["launch_s"] += time.perf_counter() - t0

    def reduce_end(self):
        """Split form: wait for the collectives begun this step (RCCL: the current stream waits, not the host), finish the average
        where the backend had no ReduceOp.AVG, point every param.grad at its slot."""
        if not self.active:
            return
        t0 = time.perf_counter()
        for h, buf, div in self._pending:
            if h is not None:
                h.wait()
            if div:
                buf.div_(self.world)
        self._pending = []
        for st in self.stages.values():
            for p, v in zip(st["params"], st["views"]):
                p.grad = v
            st["count"], st["handle"], st["ready"] = 0, None, False
        if self.prof is not None:
            self.prof["finish_s"] += time.perf_counter() - t0; self.prof["steps"] += 1

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
            st["count"], st["handle"], st["ready"] = 0, None, False
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
        while self._cursor < len(self._sequence):          # what the hooks did not reach, in the same fixed order
            self._launch(self._sequence[self._cursor]); self._cursor += 1
        for st in self._order:
            st["handle"].wait()
            if self.average and self.world > 1:
                st["flat"].div_(self.world)
            for p, v in zip(st["params"], st["views"]):
                p.grad = v
            st["count"], st["handle"], st["ready"] = 0, None, False
        self._order, self._cursor = [], 0
        if self.prof is not None:
            self.prof["finish_s"] += time.perf_counter() - t0; self.prof["steps"] += 1

    def zero_grad(self):
        """set_to_none for every managed parameter (the slots are overwritten by the next pack)."""
        for st in self.stages.values():
            for p in st["params"]:
                p.grad = None
            st["count"], st["ready"] = 0, False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def describe(self):
        return {k: dict(params=len(st["params"]), mbytes=st["flat"].numel() * st["flat"].element_size() / 1e6) for k, st in self.stages.items()}
