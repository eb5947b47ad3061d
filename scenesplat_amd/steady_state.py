"""Steady-state replay: forward + backward of a batch whose integer plan has a shape seen before, as ONE hipGraph launch.

The float pipeline of a PT-v3m1 step is ~1,100 kernel launches whose shapes depend only on the host-side shape of the
batch's ScenePlan (level sizes, batch offsets, window layout).  When batches repeat that shape -- fixed-size chunks of one
scene, the evaluator's chunk_size slices, the synthetic room of bench.py -- the launches are captured once with
`torch.cuda.graph` (a hipGraph on ROCm) and later steps copy their plan tensors and inputs into the captured addresses
(ScenePlan.load_from: one multi-tensor copy per dtype) and replay: the host enqueues a plan build, two copies and one
graph launch instead of a thousand kernels.  Batches with any other shape run eagerly, as does everything when
capture is refused (a host read inside the step: found by one eager step per signature under torch's sync detector before any
capture is attempted).

What is inside the graph is exactly what the eager step launches: the bf16 parameter shadows are re-cast from the fp32
parameters at the top of every replay (functional.refresh_shadows), dropout / DropPath draws advance the Philox offset of
the default generator per replay (torch registers it with the graph), and the parameter gradients are (re)written -- not
accumulated -- by every replay into the tensors the capture allocated; `param.grad` is pointed at them after each replay,
so an optimizer step between replays works as usual.  Gradient accumulation over several micro-batches and
DistributedDataParallel (its bucket hooks are host callbacks) are outside this path: the caller uses the eager step.
Contract: between steps the caller keeps no tensor that still carries an autograd graph of the model (the step itself
returns detached outputs): such a graph holds AccumulateGrad nodes of the default stream, which a capturing stream must not
wait for.  The checked eager step reports it when torch warns about it (it warns once per process).

No part of the reference corresponds to this file: the reference launches every kernel of every step from Python
(pointcept/engines/train.py:142-196)."""
import gc
import os
import warnings

import torch

from . import functional as SF
from . import native as nv
from .plan import HOST_SYNC_GATE


class _Captured:
    __slots__ = ("graph", "graph2", "plan", "inputs", "outputs", "grads", "keep")


class CaptureInvalidated(RuntimeError):
    """A hipGraph capture was invalidated half way (an illegal call while the stream captured).  Such a capture cannot be ended
    -- capture_end() on it crashes inside the runtime -- so it is abandoned, every replay path of the step is switched off and
    this error travels to the caller instead of a silent eager retry on a stream in an unknown state."""


_ABANDONED = []     # graph objects of invalidated captures: kept alive so that nothing ever calls into them again


class SteadyStateStep:
    """step = SteadyStateStep(fn, params); out = step(plan, {"feat": ..., ...})

    fn(plan, inputs) -> dict of tensors: the eager forward + backward (it must not synchronise with the host); the step
        returns them detached;
    params: the parameters whose .grad the step writes (their .grad is cleared before a capture);
    warmup: eager runs of a signature before it is captured (allocator pools, hipBLASLt heuristics, lazy code objects);
    max_graphs: signatures kept (least recently used is dropped; a captured step owns its activations' memory);
    tail, between (round 4, the split step of a data-parallel run): the step is fn(plan, inputs) -- forward + the first part of
        the backward --, then between() ON THE HOST, OUTSIDE any graph (it starts the gradient all-reduce of the stages that are
        final: a collective is never captured), then tail() -- the rest of the backward, working on what fn left in the
        caller's closure.  Captured as TWO graphs sharing one memory pool; a replay is graph 1, between(), graph 2."""

    def __init__(self, fn, params, warmup=2, max_graphs=2, enabled=True, tail=None, between=None):
        self.fn, self.params = fn, list(params)
        self.tail, self.between = tail, between
        self.warmup, self.max_graphs, self.enabled = int(warmup), int(max_graphs), bool(enabled)
        self._seen, self._graphs = {}, {}
        self._refused = {}             # signature -> repr of what refused its capture: THAT signature runs eagerly from then on
        self.refused = None            # the most recent refusal (None: none so far)
        self.poisoned = None           # set when a capture was invalidated: no further capture is attempted by this object
        self._checked = set()          # signatures whose eager step passed the sync check (every new signature is checked: a host-side
                                       # key may open a branch that reads values on the host)
        self.replays = self.eager_steps = 0

    @staticmethod
    def _signature(plan, inputs, key=None):
        return (plan.signature(), key) + tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(inputs.items()))

    def _eager(self, plan, inputs, arm=None):
        # outputs are handed out DETACHED: a caller holding on to them must not keep this step's autograd graph alive
        self.eager_steps += 1
        out = {k: v.detach() for k, v in self.fn(plan, inputs).items()}
        if self.tail is not None:
            if arm is None:
                if self.between is not None:
                    self.between()
                self.tail()
            else:
                # the sync-checked step: fn and tail run under torch's PROCESS-wide sync detector; between() -- host code outside the
                # graphs, typically an asynchronous collective whose backend may work on a thread of its own (gloo reads device
                # tensors on the host there) -- is called after the detector is disarmed, i.e. AFTER tail() in this one step.
                # (Legal for a gradient exchange: what between() reduces was packed by fn, tail() does not touch it.)
                self.tail()
                arm(False)
                if self.between is not None:
                    self.between()
        return out

    def __call__(self, plan, inputs, key=None):
        """key: hashable summary of every HOST-side value the step's control flow depends on (a loss schedule gate, a mode
        flag): a captured graph is replayed only for the key it was captured under."""
        if not self.enabled or self.poisoned is not None:
            return self._eager(plan, inputs)
        sig = self._signature(plan, inputs, key)
        if sig in self._refused:
            return self._eager(plan, inputs)
        cap = self._graphs.get(sig)
        if cap is None:
            if len(self._seen) > 256:                          # batches of ever-changing shape: forget the counts
                self._seen.clear(); self._checked.clear()
            self._seen[sig] = self._seen.get(sig, 0) + 1
            if self._seen[sig] <= self.warmup:
                return self._eager(plan, inputs)
            if sig not in self._checked:
                # One eager step on a SIDE stream with torch's sync detector armed, BEFORE any capture is attempted: what
                # would poison a capture -- a host read (.item(), .cpu(), a blocking copy), or an autograd graph of an
                # earlier step that is still referenced (its AccumulateGrad nodes belong to the default stream, which a
                # capturing stream must not touch) -- shows up here, where nothing needs unwinding.  (A capture that fails
                # half way cannot be unwound on this stack: ending an invalidated capture crashes inside the runtime.)
                self._checked.add(sig)
                main, side = torch.cuda.current_stream(), torch.cuda.Stream()
                side.wait_stream(main)
                # the detector is process-wide: plan builds on other threads (plan.PlanAhead) wait at the gate while it is armed
                armed = [False]

                def arm(on):
                    if on and not armed[0]:
                        HOST_SYNC_GATE.acquire(); torch.cuda.set_sync_debug_mode("error"); armed[0] = True
                    elif not on and armed[0]:
                        torch.cuda.set_sync_debug_mode("default"); HOST_SYNC_GATE.release(); armed[0] = False
                arm(True)
                # torch reports a stale AccumulateGrad node with a warn-ONCE warning: once per process unless warn_always is on, and a
                # capture that runs into such a node pulls the default stream into the capture -- ending that capture crashes inside the
                # runtime (seen: segmentation fault in capture_end).  Every checked step must see the warning, not only the first one.
                warn_always = torch.is_warn_always_enabled()
                torch.set_warn_always(True)
                try:
                    with warnings.catch_warnings(record=True) as seen, torch.cuda.stream(side):
                        warnings.simplefilter("always")
                        out = self._eager(plan, inputs, arm)
                    stale = [w for w in seen if "AccumulateGrad node's stream" in str(w.message)]
                    if stale:
                        raise RuntimeError("an autograd graph of an earlier step is still referenced (keep only detached "
                                           "outputs between steps): " + str(stale[0].message)[:120])
                    main.wait_stream(side)
                    return out
                except Exception as e:  # noqa: BLE001
                    arm(False)
                    main.wait_stream(side)
                    self._refuse(e, sig)
                    return self._eager(plan, inputs)
                finally:
                    arm(False)
                    torch.set_warn_always(warn_always)
            try:
                cap = self._capture(plan, inputs)
            except CaptureInvalidated as e:
                # no device-wide synchronisation here: this thread still owns the (dead) capture, the runtime would refuse it
                self.poisoned = repr(e)
                self._refuse(e, sig, sync=False)
                raise
            except Exception as e:  # noqa: BLE001 -- the capture was still valid and has been ended: the eager step works
                self._refuse(e, sig)
                return self._eager(plan, inputs)
            self._graphs[sig] = cap
            while len(self._graphs) > self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
        else:
            self._graphs[sig] = self._graphs.pop(sig)          # most recently used last
            try:
                cap.plan.load_from(plan)
            except ValueError as e:
                # the new plan holds other lazily built pieces than the captured one (e.g. a walk-order rulebook the model's
                # plan_specs did not announce): nothing has been launched yet -- this signature runs eagerly from now on
                self._graphs.pop(sig, None)
                self._refuse(e, sig, sync=False)
                return self._eager(plan, inputs)
            names = sorted(inputs)
            torch._foreach_copy_([cap.inputs[k] for k in names], [inputs[k] for k in names])
        cap.graph.replay()
        if cap.graph2 is not None:
            if self.between is not None:
                self.between()
            cap.graph2.replay()
        for p, g in cap.grads:
            p.grad = g
        self.replays += 1
        return cap.outputs

    def _refuse(self, exc, sig=None, sync=True):
        self.refused = repr(exc)
        if sig is not None:
            if len(self._refused) > 256:
                self._refused.clear()
            self._refused[sig] = self.refused
        SF.reset_state()
        for p in self.params:
            p.grad = None
        if sync:
            torch.cuda.synchronize()

    def _capture(self, plan, inputs):
        cap = _Captured()
        if plan.ready_event is not None:
            torch.cuda.current_stream().wait_event(plan.ready_event)
            plan.record_stream(torch.cuda.current_stream())
            plan.ready_event = None
        cap.plan = plan.own_storage()
        cap.inputs = {k: v.clone() for k, v in inputs.items()}
        gc.collect()      # unreachable autograd graphs of earlier steps hold AccumulateGrad nodes of the default stream
        for p in self.params:
            p.grad = None
        cap.keep = nv.DescriptorPool()
        cap.graph = torch.cuda.CUDAGraph()
        cap.graph2 = torch.cuda.CUDAGraph() if self.tail is not None else None
        main = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        nv.CAPTURE_POOL = cap.keep
        try:
            # capture_begin / capture_end by hand (not `with torch.cuda.graph`).  thread_local error mode: HIP calls of OTHER
            # threads (a pin-memory thread, a plan build on its own stream) no longer invalidate this capture.  When the step
            # raises, the capture is ended ONLY while the runtime still reports it active: ending an invalidated capture is the
            # call that crashed the process in round 2 (Fatal Python error in capture_end after hipErrorStreamCaptureImplicit).
            with torch.cuda.stream(side):
                cap.graph.capture_begin(capture_error_mode="thread_local")
                try:
                    cap.keep.open_capture(torch.cuda.current_device())     # one upload node for every descriptor table of the step
                    cap.outputs = {k: v.detach() for k, v in self.fn(cap.plan, cap.inputs).items()}
                except BaseException as e:
                    if os.environ.get("SS_STEADY_DEBUG"):
                        import traceback
                        traceback.print_exc()
                    status = nv.stream_capture_status(side)
                    if status == 1:
                        cap.graph.capture_end()          # a Python-level error inside a healthy capture: end it, drop the graph
                        raise
                    # invalidated (2) or unknown: abandon the graph object WITHOUT ending / resetting it (its destructor does
                    # not touch the stream) and surface the original error
                    _ABANDONED.append(cap.graph)
                    raise CaptureInvalidated(f"hipGraph capture invalidated (status {status}) by: {e!r}") from e
                if nv.stream_capture_status(side) != 1:
                    _ABANDONED.append(cap.graph)
                    raise CaptureInvalidated("hipGraph capture invalidated by a call inside the step (no Python error)")
                cap.graph.capture_end()
                if cap.graph2 is not None:
                    # second graph of a split step: the rest of the backward, in the SAME memory pool (it reads what graph 1
                    # saved for it and frees it as it goes).  between() is not called here: a capture executes nothing.
                    cap.graph2.capture_begin(pool=cap.graph.pool(), capture_error_mode="thread_local")
                    try:
                        self.tail()
                    except BaseException as e:
                        status = nv.stream_capture_status(side)
                        if status == 1:
                            cap.graph2.capture_end()
                            raise
                        _ABANDONED.extend([cap.graph, cap.graph2])
                        raise CaptureInvalidated(f"hipGraph capture (second graph) invalidated (status {status}) by: {e!r}") from e
                    if nv.stream_capture_status(side) != 1:
                        _ABANDONED.extend([cap.graph, cap.graph2])
                        raise CaptureInvalidated("hipGraph capture (second graph) invalidated by a call inside the step")
                    cap.graph2.capture_end()
        finally:
            nv.CAPTURE_POOL = None
        main.wait_stream(side)
        cap.grads = [(p, p.grad) for p in self.params if p.grad is not None]
        return cap
