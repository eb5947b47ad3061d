"""ScenePlan: the integer structure of one forward pass, built once on the GPU.

Everything PTv3 derives from grid coordinates alone -- 4-curve codes/orders per resolution
level, the grid-pool partitions, attention window indices with duplicate padding, submanifold
rulebooks -- is computed here by the HIP kernels of csrc/serialize.hip before any float work
starts, so the float pipeline runs without host round trips.  Host syncs: one for
(serialization depth, batch offsets) and one per pooling level (pooled size + offsets).

Mirrors, for the structure it produces (reference file:line):
  Point.serialization            pointcept/models/utils/structure.py:47-102
  SerializedPooling (int part)   point_transformer_v3m1_base.py:372-428
  get_padding_and_inverse        point_transformer_v3m1_base.py:114-170
  Point.sparsify / spconv pairs  structure.py:104-140
"""
import os

import torch

from . import native as nv


class WindowIndex:
    """Padded-slot -> row maps of one (level, curve, patch size): gidx/sidx (n_pad) and
    win_start == cu_seqlens (ptv3:165-169)."""

    __slots__ = ("gidx", "sidx", "win_start", "num_windows", "max_window", "n", "n_pad", "offsets_pad")

    def __init__(self, gidx, sidx, win_start, num_windows, max_window, n, n_pad, offsets_pad):
        self.gidx, self.sidx, self.win_start = gidx, sidx, win_start
        self.num_windows, self.max_window, self.n, self.n_pad = num_windows, max_window, n, n_pad
        self.offsets_pad = offsets_pad


def window_layout(counts, patch):
    """Host-side closed form of the reference padding rule: returns (offsets_pad, win_start).
    An element with more than `patch` points is padded to a multiple of `patch`; smaller
    elements form one short window (ptv3:126-136, 156-164)."""
    off_pad, win = [0], []
    for c in counts:
        cp = (c + patch - 1) // patch * patch if c > patch else c
        win.extend(range(off_pad[-1], off_pad[-1] + cp, patch) if cp > 0 else [])
        off_pad.append(off_pad[-1] + cp)
    win.append(off_pad[-1])
    return off_pad, win


CONV_ORDER = os.environ.get("SS_CONV_ORDER", "mask")   # "z": plain curve order
CONV_COARSE_BITS = 15        # regroup inside blocks of 32768 curve positions (in-process A/B: 15 beats 13 by 0.2 ms, 11 loses 0.4)
CONV_MASK_MIN_SITES = 16384  # small levels are latency-bound; the regrouping costs more than it saves


class Level:
    def __init__(self, gc32, batch32, offsets, depth, codes, order, inverse, codes_sorted, curves):
        self.grid_coord, self.batch = gc32, batch32
        self.offsets = list(offsets)  # host, leading 0, len B+1
        self.depth = depth
        self.n = gc32.shape[0]
        self.device = gc32.device
        # rows of these (K, n) buffers are in the config's curve order; `curves[j]` = row used at index j
        self.codes, self.order, self.inverse, self.codes_sorted = codes, order, inverse, codes_sorted
        self.curves = list(curves)
        self.curve_names = None
        # link to the finer level (set on pooled levels): segment_csr operands
        self.cluster = self.idx_ptr = self.indices = None
        self.has_duplicates = False
        self._windows, self._nbr = {}, {}

    # -- accessors in *current* (possibly shuffled) curve order --------------------------------
    def order_row(self, j):
        return self.order[self.curves[j]]

    def inverse_row(self, j):
        return self.inverse[self.curves[j]]

    def code_row(self, j):
        return self.codes[self.curves[j]]

    @property
    def counts(self):
        return [self.offsets[i + 1] - self.offsets[i] for i in range(len(self.offsets) - 1)]

    def _layout(self, patch):
        """(offsets, offsets_pad, win_start) device int32 vectors + host copies for one patch size.
        Uploaded in one pinned, non-blocking copy per patch size (a blocking torch.tensor(..., device=)
        upload would synchronise the stream in the middle of the forward)."""
        lay = self._windows.get(("layout", int(patch)))
        if lay is None:
            off_pad, win = window_layout(self.counts, int(patch))
            packed = torch.tensor(self.offsets + off_pad + win, dtype=torch.int32).pin_memory().to(self.device, non_blocking=True)
            a, b = len(self.offsets), len(self.offsets) + len(off_pad)
            lay = (packed[:a], packed[a:b], packed[b:], off_pad, win)
            self._windows[("layout", int(patch))] = lay
        return lay

    def window(self, j, patch):
        key = (self.curves[j], int(patch))
        w = self._windows.get(key)
        if w is None:
            t_off, t_offp, t_win, off_pad, win = self._layout(patch)
            gidx, sidx = nv.window_index(self.order_row(j), t_off, t_offp, patch, off_pad[-1])
            maxw = max([win[i + 1] - win[i] for i in range(len(win) - 1)] or [0])
            w = WindowIndex(gidx, sidx, t_win, len(win) - 1, maxw, self.n, off_pad[-1], off_pad)
            self._windows[key] = w
        return w

    def conv_rowperm(self):
        """Site walk order of the fused conv kernels.  Base: a curve order (z if present), so that a tile of 128 /
        256 consecutive sites is spatially compact.  CONV_ORDER == "mask": inside blocks of 2^CONV_COARSE_BITS
        consecutive curve positions the sites are regrouped (stably) by their 27-bit neighbour mask, so a tile
        holds sites of ONE surface orientation and skips every tap that orientation lacks -- on room-102400 the
        256-site tiles issue 1.05x the useful tap work instead of 1.29x; the coarse blocks keep the gathers local."""
        rp = self._nbr.get("rowperm")
        if rp is not None:
            return rp
        base = self.order[0]
        for r, name in enumerate(self.curve_names):
            if name == "z":
                base = self.order[r]
                break
        rp = base
        if CONV_ORDER == "mask" and self.n >= CONV_MASK_MIN_SITES and self.n < (1 << 30):
            nbr = self.neighbors(3)
            key = nv.subm_tap_mask_keys(nbr, base.contiguous(), CONV_COARSE_BITS)
            bits = 27 + max(1, (self.n >> CONV_COARSE_BITS).bit_length())
            perm, _, _ = nv.argsort_i64(key.unsqueeze(0), bits, want_inverse=False, want_sorted=False)
            rp = base[perm[0].long()].contiguous()
        self._nbr["rowperm"] = rp
        return rp

    def conv_blocks(self, ksize):
        """Per-tap lists of the 64-site blocks (in conv_rowperm order) that hold a pair: wgrad work list."""
        key = ("blocks", ksize)
        b = self._nbr.get(key)
        if b is None:
            b = nv.subm_block_lists(self.neighbors(ksize), self.conv_rowperm())
            self._nbr[key] = b
        return b

    def neighbors_walk(self, ksize):
        """The rulebook in conv_rowperm order, [t][k] = neighbors(ksize)[t][rowperm[k]]: what the fp32-MFMA first-stage conv reads
        (csrc/subm_f32.hip) -- 128 contiguous bytes per tap and 32-site tile instead of 32 scattered words."""
        key = ("walk", ksize)
        w = self._nbr.get(key)
        if w is None:
            w = nv.subm_walk_rulebook(self.neighbors(ksize), self.conv_rowperm())
            self._nbr[key] = w
        return w

    def dup_runs(self):
        """(sorted codes (n) int64, order (n) int32) of the first curve: runs of equal codes = the rows of one voxel, ascending, the
        winner (lowest row, what the rulebook resolves a voxel to) first -- what ss_dup_fold_rows / ss_dup_zero_rows walk on a
        level with duplicate voxels (Mix3D batches).  Rows of the plan's own buffers: nothing to build, replay-safe."""
        return self.codes_sorted[0], self.order[0]

    def neighbors(self, ksize):
        """(k^3, n) int32 tap-major rulebook, shared by every conv of this level (indice_key)."""
        nb = self._nbr.get(ksize)
        if nb is None:
            zrow, swap = None, 0
            for r, name in enumerate(self.curve_names):
                if name == "z":
                    zrow, swap = r, 0
                    break
                if name == "z-trans" and zrow is None:
                    zrow, swap = r, 1
            if zrow is None:
                codes = nv.serialize_encode(self.grid_coord, self.batch, self.depth, ["z"])
                bits = _key_bits(self.depth, len(self.offsets) - 1)
                zo, _, zs = nv.argsort_i64(codes, bits, want_inverse=False)
                zkeys, zorder, swap = zs[0], zo[0], 0
            else:
                zkeys, zorder = self.codes_sorted[zrow], self.order[zrow]
            nb = nv.subm_rulebook(self.grid_coord, self.batch, self.depth, zkeys, zorder, swap, ksize)
            self._nbr[ksize] = nb
        return nb


class ScenePlan:
    def __init__(self, levels, order_names):
        self.levels, self.order_names = levels, order_names
        self.ready_event = None     # set when the plan was built on a side stream

    def tensors(self):
        for lv in self.levels:
            for t in (lv.grid_coord, lv.batch, lv.codes, lv.order, lv.inverse, lv.codes_sorted, lv.cluster, lv.idx_ptr):
                if t is not None:
                    yield t
            for v in list(lv._windows.values()) + list(lv._nbr.values()):
                for t in (v if isinstance(v, tuple) else (v,)):
                    if isinstance(t, torch.Tensor):
                        yield t
                    elif isinstance(t, WindowIndex):
                        yield t.gidx; yield t.sidx

    # -- steady-state replay (scenesplat_amd/steady_state.py) ------------------------------------------------------
    # A captured hipGraph holds the ADDRESSES of the plan tensors it was captured with.  A later batch whose plan has the
    # same host-side shape (level sizes, offsets, window layout: `signature`) is run by copying its tensors into those
    # addresses (`load_from`) and replaying.  Tensors are matched by ROLE.  The float pipeline reads a plan through
    # window(slot j, patch), indices / idx_ptr / cluster and the conv tables only; window indices are matched by SLOT
    # (shuffle_orders draws a new curve permutation per batch: slot j of the captured plan receives slot j of the new
    # one, whatever curve that is this step), everything else by name.
    def _slots(self):
        """-> list of (role, tensor)."""
        out = []
        for li, lv in enumerate(self.levels):
            for name in ("grid_coord", "batch", "codes", "order", "inverse", "codes_sorted", "cluster", "idx_ptr", "indices"):
                t = getattr(lv, name)
                if t is not None:
                    out.append(((li, name), t))
            for key, v in lv._windows.items():
                if key[0] == "layout":
                    for i, t in enumerate(v[:3]):
                        out.append(((li, "layout", key[1], i), t))
                else:
                    slot = lv.curves.index(key[0])
                    out.append(((li, "window", slot, key[1], "gidx", v.num_windows, v.max_window, v.n_pad), v.gidx))
                    out.append(((li, "window", slot, key[1], "sidx"), v.sidx))
            for key, v in lv._nbr.items():
                for i, t in enumerate(v if isinstance(v, tuple) else (v,)):
                    if isinstance(t, torch.Tensor):
                        out.append(((li, "nbr", key, i), t))
        return out

    def signature(self):
        """Hashable host-side shape of the plan; equal signatures <=> `load_from` is valid."""
        head = (tuple(self.order_names),) + tuple((lv.n, tuple(lv.offsets), lv.depth, bool(lv.has_duplicates), len(lv.curves))
                                                  for lv in self.levels)
        return head + tuple((role, tuple(t.shape), str(t.dtype)) for role, t in self._slots())

    def own_storage(self):
        """Give `indices` (a row of the finer level's order buffer) and the conv walk order (a row of `order` on small
        levels) storage of their own, so that `load_from` can fill every role independently.  Called once on the plan a
        graph is captured with."""
        for lv in self.levels:
            if lv.indices is not None:
                lv.indices = lv.indices.clone()
            rp = lv._nbr.get("rowperm")
            if rp is not None:
                lv._nbr["rowperm"] = rp.clone()
        return self

    @torch.no_grad()
    def load_from(self, other):
        """Copy the tensors of `other` (same signature) into this plan's tensors on the current stream, one multi-tensor
        copy per dtype.  Host-side fields stay: they are equal by signature, except the curve permutation, which only
        names the window slots."""
        if other.ready_event is not None:
            torch.cuda.current_stream().wait_event(other.ready_event)
            other.record_stream(torch.cuda.current_stream())
            other.ready_event = None
        mine, theirs = self._slots(), other._slots()
        if len(mine) != len(theirs):
            raise ValueError("load_from: plans differ in structure")
        by_dtype = {}
        for (ra, ta), (rb, tb) in zip(mine, theirs):
            if ra != rb or ta.shape != tb.shape or ta.dtype != tb.dtype:
                raise ValueError(f"load_from: plans differ at {ra} vs {rb}")
            if ta.numel():
                d_list, s_list = by_dtype.setdefault(ta.dtype, ([], []))
                d_list.append(ta); s_list.append(tb)
        for d_list, s_list in by_dtype.values():
            torch._foreach_copy_(d_list, s_list)

    def materialize(self, window_specs=(), kernel_sizes=()):
        """Build the lazily cached pieces now (on the current stream): window_specs = [(level, curve index,
        patch)], kernel_sizes = [(level, k) | (level, k, walk)], walk = True (also the walk-order rulebook) or the list of the
        channel widths of the level's convs: the walk-order rulebook is built when one of them will read it (the forward's own
        predicate, pointcept_api.ptv3.conv_wants_walk -- so a captured plan and every later plan hold the same pieces)."""
        for li, j, patch in window_specs:
            self.levels[li].window(j, patch)
        for li, k, *walk in kernel_sizes:
            lv = self.levels[li]
            lv.neighbors(k); lv.conv_blocks(k)
            want = walk[0] if walk else False
            if isinstance(want, (list, tuple)):
                from .pointcept_api.ptv3 import conv_wants_walk
                want = any(conv_wants_walk(lv.n, c, k) for c in want)
            if want:
                lv.neighbors_walk(k)

    def record_stream(self, stream):
        """The plan was allocated on another stream: keep the caching allocator from reusing its memory
        before `stream`'s queued work is done."""
        for t in self.tensors():
            t.record_stream(stream)


def _key_bits(depth, num_batches):
    return max(1, 3 * depth + max(0, (int(num_batches) - 1).bit_length()))


@torch.no_grad()
def build_plan(grid_coord, offset, order_names, strides, perms=None, depth=None):
    """grid_coord (N,3) integer GPU tensor; offset (B,) cumulative counts (GPU or CPU).
    perms: list (1 + len(strides)) of curve permutations (level 0 first) or None for identity
    at level 0 and identity at pooled levels.  The caller draws them (torch.randperm on the
    CPU RNG, as the reference does: structure.py:94-98, ptv3:408-412)."""
    if not grid_coord.is_cuda:
        raise RuntimeError("build_plan: grid_coord must live on the GPU (no CPU fallback)")
    dev = grid_coord.device
    K = len(order_names)
    gc32 = grid_coord.to(torch.int32).contiguous()
    n = gc32.shape[0]
    off_dev = offset.to(device=dev, dtype=torch.int32).contiguous()
    B = off_dev.numel()
    gmax = nv.grid_coord_max(gc32)
    host = torch.cat([gmax, off_dev]).cpu().tolist()            # sync 1: depth + offsets
    if depth is None:
        depth = int(host[0]).bit_length()
    if depth > 16 or 3 * depth + int(B).bit_length() > 63:
        raise ValueError("serialization depth out of range (structure.py:69,74)")
    offsets = [0] + [int(v) for v in host[1:]]
    if offsets[-1] != n:
        raise ValueError(f"offset[-1]={offsets[-1]} != number of points {n}")
    batch32 = nv.offsets_to_batch(off_dev, n)
    codes = nv.serialize_encode(gc32, batch32, depth, order_names)
    order, inverse, csorted = nv.argsort_i64(codes, _key_bits(depth, B))
    curves = list(range(K)) if perms is None else [int(v) for v in perms[0]]
    lv = Level(gc32, batch32, offsets, depth, codes, order, inverse, csorted, curves)
    lv.curve_names = list(order_names)
    dup = nv.count_duplicates(csorted[0])
    levels = [lv]
    for s, stride in enumerate(strides):
        prev = levels[-1]
        pd = (int(stride) - 1).bit_length()
        if pd > prev.depth:
            pd = 0
        cluster, idx_ptr, head, _ = nv.pool_partition(prev.code_row(0), prev.order_row(0), 3 * pd)
        ends = torch.tensor([max(e - 1, 0) for e in prev.offsets[1:]], dtype=torch.int64, device=dev)
        o0 = prev.order_row(0)
        new_off = (cluster[o0[ends].long()] + 1)
        if s == 0:
            new_off = torch.cat([new_off, dup])
        new_off = new_off.cpu().tolist()                              # sync per level: pooled offsets
        if s == 0:
            levels[0].has_duplicates = new_off.pop() > 0
        for b in range(B):  # empty batch elements own no cluster
            if prev.offsets[b + 1] == prev.offsets[b]:
                new_off[b] = new_off[b - 1] if b > 0 else 0
        n_out = int(new_off[-1])
        gco, bo, co = nv.pool_level_attrs(head, n_out, prev.grid_coord, prev.batch, prev.codes, pd)
        ndepth = prev.depth - pd
        o, inv, cs = nv.argsort_i64(co, _key_bits(ndepth, B))
        base = prev.curves
        curves = list(base) if perms is None else [base[int(v)] for v in perms[s + 1]]
        nl = Level(gco, bo, [0] + [int(v) for v in new_off], ndepth, co, o, inv, cs, curves)
        nl.curve_names = list(order_names)
        nl.cluster, nl.idx_ptr, nl.indices = cluster, idx_ptr, o0
        nl.has_duplicates = False  # pooled levels have unique voxels by construction
        levels.append(nl)
    if len(strides) == 0:
        levels[0].has_duplicates = int(dup.item()) > 0
    return ScenePlan(levels, list(order_names))


# torch's synchronisation detector (torch.cuda.set_sync_debug_mode) is PROCESS-wide.  steady_state.py arms it for one eager step per
# plan signature; a plan build on another thread reads sizes on the host and would trip it.  Both sides take this gate: a build holds
# it while it runs (so the detector is never armed in the middle of one), steady_state holds it while the detector is armed (so no
# build starts inside the window) -- the plan thread simply waits at the gate for that one step.
HOST_SYNC_GATE = __import__("threading").Lock()
PLAN_BUILD_MAX_RETRIES = 200     # x 5 ms: a build that still trips a detector somebody else left armed becomes an error, not a hang


class PlanAhead:
    """Integer plans built by a HOST THREAD, `depth` steps ahead of the float pipeline.

    A plan build has host round trips (depth, pooled sizes: structure.py:69-74, ptv3:392-400 need them on the host to size
    tensors).  Its small kernels share the GPU with a step that fills every CU, so they are often scheduled only when the
    running step thins out at its end -- and a loop that builds the next plan between two steps then blocks until the
    current step is over, enqueues the next one late and leaves the GPU idle for the host's latency (1.0-1.5 ms per 39 ms
    step measured on the replayed bench step, 2.5 ms in a cold process: scripts/trace_step.sh).  With the builds on their
    own thread (and their own stream) the step loop never waits: it takes a finished plan and enqueues.  The role of a
    data-loader worker; the reference does the same work inside the model's forward (structure.py:47-102).

        ahead = PlanAhead(lambda: model.prepare_plan(next_batch(), stream=side), depth=2)
        plan = ahead.get()            # every step
        ahead.close()
    """

    def __init__(self, build, depth=2, device=None):
        import queue
        import threading
        self._build, self._q, self._stop, self._err = build, queue.Queue(maxsize=max(1, int(depth))), False, None
        self._device = torch.cuda.current_device() if device is None and torch.cuda.is_available() else device
        self._t = threading.Thread(target=self._run, name="ss-plan-ahead", daemon=True)
        self._t.start()

    def _run(self):
        import queue
        import time
        try:
            if self._device is not None:
                torch.cuda.set_device(self._device)
            retries = 0
            while not self._stop:
                try:
                    with HOST_SYNC_GATE:            # never while steady_state has torch's process-wide sync detector armed
                        plan = self._build()
                    retries = 0
                except RuntimeError as e:
                    # The gate keeps this library's own detector window away from the build.  A detector armed by somebody else
                    # (user code calling torch.cuda.set_sync_debug_mode("error")) still makes the build's host round trips raise:
                    # retried for a bounded time (it may be a short window), then handed to the consumer as the error it is.
                    if "synchronizing" in str(e) and not self._stop and retries < PLAN_BUILD_MAX_RETRIES:
                        retries += 1
                        time.sleep(0.005)
                        continue
                    raise
                while not self._stop:
                    try:
                        self._q.put(plan, timeout=0.05)
                        break
                    except queue.Full:
                        pass
        except BaseException as e:  # noqa: BLE001 -- handed to the consumer
            self._err = e
            self._q.put(None)

    def get(self):
        plan = self._q.get()
        if plan is None:
            self._stop = True
            raise RuntimeError("plan build thread failed") from self._err
        return plan

    def close(self):
        import queue
        self._stop = True
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._t.join(timeout=30)
