"""Tensor-level wrappers over the C-ABI (include/scenesplat_hip.h).

PyTorch supplies device memory and the current HIP stream only; every call here lands in a
hand-written HIP kernel.  Operands are validated on the host (device, dtype, contiguity,
shape) before any launch; a CPU tensor is an error -- there is no fallback path."""
import ctypes
import os

import torch

from . import _lib
from ._lib import check

F32, BF16 = 0, 1
ATTN_SIMT, ATTN_MFMA = 0, 1
HAVE_MFMA_ATTN = True  # bf16 MFMA window attention (csrc/attention_mfma.hip)
ORDER_IDS = {"z": 0, "z-trans": 1, "hilbert": 2, "hilbert-trans": 3}


_GROUP_WIDE_SHARES = os.environ.get("SS_WGRAD_GROUP_WIDE", "1") != "0"   # 0: plan every problem of a group as if it had the chip to itself (diagnostic A/B)
_GROUP_KEEP = []      # (event, pinned sources, device copies) of grouped launches whose upload may still be in flight
_GROUP_LOCK = __import__("threading").Lock()


class DescriptorPool:
    """Pinned host memory for the descriptors of grouped launches issued INSIDE a hipGraph capture (steady_state.py).
    The captured host-to-device copy re-reads its source on every replay, so the source must live, unchanged, as long as
    the graph does; and pinned memory cannot be allocated while a stream is capturing.  The pool is allocated before the
    capture begins and owned by the captured step.

    open_capture(device), called right after capture_begin, records ONE copy of the whole pool into a device mirror: a copy
    node reads its source at REPLAY time, so the tables that the step writes into the pool later in the capture are all
    uploaded by that single node (34 separate 5-us copies per step before)."""

    def __init__(self, nbytes=1 << 20):
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8).pin_memory()
        self.off = 0
        self.device_side = []
        self.mirror = None

    def open_capture(self, device):
        self.mirror = torch.empty(self.buf.numel(), dtype=torch.uint8, device=device)
        self.mirror.copy_(self.buf, non_blocking=True)

    def take(self, arr):
        """-> (pinned host view holding arr, device view of the same bytes in the mirror | None)"""
        nb = arr.nbytes
        off = (self.off + 63) & ~63
        if off + nb > self.buf.numel():
            raise RuntimeError("descriptor pool of the captured step is full")
        self.off = off + nb
        view = self.buf[off:off + nb]
        view.numpy()[:] = arr.reshape(-1).view("uint8")
        host = view.view(_NP2T[arr.dtype.name]).view(arr.shape)
        dev = None if self.mirror is None else self.mirror[off:off + nb].view(_NP2T[arr.dtype.name]).view(arr.shape)
        return host, dev


_NP2T = {"int64": torch.int64, "int32": torch.int32}
CAPTURE_POOL = None   # a DescriptorPool while a capture is open


def _upload_descriptors(desc, starts, dev):
    """(descriptor rows, first-workgroup table) -> device tensors through pinned memory, without a stream sync."""
    import numpy as np
    starts = np.asarray(starts, dtype=np.int32)
    if CAPTURE_POOL is not None and torch.cuda.is_current_stream_capturing():
        (d_host, d_dev), (s_host, s_dev) = CAPTURE_POOL.take(desc), CAPTURE_POOL.take(starts)
        if d_dev is None or d_dev.device != torch.device(dev):
            d_dev, s_dev = d_host.to(dev, non_blocking=True), s_host.to(dev, non_blocking=True)
            CAPTURE_POOL.device_side.append((d_dev, s_dev))
        return d_dev, s_dev
    d_host, s_host = torch.from_numpy(desc).pin_memory(), torch.from_numpy(starts).pin_memory()
    d_dev, s_dev = d_host.to(dev, non_blocking=True), s_host.to(dev, non_blocking=True)
    # the pinned sources must outlive their asynchronous copies: each upload is kept until an event recorded behind it has passed
    # (a thread whose stream is capturing must not query events: it only appends)
    ev = None
    with _GROUP_LOCK:                         # (grouped launches may come from more than one host thread)
        if not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event(); ev.record()
            while _GROUP_KEEP and _GROUP_KEEP[0][0] is not None and _GROUP_KEEP[0][0].query():
                _GROUP_KEEP.pop(0)
        _GROUP_KEEP.append((ev, d_host, s_host, d_dev, s_dev))
        if len(_GROUP_KEEP) > 4096:           # (never reached in practice: a stalled stream would have to hold thousands of launches)
            _GROUP_KEEP[0][0].synchronize() if _GROUP_KEEP[0][0] is not None else None
            _GROUP_KEEP.pop(0)
    return d_dev, s_dev


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype=None, name="tensor", shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: scenesplat_amd ops need a GPU tensor (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise RuntimeError(f"unsupported feature dtype {t.dtype} (float32 / bfloat16 only)")


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def lib():
    return _lib.load()


# ---- serialization ------------------------------------------------------------------------
def grid_coord_max(gc32):
    _req(gc32, torch.int32, "grid_coord")
    out = torch.empty(1, dtype=torch.int32, device=gc32.device)
    check(lib().ss_grid_coord_max(_p(gc32), gc32.shape[0], _p(out), _stream()), "ss_grid_coord_max")
    return out


def offsets_to_batch(offsets32, n):
    _req(offsets32, torch.int32, "offsets")
    batch = torch.empty(n, dtype=torch.int32, device=offsets32.device)
    check(lib().ss_offsets_to_batch(_p(offsets32), offsets32.numel(), n, _p(batch), _stream()), "ss_offsets_to_batch")
    return batch


def serialize_encode(gc32, batch32, depth, order_names):
    n = gc32.shape[0]
    _req(gc32, torch.int32, "grid_coord", (n, 3))
    if batch32 is not None:
        _req(batch32, torch.int32, "batch", (n,))
    k = len(order_names)
    ids = (ctypes.c_int * k)(*[ORDER_IDS[o] for o in order_names])
    codes = torch.empty((k, n), dtype=torch.int64, device=gc32.device)
    check(lib().ss_serialize_encode(_p(gc32), _p(batch32), n, int(depth), ids, k, _p(codes), _stream()),
          "ss_serialize_encode")
    return codes


def argsort_i64(keys, key_bits, want_inverse=True, want_sorted=True):
    """keys (K, n) int64 >= 0 -> order (K,n) int32 [, inverse (K,n) int32, sorted keys (K,n) int64]; stable."""
    _req(keys, torch.int64, "keys")
    K, n = keys.shape
    dev = keys.device
    order = torch.empty((K, n), dtype=torch.int32, device=dev)
    inverse = torch.empty((K, n), dtype=torch.int32, device=dev) if want_inverse else None
    skeys = torch.empty((K, n), dtype=torch.int64, device=dev) if want_sorted else None
    nb = lib().ss_argsort_workspace_bytes(n, K)
    ws = _ws(nb, dev)
    check(lib().ss_argsort_i64(_p(keys), K, n, int(max(1, min(64, key_bits))), _p(order), _p(inverse), _p(skeys),
                               _p(ws), ws.numel(), _stream()), "ss_argsort_i64")
    return order, inverse, skeys


def count_duplicates(sorted_keys_row):
    _req(sorted_keys_row, torch.int64, "sorted_keys")
    cnt = torch.empty(1, dtype=torch.int32, device=sorted_keys_row.device)
    check(lib().ss_count_duplicates(_p(sorted_keys_row), sorted_keys_row.numel(), _p(cnt), _stream()),
          "ss_count_duplicates")
    return cnt


def pool_partition(code0, order0, shift_bits):
    n = code0.numel()
    _req(code0, torch.int64, "code0", (n,)); _req(order0, torch.int32, "order0", (n,))
    dev = code0.device
    cluster = torch.empty(n, dtype=torch.int32, device=dev)
    idx_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    head = torch.empty(n, dtype=torch.int32, device=dev)
    n_out = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _ws(lib().ss_pool_partition_workspace_bytes(n), dev)
    check(lib().ss_pool_partition(_p(code0), _p(order0), n, int(shift_bits), _p(cluster), _p(idx_ptr), _p(head),
                                  _p(n_out), _p(ws), ws.numel(), _stream()), "ss_pool_partition")
    return cluster, idx_ptr, head, n_out


def pool_level_attrs(head, n_out, gc32, batch32, codes, pool_depth):
    K, n_in = codes.shape
    _req(codes, torch.int64, "codes"); _req(gc32, torch.int32, "grid_coord", (n_in, 3)); _req(batch32, torch.int32, "batch", (n_in,))
    _req(head, torch.int32, "head")
    if head.numel() < n_out:
        raise RuntimeError("head shorter than n_out")
    dev = codes.device
    gco = torch.empty((n_out, 3), dtype=torch.int32, device=dev)
    bo = torch.empty(n_out, dtype=torch.int32, device=dev)
    co = torch.empty((K, n_out), dtype=torch.int64, device=dev)
    check(lib().ss_pool_level_attrs(_p(head), n_out, n_in, _p(gc32), _p(batch32), _p(codes), K, int(pool_depth),
                                    _p(gco), _p(bo), _p(co), _stream()), "ss_pool_level_attrs")
    return gco, bo, co


def window_index(order_row, offsets, offsets_pad, patch, n_pad):
    _req(order_row, torch.int32, "order"); _req(offsets, torch.int32, "offsets"); _req(offsets_pad, torch.int32, "offsets_pad")
    B = offsets.numel() - 1
    dev = order_row.device
    gidx = torch.empty(n_pad, dtype=torch.int32, device=dev)
    sidx = torch.empty(n_pad, dtype=torch.int32, device=dev)
    check(lib().ss_window_index(_p(order_row), _p(offsets), _p(offsets_pad), B, int(patch), n_pad, _p(gidx), _p(sidx),
                                _stream()), "ss_window_index")
    return gidx, sidx


RULEBOOK_HASHED = True    # hash-table lookups (ss_subm_rulebook_hashed) instead of binary search in the sorted z keys


def subm_rulebook(gc32, batch32, depth, zkeys_sorted, zorder, swap_xy, ksize):
    n = gc32.shape[0]
    _req(gc32, torch.int32, "grid_coord", (n, 3)); _req(batch32, torch.int32, "batch", (n,))
    nbr = torch.empty((ksize ** 3, n), dtype=torch.int32, device=gc32.device)
    if RULEBOOK_HASHED and n < (1 << 30):
        ws = _ws(12 * lib().ss_subm_rulebook_table_size(n), gc32.device)
        check(lib().ss_subm_rulebook_hashed(_p(gc32), _p(batch32), n, int(depth), int(ksize), _p(nbr), _p(ws), ws.numel(), _stream()),
              "ss_subm_rulebook_hashed")
        return nbr
    _req(zkeys_sorted, torch.int64, "zkeys_sorted", (n,)); _req(zorder, torch.int32, "zorder", (n,))
    check(lib().ss_subm_rulebook(_p(gc32), _p(batch32), n, int(depth), _p(zkeys_sorted), _p(zorder), int(swap_xy),
                                 int(ksize), _p(nbr), _stream()), "ss_subm_rulebook")
    return nbr


def subm_tap_mask_keys(nbr, order, coarse_bits):
    """(taps,n) rulebook + walk order -> (n,) int64 sort keys: tap mask | coarse block id << taps."""
    taps, n = nbr.shape
    _req(nbr, torch.int32, "nbr"); _req(order, torch.int32, "order", (n,))
    keys = torch.empty(n, dtype=torch.int64, device=nbr.device)
    check(lib().ss_subm_tap_mask_keys(_p(nbr), _p(order), n, taps, int(coarse_bits), _p(keys), _stream()), "ss_subm_tap_mask_keys")
    return keys


def subm_conv_fwd_uses_pipe(n, cin, cout, taps):
    """Does subm_conv_fwd run this shape on the pipeline kernel (the one that profits from a walk-order rulebook)?"""
    return bool(lib().ss_subm_conv_fwd_uses_pipe(n, cin, cout, taps)) and lib().ss_subm_conv_splits(n, cout, taps) <= 1


def subm_conv_fwd(x, w, bias, nbr, rowperm, out_dtype=torch.bfloat16, nbr_walk=None):
    """x (n,cin) bf16, w (cout,taps,cin) bf16, bias (cout) f32|None, nbr (taps,n) -> (n,cout).
    nbr_walk: the rulebook in walk order (subm_walk_rulebook), read by the pipeline kernel where the shape runs on it."""
    n, cin = x.shape
    cout, taps, cin2 = w.shape
    _req(x, torch.bfloat16, "x"); _req(w, torch.bfloat16, "w"); _req(nbr, torch.int32, "nbr", (taps, n))
    if cin2 != cin or cin % 8:
        raise RuntimeError(f"subm_conv_fwd: cin mismatch / not a multiple of 8 ({cin}, {cin2})")
    if bias is not None:
        _req(bias, torch.float32, "bias", (cout,))
    if rowperm is not None:
        _req(rowperm, torch.int32, "rowperm", (n,))
    splits = lib().ss_subm_conv_splits(n, cout, taps)
    if splits > 1:      # small level: tap ranges in parallel, fp32 atomics, one conversion pass
        acc = torch.zeros((n, cout), dtype=torch.float32, device=x.device)
        check(lib().ss_subm_conv_fwd_splitk(_p(x), _p(w), _p(bias), _p(nbr), _p(rowperm), _p(acc), n, cin, cout, taps, splits,
                                            _stream()), "ss_subm_conv_fwd_splitk")
        return acc if out_dtype == torch.float32 else acc.to(out_dtype)
    out = torch.empty((n, cout), dtype=out_dtype, device=x.device)
    if nbr_walk is not None:
        _req(nbr_walk, torch.int32, "nbr_walk", (taps, n))
        check(lib().ss_subm_conv_fwd_walk(_p(x), _p(w), _p(bias), _p(nbr), _p(nbr_walk), _p(rowperm), _p(out), n, cin, cout, taps,
                                          dtype_code(out), _stream()), "ss_subm_conv_fwd_walk")
        return out
    check(lib().ss_subm_conv_fwd(_p(x), _p(w), _p(bias), _p(nbr), _p(rowperm), _p(out), n, cin, cout, taps,
                                 dtype_code(out), _stream()), "ss_subm_conv_fwd")
    return out


def subm_weight_mirror(w):
    """w (cout,taps,cin) bf16 -> wt (cin,taps,cout) bf16 with wt[ci][T-1-t][co] = w[co][t][ci] (dgrad weight)."""
    cout, taps, cin = w.shape
    _req(w, torch.bfloat16, "w")
    wt = torch.empty((cin, taps, cout), dtype=torch.bfloat16, device=w.device)
    check(lib().ss_subm_weight_mirror(_p(w), _p(wt), cout, taps, cin, _stream()), "ss_subm_weight_mirror")
    return wt


def subm_im2col(x, nbr):
    """x (n,c) bf16 with c % 8 == 0, nbr (taps,n) -> (n, taps*c): neighbour rows side by side, zeros where missing."""
    n, c = x.shape
    taps = nbr.shape[0]
    _req(x, torch.bfloat16, "x"); _req(nbr, torch.int32, "nbr", (taps, n))
    if c % 8:
        raise RuntimeError("subm_im2col: channels must be a multiple of 8")
    out = torch.empty((n, taps * c), dtype=torch.bfloat16, device=x.device)
    check(lib().ss_subm_im2col(_p(x), _p(nbr), _p(out), n, taps, c * 2, _stream()), "ss_subm_im2col")
    return out


def subm_conv_fwd_pipe(x, w, bias, nbr, rowperm, out_dtype=torch.bfloat16):
    """The LDS-DMA pipeline kernel directly (tests / benches); subm_conv_fwd dispatches to it for wide, large levels."""
    n, cin = x.shape
    cout, taps = w.shape[0], w.shape[1]
    out = torch.empty((n, cout), dtype=out_dtype, device=x.device)
    check(lib().ss_subm_conv_fwd_pipe(_p(x), _p(w), _p(bias), _p(nbr), _p(rowperm), _p(out), n, cin, cout, taps,
                                      dtype_code(out), _stream()), "ss_subm_conv_fwd_pipe")
    return out


def linear_fwd(x, w, bias=None, out_dtype=torch.bfloat16):
    """out = x @ w.T + bias on the pipeline GEMM.  x (m,k) bf16, w (n,k) bf16, bias (n) f32 or None."""
    m, k = x.shape
    n = w.shape[0]
    out = torch.empty((m, n), dtype=out_dtype, device=x.device)
    check(lib().ss_linear_fwd(_p(x), _p(w), _p(bias), _p(out), m, k, n, dtype_code(out), _stream()), "ss_linear_fwd")
    return out


def subm_block_lists(nbr, rowperm):
    """Per tap the compacted list of 64-site blocks with at least one pair: (count (taps), list (taps, nblocks))."""
    taps, n = nbr.shape
    _req(nbr, torch.int32, "nbr")
    if rowperm is not None:
        _req(rowperm, torch.int32, "rowperm", (n,))
    nb = (n + 63) // 64
    cnt = torch.empty(taps, dtype=torch.int32, device=nbr.device)
    lst = torch.empty((taps, nb), dtype=torch.int32, device=nbr.device)
    check(lib().ss_subm_block_lists(_p(nbr), _p(rowperm), n, taps, _p(cnt), _p(lst), _stream()), "ss_subm_block_lists")
    return cnt, lst


# ---- zero-initialised fp32 arena for the weight-gradient accumulators -------------------------------
# The wgrad kernels accumulate with fp32 atomics into ZEROED outputs; ~80 separate torch.zeros per backward cost
# ~0.9 ms of fill kernels and as many launches.  A model may reserve one zero-filled block per step
# (zero_arena_begin) from which the accumulators are carved; without a reservation each call allocates its own.
_ARENA = {"buf": None, "off": 0}


def zero_arena_begin(numel, device):
    """Reserve `numel` zero-filled fp32 elements (one fill kernel) for this step's wgrad accumulators."""
    _ARENA["buf"] = torch.zeros(int(numel), dtype=torch.float32, device=device) if numel > 0 else None
    _ARENA["off"] = 0


def zeros_f32(numel, device):
    """numel zero fp32 elements: a slice of the step's arena when it has room (16-byte aligned), else a fresh tensor."""
    buf, off = _ARENA["buf"], _ARENA["off"]
    if buf is not None and buf.device == device and off + numel <= buf.numel():
        _ARENA["off"] = off + ((numel + 3) & ~3)
        return buf[off:off + numel]
    return torch.zeros(int(numel), dtype=torch.float32, device=device)


def subm_conv_wgrad(x, dout, nbr, rowperm, blocks, out=None, nbr_walk=None):
    """-> dW (cout,taps,cin) f32 = sum_i dout[i] (x) x[nbr[t][i]].  blocks = subm_block_lists(nbr, rowperm).
    out: a ZEROED (cout,taps,cin) f32 accumulator to add into (deferred launches).
    nbr_walk: the rulebook in walk order (subm_walk_rulebook), read by the pipeline kernel where the shape runs on it."""
    n, cin = x.shape
    cout = dout.shape[1]
    taps = nbr.shape[0]
    _req(x, torch.bfloat16, "x"); _req(dout, torch.bfloat16, "dout", (n, cout)); _req(nbr, torch.int32, "nbr", (taps, n))
    if cin % 8 or cout % 8:
        raise RuntimeError("subm_conv_wgrad: channels must be multiples of 8")
    if rowperm is not None:
        _req(rowperm, torch.int32, "rowperm", (n,))
    dw = zeros_f32(cout * taps * cin, x.device).view(cout, taps, cin) if out is None else _req(out, torch.float32, "out", (cout, taps, cin))
    cnt, lst = blocks
    _req(cnt, torch.int32, "blk_count", (taps,)); _req(lst, torch.int32, "blk_list", (taps, (n + 63) // 64))
    if nbr_walk is not None:
        _req(nbr_walk, torch.int32, "nbr_walk", (taps, n))
    check(lib().ss_subm_conv_wgrad_walk(_p(x), _p(dout), _p(nbr), _p(nbr_walk), _p(rowperm), _p(cnt), _p(lst), _p(dw), n, cin, cout, taps,
                                        _stream()), "ss_subm_conv_wgrad")
    return dw


def subm_f32_weight_layout(w, mirror=False):
    """w (32, taps, c) f32 -> the fp32-MFMA conv's operand layout [tap][cp / 8][2][32][4] (cp = c padded to 16 or 32).
    mirror: the dgrad operand, w'[ci][T-1-t][co] = w[co][t][ci] (needs c == 32)."""
    w = w.float()
    if mirror:
        w = w.flip(1).permute(2, 1, 0)
    co, taps, c = w.shape
    cp = 16 if c <= 16 else 32
    if co != 32 or c > 32:
        raise RuntimeError("subm_f32: 32 output channels, at most 32 input channels")
    if cp != c:
        w = torch.nn.functional.pad(w, (0, cp - c))
    return w.reshape(32, taps, cp // 8, 2, 4).permute(1, 2, 3, 0, 4).contiguous()


def subm_walk_rulebook(nbr, rowperm):
    """The rulebook in walk order: (taps, n) with [t][k] = nbr[t][rowperm[k]] (rowperm None: nbr itself)."""
    return nbr if rowperm is None else nbr.index_select(1, rowperm).contiguous()


def subm_f32_fwd(x, wq, bias, nbr, rowperm, winners_only=False):
    """x (n, cp) f32 (cp 16 | 32), wq from subm_f32_weight_layout, nbr = subm_walk_rulebook(...) -> (n, 32) f32, exact fp32
    products (v_mfma_f32_32x32x2_f32).  winners_only: the dgrad form for a level with duplicate voxels -- x = dup_fold_rows(dout),
    rows that are not the winner of their voxel come out as zeros."""
    n, cp = x.shape
    taps = nbr.shape[0]
    _req(x, torch.float32, "x"); _req(wq, torch.float32, "wq", (taps, cp // 8, 2, 32, 4)); _req(nbr, torch.int32, "nbr", (taps, n))
    if bias is not None:
        _req(bias, torch.float32, "bias", (32,))
    if rowperm is not None:
        _req(rowperm, torch.int32, "rowperm", (n,))
    out = torch.empty((n, 32), dtype=torch.float32, device=x.device)
    if winners_only:
        if bias is not None:
            raise RuntimeError("subm_f32_fwd: the dgrad form takes no bias")
        check(lib().ss_subm_f32_dgrad_dup(_p(x), _p(wq), _p(nbr), _p(rowperm), _p(out), n, cp, 32, taps, _stream()), "ss_subm_f32_dgrad_dup")
        return out
    check(lib().ss_subm_f32_fwd(_p(x), _p(wq), _p(bias), _p(nbr), _p(rowperm), _p(out), n, cp, 32, taps, _stream()), "ss_subm_f32_fwd")
    return out


def dup_runs_from_rulebook(nbr):
    """(sorted keys (n) int64, order (n) int32) naming the duplicate-voxel runs of a level, derived from a rulebook alone: the centre
    tap of nbr (taps, n) is every site's winner row, a stable sort by it lists each voxel's rows ascending behind their winner.
    (A ScenePlan level hands out its curve codes instead: Level.dup_runs.)"""
    keys, order = torch.sort(nbr[nbr.shape[0] // 2].long(), stable=True)
    return keys.contiguous(), order.to(torch.int32).contiguous()


def dup_fold_rows(src, runs):
    """Adjoint of "every site of a voxel reads the voxel's winner row": out[winner] = sum of src over the voxel's rows (fp32
    accumulate, deterministic), out[other rows] = 0.  src (n, C) f32 | bf16 with 16-byte rows; runs = (sorted keys, order)."""
    keys, order = runs
    n, C = src.shape
    _req(src, None, "src"); _req(keys, torch.int64, "sorted_keys", (n,)); _req(order, torch.int32, "order", (n,))
    out = torch.empty_like(src)
    check(lib().ss_dup_fold_rows(_p(src), _p(keys), _p(order), _p(out), n, C, dtype_code(src), _stream()), "ss_dup_fold_rows")
    return out


def dup_zero_rows_(x, runs):
    """In place: rows of x (n, C) that are not the winner of their voxel := 0."""
    keys, order = runs
    n = x.shape[0]
    _req(x, None, "x"); _req(keys, torch.int64, "sorted_keys", (n,)); _req(order, torch.int32, "order", (n,))
    check(lib().ss_dup_zero_rows(_p(keys), _p(order), _p(x), n, x.shape[1] * x.element_size(), _stream()), "ss_dup_zero_rows")
    return x


def subm_f32_wgrad(x, dout, nbr, rowperm, blocks, cin):
    """-> dW (32, taps, cin) f32 = sum_i dout[i] (x) x[nbr[t][i]][:cin]; x (n, cp) f32, dout (n, 32) f32; nbr in walk order
    (subm_walk_rulebook), blocks = subm_block_lists(original nbr, rowperm)."""
    n, cp = x.shape
    taps = nbr.shape[0]
    _req(x, torch.float32, "x"); _req(dout, torch.float32, "dout", (n, 32)); _req(nbr, torch.int32, "nbr", (taps, n))
    if rowperm is not None:
        _req(rowperm, torch.int32, "rowperm", (n,))
    cnt, lst = blocks
    _req(cnt, torch.int32, "blk_count", (taps,)); _req(lst, torch.int32, "blk_list", (taps, (n + 63) // 64))
    dw = zeros_f32(32 * taps * cin, x.device).view(32, taps, cin)
    check(lib().ss_subm_f32_wgrad(_p(x), _p(dout), _p(nbr), _p(rowperm), _p(cnt), _p(lst), _p(dw), n, cp, cin, 32, taps, _stream()),
          "ss_subm_f32_wgrad")
    return dw


def subm_conv_wgrad_pipe(x, dout, nbr, rowperm, blocks):
    """The pipeline weight-gradient kernel directly (tests / benches)."""
    n, cin = x.shape
    cout, taps = dout.shape[1], nbr.shape[0]
    dw = torch.zeros((cout, taps, cin), dtype=torch.float32, device=x.device)
    cnt, lst = blocks
    check(lib().ss_subm_conv_wgrad_pipe(_p(x), _p(dout), _p(nbr), _p(rowperm), _p(cnt), _p(lst), _p(dw), n, cin, cout, taps,
                                        _stream()), "ss_subm_conv_wgrad_pipe")
    return dw


def linear_wgrad_alloc(k, nout, want_bias, device):
    """Zeroed fp32 accumulators (dW (nout,k), db (nout) | None) sharing one allocation."""
    buf = zeros_f32(nout * k + (nout if want_bias else 0), device)
    return buf[:nout * k].view(nout, k), (buf[nout * k:] if want_bias else None)


def linear_wgrad_into(x, dy, dw, db):
    """dw += dy^T @ x, db += column sums of dy (db may be None); dw / db must be zero on the first call."""
    m, k = x.shape
    nout = dy.shape[1]
    _req(x, torch.bfloat16, "x"); _req(dy, torch.bfloat16, "dy", (m, nout)); _req(dw, torch.float32, "dw", (nout, k))
    check(lib().ss_linear_wgrad(_p(x), _p(dy), _p(dw), _p(db), m, k, nout, _stream()), "ss_linear_wgrad")


def linear_wgrad_group(items):
    """ONE launch for many nn.Linear weight gradients.  items: list of (x (m,k) bf16, dy (m,n) bf16, dw (n,k) f32 ZEROED,
    db (n) f32 ZEROED | None).  Problems the pipeline kernel cannot take are run one by one."""
    import numpy as np
    if not items:
        return
    dev = items[0][0].device
    desc = np.zeros((len(items), 8), dtype=np.int64)
    starts = [0]
    rows = []
    # the CUs are shared out over the output tiles of the whole group (one round of workgroups for the launch)
    launch_tiles = sum(lib().ss_linear_wgrad_tiles(x.shape[1], dy.shape[1]) for x, dy, _, _ in items) if _GROUP_WIDE_SHARES else 0
    for x, dy, dw, db in items:
        _req(x, torch.bfloat16, "x"); _req(dy, torch.bfloat16, "dy", (x.shape[0], dy.shape[1]))
        _req(dw, torch.float32, "dw", (dy.shape[1], x.shape[1]))
        if db is not None:
            _req(db, torch.float32, "db", (dy.shape[1],))
        j = len(rows)
        nwg = lib().ss_linear_wgrad_group_plan2(x.shape[0], x.shape[1], dy.shape[1], launch_tiles, ctypes.c_void_p(desc[j].ctypes.data))
        if nwg <= 0 or len(rows) >= 128:
            linear_wgrad_into(x, dy, dw, db)
            continue
        desc[j, 0], desc[j, 1], desc[j, 2] = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
        desc[j, 3] = db.data_ptr() if db is not None else 0
        desc[j, 4] = x.shape[0]
        rows.append(j); starts.append(starts[-1] + nwg)
    if not rows:
        return
    nprob = len(rows)
    d_dev, s_dev = _upload_descriptors(desc[:nprob].copy(), starts, dev)
    check(lib().ss_linear_wgrad_group(_p(d_dev), _p(s_dev), nprob, starts[-1], _stream()), "ss_linear_wgrad_group")


def linear_wgrad(x, dy, want_bias=False):
    """dW (n_out,k_in) f32 = dy^T @ x on the pipeline kernel (and db (n_out) f32 = column sums of dy when want_bias).
    x (m,k_in) bf16, dy (m,n_out) bf16.  dW and db share one zero-filled allocation."""
    m, k = x.shape
    nout = dy.shape[1]
    _req(x, torch.bfloat16, "x"); _req(dy, torch.bfloat16, "dy", (m, nout))
    buf = zeros_f32(nout * k + (nout if want_bias else 0), x.device)
    dw = buf[:nout * k].view(nout, k)
    db = buf[nout * k:] if want_bias else None
    check(lib().ss_linear_wgrad(_p(x), _p(dy), _p(dw), _p(db), m, k, nout, _stream()), "ss_linear_wgrad")
    return (dw, db) if want_bias else dw


# ---- fused add + layernorm ----------------------------------------------------------------------
def _dt(t):
    return dtype_code(t) if t is not None else 0


def add_layernorm_fwd(x, y, rowscale, gamma, beta, eps, want_xout, want_copy, h_dtype):
    """v = x + rowscale*y -> (xout f32 | None, xcopy bf16 | None, h | None, mean, rstd)."""
    n, C = x.shape
    _req(x, None, "x")
    if y is not None:
        _req(y, None, "y", (n, C))
    if rowscale is not None:
        _req(rowscale, torch.float32, "rowscale", (n,))
    dev = x.device
    xout = torch.empty((n, C), dtype=torch.float32, device=dev) if want_xout else None
    xcopy = torch.empty((n, C), dtype=torch.bfloat16, device=dev) if want_copy else None
    h = mean = rstd = None
    if gamma is not None:
        _req(gamma, torch.float32, "gamma", (C,)); _req(beta, torch.float32, "beta", (C,))
        h = torch.empty((n, C), dtype=h_dtype, device=dev)
        mean = torch.empty(n, dtype=torch.float32, device=dev); rstd = torch.empty(n, dtype=torch.float32, device=dev)
    check(lib().ss_add_layernorm_fwd(_p(x), _dt(x), _p(y), _dt(y), _p(rowscale), _p(gamma), _p(beta), float(eps), _p(xout), F32,
                                     _p(xcopy), _p(h), _dt(h), _p(mean), _p(rstd), n, C, _stream()), "ss_add_layernorm_fwd")
    return xout, xcopy, h, mean, rstd


def ln_add_ln_fwd(x, t, gamma0, beta0, eps0, gamma1, beta1, eps1, h_dtype):
    """xout = x + LN0(t), h = LN1(xout) in one pass -> (xout f32, h, stats (n,4) f32)."""
    n, C = x.shape
    _req(x, None, "x"); _req(t, None, "t", (n, C))
    for p_, nm in ((gamma0, "gamma0"), (beta0, "beta0"), (gamma1, "gamma1"), (beta1, "beta1")):
        _req(p_, torch.float32, nm, (C,))
    dev = x.device
    xout = torch.empty((n, C), dtype=torch.float32, device=dev)
    h = torch.empty((n, C), dtype=h_dtype, device=dev)
    stats = torch.empty((n, 4), dtype=torch.float32, device=dev)
    check(lib().ss_ln_add_ln_fwd(_p(x), _dt(x), _p(t), _dt(t), _p(gamma0), _p(beta0), float(eps0), _p(gamma1), _p(beta1), float(eps1),
                                 _p(xout), _p(h), _dt(h), _p(stats), n, C, _stream()), "ss_ln_add_ln_fwd")
    return xout, h, stats


def group_partial_sums(items):
    """ONE launch for many partial-sum reductions.  items: list of (part (K, nb, C) f32, dst (K, C) f32): dst = part.sum(1)."""
    import numpy as np
    if not items:
        return
    dev = items[0][0].device
    desc = np.zeros((len(items), 4), dtype=np.int64)
    starts = [0]
    per = lib().ss_group_partial_sums_outputs_per_workgroup()
    for j, (part, dst) in enumerate(items):
        K, nb, C = part.shape
        _req(part, torch.float32, "part"); _req(dst, torch.float32, "dst", (K, C))
        desc[j] = (part.data_ptr(), dst.data_ptr(), nb, C | ((K * C) << 32))
        starts.append(starts[-1] + (K * C + per - 1) // per)
    d_dev, s_dev = _upload_descriptors(desc, starts, dev)
    check(lib().ss_group_partial_sums(_p(d_dev), _p(s_dev), len(items), starts[-1], _stream()), "ss_group_partial_sums")


def subm_weight_mirror_group(pairs):
    """ONE launch: wt (cin, taps, cout) = tap-mirrored transpose of w (cout, taps, cin) for every (w, wt) bf16 pair."""
    import numpy as np
    if not pairs:
        return
    dev = pairs[0][0].device
    desc = np.zeros((len(pairs), 5), dtype=np.int64)
    starts = [0]
    tile = lib().ss_subm_weight_mirror_group_tile()
    for j, (w, wt) in enumerate(pairs):
        cout, taps, cin = w.shape
        _req(w, torch.bfloat16, "w"); _req(wt, torch.bfloat16, "wt", (cin, taps, cout))
        desc[j] = (w.data_ptr(), wt.data_ptr(), cout, taps, cin)
        starts.append(starts[-1] + taps * ((cout + tile - 1) // tile) * ((cin + tile - 1) // tile))
    d_dev, s_dev = _upload_descriptors(desc, starts, dev)
    check(lib().ss_subm_weight_mirror_group(_p(d_dev), _p(s_dev), len(pairs), starts[-1], _stream()), "ss_subm_weight_mirror_group")


def transpose16_group(pairs):
    """ONE launch: dst (cols, rows) = src (rows, cols)^T for every (src, dst) pair of 2-byte tensors (bf16 weight copies)."""
    import numpy as np
    if not pairs:
        return
    dev = pairs[0][0].device
    desc = np.zeros((len(pairs), 4), dtype=np.int64)
    starts = [0]
    for j, (src, dst) in enumerate(pairs):
        r, c = src.shape
        if src.element_size() != 2 or dst.element_size() != 2 or tuple(dst.shape) != (c, r) or not (src.is_contiguous() and dst.is_contiguous()):
            raise RuntimeError("transpose16_group: contiguous 2-byte (rows, cols) -> (cols, rows) pairs")
        desc[j] = (src.data_ptr(), dst.data_ptr(), r, c)
        starts.append(starts[-1] + ((r + 63) // 64) * ((c + 63) // 64))
    d_dev, s_dev = _upload_descriptors(desc, starts, dev)
    check(lib().ss_transpose16_group(_p(d_dev), _p(s_dev), len(pairs), starts[-1], _stream()), "ss_transpose16_group")


def ln_add_ln_bwd(g_xout, g_h, xout, t, stats, gamma0, gamma1, gx_dtype, gt_dtype, reduce=True):
    """-> (g_x, g_t, dgamma0, dbeta0, dgamma1, dbeta1); reduce=False: (g_x, g_t, part (4, nb, C)) for a grouped reduction"""
    n, C = xout.shape
    dev = xout.device
    if g_xout is not None:
        _req(g_xout, torch.float32, "g_xout", (n, C))
    if g_h is not None:
        _req(g_h, None, "g_h", (n, C))
    g_x = torch.empty((n, C), dtype=gx_dtype, device=dev)
    g_t = torch.empty((n, C), dtype=gt_dtype, device=dev)
    nb = lib().ss_add_layernorm_bwd_blocks(n)
    part = torch.empty((4, nb, C), dtype=torch.float32, device=dev)
    check(lib().ss_ln_add_ln_bwd(_p(g_xout), _p(g_h), _dt(g_h), _p(xout), _p(t), _dt(t), _p(stats), _p(gamma0), _p(gamma1), _p(g_x),
                                 _dt(g_x), _p(g_t), _dt(g_t), _p(part), n, C, nb, _stream()), "ss_ln_add_ln_bwd")
    if not reduce:
        return g_x, g_t, part
    red = part.sum(1)
    return g_x, g_t, red[0], red[1], red[2], red[3]


def add_layernorm_bwd(g_xout, g_xcopy, g_h, v, mean, rstd, gamma, rowscale, gx_dtype, gy_dtype, reduce=True):
    """-> (g_x | None, g_y | None, dgamma | None, dbeta | None); reduce=False: (g_x, g_y, part (2, nb, C) | None, None)"""
    ref = g_h if g_h is not None else (g_xout if g_xout is not None else g_xcopy)
    n, C = ref.shape
    dev = ref.device
    for t, nm in ((g_xout, "g_xout"), (g_xcopy, "g_xcopy"), (g_h, "g_h")):
        if t is not None:
            _req(t, None, nm, (n, C))
    g_x = torch.empty((n, C), dtype=gx_dtype, device=dev) if gx_dtype is not None else None
    g_y = torch.empty((n, C), dtype=gy_dtype, device=dev) if gy_dtype is not None else None
    nb = lib().ss_add_layernorm_bwd_blocks(n)
    dgp = dbp = None
    if g_h is not None:
        part = torch.empty((2, nb, C), dtype=torch.float32, device=dev)     # one allocation, ONE reduction for both
        dgp, dbp = part[0], part[1]
        _req(v, None, "v", (n, C))
    check(lib().ss_add_layernorm_bwd(_p(g_xout), _dt(g_xout), _p(g_xcopy), _dt(g_xcopy), _p(g_h), _dt(g_h), _p(v), _dt(v),
                                     _p(mean), _p(rstd), _p(gamma), _p(rowscale), _p(g_x), _dt(g_x), _p(g_y), _dt(g_y),
                                     _p(dgp), _p(dbp), n, C, nb, _stream()), "ss_add_layernorm_bwd")
    if g_h is not None:
        if not reduce:
            return g_x, g_y, part, None
        red = part.sum(1)
        return g_x, g_y, red[0], red[1]
    return g_x, g_y, None, None


# ---- fused batchnorm (+GELU) ----------------------------------------------------------------------
def col_stats(x, shift=None):
    """-> (sum, sumsq) of (x - shift) per column, fp32 (partials reduced here)."""
    n, C = x.shape
    _req(x, None, "x")
    nb = lib().ss_add_layernorm_bwd_blocks(n)
    part = torch.empty((2, nb, C), dtype=torch.float32, device=x.device)
    ps, pq = part[0], part[1]
    check(lib().ss_col_stats(_p(x), _dt(x), _p(shift), _p(ps), _p(pq), n, C, nb, _stream()), "ss_col_stats")
    red = part.sum(1)
    return red[0], red[1]


def bn_batch_stats(x, running_mean, running_var, num_batches, momentum, eps):
    """Training-mode BatchNorm statistics in two launches: -> (mean, rstd) of the batch (fp32, biased variance), with the
    running statistics (fp32 buffers, in place; momentum update with the unbiased variance) and the batch counter updated
    the way nn.BatchNorm1d does.  running_mean also conditions the one-pass variance (sums of x - running_mean)."""
    n, C = x.shape
    _req(x, None, "x")
    _req(running_mean, torch.float32, "running_mean", (C,)); _req(running_var, torch.float32, "running_var", (C,))
    nb = lib().ss_add_layernorm_bwd_blocks(n)
    part = torch.empty((2, nb, C), dtype=torch.float32, device=x.device)
    check(lib().ss_col_stats(_p(x), _dt(x), _p(running_mean), _p(part[0]), _p(part[1]), n, C, nb, _stream()), "ss_col_stats")
    out = torch.empty((2, C), dtype=torch.float32, device=x.device)
    if num_batches is not None:
        _req(num_batches, torch.int64, "num_batches_tracked")
    check(lib().ss_bn_stats_finish(_p(part), _p(running_mean), nb, C, n, float(momentum), float(eps), _p(running_mean),
                                   _p(running_var), _p(num_batches), _p(out[0]), _p(out[1]), _stream()), "ss_bn_stats_finish")
    return out[0], out[1]


def bn_act_fwd(x, mean, rstd, gamma, beta, act, out_dtype):
    n, C = x.shape
    y = torch.empty((n, C), dtype=out_dtype, device=x.device)
    check(lib().ss_bn_act_fwd(_p(x), _dt(x), _p(mean), _p(rstd), _p(gamma), _p(beta), int(act), _p(y), _dt(y), n, C, _stream()),
          "ss_bn_act_fwd")
    return y


def bn_act_bwd(dy, x, mean, rstd, gamma, beta, act, training):
    """-> dx (x.dtype), dgamma, dbeta (fp32)"""
    n, C = x.shape
    _req(dy, None, "dy", (n, C))
    nb = lib().ss_add_layernorm_bwd_blocks(n)
    part = torch.empty((2, nb, C), dtype=torch.float32, device=x.device)
    pz, pzx = part[0], part[1]
    check(lib().ss_bn_act_bwd_reduce(_p(dy), _dt(dy), _p(x), _dt(x), _p(mean), _p(rstd), _p(gamma), _p(beta), int(act), _p(pz),
                                     _p(pzx), n, C, nb, _stream()), "ss_bn_act_bwd_reduce")
    sums = torch.empty((2 if not training else 4, C), dtype=torch.float32, device=x.device)
    check(lib().ss_bn_bwd_finish(_p(part), nb, C, n, _p(sums), _p(sums[2:]) if training else None, _stream()), "ss_bn_bwd_finish")
    sdz, sdzx = sums[0], sums[1]
    c1, c2 = (sums[2], sums[3]) if training else (None, None)
    dx = torch.empty_like(x)
    check(lib().ss_bn_act_bwd_apply(_p(dy), _dt(dy), _p(x), _dt(x), _p(mean), _p(rstd), _p(gamma), _p(beta), int(act), _p(c1),
                                    _p(c2), _p(dx), _dt(dx), n, C, _stream()), "ss_bn_act_bwd_apply")
    return dx, sdzx, sdz


# ---- open-vocabulary scan ----------------------------------------------------------------------------
def feat_text_scan(feat, text, want_max=True, idx=None, pred_accum=None):
    """feat (n, D), text (C, D) -> (max_prob (n) f32, argmax (n) int32) and/or pred_accum[idx] += sigmoid(feat text^T)."""
    f = _req(feat.to(torch.bfloat16).contiguous(), torch.bfloat16, "feat")
    t = _req(text.to(torch.bfloat16).contiguous(), torch.bfloat16, "text")
    n, D = f.shape
    C = t.shape[0]
    if t.shape[1] != D or D % 8 or C > 256:
        raise RuntimeError("feat_text_scan: need matching dim (multiple of 8) and <= 256 classes")
    mp = torch.empty(n, dtype=torch.float32, device=f.device) if want_max else None
    am = torch.empty(n, dtype=torch.int32, device=f.device) if want_max else None
    if pred_accum is not None:
        _req(pred_accum, torch.float32, "pred_accum")
        if pred_accum.shape[1] != C:
            raise RuntimeError("pred_accum must be (rows, num_classes)")
    if idx is not None:
        _req(idx, torch.int32, "idx", (n,))
    check(lib().ss_feat_text_scan(_p(f), _p(t), n, D, C, _p(mp), _p(am), _p(idx), _p(pred_accum), _stream()), "ss_feat_text_scan")
    return mp, am


# ---- distillation head ------------------------------------------------------------------------
def _mask_bytes(mask, n):
    m = mask if mask.dtype in (torch.bool, torch.uint8) else (mask > 0)
    m = _req(m.contiguous(), None, "valid_feat_mask", (n,))
    return m.view(torch.uint8) if m.dtype == torch.bool else m


def lang_head_fwd(feat, target, mask, normalize, want_p=True, p_dtype=torch.float32):
    """-> (p | None, sums (3) f32 | None, rowstat (n,4) f32).  feat / target (n, C) f32 | bf16."""
    _req(feat, None, "feat")
    n, C = feat.shape
    if C % 4 or C > 2048:
        raise RuntimeError("lang_head: channels must be a multiple of 4 and <= 2048")
    dev = feat.device
    p = torch.empty((n, C), dtype=p_dtype, device=dev) if want_p else None
    rowstat = torch.empty((n, 4), dtype=torch.float32, device=dev)
    sums = part = m8 = None
    if target is not None:
        _req(target, None, "target", (n, C))
        m8 = _mask_bytes(mask, n)
        part = torch.empty(lib().ss_lang_head_blocks(n) * 3, dtype=torch.float32, device=dev)
        sums = torch.empty(3, dtype=torch.float32, device=dev)
    check(lib().ss_lang_head_fwd(_p(feat), dtype_code(feat), _p(target), dtype_code(target) if target is not None else 0, _p(m8),
                                 int(bool(normalize)), _p(p), dtype_code(p) if p is not None else 0, _p(rowstat), _p(part), _p(sums),
                                 n, C, _stream()), "ss_lang_head_fwd")
    return p, sums, rowstat


def lang_head_bwd(feat, target, mask, normalize, rowstat, coef, dp_extra):
    """-> dfeat (n, C) in feat's dtype.  coef (2) f32 on the device (dL/dsums[0:2]) or None; dp_extra (n, C) or None."""
    _req(feat, None, "feat"); _req(rowstat, torch.float32, "rowstat")
    n, C = feat.shape
    m8 = None
    if target is not None:
        _req(target, None, "target", (n, C)); _req(coef, torch.float32, "coef")
        m8 = _mask_bytes(mask, n)
    if dp_extra is not None:
        _req(dp_extra, None, "dp", (n, C))
    dfeat = torch.empty_like(feat)
    check(lib().ss_lang_head_bwd(_p(feat), dtype_code(feat), _p(target), dtype_code(target) if target is not None else 0, _p(m8),
                                 int(bool(normalize)), _p(rowstat), _p(coef), _p(dp_extra),
                                 dtype_code(dp_extra) if dp_extra is not None else 0, _p(dfeat), dtype_code(dfeat), n, C, _stream()),
          "ss_lang_head_bwd")
    return dfeat


# ---- rows ------------------------------------------------------------------------------------
def gather_rows(src, idx, out=None):
    """out[i] = src[idx[i]] (zero row where idx < 0).  src (m, C)."""
    _req(src, None, "src"); _req(idx, torch.int32, "idx")
    n = idx.numel()
    if src.dim() != 2:
        raise RuntimeError("src must be (rows, C)")
    rb = src.shape[1] * src.element_size()
    if out is None:
        out = torch.empty((n, src.shape[1]), dtype=src.dtype, device=src.device)
    else:
        _req(out, src.dtype, "out", (n, src.shape[1]))
    check(lib().ss_gather_rows(_p(src), _p(idx), _p(out), n, rb, _stream()), "ss_gather_rows")
    return out


def scatter_rows(src, idx, dst):
    """dst[idx[i]] = src[i] for idx[i] >= 0 (idx unique)."""
    _req(src, None, "src"); _req(idx, torch.int32, "idx", (src.shape[0],)); _req(dst, src.dtype, "dst")
    if dst.shape[1] != src.shape[1]:
        raise RuntimeError("row width mismatch")
    rb = src.shape[1] * src.element_size()
    check(lib().ss_scatter_rows(_p(src), _p(idx), _p(dst), src.shape[0], rb, _stream()), "ss_scatter_rows")
    return dst


def gather_add_rows(a, b, idx):
    n, C = a.shape
    _req(a, None, "a"); _req(b, a.dtype, "b"); _req(idx, torch.int32, "idx", (n,))
    if b.shape[1] != C:
        raise RuntimeError("channel mismatch")
    out = torch.empty_like(a)
    check(lib().ss_gather_add_rows(_p(a), _p(b), _p(idx), _p(out), n, C, dtype_code(a), _stream()), "ss_gather_add_rows")
    return out


def segment_reduce(src, indices, idx_ptr, n_seg, mean):
    _req(src, None, "src"); _req(idx_ptr, torch.int32, "idx_ptr")
    if indices is not None:
        _req(indices, torch.int32, "indices")
    if idx_ptr.numel() < n_seg + 1:
        raise RuntimeError("idx_ptr shorter than n_seg+1")
    C = src.shape[1]
    out = torch.empty((n_seg, C), dtype=src.dtype, device=src.device)
    check(lib().ss_segment_reduce(_p(src), _p(indices), _p(idx_ptr), _p(out), n_seg, C, dtype_code(src), int(mean),
                                  _stream()), "ss_segment_reduce")
    return out


def segment_minmax(src, indices, idx_ptr, n_seg, is_max):
    """-> (out (n_seg,C), arg (n_seg,C) int32 source rows)"""
    _req(src, None, "src"); _req(idx_ptr, torch.int32, "idx_ptr")
    if indices is not None:
        _req(indices, torch.int32, "indices")
    if idx_ptr.numel() < n_seg + 1:
        raise RuntimeError("idx_ptr shorter than n_seg+1")
    C = src.shape[1]
    out = torch.empty((n_seg, C), dtype=src.dtype, device=src.device)
    arg = torch.empty((n_seg, C), dtype=torch.int32, device=src.device)
    check(lib().ss_segment_minmax(_p(src), _p(indices), _p(idx_ptr), _p(out), _p(arg), n_seg, C, dtype_code(src), int(is_max),
                                  _stream()), "ss_segment_minmax")
    return out, arg


def segment_minmax_bwd(dout, arg, n_src):
    _req(dout, None, "dout"); _req(arg, torch.int32, "arg", tuple(dout.shape))
    n_seg, C = dout.shape
    dsrc = torch.zeros((n_src, C), dtype=dout.dtype, device=dout.device)
    check(lib().ss_segment_minmax_bwd(_p(dout), _p(arg), _p(dsrc), n_seg, C, dtype_code(dout), _stream()), "ss_segment_minmax_bwd")
    return dsrc


def segment_bcast(dout, cluster, idx_ptr, mean):
    _req(dout, None, "dout"); _req(cluster, torch.int32, "cluster"); _req(idx_ptr, torch.int32, "idx_ptr")
    n, C = cluster.numel(), dout.shape[1]
    dsrc = torch.empty((n, C), dtype=dout.dtype, device=dout.device)
    check(lib().ss_segment_bcast(_p(dout), _p(cluster), _p(idx_ptr), _p(dsrc), n, C, dtype_code(dout), int(mean),
                                 _stream()), "ss_segment_bcast")
    return dsrc


# ---- attention -------------------------------------------------------------------------------
def window_attn_fwd(qkv, win, num_heads, scale, impl):
    n, C3 = qkv.shape
    C = C3 // 3
    _req(qkv, None, "qkv")
    if n != win.n:
        raise RuntimeError(f"qkv rows {n} != window index rows {win.n}")
    out = torch.empty((n, C), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((win.n_pad, num_heads), dtype=torch.float32, device=qkv.device)
    check(lib().ss_window_attn_fwd(_p(qkv), _p(win.gidx), _p(win.sidx), _p(win.win_start), win.num_windows,
                                   win.max_window, n, win.n_pad, C, num_heads, float(scale), dtype_code(qkv), int(impl),
                                   _p(out), _p(lse), _stream()), "ss_window_attn_fwd")
    return out, lse


def window_attn_bwd(qkv, out, dout, lse, win, num_heads, scale, impl):
    n, C3 = qkv.shape
    C = C3 // 3
    _req(qkv, None, "qkv"); _req(out, qkv.dtype, "out", (n, C)); _req(dout, qkv.dtype, "dout", (n, C))
    _req(lse, torch.float32, "lse", (win.n_pad, num_heads))
    dqkv = torch.empty_like(qkv)
    nb = lib().ss_window_attn_bwd_workspace_bytes(n, win.n_pad, C, num_heads, dtype_code(qkv))
    ws = _ws(nb, qkv.device)
    check(lib().ss_window_attn_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(win.gidx), _p(win.sidx), _p(win.win_start),
                                   win.num_windows, win.max_window, n, win.n_pad, C, num_heads, float(scale),
                                   dtype_code(qkv), int(impl), _p(dqkv), _p(ws), ws.numel(), _stream()),
          "ss_window_attn_bwd")
    return dqkv


# ---- head-major window attention (csrc/attention_hm.hip, round 3) ------------------------------------------------------
LOG2E = 1.4426950408889634


def headmajor_eligible(k):
    """The projection can write the head-major layout from its own epilogue (gemm8.hip) when k is a multiple of 64."""
    return k >= 64 and k % 64 == 0


def linear_fwd_headmajor(x, win, w, bias, num_heads, sec0_scale):
    """hm (sections, H, n_pad, D) bf16 = head-major, window-ordered x[gidx] @ w.T + bias; section 0 times sec0_scale (fp32,
    before the one bf16 rounding).  x (n, k) bf16, w (sections * C, k) bf16, bias (sections * C) f32 or None."""
    n, k = x.shape
    nout = w.shape[0]
    _req(x, torch.bfloat16, "x"); _req(w, torch.bfloat16, "weight", (nout, k))
    if bias is not None:
        _req(bias, torch.float32, "bias", (nout,))
    if n != win.n:
        raise RuntimeError(f"x rows {n} != window index rows {win.n}")
    C = k
    if nout % C:
        raise RuntimeError("output width must be a multiple of the channel count")
    hm = torch.empty((nout // C, num_heads, win.n_pad, C // num_heads), dtype=torch.bfloat16, device=x.device)
    check(lib().ss_linear_fwd_headmajor(_p(x), _p(win.gidx), _p(w), _p(bias), _p(hm), win.n_pad, k, nout, C, C // num_heads,
                                        float(sec0_scale), _stream()), "ss_linear_fwd_headmajor")
    return hm


def headmajor_pack(src, win, num_heads, sections, sec0_scale):
    """hm (sections, H, n_pad, D) bf16 from an (n, sections * C) projection (fp32 or bf16) in memory row order."""
    n, w = src.shape
    _req(src, None, "src")
    C = w // sections
    hm = torch.empty((sections, num_heads, win.n_pad, C // num_heads), dtype=torch.bfloat16, device=src.device)
    check(lib().ss_headmajor_pack(_p(src), dtype_code(src), _p(win.gidx), _p(hm), win.n_pad, C, num_heads, sections,
                                  float(sec0_scale), _stream()), "ss_headmajor_pack")
    return hm


def window_attn_hm_fwd(hm, win, num_heads):
    _req(hm, torch.bfloat16, "hm")
    sections, H, n_pad, D = hm.shape
    if sections != 3 or H != num_heads or n_pad != win.n_pad:
        raise RuntimeError(f"hm shape {tuple(hm.shape)} does not match the window index (n_pad {win.n_pad}, heads {num_heads})")
    C = H * D
    out = torch.empty((win.n, C), dtype=torch.bfloat16, device=hm.device)
    nlse2 = torch.empty((H, n_pad), dtype=torch.float32, device=hm.device)
    check(lib().ss_window_attn_hm_fwd(_p(hm), _p(win.sidx), _p(win.win_start), win.num_windows, win.max_window, win.n, n_pad, C,
                                      H, _p(out), _p(nlse2), _stream()), "ss_window_attn_hm_fwd")
    return out, nlse2


def window_attn_hm_bwd(hm, out, dout, nlse2, win, num_heads, scale):
    sections, H, n_pad, D = hm.shape
    C = H * D
    _req(out, torch.bfloat16, "out", (win.n, C)); _req(dout, torch.bfloat16, "dout", (win.n, C))
    _req(nlse2, torch.float32, "nlse2", (H, n_pad))
    dqkv = torch.empty((win.n, 3 * C), dtype=torch.bfloat16, device=hm.device)
    nb = lib().ss_window_attn_hm_bwd_workspace_bytes(win.n, n_pad, C, H)
    ws = _ws(nb, hm.device)
    check(lib().ss_window_attn_hm_bwd(_p(hm), _p(out), _p(dout), _p(nlse2), _p(win.gidx), _p(win.sidx), _p(win.win_start),
                                      win.num_windows, win.max_window, win.n, n_pad, C, H, float(scale), _p(dqkv), _p(ws),
                                      ws.numel(), _stream()), "ss_window_attn_hm_bwd")
    return dqkv


def stream_capture_status(stream=None):
    """0 = not capturing, 1 = capturing, 2 = the capture was invalidated (it must be abandoned, never ended), -1 = unknown."""
    st = stream if stream is not None else torch.cuda.current_stream()
    return int(lib().ss_stream_capture_status(ctypes.c_void_p(st.cuda_stream)))


def gelu(x, dy=None):
    """Exact-erf GELU on the HIP kernel: gelu(x), or dy * gelu'(x) when dy is given.  x (any shape) bf16 | f32, contiguous."""
    _req(x, None, "x")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"gelu: float32 / bfloat16 only, got {x.dtype}")
    if dy is not None:
        _req(dy, x.dtype, "dy", x.shape)
    out = torch.empty_like(x)
    check(lib().ss_gelu(_p(x), _p(dy), _p(out), x.numel(), dtype_code(x), _stream()), "ss_gelu")
    return out


def row_keep_scales(keep, seed=None):
    """keep (n) f32 in (0, 1] -> (n) f32 of Bernoulli(keep) / keep, one Philox draw per row.  seed: (1,) int64 DEVICE tensor; None draws
    it from torch's generator of the device (graph-safe: a captured call gets a fresh seed on every replay)."""
    _req(keep, torch.float32, "keep")
    if seed is None:
        seed = torch.randint(-(1 << 62), 1 << 62, (1,), dtype=torch.int64, device=keep.device)
    _req(seed, torch.int64, "seed", (1,))
    out = torch.empty_like(keep)
    check(lib().ss_row_keep_scales(_p(seed), _p(keep), _p(out), keep.numel(), _stream()), "ss_row_keep_scales")
    return out


def stream_capture_id(stream=None):
    """> 0: identity of the capture the stream is recording into; 0: not capturing."""
    st = stream if stream is not None else torch.cuda.current_stream()
    return int(lib().ss_stream_capture_id(ctypes.c_void_p(st.cuda_stream)))


def cast_bf16_group(srcs, dsts):
    """ONE launch: dst (bf16) = src (fp32) for every pair of equally shaped contiguous tensors (the weight shadows)."""
    import numpy as np
    if not srcs:
        return
    dev = srcs[0].device
    per = lib().ss_cast_bf16_group_elems_per_workgroup()
    desc = np.zeros((len(srcs), 3), dtype=np.int64)
    starts = [0]
    for j, (s, d) in enumerate(zip(srcs, dsts)):
        if s.dtype != torch.float32 or d.dtype != torch.bfloat16 or s.numel() != d.numel() or not (s.is_contiguous() and d.is_contiguous()):
            raise RuntimeError("cast_bf16_group: contiguous fp32 -> bf16 pairs of equal size")
        desc[j] = (s.data_ptr(), d.data_ptr(), s.numel())
        starts.append(starts[-1] + (s.numel() + per - 1) // per)
    d_dev, s_dev = _upload_descriptors(desc, starts, dev)
    check(lib().ss_cast_bf16_group(_p(d_dev), _p(s_dev), len(srcs), starts[-1], _stream()), "ss_cast_bf16_group")
