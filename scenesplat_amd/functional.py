"""Differentiable ops of the PTv3 hot path: torch.autograd.Function shells around the HIP
kernels (forward and backward are separate C-ABI entry points).

  window_attention   <- flash_attn_varlen_qkvpacked_func + the [order]/[inverse] row gathers
                        around it (ptv3:184-216), gather/scatter fused into the kernels
  segment_mean       <- torch_scatter.segment_csr(proj(feat)[indices], idx_ptr, "mean") (ptv3:416-418)
  unpool_add         <- parent.feat + point.feat[pooling_inverse]                      (ptv3:478)
  subm_conv3d        <- spconv.SubMConv3d on a cached rulebook                         (ptv3:278-284,499-506)
"""
import os
import weakref

import torch
from torch.utils.weak import WeakIdKeyDictionary

from . import native as nv


class _WindowAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, win, num_heads, scale, impl):
        qkv = qkv.contiguous()
        out, lse = nv.window_attn_fwd(qkv, win, num_heads, scale, impl)
        ctx.save_for_backward(qkv, out, lse)
        ctx.win, ctx.num_heads, ctx.scale, ctx.impl = win, num_heads, scale, impl
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        dqkv = nv.window_attn_bwd(qkv, out, dout.contiguous().to(qkv.dtype), lse, ctx.win, ctx.num_heads, ctx.scale, ctx.impl)
        return dqkv, None, None, None, None


def window_attention(qkv, win, num_heads, scale, impl=nv.ATTN_SIMT):
    """qkv (n, 3C) in memory row order -> (n, C); windows/padding per `win` (plan.WindowIndex)."""
    return _WindowAttention.apply(qkv, win, num_heads, scale, impl)


class _SegmentMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, level, mean):
        src = src.contiguous()
        ctx.level, ctx.mean = level, mean
        return nv.segment_reduce(src, level.indices, level.idx_ptr, level.n, mean)

    @staticmethod
    def backward(ctx, dout):
        lv = ctx.level
        return nv.segment_bcast(dout.contiguous(), lv.cluster, lv.idx_ptr, ctx.mean), None, None


class _SegmentMinMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, level, is_max):
        src = src.contiguous()
        out, arg = nv.segment_minmax(src, level.indices, level.idx_ptr, level.n, is_max)
        ctx.save_for_backward(arg)
        ctx.n_src = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        return nv.segment_minmax_bwd(dout.contiguous(), arg, ctx.n_src), None, None


def segment_minmax(src, level, is_max):
    """reduce="min" / "max" of torch_scatter.segment_csr over each cluster of `level`; the gradient goes to the row that
    attained the extremum (first one on ties)."""
    return _SegmentMinMax.apply(src, level, is_max)


def segment_mean(src, level, mean=True):
    """Pool rows of the finer level into `level` (the coarser one): mean (or sum) over each cluster."""
    return _SegmentMean.apply(src, level, mean)


class _UnpoolAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, up, level):
        ctx.level = level
        return nv.gather_add_rows(skip.contiguous(), up.contiguous(), level.cluster)

    @staticmethod
    def backward(ctx, g):
        lv = ctx.level
        g = g.contiguous()
        return g, nv.segment_reduce(g, lv.indices, lv.idx_ptr, lv.n, False), None


def unpool_add(skip, up, level):
    """skip (n_fine, C) + up (n_coarse, C)[pooling_inverse]; `level` is the coarse level."""
    return _UnpoolAdd.apply(skip, up, level)


class _SubMConv3d(torch.autograd.Function):
    """Per-tap gather (HIP) + dense GEMM accumulate (hipBLASLt through torch.addmm_).
    Rulebook symmetry (nbr[i][t]=j <=> nbr[j][T-1-t]=i) turns dgrad into the same gather form;
    it holds when voxels are unique per batch element.  With duplicate voxels (Mix3D batches)
    dgrad falls back to an index_add scatter so the gradient stays exact."""

    @staticmethod
    def forward(ctx, feat, weight, bias, nbr, has_dup, compute_dtype):
        taps, n = nbr.shape
        cout = weight.shape[0]
        x = feat.to(compute_dtype).contiguous()
        w = weight.reshape(cout, taps, -1).to(compute_dtype)
        out = torch.zeros((n, cout), dtype=compute_dtype, device=feat.device) if bias is None else \
            bias.to(compute_dtype).unsqueeze(0).repeat(n, 1)
        buf = torch.empty_like(x)
        for t in range(taps):
            nv.gather_rows(x, nbr[t], out=buf)
            out.addmm_(buf, w[:, t, :].t())
        ctx.save_for_backward(x, weight, nbr)
        ctx.has_bias, ctx.has_dup, ctx.compute_dtype, ctx.in_dtype = bias is not None, has_dup, compute_dtype, feat.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, nbr = ctx.saved_tensors
        taps, n = nbr.shape
        cout = weight.shape[0]
        cd = ctx.compute_dtype
        g = dout.to(cd).contiguous()
        w = weight.reshape(cout, taps, -1).to(cd)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.zeros_like(x)
            if not ctx.has_dup:
                buf = torch.empty_like(g)
                for t in range(taps):
                    nv.gather_rows(g, nbr[t], out=buf)
                    dx.addmm_(buf, w[:, taps - 1 - t, :])
            else:
                for t in range(taps):
                    rows = torch.nonzero(nbr[t] >= 0, as_tuple=True)[0]
                    if rows.numel():
                        dx.index_add_(0, nbr[t][rows].long(), g[rows] @ w[:, t, :])
            dx = dx.to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            dw = torch.empty((taps, cout, x.shape[1]), dtype=cd, device=x.device)
            buf = torch.empty_like(x)
            gt = g.t()
            for t in range(taps):
                nv.gather_rows(x, nbr[t], out=buf)
                torch.mm(gt, buf, out=dw[t])
            dw = dw.permute(1, 0, 2).reshape(weight.shape).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g.sum(0).to(weight.dtype)
        return dx, dw, db, None, None, None, None


class _SubMConv3dFused(torch.autograd.Function):
    """bf16 MFMA implicit-GEMM path (csrc/subm_conv.hip): one launch each for forward, dgrad
    (same kernel, tap-mirrored transposed weights) and wgrad; duplicate voxels handled exactly."""

    @staticmethod
    def forward(ctx, feat, weight, bias, nbr, rowperm, blocks_fn, has_dup, walk_fn=None, dup_fn=None):
        taps, n = nbr.shape
        cout, cin = weight.shape[0], weight.shape[-1]
        pad = (-cin) % 8
        ctx.dup_fn = dup_fn
        x = feat.to(torch.bfloat16)
        w = bf16_of(weight).reshape(cout, taps, cin)
        if pad:
            x = torch.nn.functional.pad(x, (0, pad)); w = torch.nn.functional.pad(w, (0, pad))
        x, w = x.contiguous(), w.contiguous()
        ctx.meta = (feat.dtype, weight.dtype, weight.shape, cin, bias is not None)
        ctx.blocks_fn, ctx.has_dup = blocks_fn, has_dup
        ctx.wt = mirrored_of(weight) if not pad else None      # dgrad weight kept beside the shadow (functional.register_mirrored)
        ctx.im2col = n <= CONV_IM2COL_MAX_SITES
        # bf16 out under autocast (the next op is a bf16 GEMM); outside autocast (the evaluator's call form) the output
        # keeps the input's dtype so that the fp32 Linear that follows sees what the reference's fp32 conv would hand it
        out_dtype = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        if ctx.im2col:
            # small level: 26 tiles for 256 CUs and a serial 27-tap loop made the implicit-GEMM kernel latency-bound
            # (70-90 us); neighbour rows side by side + ONE long-K library GEMM takes ~30 us
            cols = nv.subm_im2col(x, nbr)
            out = torch.nn.functional.linear(cols, w.view(cout, -1), bf16_of(bias))
            ctx.save_for_backward(cols, w, nbr, rowperm)
            return out.to(out_dtype)
        # the pipeline kernel (wide, large levels) reads its rulebook slice in walk order when the level has one (plan.neighbors_walk)
        walk = None
        if CONV_WALK_RULEBOOK and walk_fn is not None and rowperm is not None and nv.subm_conv_fwd_uses_pipe(n, cin + pad, cout, taps):
            walk = walk_fn()
        ctx.has_walk = walk is not None
        out = nv.subm_conv_fwd(x, w, None if bias is None else bias.float().contiguous(), nbr, rowperm, out_dtype, nbr_walk=walk)
        ctx.save_for_backward(x, w, nbr, rowperm, *(() if walk is None else (walk,)))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, nbr, rowperm = ctx.saved_tensors[:4]
        walk = ctx.saved_tensors[4] if getattr(ctx, "has_walk", False) else None
        in_dtype, w_dtype, w_shape, cin, has_bias = ctx.meta
        g = dout.to(torch.bfloat16).contiguous()
        dx = dw = db = None
        # duplicate voxels (Mix3D batches): every site at a voxel reads the voxel's WINNER row, so the adjoint first folds the
        # gradients of all duplicates onto their winner (ss_dup_fold_rows), runs the symmetric gather on that, and leaves the
        # non-winner rows (which nobody reads) with zero gradient
        runs = gd = None
        if ctx.has_dup and ctx.needs_input_grad[0]:
            runs = ctx.dup_fn() if ctx.dup_fn is not None else nv.dup_runs_from_rulebook(nbr)
            gd = nv.dup_fold_rows(g, runs)
        if ctx.im2col:
            cols = x                                               # saved im2col(x): (n, taps * cin_padded)
            taps = nbr.shape[0]
            if ctx.needs_input_grad[0]:
                wt = ctx.wt if ctx.wt is not None else nv.subm_weight_mirror(w)   # [ci][t'][co] = w[co][T-1-t'][ci]
                dx = torch.nn.functional.linear(nv.subm_im2col(g if gd is None else gd, nbr), wt.view(wt.shape[0], -1))
                if runs is not None:
                    nv.dup_zero_rows_(dx, runs)
                dx = dx[:, :cin].to(in_dtype)
            if ctx.needs_input_grad[1]:
                dw = _mm_f32(g.t(), cols).view(w.shape)[:, :, :cin].reshape(w_shape).to(w_dtype)
            if has_bias and ctx.needs_input_grad[2]:
                db = g.sum(0, dtype=torch.float32).to(w_dtype)
            return dx, dw, db, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            wt = ctx.wt if ctx.wt is not None else nv.subm_weight_mirror(w)       # [ci][t'][co] = w[co][T-1-t'][ci]
            dx = nv.subm_conv_fwd(g if gd is None else gd, wt, None, nbr, rowperm, nbr_walk=walk)
            if runs is not None:
                nv.dup_zero_rows_(dx, runs)
            dx = dx[:, :cin].to(in_dtype)
        if ctx.needs_input_grad[1]:
            blocks = ctx.blocks_fn() if ctx.blocks_fn is not None else nv.subm_block_lists(nbr, rowperm)
            dw = nv.subm_conv_wgrad(x, g, nbr, rowperm, blocks, nbr_walk=walk)[:, :, :cin].reshape(w_shape).to(w_dtype)
        if has_bias and ctx.needs_input_grad[2]:
            db = g.sum(0, dtype=torch.float32).to(w_dtype)
        return dx, dw, db, None, None, None, None, None, None


def _split_bf16(t):
    """t (fp32) -> (hi, lo) bf16 with hi + lo = t to ~2^-17 relative (hi = RNE(t), lo = RNE(t - hi))."""
    t = t.float()
    hi = t.to(torch.bfloat16)
    return hi, (t - hi.float()).to(torch.bfloat16)


class _SubMConv3dSplit(torch.autograd.Function):
    """Submanifold conv at (near) fp32 precision ON THE MFMA KERNELS: the reference keeps this one op in fp32 under AMP
    (pointcept/models/modules.py:64-75); bf16 operands cost ~2e-5 of per-Gaussian cosine distance on the full model, the
    per-tap fp32 path is 8x slower.  Both operands are split into bf16 hi + lo parts and the product is evaluated as
    xh wh + xl wh + xh wl (the dropped xl wl term is ~2^-18 relative) -- as ONE fused launch on channel-concatenated operands
    [xh | xl | xh] x [wh | wh | wl], fp32 accumulate, fp32 out.  dgrad uses the same form on (g, mirrored w); wgrad is
    (xh + xl) (x) gh + xh (x) gl.  3x the conv FLOPs of the bf16 mode."""

    @staticmethod
    def forward(ctx, feat, weight, bias, nbr, rowperm, blocks_fn, has_dup=False, dup_fn=None):
        taps, n = nbr.shape
        cout, cin = weight.shape[0], weight.shape[-1]
        pad = (-cin) % 8
        ctx.has_dup, ctx.dup_fn = has_dup, dup_fn
        x = feat.float()
        w = weight.float().reshape(cout, taps, cin)
        if pad:
            x = torch.nn.functional.pad(x, (0, pad)); w = torch.nn.functional.pad(w, (0, pad))
        xh, xl = _split_bf16(x)
        wh, wl = _split_bf16(w)
        x3 = torch.cat([xh, xl, xh], 1).contiguous()
        w3 = torch.cat([wh, wh, wl], 2).contiguous()
        out = nv.subm_conv_fwd(x3, w3, None if bias is None else bias.float().contiguous(), nbr, rowperm, torch.float32)
        ctx.save_for_backward(xh, xl, wh, wl, nbr, rowperm)
        ctx.meta = (feat.dtype, weight.dtype, weight.shape, cin, bias is not None)
        ctx.blocks_fn = blocks_fn
        return out

    @staticmethod
    def backward(ctx, dout):
        xh, xl, wh, wl, nbr, rowperm = ctx.saved_tensors
        in_dtype, w_dtype, w_shape, cin, has_bias = ctx.meta
        gh, gl = _split_bf16(dout.contiguous())
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt3 = torch.cat([nv.subm_weight_mirror(wh), nv.subm_weight_mirror(wh), nv.subm_weight_mirror(wl)], 2).contiguous()
            runs = None
            fh, fl = gh, gl
            if ctx.has_dup:      # duplicate voxels: fold the gradients of a voxel's sites onto its winner (fp32), then split
                runs = ctx.dup_fn() if ctx.dup_fn is not None else nv.dup_runs_from_rulebook(nbr)
                fh, fl = _split_bf16(nv.dup_fold_rows(dout.float().contiguous(), runs))
            g3 = torch.cat([fh, fl, fh], 1).contiguous()
            dx = nv.subm_conv_fwd(g3, wt3, None, nbr, rowperm, torch.float32)
            if runs is not None:
                nv.dup_zero_rows_(dx, runs)
            dx = dx[:, :cin].to(in_dtype)
        if ctx.needs_input_grad[1]:
            blocks = ctx.blocks_fn() if ctx.blocks_fn is not None else nv.subm_block_lists(nbr, rowperm)
            cp = xh.shape[1]
            d2 = nv.subm_conv_wgrad(torch.cat([xh, xl], 1).contiguous(), gh, nbr, rowperm, blocks)      # (cout, taps, 2 cp)
            d1 = nv.subm_conv_wgrad(xh, gl, nbr, rowperm, blocks)
            dw = (d2[:, :, :cp] + d2[:, :, cp:] + d1)[:, :, :cin].reshape(w_shape).to(w_dtype)
        if has_bias and ctx.needs_input_grad[2]:
            db = dout.sum(0, dtype=torch.float32).to(w_dtype)
        return dx, dw, db, None, None, None, None, None


class _SubMConv3dF32(torch.autograd.Function):
    """The 32-channel first stage in EXACT fp32 on the matrix cores (csrc/subm_f32.hip, v_mfma_f32_32x32x2_f32): what the reference
    computes for this op under AMP (spconv in fp32, pointcept/models/modules.py:64-75) without the 3x products and the hi/lo operand
    copies of _SubMConv3dSplit.  Eligible: 32 output channels, at most 32 input channels; the input gradient needs 32 input channels
    (the stem's input is data and takes none).  Duplicate voxels (Mix3D batches, datasets/utils.py:43-47 -- 80 % of the reference's
    training steps): the forward and the weight gradient read winner rows through the rulebook as they are; the input gradient
    folds the output gradients of a voxel's sites onto its winner first (ss_dup_fold_rows) and the kernel writes zeros to the
    other rows (ss_subm_f32_dgrad_dup)."""

    @staticmethod
    def forward(ctx, feat, weight, bias, nbr, rowperm, blocks_fn, walk_fn, has_dup=False, dup_fn=None):
        taps, n = nbr.shape
        cout, cin = weight.shape[0], weight.shape[-1]
        cp = 16 if cin <= 16 else 32
        ctx.has_dup, ctx.dup_fn, ctx.plain_nbr = has_dup, dup_fn, nbr
        ctx.orig_nbr = nbr if blocks_fn is None else None
        nbr = walk_fn() if walk_fn is not None else nv.subm_walk_rulebook(nbr, rowperm)     # walk order from here on
        x = feat.float()
        if cp != cin:
            x = torch.nn.functional.pad(x, (0, cp - cin))
        x = x.contiguous()
        w = weight.float().reshape(cout, taps, cin)
        out = nv.subm_f32_fwd(x, nv.subm_f32_weight_layout(w), None if bias is None else bias.float().contiguous(), nbr, rowperm)
        ctx.save_for_backward(x, w, nbr, rowperm)
        ctx.meta = (feat.dtype, weight.dtype, weight.shape, cin, bias is not None)
        ctx.blocks_fn = blocks_fn
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, nbr, rowperm = ctx.saved_tensors
        in_dtype, w_dtype, w_shape, cin, has_bias = ctx.meta
        g = dout.float().contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.has_dup:
                runs = ctx.dup_fn() if ctx.dup_fn is not None else nv.dup_runs_from_rulebook(ctx.plain_nbr)
                dx = nv.subm_f32_fwd(nv.dup_fold_rows(g, runs), nv.subm_f32_weight_layout(w, mirror=True), None, nbr, rowperm,
                                     winners_only=True).to(in_dtype)
            else:
                dx = nv.subm_f32_fwd(g, nv.subm_f32_weight_layout(w, mirror=True), None, nbr, rowperm).to(in_dtype)
        if ctx.needs_input_grad[1]:
            blocks = ctx.blocks_fn() if ctx.blocks_fn is not None else nv.subm_block_lists(ctx.orig_nbr, rowperm)
            dw = nv.subm_f32_wgrad(x, g, nbr, rowperm, blocks, cin).reshape(w_shape).to(w_dtype)
        if has_bias and ctx.needs_input_grad[2]:
            db = g.sum(0, dtype=torch.float32).to(w_dtype)
        return dx, dw, db, None, None, None, None, None, None


CONV_WALK_RULEBOOK = os.environ.get("SS_CONV_WALK", "1") != "0"    # 0: the pipeline conv kernels read the plain rulebook (A/B: scripts/ab_step.py conv_walk)
CONV_F32_MFMA = os.environ.get("SS_CONV_F32_MFMA", "1") != "0"      # A/B switch against the bf16x3 split (scripts/ab_step.py)
CONV_IM2COL_MAX_SITES = int(os.environ.get("SS_CONV_IM2COL_MAX", "8192"))


def _mm_f32(a, b):
    """bf16 x bf16 -> fp32 (fp32 accumulate, no bf16 rounding of the result) where the backend has aten::mm.dtype."""
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except (RuntimeError, NotImplementedError, TypeError):
        return torch.mm(a, b).float()


def subm_conv3d(feat, weight, bias, nbr, has_dup=False, compute_dtype=torch.float32, rowperm=None, blocks_fn=None, walk_fn=None,
                dup_fn=None):
    """weight (Cout, k, k, k, Cin) as in the reference checkpoints; nbr (k^3, n) tap-major.
    compute_dtype: torch.bfloat16 -> fused MFMA kernels on bf16 operands; "bf16x3" -> the reference's fp32 precision for this op on the
    matrix cores: exact fp32 MFMA for the 32-channel stage (walk_fn: the level's cached walk-order rulebook), else the bf16
    kernels on hi/lo-split operands; torch.float32 -> per-tap gather + fp32 GEMM.
    has_dup: the level holds duplicate voxels (Mix3D batches); dup_fn() -> (sorted keys, order) of the level (plan.Level.dup_runs),
    derived from the rulebook when absent.  Every MFMA path handles duplicates itself (no per-tap detour)."""
    if compute_dtype == torch.bfloat16 and weight.shape[0] % 8 == 0:
        return _SubMConv3dFused.apply(feat, weight, bias, nbr, rowperm, blocks_fn, has_dup, walk_fn, dup_fn)
    if compute_dtype == "bf16x3":
        if (CONV_F32_MFMA and weight.shape[0] == 32 and weight.shape[-1] <= 32 and feat.is_cuda
                and (weight.shape[-1] == 32 or not feat.requires_grad)):
            return _SubMConv3dF32.apply(feat, weight, bias, nbr, rowperm, blocks_fn, walk_fn, has_dup, dup_fn)
        if weight.shape[0] % 8 == 0 and feat.is_cuda:
            return _SubMConv3dSplit.apply(feat, weight, bias, nbr, rowperm, blocks_fn, has_dup, dup_fn)
        compute_dtype = torch.float32            # odd widths: the per-tap fp32 path
    return _SubMConv3d.apply(feat, weight, bias, nbr, has_dup, compute_dtype)


class _Gelu(torch.autograd.Function):
    """nn.GELU() (exact erf) of the MLP (ptv3:225-248) on the HIP kernel pair of csrc/norm.hip: the fp32 formulas torch evaluates,
    one rounding to the storage type."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return nv.gelu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous().to(x.dtype)
        if not _aligned16(dy):
            dy = dy.clone()
        return nv.gelu(x, dy)


GELU_HIP = os.environ.get("SS_GELU_HIP", "1") != "0"       # 0: torch's elementwise GELU kernels (A/B: scripts/ab_step.py gelu)


def _aligned16(t):
    return t.data_ptr() % 16 == 0 and t.storage_offset() * t.element_size() % 16 == 0


def gelu(x):
    # (ss_gelu moves 16-byte lanes: a contiguous slice at an odd storage offset takes torch's kernel instead of an argument error)
    if GELU_HIP and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and _aligned16(x):
        return _Gelu.apply(x)
    return torch.nn.functional.gelu(x)


def lang_head_release():
    """Drop the remembered sums (and with them the autograd graph they hold) once the criteria have run."""
    _HEAD_CACHE.clear()


class _LangHead(torch.autograd.Function):
    """Fused distillation head (csrc/head.hip): (feat, target, mask) -> (p, sums) with p = normalize(feat) (or feat) and
    sums = [sum_valid (1 - cos(p, t)), sum_valid |p - t|^2, #valid]; one kernel each way."""

    @staticmethod
    def forward(ctx, feat, target, mask, normalize, want_p):
        feat = feat.contiguous()
        if target is not None:
            target = target.contiguous()
            if target.dtype not in (torch.float32, torch.bfloat16):
                target = target.float()
        p, sums, rowstat = nv.lang_head_fwd(feat, target, mask, normalize, want_p=want_p)
        ctx.save_for_backward(feat, target, mask, rowstat)
        ctx.normalize = normalize
        ctx.set_materialize_grads(False)
        return p, sums

    @staticmethod
    def backward(ctx, dp, dsums):
        feat, target, mask, rowstat = ctx.saved_tensors
        if dp is None and (dsums is None or target is None):
            return None, None, None, None, None
        coef = None
        if target is not None:
            coef = (dsums[:2] if dsums is not None else torch.zeros(2, device=feat.device)).float().contiguous()
        dp = dp.contiguous() if dp is not None else None
        return nv.lang_head_bwd(feat, target, mask, ctx.normalize, rowstat, coef, dp), None, None, None, None


_HEAD_CACHE = {}


def lang_head(feat, target=None, mask=None, normalize=True):
    """-> (p, sums).  The result is remembered by identity of (p, target, mask): CosineSimilarity / L2Loss called on that
    p with that target / mask read their sums from it instead of making their own passes (lang_head_sums)."""
    p, sums = _LangHead.apply(feat, target, mask, normalize, True)
    if target is not None:
        _HEAD_CACHE["last"] = (weakref.ref(p), weakref.ref(target), weakref.ref(mask), target._version, mask._version, sums)
    return p, sums


def lang_head_sums(pred, target, mask):
    """sums for (pred, target, mask): the fused head's when pred came out of lang_head() with the same target / mask
    (criteria fan-out of LangPretrainer), else one un-normalised pass -- shared by both losses through the same cache."""
    ent = _HEAD_CACHE.get("last")
    if ent is not None and ent[0]() is pred and ent[1]() is target and ent[2]() is mask and ent[3] == target._version \
            and ent[4] == mask._version:
        return ent[5]
    _, sums = _LangHead.apply(pred, target, mask, False, False)
    _HEAD_CACHE["last"] = (weakref.ref(pred), weakref.ref(target), weakref.ref(mask), target._version, mask._version, sums)
    return sums


# ---- bf16 shadows of fp32 parameters ------------------------------------------------------------------
# torch autocast casts every fp32 weight / bias with its own ~4 us kernel (PT-v3m1: ~250 launches per step).  A model
# can register its GEMM operands here and refresh ALL shadows with a few multi-tensor copies at the start of forward.
# Every shadow carries the (version counter, storage address) of the parameter it was cast from: bf16_of() hands a
# shadow out only while that stamp still matches, so a forward that nobody refreshed for (eval without autocast after
# an optimizer step, load_state_dict, RUNTIME["param_shadows"] = False) gets a fresh cast instead of stale weights,
# and refresh_shadows() is a no-op while no parameter changed -- a second forward before the backward of the first
# (LangPretrainer._chunked_forward in training, multi-view losses) does not overwrite tensors an autograd graph saved.
_SHADOW = WeakIdKeyDictionary()          # parameter -> [bf16 tensor, stamp]; keyed by identity (Tensor.__eq__ is elementwise)
_SHADOW_T = WeakIdKeyDictionary()        # parameter -> (in, out) bf16 copy (register_transposed)
_SHADOW_M = WeakIdKeyDictionary()        # conv parameter -> tap-mirrored (cin, taps, cout) bf16 copy (register_mirrored)


def _stamp(p):
    return (p._version, p.data_ptr())


def _shadow_current(ent, p):
    """Is the shadow entry current: written eagerly for this parameter version, or recorded inside the capture now running."""
    tag = ent[1]
    if tag is None:
        return False
    if tag[0] == "capture":
        return tag[1] != 0 and torch.cuda.is_current_stream_capturing() and tag[1] == nv.stream_capture_id()
    return tag == _stamp(p)


def register_shadows(params):
    """-> (sources, shadows) lists for refresh_shadows; idempotent per parameter."""
    src, dst = [], []
    for p in params:
        if p is None or p.dtype != torch.float32 or not p.is_cuda:
            continue
        ent = _SHADOW.get(p)
        if ent is None or ent[0].shape != p.shape or ent[0].device != p.device:
            ent = [torch.empty_like(p, dtype=torch.bfloat16), None]
            _SHADOW[p] = ent
        src.append(p); dst.append(ent[0])
    return src, dst


SHADOW_GROUP_CAST = os.environ.get("SS_SHADOW_GROUP_CAST", "1") != "0"     # 0: torch._foreach_copy_ (one copy kernel per tensor)


@torch.no_grad()
def refresh_shadows(src, dst):
    """Re-cast the parameters whose value changed since their shadow was written (all of them after an optimizer
    step: one multi-tensor copy; none on a repeated forward)."""
    todo_s, todo_d = [], []
    # inside a hipGraph capture every shadow is re-cast: the replayed step must pick up whatever the optimizer wrote
    # since the previous replay, and a copy skipped now would be missing from the graph for good
    everything = torch.cuda.is_current_stream_capturing()
    cap = ("capture", nv.stream_capture_id()) if everything else None
    for p, d in zip(src, dst):
        ent = _SHADOW.get(p)
        st = _stamp(p)
        if everything or ent is None or ent[0] is not d or ent[1] != st:
            todo_s.append(p); todo_d.append(d)
            if ent is not None and ent[0] is d:
                # while capturing the copy is only RECORDED (it runs at replay): the shadow is current for THIS capture only
                # (bf16_of / mirrored_of / transposed_of compare the capture's identity) and stale for everything else, so
                # that an eager step after a refused capture re-casts instead of trusting weights one optimizer step old
                ent[1] = cap if everything else st
    if todo_s:
        # one launch for all of them (torch._foreach_copy_ with a dtype change is one copy kernel PER TENSOR: 206 launches and
        # 1.2 ms per step on the lang-pretrain model); anything that is not a contiguous fp32 -> bf16 pair takes the torch path
        fast = [(s_, d_) for s_, d_ in zip(todo_s, todo_d) if s_.is_cuda and s_.dtype == torch.float32 and d_.dtype == torch.bfloat16
                and s_.is_contiguous() and d_.is_contiguous() and s_.numel() == d_.numel()
                and s_.data_ptr() % 16 == 0 and d_.data_ptr() % 16 == 0]      # (the group kernel moves 16-byte lanes from the tensor base)
        if SHADOW_GROUP_CAST and fast:
            nv.cast_bf16_group([a for a, _ in fast], [b for _, b in fast])
            if len(fast) != len(todo_s):
                done = {id(d_) for _, d_ in fast}
                rest = [(s_, d_) for s_, d_ in zip(todo_s, todo_d) if id(d_) not in done]
                torch._foreach_copy_([d_ for _, d_ in rest], [s_ for s_, _ in rest])
        else:
            torch._foreach_copy_(todo_d, todo_s)
        pairs = [(d, _SHADOW_T[p]) for p, d in zip(todo_s, todo_d) if p in _SHADOW_T]
        if pairs:
            nv.transpose16_group(pairs)       # the (in, out) copies the dgrad GEMM reads, refreshed with their shadows
        pairs = [(d.view(d.shape[0], -1, d.shape[-1]), _SHADOW_M[p]) for p, d in zip(todo_s, todo_d) if p in _SHADOW_M]
        if pairs:
            nv.subm_weight_mirror_group(pairs)   # the tap-mirrored (cin, taps, cout) copies the conv dgrad reads


def register_mirrored(weights):
    """SubMConv3d weights (cout, k, k, k, cin): keep the tap-mirrored transpose the dgrad conv reads beside the bf16 shadow, all of them
    rewritten by one launch when the shadows are re-cast (22 small launches per backward pass otherwise).  Widths that need padding
    (cin % 8) and the split-precision convs build theirs on the fly as before."""
    for p in weights:
        ent = _SHADOW.get(p)
        if ent is None or p.dim() != 5 or p.shape[-1] % 8:
            continue
        cout, cin = p.shape[0], p.shape[-1]
        taps = p.numel() // (cout * cin)
        m = _SHADOW_M.get(p)
        if m is None or m.shape != (cin, taps, cout) or m.device != p.device:
            _SHADOW_M[p] = torch.empty((cin, taps, cout), dtype=torch.bfloat16, device=p.device)
            ent[1] = None


def mirrored_of(p):
    """The mirrored bf16 copy of a registered conv weight while its shadow is current, else None."""
    if not isinstance(p, torch.nn.Parameter):
        return None
    m = _SHADOW_M.get(p)
    if m is None:
        return None
    ent = _SHADOW.get(p)
    return m if (ent is not None and _shadow_current(ent, p)) else None


def register_transposed(weights):
    """nn.Linear weights (out, in) whose dgrad dx = dy @ W should run in hipBLASLt's NT form: keep an (in, out) bf16 copy beside the
    registered shadow (12-17 % faster than the NN form at the wide full-resolution layers, bit-identical results).  Call after
    register_shadows and before the refresh that follows it."""
    for p in weights:
        ent = _SHADOW.get(p)
        if ent is None or p.dim() != 2:
            continue
        t = _SHADOW_T.get(p)
        if t is None or t.shape != (p.shape[1], p.shape[0]) or t.device != p.device:
            _SHADOW_T[p] = torch.empty((p.shape[1], p.shape[0]), dtype=torch.bfloat16, device=p.device)
            ent[1] = None                     # force the next refresh to write both copies


def bf16_t_of(p):
    """The transposed bf16 copy of a registered Linear weight while its shadow is current, else None."""
    if not isinstance(p, torch.nn.Parameter):
        return None
    t = _SHADOW_T.get(p)
    if t is None:
        return None
    ent = _SHADOW.get(p)
    return t if (ent is not None and _shadow_current(ent, p)) else None


def bf16_of(p):
    """bf16 copy of an fp32 operand: the registered shadow while it is current, otherwise a fresh cast."""
    if p is None:
        return None
    if p.dtype == torch.bfloat16:
        return p
    ent = _SHADOW.get(p) if isinstance(p, torch.nn.Parameter) else None
    if ent is not None and _shadow_current(ent, p):
        return ent[0]
    return p.to(torch.bfloat16)


# ---- grouped weight gradients of a stage ------------------------------------------------------------------------------
# On the pooled levels a Linear weight gradient is a 20-30 us, latency-bound launch on 12-20 workgroups, and a stage has
# 5 per block (86 such launches = 2.7 ms per step).  They do not feed the backward chain, so the blocks of a stage QUEUE
# them and ONE grouped launch (ss_linear_wgrad_group) computes them when the chain leaves the stage.  DDP-safe by
# construction: the stage's parameters enter the stage through an identity node (_StageParams) whose backward runs after
# every consumer inside the stage has returned -- it launches the group and only then hands the (already returned,
# zero-initialised, stream-ordered) gradient buffers on to AccumulateGrad and the bucket hooks.
# every stage is grouped: with the CUs shared out over the tiles of the whole group (ss_linear_wgrad_group_plan2) the full-resolution stage gains
# too (room-102400 40.4 -> 40.0 ms/step, uniform-102400 68.0 -> 65.6 against the round-2 cap of 32,768 rows); the queued dy tensors of a stage
# stay alive until its end (3 GB at dec0 for one chunk)
WGRAD_GROUP_MAX_ROWS = int(os.environ.get("SS_WGRAD_GROUP_MAX", str(1 << 22)))    # 0 disables grouping
_STAGE = {"cur": None, "route": {}}


# queued (x, dy) operands of a stage stay alive until its group is launched: 3 GB at dec0 for one 102,400-row chunk, 8x that for the
# B = 8 step of config 3.  Above this many queued bytes the stage launches what it holds (a smaller group) and goes on queueing.
WGRAD_GROUP_MAX_BYTES = int(float(os.environ.get("SS_WGRAD_GROUP_MAX_GIB", "8")) * (1 << 30))


class _WgradStage:
    def __init__(self):
        self.queue = []          # (x, dy, dW, db): nn.Linear weight gradients
        self.redq = []           # (part (K, nb, C), dst (K, C)): LayerNorm dgamma / dbeta partial sums
        self.bytes = 0

    def add(self, x, dy, dw, db):
        self.queue.append((x, dy, dw, db))
        self.bytes += x.numel() * x.element_size() + dy.numel() * dy.element_size()
        if self.bytes > WGRAD_GROUP_MAX_BYTES:
            q, self.queue, self.bytes = self.queue, [], 0
            nv.linear_wgrad_group(q)

    def flush(self):
        q, self.queue, self.bytes = self.queue, [], 0
        r, self.redq = self.redq, []
        nv.linear_wgrad_group(q)
        nv.group_partial_sums(r)


class _StageParams(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stage, *params):
        ctx.stage = stage
        ctx.set_materialize_grads(False)
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    def backward(ctx, *grads):
        ctx.stage.flush()
        return (None,) + grads


def stage_begin(linears, n_rows):
    """Route the parameters of a stage's nn.Linear modules through one identity node (training under bf16 autocast only)."""
    _STAGE["cur"], _STAGE["route"] = None, {}
    if not (WGRAD_GROUP_MAX_ROWS and LINEAR_WGRAD_MIN_ROWS <= n_rows <= WGRAD_GROUP_MAX_ROWS and torch.is_grad_enabled()
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return
    params = [p for m in linears for p in (m.weight, m.bias) if p is not None and p.requires_grad and p.dtype == torch.float32 and p.is_cuda]
    # (the list holds the stage's nn.Linear AND nn.LayerNorm modules: both have .weight / .bias)
    if not params:
        return
    stage = _WgradStage()
    aliases = _StageParams.apply(stage, *params)
    _STAGE["cur"], _STAGE["route"] = stage, {id(p): a for p, a in zip(params, aliases)}


def stage_end():
    _STAGE["cur"], _STAGE["route"] = None, {}


def reset_state():
    """Forget everything a forward pass leaves between its calls (the open stage, queued weight gradients, the remembered
    head sums): called after a forward / backward that ended in an exception, e.g. a refused hipGraph capture."""
    stage_end()
    _HEAD_CACHE.clear()


def _routed(p):
    """(alias | p, stage | None) for a parameter that may be routed through the current stage's identity node."""
    a = _STAGE["route"].get(id(p)) if (p is not None and _STAGE["route"]) else None
    return (a, _STAGE["cur"]) if a is not None else (p, None)


class _Linear(torch.autograd.Function):
    """nn.Linear under bf16 autocast.  Forward and dgrad stay on hipBLASLt (NT / NN forms run at 0.9-1.3 PFLOP/s
    there); the weight gradient dy^T x -- K = sites, 0.25-0.7 PFLOP/s in hipBLASLt's TN form -- runs on the
    LDS-DMA pipeline kernel (csrc/wgrad8.hip) and lands in fp32 directly, without the bf16 -> fp32 grad cast.
    weight / bias are the parameters or their stage aliases (gradient routing only); w16 / b16 the bf16 operands."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, b16, stage, w16t=None):
        y = torch.nn.functional.linear(x, w16, b16)
        ctx.save_for_backward(x, w16 if w16t is None else w16t)
        ctx.meta = (weight.dtype, bias is not None)
        ctx.dgrad_nt = w16t is not None
        ctx.stage = stage
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        w_dtype, has_bias = ctx.meta
        dx, dw, db = _linear_backward(x, w16, ctx.dgrad_nt, w_dtype, has_bias, ctx.stage, dy.contiguous(), ctx.needs_input_grad[0],
                                      ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw, db, None, None, None, None


class _LinearGelu(torch.autograd.Function):
    """fc1 + nn.GELU() of the MLP (ptv3:225-248) as ONE autograd node: library GEMM, then the exact-erf GELU on the HIP kernel
    (csrc/norm.hip); backward = gelu' on the HIP kernel, then the Linear's own backward (_linear_backward).  One node instead of two: an
    eager step crosses Python once less per block and direction (a Python autograd Function costs ~50 us of host time per apply +
    backward pair; the eager variable-size line is host-bound)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, b16, stage, w16t=None):
        u = torch.nn.functional.linear(x, w16, b16)
        ctx.save_for_backward(x, w16 if w16t is None else w16t, u)
        ctx.meta = (weight.dtype, bias is not None)
        ctx.dgrad_nt = w16t is not None
        ctx.stage = stage
        return nv.gelu(u)

    @staticmethod
    def backward(ctx, dy):
        x, w16, u = ctx.saved_tensors
        w_dtype, has_bias = ctx.meta
        dy = dy.contiguous()
        if not _aligned16(dy):
            dy = dy.clone()                     # a fresh allocation is 16-byte aligned (ss_gelu's lanes)
        du = nv.gelu(u, dy)
        dx, dw, db = _linear_backward(x, w16, ctx.dgrad_nt, w_dtype, has_bias, ctx.stage, du, ctx.needs_input_grad[0],
                                      ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw, db, None, None, None, None


def _linear_backward(x, w16, dgrad_nt, w_dtype, has_bias, stage, dy, need_x, need_w, need_b):
    """dx, dW, db of y = x @ W.T + b (the backward of _Linear and of the fused qkv + attention function)."""
    dx = dw = db = None
    if need_x:
        # w16 is the (in, out) copy when the layer registered one: hipBLASLt's NT form (functional.register_transposed)
        dx = torch.nn.functional.linear(dy, w16) if dgrad_nt else dy @ w16
    want_db = has_bias and need_b
    if need_w:
        m, k = x.shape
        if m >= LINEAR_WGRAD_MIN_ROWS and nv.lib().ss_wgrad8_ok(m, k, dy.shape[1], 1):
            if stage is not None and w_dtype == torch.float32:
                # queued: the stage's identity node launches the whole group when the backward chain leaves the stage
                dw, db_ = nv.linear_wgrad_alloc(k, dy.shape[1], want_db, x.device)
                stage.add(x, dy, dw, db_)
                if want_db:
                    db, want_db = db_, False
            elif want_db:                    # column sums of dy ride along in the wgrad kernel
                dw, db = nv.linear_wgrad(x, dy, True)
                dw, db = dw.to(w_dtype), db.to(w_dtype)
                want_db = False
            else:
                dw = nv.linear_wgrad(x, dy).to(w_dtype)
        else:
            dw = _mm_f32(dy.t(), x).to(w_dtype)     # small levels: library GEMM, fp32 out where aten::mm.dtype exists
    if want_db:
        db = dy.sum(0, dtype=torch.float32).to(w_dtype)
    return dx, dw, db


LINEAR_WGRAD_MIN_ROWS = 1024     # in-process A/B on room-102400: 1024 beats 4096 by 0.5 ms/step


def linear(x, weight, bias=None):
    """torch.nn.functional.linear; under CUDA bf16 autocast on 2-D input the backward uses the pipeline wgrad kernel."""
    if x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16 \
            and weight.dtype == torch.float32:
        route = _STAGE["route"]
        wr = route.get(id(weight)) if route else None
        if wr is not None and torch.is_grad_enabled():
            br = route.get(id(bias)) if bias is not None else None
            return _Linear.apply(x.to(torch.bfloat16).contiguous(), wr, br if br is not None else bias, bf16_of(weight), bf16_of(bias),
                                 _STAGE["cur"], bf16_t_of(weight))
        return _Linear.apply(x.to(torch.bfloat16).contiguous(), weight, bias, bf16_of(weight), bf16_of(bias), None, bf16_t_of(weight))
    return torch.nn.functional.linear(x, weight, bias)


def linear_gelu(x, weight, bias=None):
    """gelu(linear(x)) with the exact erf form; under CUDA bf16 autocast on 2-D input one autograd node (_LinearGelu)."""
    if GELU_HIP and x.is_cuda and x.dim() == 2 and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16 \
            and weight.dtype == torch.float32 and torch.is_grad_enabled():
        route = _STAGE["route"]
        wr = route.get(id(weight)) if route else None
        if wr is not None:
            br = route.get(id(bias)) if bias is not None else None
            return _LinearGelu.apply(x.to(torch.bfloat16).contiguous(), wr, br if br is not None else bias, bf16_of(weight), bf16_of(bias),
                                     _STAGE["cur"], bf16_t_of(weight))
        return _LinearGelu.apply(x.to(torch.bfloat16).contiguous(), weight, bias, bf16_of(weight), bf16_of(bias), None, bf16_t_of(weight))
    return gelu(linear(x, weight, bias))


class _QkvWindowAttentionHM(torch.autograd.Function):
    """qkv projection + serialized-window attention on the head-major layout (csrc/attention_hm.hip; ptv3:172-216).
    Forward: the projection writes q / k / v head-major and window-ordered from its own epilogue (gemm8.hip; narrow levels:
    fp32 library GEMM + ss_headmajor_pack, the same single rounding), the attention kernels stage contiguous tiles by LDS-DMA.
    Backward: dqkv (n, 3C) in memory row order, then exactly _Linear's dgrad / wgrad paths."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, stage, w16t, win, num_heads, scale):
        sec0 = float(scale) * nv.LOG2E
        b32 = bias.detach() if bias is not None else None
        if nv.headmajor_eligible(x.shape[1]) and x.shape[0] >= HM_FUSED_MIN_ROWS and x.shape[1] >= HM_FUSED_MIN_CHANNELS:
            hm = nv.linear_fwd_headmajor(x, win, w16, b32.float() if b32 is not None else None, num_heads, sec0)
        else:
            qkv = _mm_f32(x, w16.t())
            if b32 is not None:
                qkv = qkv + b32.float()
            hm = nv.headmajor_pack(qkv.contiguous(), win, num_heads, 3, sec0)
        out, nlse2 = nv.window_attn_hm_fwd(hm, win, num_heads)
        ctx.save_for_backward(x, w16 if w16t is None else w16t, hm, out, nlse2)
        ctx.meta = (weight.dtype, bias is not None)
        ctx.dgrad_nt = w16t is not None
        ctx.stage, ctx.win, ctx.num_heads, ctx.scale = stage, win, num_heads, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w16, hm, out, nlse2 = ctx.saved_tensors
        w_dtype, has_bias = ctx.meta
        dqkv = nv.window_attn_hm_bwd(hm, out, dout.contiguous().to(torch.bfloat16), nlse2, ctx.win, ctx.num_heads, ctx.scale)
        dx, dw, db = _linear_backward(x, w16, ctx.dgrad_nt, w_dtype, has_bias, ctx.stage, dqkv, ctx.needs_input_grad[0],
                                      ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw, db, None, None, None, None, None, None


# below these: fp32 library GEMM + bias add + pack kernel (3 launches).  Round 4: 1,024 rows / 64 channels instead of 4,096 / 128 -- the
# fused epilogue also on enc3 (1,600 x 256, 6 blocks) and enc1 (25,600 x 64): two launches fewer per block, -0.08 ms each in the
# in-process A/B (scripts/ab_step.py hm_rows | hm_ch: 37.10 vs 37.18, 36.94 vs 37.02 ms/step)
HM_FUSED_MIN_ROWS = int(os.environ.get("SS_HM_FUSED_MIN_ROWS", "1024"))
HM_FUSED_MIN_CHANNELS = int(os.environ.get("SS_HM_FUSED_MIN_CHANNELS", "64"))


def qkv_window_attention(x, weight, bias, win, num_heads, scale):
    """SerializedAttention's qkv Linear + attention (ptv3:172-216) under CUDA bf16 autocast, MFMA path: (n, C) -> (n, C)."""
    route = _STAGE["route"]
    wr = route.get(id(weight)) if (route and torch.is_grad_enabled()) else None
    br = route.get(id(bias)) if (wr is not None and bias is not None) else None
    stage = _STAGE["cur"] if wr is not None else None
    return _QkvWindowAttentionHM.apply(x.to(torch.bfloat16).contiguous(), wr if wr is not None else weight,
                                       br if br is not None else bias, bf16_of(weight), stage, bf16_t_of(weight), win, num_heads,
                                       scale)


def window_attention_hm(qkv, win, num_heads, scale):
    """The head-major kernels behind the (n, 3C) interface of window_attention (tests, narrow callers): pack + attention."""
    return _WindowAttentionHM.apply(qkv, win, num_heads, scale)


class _WindowAttentionHM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, win, num_heads, scale):
        hm = nv.headmajor_pack(qkv.contiguous(), win, num_heads, 3, float(scale) * nv.LOG2E)
        out, nlse2 = nv.window_attn_hm_fwd(hm, win, num_heads)
        ctx.save_for_backward(hm, out, nlse2)
        ctx.win, ctx.num_heads, ctx.scale, ctx.in_dtype = win, num_heads, scale, qkv.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        hm, out, nlse2 = ctx.saved_tensors
        dqkv = nv.window_attn_hm_bwd(hm, out, dout.contiguous().to(torch.bfloat16), nlse2, ctx.win, ctx.num_heads, ctx.scale)
        return dqkv.to(ctx.in_dtype), None, None, None


class _LayerNorm(torch.autograd.Function):
    """h = LN(x) in one pass, bf16 or fp32 out (csrc/norm.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype):
        x = x.contiguous()
        _, _, h, mean, rstd = nv.add_layernorm_fwd(x, None, None, gamma.float().contiguous(), beta.float().contiguous(), eps,
                                                   False, False, out_dtype)
        ctx.save_for_backward(x, mean, rstd, gamma)
        return h

    @staticmethod
    def backward(ctx, g_h):
        x, mean, rstd, gamma = ctx.saved_tensors
        g_x, _, dg, db = nv.add_layernorm_bwd(None, None, g_h.contiguous(), x, mean, rstd, gamma.float().contiguous(), None,
                                              x.dtype, None)
        return g_x, dg.to(gamma.dtype), db.to(gamma.dtype), None, None


def layer_norm(x, gamma, beta, eps=1e-5, out_dtype=None):
    return _LayerNorm.apply(x, gamma, beta, eps, out_dtype or x.dtype)


class _AddLayerNorm(torch.autograd.Function):
    """(x, y) -> xout = x + rowscale*y [fp32], h = LN(xout) [optional], xcopy = bf16(xout) [optional].  stage: the
    _WgradStage whose grouped launch reduces this seam's dgamma / dbeta partials (None: reduced here)."""

    @staticmethod
    def forward(ctx, x, y, rowscale, gamma, beta, eps, want_copy, h_dtype, stage):
        x, y = x.contiguous(), y.contiguous()
        g32 = gamma.float().contiguous() if gamma is not None else None
        b32 = beta.float().contiguous() if beta is not None else None
        xout, xcopy, h, mean, rstd = nv.add_layernorm_fwd(x, y, rowscale, g32, b32, eps, True, want_copy, h_dtype)
        ctx.save_for_backward(xout, mean, rstd, g32, rowscale)
        ctx.meta = (x.dtype, y.dtype, gamma.dtype if gamma is not None else None)
        ctx.stage = stage
        ctx.set_materialize_grads(False)
        return xout, h, xcopy

    @staticmethod
    def backward(ctx, g_xout, g_h, g_xcopy):
        xout, mean, rstd, g32, rowscale = ctx.saved_tensors
        x_dt, y_dt, g_dt = ctx.meta
        g_xout = g_xout.contiguous() if g_xout is not None else None
        g_h = g_h.contiguous() if g_h is not None else None
        g_xcopy = g_xcopy.contiguous() if g_xcopy is not None else None
        if g_xout is None and g_h is None and g_xcopy is None:
            return (None,) * 9
        if ctx.stage is not None and g_dt == torch.float32:
            g_x, g_y, part, _ = nv.add_layernorm_bwd(g_xout, g_xcopy, g_h, xout, mean, rstd, g32, rowscale, x_dt, y_dt, reduce=False)
            dg = db = None
            if part is not None:
                dst = torch.empty((2, part.shape[2]), dtype=torch.float32, device=part.device)
                ctx.stage.redq.append((part, dst))
                dg, db = dst[0], dst[1]
            return g_x, g_y, None, dg, db, None, None, None, None
        g_x, g_y, dg, db = nv.add_layernorm_bwd(g_xout, g_xcopy, g_h, xout, mean, rstd, g32, rowscale, x_dt, y_dt)
        return (g_x, g_y, None, dg.to(g_dt) if dg is not None else None, db.to(g_dt) if db is not None else None,
                None, None, None, None)


class _LnAddLn(torch.autograd.Function):
    """First seam of a pre-norm Block: (x, t) -> xout = x + LN0(t) [fp32], h = LN1(xout); one kernel each way."""

    @staticmethod
    def forward(ctx, x, t, g0, b0, eps0, g1, b1, eps1, h_dtype, stage):
        x, t = x.contiguous(), t.contiguous()
        p32 = [p.float().contiguous() for p in (g0, b0, g1, b1)]
        xout, h, stats = nv.ln_add_ln_fwd(x, t, p32[0], p32[1], eps0, p32[2], p32[3], eps1, h_dtype)
        ctx.save_for_backward(xout, t, stats, p32[0], p32[2])
        ctx.meta = (x.dtype, t.dtype, g0.dtype)
        ctx.stage = stage
        ctx.set_materialize_grads(False)
        return xout, h

    @staticmethod
    def backward(ctx, g_xout, g_h):
        xout, t, stats, g0, g1 = ctx.saved_tensors
        x_dt, t_dt, p_dt = ctx.meta
        if g_xout is None and g_h is None:
            return (None,) * 10
        g_xout = g_xout.float().contiguous() if g_xout is not None else None
        g_h = g_h.contiguous() if g_h is not None else None
        if ctx.stage is not None and p_dt == torch.float32:
            g_x, g_t, part = nv.ln_add_ln_bwd(g_xout, g_h, xout, t, stats, g0, g1, x_dt, t_dt, reduce=False)
            dst = torch.empty((4, part.shape[2]), dtype=torch.float32, device=part.device)
            ctx.stage.redq.append((part, dst))
            return g_x, g_t, dst[0], dst[1], None, dst[2], dst[3], None, None, None
        g_x, g_t, dg0, db0, dg1, db1 = nv.ln_add_ln_bwd(g_xout, g_h, xout, t, stats, g0, g1, x_dt, t_dt)
        return g_x, g_t, dg0.to(p_dt), db0.to(p_dt), None, dg1.to(p_dt), db1.to(p_dt), None, None, None


def ln_add_ln(x, t, ln0, ln1, h_dtype=torch.float32):
    """x + LayerNorm0(t) and LayerNorm1 of the sum, fused (ln0 / ln1: nn.LayerNorm modules with affine parameters)."""
    (g0, s0), (b0, s1), (g1, s2), (b1, s3) = _routed(ln0.weight), _routed(ln0.bias), _routed(ln1.weight), _routed(ln1.bias)
    stage = s0 if (s0 is not None and s0 is s1 and s0 is s2 and s0 is s3) else None
    if stage is None:
        g0, b0, g1, b1 = ln0.weight, ln0.bias, ln1.weight, ln1.bias
    return _LnAddLn.apply(x, t, g0, b0, ln0.eps, g1, b1, ln1.eps, h_dtype, stage)


def add_layer_norm(x, y, rowscale=None, gamma=None, beta=None, eps=1e-5, want_copy=False, h_dtype=torch.float32):
    """Fused residual seam: returns (xout fp32, h or None, bf16 copy of xout or None)."""
    (ga, s0), (be, s1) = _routed(gamma), _routed(beta)
    stage = s0 if (s0 is not None and s0 is s1) else None
    if stage is None:
        ga, be = gamma, beta
    return _AddLayerNorm.apply(x, y, rowscale, ga, be, eps, want_copy, h_dtype, stage)


class _BatchNormAct(torch.autograd.Function):
    """BatchNorm1d (training: batch statistics + running-stat update; eval: running statistics) fused with
    an optional exact GELU, 2 passes forward / 2 passes backward (csrc/norm.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches, training, momentum, eps, act):
        x = x.contiguous()
        n = x.shape[0]
        if training and running_mean.dtype == torch.float32 and running_var.dtype == torch.float32 and momentum is not None:
            # statistics pass + ONE finishing launch: batch mean / rstd, running-stat update and batch counter
            with torch.no_grad():
                mean, rstd = nv.bn_batch_stats(x, running_mean, running_var, num_batches, momentum, eps)
        else:
            if training:
                shift = running_mean.float().contiguous()              # conditioning of the one-pass variance
                s, q = nv.col_stats(x, shift)
                d = s / n
                mean = shift + d
                var = (q / n - d * d).clamp_(min=0.0)
                with torch.no_grad():
                    mom = momentum if momentum is not None else 1.0 / float(num_batches.item() + 1 if num_batches is not None else 1)
                    running_mean.mul_(1 - mom).add_(mean.to(running_mean.dtype), alpha=mom)
                    running_var.mul_(1 - mom).add_((var * (n / max(n - 1, 1))).to(running_var.dtype), alpha=mom)
                    if num_batches is not None:
                        num_batches.add_(1)
            else:
                mean, var = running_mean.float(), running_var.float()
            rstd = torch.rsqrt(var + eps).contiguous()
            mean = mean.contiguous()
        g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
        y = nv.bn_act_fwd(x, mean, rstd, g32, b32, act, x.dtype)
        ctx.save_for_backward(x, mean, rstd, g32, b32)
        ctx.meta = (training, act, gamma.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, g32, b32 = ctx.saved_tensors
        training, act, pdt = ctx.meta
        dx, dg, db = nv.bn_act_bwd(dy.contiguous(), x, mean, rstd, g32, b32, act, training)
        return dx, dg.to(pdt), db.to(pdt), None, None, None, None, None, None, None


def batch_norm_act(x, bn, act=False):
    """x (n, C) through an nn.BatchNorm1d's parameters / buffers (updates running stats in training)."""
    return _BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked if bn.training else None,
                               bn.training, bn.momentum, bn.eps, act)
