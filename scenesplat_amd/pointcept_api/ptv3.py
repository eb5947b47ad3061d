"""PT-v3m1 on the HIP hot path.

Same registry name, constructor kwargs, ``forward(data_dict) -> Point`` contract and
state-dict keys/shapes as the reference model
(pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:518-714; keys: SURVEY
Appendix D), so released checkpoints load unchanged.  The execution model is different: the
integer structure of the whole forward is planned once on the GPU (scenesplat_amd.plan) and
each Block runs fused HIP ops -- rulebook conv, serialized-window attention with the
[order]/[inverse] gathers folded into the kernel, CSR pooling -- around hipBLASLt GEMMs.

Replicated reference behaviours that look like accidents but are what the released weights
were trained with:
  * the first Block after every unpooling convolves the *stale* sparse features
    (proj_skip(skip) only): modules.py:64-75 + ptv3:476-478;
  * SerializedPooling shuffles the curve order with torch.randperm even when the model's
    shuffle_orders=False (ptv3:350,408-412,614-620);
  * tail windows are topped up with points borrowed from the previous window (ptv3:145-154).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as SF
from .. import native as nv
from ..plan import build_plan
from .registry import MODELS
from .structure import Point

# knobs of the execution (not of the model): attention kernel family and conv compute dtype
RUNTIME = dict(attn_impl=nv.ATTN_SIMT, conv_dtype=None,  # conv_dtype: None = fp32 per-tap path (reference), torch.bfloat16, or "bf16x3"
               attn_headmajor=os.environ.get("SS_ATTN_HM", "1") != "0",   # MFMA attention on the head-major layout (round 3)
               param_shadows=os.environ.get("SS_PARAM_SHADOWS", "1") != "0")   # bf16 weight shadows under autocast


def bench_runtime():
    """The execution knobs bench.py times (and tests/test_hip_prod.py holds to the north-star cosine bar): MFMA window
    attention; submanifold conv on bf16 operands -- except the 32-channel stage (stem + the two enc0 blocks), which runs on
    hi/lo-split operands ("bf16x3", the reference's fp32 precision for this op).  Rounding the conv operands of the FIRST
    stage is what cost the cosine budget (it propagates through all 22 blocks): split there, the training-mode maximum
    drops from 1.0e-4 to 7e-5 for +0.8 ms per step; splitting every stage up to 256 channels buys 0.5e-5 more for +3.8 ms
    (scripts/prec_probe.py --split-sweep).  All under torch bf16 autocast."""
    return dict(attn_impl=nv.ATTN_MFMA, conv_dtype=torch.bfloat16,
                conv_split_max_channels=int(os.environ.get("SS_CONV_SPLIT_MAX", "32")))


def conv_wants_walk(n_sites, channels, ksize):
    """Does a conv of this width -- an int (cin = cout) or a (cin, cout) pair -- on a level of n_sites read the walk-order rulebook?
    The forward's own decisions (functional.subm_conv3d): the exact-fp32 first-stage kernels always do, the bf16 path where the
    shape runs on the pipeline kernel."""
    cin, cout = channels if isinstance(channels, (tuple, list)) else (channels, channels)
    cd = conv_dtype_for(cout)
    if cd == "bf16x3":
        return bool(SF.CONV_F32_MFMA and cout == 32 and cin <= 32)
    if cd == torch.bfloat16 and cout % 8 == 0:
        return bool(SF.CONV_WALK_RULEBOOK and n_sites > SF.CONV_IM2COL_MAX_SITES
                    and nv.subm_conv_fwd_uses_pipe(n_sites, cin + (-cin) % 8, cout, ksize ** 3))
    return False


def conv_dtype_for(out_channels):
    """RUNTIME["conv_dtype"] for a conv of this width: bf16 convs up to conv_split_max_channels run as "bf16x3"."""
    cd = RUNTIME["conv_dtype"] or torch.float32
    if cd == torch.bfloat16 and out_channels <= RUNTIME.get("conv_split_max_channels", 0):
        return "bf16x3"
    return cd


class PointModule(nn.Module):
    """Marker base class, as pointcept/models/modules.py:8-14."""


class DropPath(nn.Module):
    """Per-row stochastic depth == timm.layers.DropPath on (N, C) features (ptv3:314-316)."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1.0 - self.p
        mask = x.new_empty((x.shape[0], 1)).bernoulli_(keep)
        if keep > 0.0:
            mask.div_(keep)
        return x * mask


class SubMConv3d(nn.Module):
    """Parameter holder with spconv's layout: weight (Cout, k, k, k, Cin), optional bias."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, indice_key=None):
        super().__init__()
        k = kernel_size
        self.kernel_size, self.indice_key = k, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=5 ** 0.5)

    def forward(self, feat, level):
        cd = conv_dtype_for(self.weight.shape[0])
        return SF.subm_conv3d(feat, self.weight, self.bias, level.neighbors(self.kernel_size),
                              level.has_duplicates, cd, level.conv_rowperm(),
                              lambda: level.conv_blocks(self.kernel_size), lambda: level.neighbors_walk(self.kernel_size),
                              level.dup_runs)


def _lin(mod, x):
    """nn.Linear through SF.linear (pipeline weight-gradient kernel under bf16 autocast)."""
    return SF.linear(x, mod.weight, mod.bias)


class SerializedAttention(PointModule):
    def __init__(self, channels, num_heads, patch_size, qkv_bias=True, qk_scale=None, attn_drop=0.0,
                 proj_drop=0.0, order_index=0, enable_rpe=False, enable_flash=True, upcast_attention=True,
                 upcast_softmax=True):
        super().__init__()
        assert channels % num_heads == 0
        if enable_rpe:
            raise NotImplementedError("enable_rpe is off in every SceneSplat language config; not on the HIP path")
        if attn_drop != 0.0 or proj_drop != 0.0:
            raise NotImplementedError("attention / projection dropout are 0 in the reference configs")
        self.channels, self.num_heads, self.patch_size = channels, num_heads, patch_size
        self.scale = qk_scale or (channels // num_heads) ** -0.5
        self.order_index = order_index
        self.qkv = nn.Linear(channels, channels * 3, bias=qkv_bias)
        self.proj = nn.Linear(channels, channels)

    def forward(self, x, level):
        win = level.window(self.order_index, self.patch_size)
        impl = RUNTIME["attn_impl"]
        if (impl == nv.ATTN_MFMA and RUNTIME.get("attn_headmajor", True) and x.is_cuda and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and self.qkv.weight.dtype == torch.float32
                and (self.channels // self.num_heads) in (16, 32, 48, 64)):
            # round 3: projection -> head-major, window-ordered q / k / v -> LDS-DMA attention kernels (csrc/attention_hm.hip)
            feat = SF.qkv_window_attention(x, self.qkv.weight, self.qkv.bias, win, self.num_heads, self.scale)
            return _lin(self.proj, feat)
        qkv = _lin(self.qkv, x)
        if impl == nv.ATTN_MFMA and qkv.dtype != torch.bfloat16:
            feat = SF.window_attention(qkv.to(torch.bfloat16), win, self.num_heads, self.scale, impl).to(qkv.dtype)
        else:
            feat = SF.window_attention(qkv, win, self.num_heads, self.scale, impl)
        return _lin(self.proj, feat)


class MLP(nn.Module):
    def __init__(self, in_channels, hidden_channels=None, out_channels=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_channels = out_channels or in_channels
        hidden_channels = hidden_channels or in_channels
        self.fc1 = nn.Linear(in_channels, hidden_channels)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_channels, out_channels)

    def forward(self, x):
        # nn.GELU() with the exact erf form runs on the HIP kernel pair (csrc/norm.hip), in one autograd node with fc1; any other
        # activation stays the module's own
        if type(self.act) is nn.GELU and getattr(self.act, "approximate", "none") == "none":
            h = SF.linear_gelu(x, self.fc1.weight, self.fc1.bias)
        else:
            h = self.act(_lin(self.fc1, x))
        return _lin(self.fc2, h)


class Block(PointModule):
    def __init__(self, channels, num_heads, patch_size=48, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, act_layer=nn.GELU,
                 pre_norm=True, order_index=0, cpe_indice_key=None, enable_rpe=False, enable_flash=True,
                 upcast_attention=True, upcast_softmax=True):
        super().__init__()
        self.channels, self.pre_norm = channels, pre_norm
        self.cpe = nn.Sequential(SubMConv3d(channels, channels, 3, bias=True, indice_key=cpe_indice_key),
                                 nn.Linear(channels, channels), norm_layer(channels))
        self.norm1 = nn.Sequential(norm_layer(channels))
        self.attn = SerializedAttention(channels, num_heads, patch_size, qkv_bias, qk_scale, attn_drop, proj_drop,
                                        order_index, enable_rpe, enable_flash, upcast_attention, upcast_softmax)
        self.norm2 = nn.Sequential(norm_layer(channels))
        self.mlp = nn.Sequential(MLP(channels, int(channels * mlp_ratio), channels, act_layer, proj_drop))
        self.drop_path = nn.Sequential(DropPath(drop_path) if drop_path > 0.0 else nn.Identity())

    def forward(self, x, conv_in, level, want_copy=False):
        """x: Point.feat; conv_in: sparse_conv_feat.features (differs from x only in the first decoder
        block of a stage).  Returns (x_out, bf16 copy of x_out or None).  The three residual seams run as
        fused add + DropPath-scale + LayerNorm kernels (csrc/norm.hip)."""
        if not self.pre_norm:
            return self._forward_post_norm(x, conv_in, level), None
        hdt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        ln0, ln1, ln2 = self.cpe[2], self.norm1[0], self.norm2[0]
        t = _lin(self.cpe[1], self.cpe[0](conv_in, level))
        if RUNTIME.get("fuse_ln_seam", True) and x.shape[1] % 4 == 0 and x.shape[1] <= 1024:
            x, h = SF.ln_add_ln(x, t, ln0, ln1, hdt)           # x += LN0(t); h = LN1(x): one pass
        else:
            t = SF.layer_norm(t, ln0.weight, ln0.bias, ln0.eps)
            x, h, _ = SF.add_layer_norm(x, t, None, ln1.weight, ln1.bias, ln1.eps, False, hdt)
        x, h, _ = SF.add_layer_norm(x, self.attn(h, level), self._row_scale(x), ln2.weight, ln2.bias, ln2.eps, False, hdt)
        x, _, xb = SF.add_layer_norm(x, self.mlp(h), self._row_scale(x), None, None, 0.0, want_copy, hdt)
        return x, xb

    def _row_scale(self, x):
        dp = self.drop_path[0]
        if not isinstance(dp, DropPath) or dp.p == 0.0 or not self.training:
            return None
        pre = self.__dict__.get("_row_scales")       # drawn for all blocks at once by PointTransformerV3.forward
        if pre:
            rs = pre.pop()
            if rs.shape[0] == x.shape[0]:
                return rs
        keep = 1.0 - dp.p
        return x.new_empty(x.shape[0], dtype=torch.float32).bernoulli_(keep).div_(keep)

    def _forward_post_norm(self, x, conv_in, level):
        c = self.cpe[0](conv_in, level)
        x = x + self.cpe[2](self.cpe[1](c))
        x = self.norm1(x + self.drop_path(self.attn(x, level)))
        return self.norm2(x + self.drop_path(self.mlp(x)))


def _norm_act(x, norm, act):
    """BatchNorm1d (+ GELU) as one fused HIP op (csrc/norm.hip); anything else falls through to the modules."""
    if isinstance(norm, nn.BatchNorm1d) and norm.affine and norm.track_running_stats and (act is None or isinstance(act, nn.GELU)) \
            and x.shape[1] % 4 == 0 and x.shape[1] <= 1024:
        return SF.batch_norm_act(x, norm, act is not None)
    if norm is not None:
        x = norm(x)
    return act(x) if act is not None else x


def _seq_lin_norm_act(seq, x):
    """nn.Sequential(Linear[, BatchNorm1d][, GELU]) with the norm/act pair fused."""
    mods = list(seq)
    x = _lin(mods[0], x) if isinstance(mods[0], nn.Linear) else mods[0](x)
    norm = mods[1] if len(mods) > 1 and isinstance(mods[1], nn.BatchNorm1d) else None
    act = mods[-1] if len(mods) > 1 and isinstance(mods[-1], nn.GELU) else None
    if norm is None and act is None:
        for mod in mods[1:]:
            x = mod(x)
        return x
    return _norm_act(x, norm, act)


class SerializedPooling(PointModule):
    def __init__(self, in_channels, out_channels, stride=2, norm_layer=None, act_layer=None, reduce="mean",
                 shuffle_orders=True, traceable=True):
        super().__init__()
        if reduce not in ("mean", "sum", "min", "max"):
            raise ValueError(f"unknown reduce {reduce!r} (torch_scatter.segment_csr: sum / mean / min / max)")
        self.reduce = reduce
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.shuffle_orders = shuffle_orders
        self.proj = nn.Linear(in_channels, out_channels)
        self.norm = nn.Sequential(norm_layer(out_channels)) if norm_layer is not None else None
        self.act = act_layer() if act_layer is not None else None

    def forward(self, x, coarse_level):
        if self.reduce in ("min", "max"):
            x = SF.segment_minmax(_lin(self.proj, x), coarse_level, self.reduce == "max")
        else:
            x = SF.segment_mean(_lin(self.proj, x), coarse_level, mean=(self.reduce == "mean"))
        return _norm_act(x, self.norm[0] if self.norm is not None else None, self.act)


class SerializedUnpooling(PointModule):
    def __init__(self, in_channels, skip_channels, out_channels, norm_layer=None, act_layer=None, traceable=False):
        super().__init__()
        self.proj = nn.Sequential(nn.Linear(in_channels, out_channels))
        self.proj_skip = nn.Sequential(nn.Linear(skip_channels, out_channels))
        if norm_layer is not None:
            self.proj.append(norm_layer(out_channels)); self.proj_skip.append(norm_layer(out_channels))
        if act_layer is not None:
            self.proj.append(act_layer()); self.proj_skip.append(act_layer())

    def forward(self, x, skip, coarse_level):
        """returns (parent.feat, parent.sparse_conv_feat.features)"""
        skip, up = _seq_lin_norm_act(self.proj_skip, skip), _seq_lin_norm_act(self.proj, x)
        return SF.unpool_add(skip, up, coarse_level), skip


class _Stem(nn.Module):
    pass


class Embedding(PointModule):
    def __init__(self, in_channels, embed_channels, norm_layer=None, act_layer=None):
        super().__init__()
        self.in_channels, self.embed_channels = in_channels, embed_channels
        self.stem = _Stem()
        self.stem.conv = SubMConv3d(in_channels, embed_channels, 5, bias=False, indice_key="stem")
        if norm_layer is not None:
            self.stem.norm = norm_layer(embed_channels)
        if act_layer is not None:
            self.stem.act = act_layer()

    def forward(self, feat, level):
        x = self.stem.conv(feat, level)
        return _norm_act(x, getattr(self.stem, "norm", None), getattr(self.stem, "act", None))


class _Stage(nn.Module):
    pass


def backward_in_two(outputs, grad_outputs, cut, between=None):
    """The backward pass of a forward that was given data_dict["backward_cut"] = cut, in two autograd calls: first down to the
    inputs of the last decoder stage (every gradient of that stage's parameters is final then), `between()` (e.g. pack that stage
    and start its all-reduce), then the rest from the gradients the cut's leaves received.  With an empty cut (evaluation, no
    decoder) it is one ordinary backward."""
    torch.autograd.backward(outputs, grad_tensors=grad_outputs)
    if between is not None:
        between()
    backward_tail(cut)


def backward_tail(cut):
    """Second call of backward_in_two: from the tensors the cut detached, with the gradients its leaves received."""
    pairs = [(o, l.grad) for o, l in cut if l.grad is not None]
    if pairs:
        torch.autograd.backward([o for o, _ in pairs], grad_tensors=[g for _, g in pairs])
    for _, l in cut:
        l.grad = None


@MODELS.register_module("PT-v3m1")
class PointTransformerV3(PointModule):
    def __init__(self, in_channels=6, order=("z", "z-trans"), stride=(2, 2, 2, 2), enc_depths=(2, 2, 2, 6, 2),
                 enc_channels=(32, 64, 128, 256, 512), enc_num_head=(2, 4, 8, 16, 32),
                 enc_patch_size=(48, 48, 48, 48, 48), dec_depths=(2, 2, 2, 2), dec_channels=(64, 64, 128, 256),
                 dec_num_head=(4, 4, 8, 16), dec_patch_size=(48, 48, 48, 48), mlp_ratio=4, qkv_bias=True,
                 qk_scale=None, attn_drop=0.0, proj_drop=0.0, drop_path=0.3, pre_norm=True, shuffle_orders=True,
                 enable_rpe=False, enable_flash=True, upcast_attention=False, upcast_softmax=False, cls_mode=False,
                 pdnorm_bn=False, pdnorm_ln=False, pdnorm_decouple=True, pdnorm_adaptive=False, pdnorm_affine=True,
                 pdnorm_conditions=("ScanNet", "S3DIS", "Structured3D")):
        super().__init__()
        if pdnorm_bn or pdnorm_ln:
            raise NotImplementedError("PDNorm is off in every SceneSplat language config; not on the HIP path")
        self.num_stages = len(enc_depths)
        self.order = [order] if isinstance(order, str) else list(order)
        self.cls_mode, self.shuffle_orders = cls_mode, shuffle_orders
        self.stride = tuple(stride)
        assert self.num_stages == len(stride) + 1 == len(enc_channels) == len(enc_num_head) == len(enc_patch_size)
        assert cls_mode or self.num_stages == len(dec_depths) + 1 == len(dec_channels) + 1
        assert cls_mode or self.num_stages == len(dec_num_head) + 1 == len(dec_patch_size) + 1

        def bn_layer(c):
            return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)

        ln_layer, act_layer = nn.LayerNorm, nn.GELU
        self.enc_channels = tuple(enc_channels)
        self.dec_channels_ = tuple(dec_channels) if not cls_mode else ()        # width of the decoder stage at level s
        self.embedding = Embedding(in_channels, enc_channels[0], bn_layer, act_layer)
        blk = dict(mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop,
                   norm_layer=ln_layer, act_layer=act_layer, pre_norm=pre_norm, enable_rpe=enable_rpe,
                   enable_flash=enable_flash, upcast_attention=upcast_attention, upcast_softmax=upcast_softmax)
        enc_dp = [x.item() for x in torch.linspace(0, drop_path, sum(enc_depths))]
        self.enc = _Stage()
        for s in range(self.num_stages):
            dp = enc_dp[sum(enc_depths[:s]):sum(enc_depths[:s + 1])]
            enc = _Stage()
            if s > 0:
                enc.down = SerializedPooling(enc_channels[s - 1], enc_channels[s], stride[s - 1], bn_layer, act_layer)
            for i in range(enc_depths[s]):
                setattr(enc, f"block{i}", Block(enc_channels[s], enc_num_head[s], enc_patch_size[s], drop_path=dp[i],
                                               order_index=i % len(self.order), cpe_indice_key=f"stage{s}", **blk))
            setattr(self.enc, f"enc{s}", enc)
        self.enc_depths, self.dec_depths = tuple(enc_depths), tuple(dec_depths)
        if not cls_mode:
            dec_dp = [x.item() for x in torch.linspace(0, drop_path, sum(dec_depths))]
            self.dec = _Stage()
            dec_channels = list(dec_channels) + [enc_channels[-1]]
            for s in reversed(range(self.num_stages - 1)):
                dp = dec_dp[sum(dec_depths[:s]):sum(dec_depths[:s + 1])]
                dp.reverse()
                dec = _Stage()
                dec.up = SerializedUnpooling(dec_channels[s + 1], enc_channels[s], dec_channels[s], bn_layer, act_layer)
                for i in range(dec_depths[s]):
                    setattr(dec, f"block{i}", Block(dec_channels[s], dec_num_head[s], dec_patch_size[s], drop_path=dp[i],
                                                   order_index=i % len(self.order), cpe_indice_key=f"stage{s}", **blk))
                setattr(self.dec, f"dec{s}", dec)

    def draw_perms(self):
        """Curve-order permutations for one forward, drawn from the CPU RNG like the reference:
        level 0 only when shuffle_orders (structure.py:94-98), every pooled level always
        (SerializedPooling.shuffle_orders defaults to True and is never overridden)."""
        K = len(self.order)
        perms = [torch.randperm(K).tolist() if self.shuffle_orders else list(range(K))]
        for s in range(1, self.num_stages):
            down = getattr(self.enc, f"enc{s}").down
            perms.append(torch.randperm(K).tolist() if down.shuffle_orders else list(range(K)))
        return perms

    def plan_specs(self):
        """Everything the float pipeline will ask the plan for: (window specs, conv kernel sizes)."""
        K = len(self.order)
        # which convs read the rulebook in WALK order is decided per level by the same predicates the forward uses
        # (ScenePlan.materialize -> conv_wants_walk below): the widths of the convs of each level travel with the spec
        wide = list(self.enc_channels)
        if not self.cls_mode:
            for s, c in enumerate(self.dec_channels_):
                wide[s] = max(wide[s], c)
        walk = [[c for c in {self.enc_channels[s], wide[s]}] for s in range(self.num_stages)]
        wins, ks = [], [(0, 5, [(self.embedding.in_channels, self.enc_channels[0])])]
        for s in range(self.num_stages):
            enc = getattr(self.enc, f"enc{s}")
            ks.append((s, 3, walk[s]))
            for i in range(self.enc_depths[s]):
                wins.append((s, i % K, getattr(enc, f"block{i}").attn.patch_size))
        if not self.cls_mode:
            for s in range(self.num_stages - 1):
                dec = getattr(self.dec, f"dec{s}")
                for i in range(self.dec_depths[s]):
                    wins.append((s, i % K, getattr(dec, f"block{i}").attn.patch_size))
        return wins, ks

    def prepare_plan(self, data_dict, perms=None, stream=None):
        """Build the integer plan of a batch ahead of time, optionally on a side stream so that its
        device->host round trips (depth, pooled sizes) never wait for the float pipeline of the
        previous step.  Pass the result as data_dict["plan"]."""
        if "grid_coord" not in data_dict:
            raise KeyError("prepare_plan needs grid_coord")
        perms = perms if perms is not None else self.draw_perms()
        if stream is None:
            plan = build_plan(data_dict["grid_coord"], data_dict["offset"], self.order, self.stride, perms)
            plan.materialize(*self.plan_specs())
            return plan
        # NB: no wait on the main stream here -- that would park the plan behind the previous step's whole
        # backward.  The caller guarantees grid_coord / offset are already materialised (as a loader's
        # copy stream would after its own event).
        with torch.cuda.stream(stream):
            plan = build_plan(data_dict["grid_coord"], data_dict["offset"], self.order, self.stride, perms)
            plan.materialize(*self.plan_specs())
            plan.ready_event = torch.cuda.Event()
            plan.ready_event.record(stream)
        return plan

    def _stage_linears(self, stage):
        cache = self.__dict__.setdefault("_stage_linear_cache", {})
        ls = cache.get(id(stage))
        if ls is None:
            ls = [m for m in stage.modules() if isinstance(m, (nn.Linear, nn.LayerNorm))]
            cache[id(stage)] = ls
        return ls

    def _refresh_shadows(self):
        """bf16 copies of every Linear / SubMConv3d weight and Linear bias, refreshed with multi-tensor copies."""
        ps = self.__dict__.get("_shadow_lists")
        if ps is None or ps[2] != next(self.parameters()).device:
            params = []
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    params += [m.weight, m.bias]
                elif isinstance(m, SubMConv3d):
                    params += [m.weight, m.bias]         # (the bias feeds the small levels' im2col GEMM as a bf16 operand)
            src, dst = SF.register_shadows(params)
            # wide Linear layers of the two finest levels: a transposed copy for the NT form of their dgrad GEMM
            wide = []
            for st in [getattr(self.enc, f"enc{s}", None) for s in (0, 1)] + [getattr(getattr(self, "dec", None), f"dec{s}", None) for s in (0, 1)]:
                if st is not None:
                    wide += [m.weight for m in st.modules() if isinstance(m, nn.Linear) and m.weight.numel() >= 65536]
            if RUNTIME.get("dgrad_nt", True):
                SF.register_transposed(wide)
            # dgrad weights of the convs that run on plain bf16 operands: mirrored once per refresh instead of once per backward call
            SF.register_mirrored([m.weight for m in self.modules() if isinstance(m, SubMConv3d)
                                  and conv_dtype_for(m.weight.shape[0]) == torch.bfloat16])
            # fp32 accumulators of the weight / bias gradients: one zero-filled arena per step
            total = sum(((p.numel() + 3) & ~3) for p in params if p is not None) + 4 * len(params)
            ps = (src, dst, next(self.parameters()).device, total)
            self.__dict__["_shadow_lists"] = ps
        SF.refresh_shadows(ps[0], ps[1])
        if torch.is_grad_enabled():
            nv.zero_arena_begin(ps[3], ps[2])

    def _draw_row_scales(self, levels, device):
        """DropPath masks (one Bernoulli(keep)/keep scale per row and residual seam, timm DropPath on (n,C) rows as
        ptv3:333-336) of ALL blocks from one launch (plus the seed draw) per forward instead of 2 per seam (88)."""
        blocks = []
        for s in range(self.num_stages):
            enc = getattr(self.enc, f"enc{s}")
            blocks += [(getattr(enc, f"block{i}"), levels[s].n) for i in range(self.enc_depths[s])]
        if not self.cls_mode:
            for s in reversed(range(self.num_stages - 1)):
                dec = getattr(self.dec, f"dec{s}")
                blocks += [(getattr(dec, f"block{i}"), levels[s].n) for i in range(self.dec_depths[s])]
        segs = []
        for blk, n in blocks:
            dp = blk.drop_path[0]
            if isinstance(dp, DropPath) and dp.p > 0.0:
                segs += [(blk, n, 1.0 - dp.p)] * 2
            blk.__dict__["_row_scales"] = []
        if not segs:
            return
        key = tuple((n, k) for _, n, k in segs) + (str(device),)
        cache = self.__dict__.get("_keep_vec")
        if cache is None or cache[0] != key:
            kv = torch.cat([torch.full((n,), k, dtype=torch.float32) for _, n, k in segs]).pin_memory().to(device, non_blocking=True)
            cache = (key, kv)
            self.__dict__["_keep_vec"] = cache
        kv = cache[1]
        scales = nv.row_keep_scales(kv)          # Bernoulli(keep) / keep per row: one Philox launch (csrc/rows.hip)
        off = 0
        for blk, n, _ in segs:
            blk.__dict__["_row_scales"].append(scales[off:off + n])
            off += n

    def forward(self, data_dict, perms=None):
        point = data_dict if isinstance(data_dict, Point) else Point(data_dict)
        feat = point["feat"]
        if not feat.is_cuda:
            raise RuntimeError("PT-v3m1 (scenesplat_amd) runs on the GPU only: the HIP path has no CPU fallback")
        if "grid_coord" not in point:
            if not {"grid_size", "coord"} <= set(point.keys()):
                raise KeyError("need grid_coord, or coord + grid_size (structure.py:54-62)")
            c = point["coord"]
            point["grid_coord"] = torch.div(c - c.min(0)[0], point["grid_size"], rounding_mode="trunc").int()
        if dict.__contains__(point, "offset"):
            offset = point["offset"]
        else:
            offset = point["offset"]  # derived lazily from batch
        # the fused conv consumes bf16: let each block hand the next one a bf16 copy of the residual stream
        self._want_copy = (RUNTIME["conv_dtype"] == torch.bfloat16)      # per stage: only where the conv takes plain bf16 operands
        # bf16 operands are consumed under autocast (Linear) and whenever the conv runs in bf16 -- also without autocast
        # (the evaluator's no_grad / chunk_size call): refresh in both cases.  Shadows are version-stamped
        # (SF.bf16_of), so a forward that skips the refresh casts afresh instead of reading stale weights.
        if (torch.is_autocast_enabled() or self._want_copy) and RUNTIME.get("param_shadows", True):
            self._refresh_shadows()
        plan = point.get("plan", None)
        if plan is None:
            plan = build_plan(point["grid_coord"], offset, self.order, self.stride,
                              perms if perms is not None else self.draw_perms())
        elif plan.ready_event is not None:
            torch.cuda.current_stream().wait_event(plan.ready_event)
            plan.record_stream(torch.cuda.current_stream())
            plan.ready_event = None
        levels = plan.levels
        if self.training:
            self._draw_row_scales(levels, feat.device)
        x = self.embedding(feat, levels[0])
        skips = []
        for s in range(self.num_stages):
            enc = getattr(self.enc, f"enc{s}")
            SF.stage_begin(self._stage_linears(enc), levels[s].n)      # grouped weight gradients of the stage's Linears
            if s > 0:
                skips.append(x)
                x = enc.down(x, levels[s])
            xb = None
            for i in range(self.enc_depths[s]):
                wc = self._want_copy and conv_dtype_for(x.shape[1]) == torch.bfloat16
                x, xb = getattr(enc, f"block{i}")(x, x if xb is None else xb, levels[s], wc and i + 1 < self.enc_depths[s])
            SF.stage_end()
        lv = self.num_stages - 1
        if not self.cls_mode:
            # backward cut (data_dict["backward_cut"] = [], training): the inputs of the LAST decoder stage -- the first one the
            # backward pass finishes, 52 % of the parameter bytes in the lang-pretrain model -- become leaves, so that the caller
            # can run the backward in two calls (backward_in_two below) and start the gradient exchange of that stage in between
            cut = point.get("backward_cut", None) if torch.is_grad_enabled() else None
            for s in reversed(range(self.num_stages - 1)):
                dec = getattr(self.dec, f"dec{s}")
                if cut is not None and s == 0 and x.requires_grad:
                    leaves = (x.detach().requires_grad_(True), skips[0].detach().requires_grad_(True))
                    cut.extend([(x, leaves[0]), (skips[0], leaves[1])])
                    x, skips[0] = leaves
                SF.stage_begin(self._stage_linears(dec), levels[s].n)
                x, conv_in = dec.up(x, skips[s], levels[s + 1])
                xb = None
                for i in range(self.dec_depths[s]):
                    wc = self._want_copy and conv_dtype_for(x.shape[1]) == torch.bfloat16
                    x, xb = getattr(dec, f"block{i}")(x, conv_in if i == 0 else (x if xb is None else xb), levels[s],
                                                      wc and i + 1 < self.dec_depths[s])
                SF.stage_end()
            lv = 0
        out = Point(feat=x, plan=plan, level=lv)
        for k in ("coord", "grid_coord", "offset"):
            if lv == 0 and dict.__contains__(point, k):
                out[k] = point[k]
        return out
