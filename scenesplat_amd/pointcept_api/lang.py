"""Vision-language distillation head: ``LangPretrainer`` and its three criteria.

Registry names, constructor kwargs and forward contracts follow the reference
(pointcept/models/default.py:77-176; pointcept/models/losses/misc.py:247-421;
pointcept/models/losses/builder.py:13-31).  The math is restated sync-free for the GPU:
boolean-mask gathers (``pred[valid]``, a device->host sync each) become masked reductions, and
AggregatedContrastiveLoss's per-class Python loop (nonzero / randperm / sum per class,
misc.py:364-388) becomes one keyed radix sort + one CSR segment-sum over (class, half) groups
with fixed-shape masked cross-entropy, keeping the reference's semantics exactly: rows with
mask & segment != -1, classes with >= 100 rows, a uniformly random permutation of each class
split at n//2, group *sums*, L2-normalise, symmetric CE at temperature tau.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as SF
from .. import native as nv
from .registry import LOSSES, MODELS, build_model
from .structure import Point


def _head_sums(pred, target, valid_feat_mask):
    """[sum_valid (1 - cos), sum_valid |pred - target|^2, #valid] from the fused head kernel (csrc/head.hip): read once
    for both losses; reuses LangPretrainer's fused normalise pass when pred came out of it."""
    if pred.is_cuda and pred.dim() == 2 and pred.shape[1] % 4 == 0 and pred.shape[1] <= 2048 \
            and pred.dtype in (torch.float32, torch.bfloat16):
        return SF.lang_head_sums(pred, target, valid_feat_mask)
    # shapes outside the kernel's contract (odd widths): the same reductions as masked PyTorch-ROCm ops
    m = (valid_feat_mask > 0).to(pred.dtype)
    t = target.to(pred.dtype)
    return torch.stack([((1 - F.cosine_similarity(pred, t, dim=1)) * m).sum(), (((pred - t) ** 2).sum(dim=1) * m).sum(), m.sum()])


@LOSSES.register_module()
class CosineSimilarity(nn.Module):
    """mean / sum over valid rows of 1 - cos(pred, target) (losses/misc.py:248-270)."""

    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, valid_feat_mask, **kwargs):
        sums = _head_sums(pred, target, valid_feat_mask)
        loss = sums[0]
        if self.reduction == "mean":
            loss = loss / sums[2].clamp(min=1.0)          # no valid row: the sum (0), as the reference
        return self.loss_weight * loss


@LOSSES.register_module()
class L2Loss(nn.Module):
    """mean / sum over valid rows of |pred - target|^2 (losses/misc.py:274-295)."""

    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, valid_feat_mask, **kwargs):
        sums = _head_sums(pred, target, valid_feat_mask)
        loss = sums[1]
        if self.reduction == "mean":
            loss = loss / sums[2].clamp(min=1.0)
        return self.loss_weight * loss


class _GroupSum(torch.autograd.Function):
    """G[g] = sum of rows listed in CSR (indices, ptr); backward broadcasts dG[group(row)]."""

    @staticmethod
    def forward(ctx, feat, indices, ptr, n_groups, row_group):
        ctx.save_for_backward(row_group)
        ctx.n = feat.shape[0]
        return nv.segment_reduce(feat.contiguous(), indices, ptr, n_groups, False)

    @staticmethod
    def backward(ctx, dG):
        (row_group,) = ctx.saved_tensors
        # rows outside every group index the extra all-zero row
        dGz = torch.cat([dG, dG.new_zeros(1, dG.shape[1])], 0).contiguous()
        return nv.gather_rows(dGz, row_group), None, None, None, None


@LOSSES.register_module()
class AggregatedContrastiveLoss(nn.Module):
    def __init__(self, temperature=0.2, reduction="mean", loss_weight=1.0, schedule="all", max_classes=256,
                 min_count=100):
        super().__init__()
        self.temperature, self.reduction, self.loss_weight, self.schedule = temperature, reduction, loss_weight, schedule
        self.max_classes, self.min_count = max_classes, min_count
        if "last_" in schedule:
            self.last_percent = float(schedule.split("_")[-1]) / 100

    def forward(self, pred, target, valid_feat_mask, segment, epoch_progress=None, rand_keys=None, **kwargs):
        dev = pred.device
        if "last_" in self.schedule and epoch_progress is not None:
            if epoch_progress <= (1 - self.last_percent):
                return torch.tensor(0.0, device=dev)
        elif self.schedule == "skip":
            return torch.tensor(0.0, device=dev)
        if segment is None:
            return torch.tensor(0.0, device=dev)
        N, Cc = pred.shape[0], self.max_classes
        valid = (valid_feat_mask > 0) & (segment != -1)
        if rand_keys is None:
            rand_keys = torch.rand(N, device=dev)
        # labels must lie in [-1, max_classes): the reference gives every distinct label its own class, a clamp would
        # silently merge the overflow into one.  Checked on the device without a host sync (the failure surfaces as
        # a device-side assert at the next synchronisation); raise max_classes for larger label spaces.
        seg64 = segment.long()
        torch._assert_async(((seg64 >= -1) & (seg64 < Cc)).all(),
                            "AggregatedContrastiveLoss: segment labels must lie in [-1, max_classes)")
        # sort rows by (class, random key); invalid rows go to a sentinel class at the end
        cls = torch.where(valid, seg64.clamp(0, Cc - 1), torch.full_like(seg64, Cc))
        keyi = (rand_keys.double() * (1 << 31)).long().clamp(0, (1 << 31) - 1)
        comp = ((cls << 31) | keyi).unsqueeze(0).contiguous()
        order, _, _ = nv.argsort_i64(comp, 31 + (Cc).bit_length(), want_inverse=False, want_sorted=False)
        order = order[0]
        counts = torch.zeros(Cc + 1, dtype=torch.int64, device=dev).scatter_add_(0, cls, torch.ones_like(cls))
        starts = torch.cumsum(counts, 0) - counts                      # class start in sorted order
        half = counts // 2
        # CSR over 2*Cc groups: [a_0, b_0, a_1, b_1, ...]
        ptr = torch.stack([starts[:Cc], starts[:Cc] + half[:Cc]], 1).reshape(-1)
        ptr = torch.cat([ptr, starts[Cc:Cc + 1]]).to(torch.int32).contiguous()
        # group of every row (for the backward broadcast); rows of the sentinel class -> 2*Cc (zero row)
        rank = torch.empty(N, dtype=torch.int64, device=dev)
        rank[order.long()] = torch.arange(N, device=dev)
        in_b = (rank - starts[cls]) >= half[cls]
        row_group = torch.where(cls < Cc, 2 * cls + in_b.long(), torch.full_like(cls, 2 * Cc)).to(torch.int32)
        G = _GroupSum.apply(pred.float(), order.contiguous(), ptr, 2 * Cc, row_group.contiguous())
        used = (counts[:Cc] >= self.min_count) & (half[:Cc] > 0)
        nused = used.sum()
        colmask = used.unsqueeze(0)
        with torch.autocast("cuda", enabled=False):      # <= 256 x 256 logits: keep the tiny tail in fp32
            A, B = F.normalize(G[0::2].float(), p=2, dim=1), F.normalize(G[1::2].float(), p=2, dim=1)
            logits = (A @ B.t()) / self.temperature

            def ce(lg):
                lg = torch.where(colmask, lg, torch.full_like(lg, -1e30))
                per = torch.logsumexp(lg, dim=1) - lg.diagonal()
                return (per * used).sum() / nused.clamp(min=1)

            loss = (ce(logits) + ce(logits.t())) / 2.0
        if self.reduction == "sum":
            loss = loss * nused
        return self.loss_weight * loss * (nused > 0)


class Criteria(object):
    """pointcept/models/losses/builder.py:13-27"""

    def __init__(self, cfg=None):
        self.cfg = cfg if cfg is not None else []
        self.criteria = [LOSSES.build(cfg=c) for c in self.cfg]

    def __call__(self, pred, target, **kwargs):
        if len(self.criteria) == 0:
            return pred
        loss = 0
        for c in self.criteria:
            loss += c(pred, target, **kwargs)
        return loss


def build_criteria(cfg):
    return Criteria(cfg)


@MODELS.register_module()
class LangPretrainer(nn.Module):
    def __init__(self, backbone=None, criteria=None):
        super().__init__()
        self.backbone = build_model(backbone)
        self.criteria = build_criteria(criteria)

    def steady_key(self, host):
        """The HOST-side decisions a training step of this model takes, as a hashable key for the steady-state replay
        (scenesplat_amd/steady_state.py): the schedule gate of every criterion, not the raw epoch_progress float (which changes
        every epoch and would force a re-capture per epoch)."""
        ep = host.get("epoch_progress", None)
        gates = []
        for c in self.criteria.criteria:
            sched = getattr(c, "schedule", None)
            if isinstance(sched, str) and "last_" in sched and ep is not None:
                gates.append(bool(ep > (1 - c.last_percent)))
            else:
                gates.append(sched)
        return tuple(gates)

    def forward(self, input_dict, chunk_size=None):
        if chunk_size is not None and chunk_size > 0 and input_dict["coord"].shape[0] > chunk_size:
            return self._chunked_forward(input_dict, chunk_size)
        point_feat = self.backbone(Point(input_dict))
        if self.training:
            return dict(loss=self._normalize_and_criteria(point_feat["feat"], input_dict, input_dict["epoch_progress"]))
        point_feat["feat"] = self._normalize(point_feat["feat"])
        return dict(point_feat=point_feat)

    @staticmethod
    def _fusable(feat):
        return feat.is_cuda and feat.shape[1] % 4 == 0 and feat.shape[1] <= 2048

    def _normalize(self, feat):
        """F.normalize(feat, p=2, dim=1) (default.py:96) as one pass of the head kernel."""
        feat = feat.float()
        if self._fusable(feat):
            return SF.lang_head(feat, None, None, True)[0]
        return F.normalize(feat, p=2, dim=1)

    def _normalize_and_criteria(self, feat, d, epoch_progress):
        """default.py:96-109: normalise, then the criteria.  The normalisation and the cosine / L2 reductions over
        (pred, lang_feat) run as ONE pass; the criteria objects find those sums by identity (SF.lang_head_sums)."""
        feat = feat.float()
        target, mask = d["lang_feat"], d["valid_feat_mask"]
        try:
            pred = SF.lang_head(feat, target, mask, True)[0] if self._fusable(feat) else F.normalize(feat, p=2, dim=1)
            return self.criteria(pred, target, valid_feat_mask=mask, segment=d["segment"] if "segment" in d.keys() else None,
                                 epoch_progress=epoch_progress)
        finally:
            SF.lang_head_release()

    def _chunked_forward(self, input_dict, chunk_size):
        """Contiguous index-range chunks processed independently (default.py:115-176): every (N, ...) tensor is sliced,
        offset is rebuilt as [len]; training averages the per-chunk losses, eval concatenates the unit features.
        As in the reference the per-chunk criteria get epoch_progress=None (the scalar is not an (N, ...) tensor, so it
        never enters the chunk dict: default.py:137-140,160)."""
        N = input_dict["coord"].shape[0]
        outs = []
        for s in range(0, N, chunk_size):
            e = min(s + chunk_size, N)
            chunk = {k: v[s:e] for k, v in input_dict.items() if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == N}
            if "condition" in input_dict.keys():
                chunk["condition"] = input_dict["condition"][0]
            chunk["offset"] = torch.tensor([e - s], device=input_dict["coord"].device)
            feat = self.backbone(Point(chunk))["feat"]
            if self.training:
                outs.append(self._normalize_and_criteria(feat, chunk, chunk.get("epoch_progress", None)))
            else:
                outs.append(self._normalize(feat))
        if self.training:
            return dict(loss=torch.stack(outs).mean())
        return dict(point_feat={"feat": torch.cat(outs, dim=0)})
