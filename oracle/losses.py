"""Oracle (test infrastructure): the vision-language distillation head, fp32 CPU.

Follows /root/reference/pointcept/models/default.py:88-113 (LangPretrainer.forward),
pointcept/models/losses/misc.py:248-270 (CosineSimilarity), :274-295 (L2Loss),
:299-421 (AggregatedContrastiveLoss), pointcept/models/losses/builder.py:20-27 (Criteria).
"""
import torch
import torch.nn.functional as F


def cosine_similarity_loss(pred, target, valid_feat_mask, loss_weight=1.0, reduction="mean"):
    m = valid_feat_mask.bool()
    loss = 1 - F.cosine_similarity(pred[m], target[m], dim=1)
    if reduction == "mean":
        cnt = m.sum()
        loss = loss.sum() / cnt if cnt > 0 else loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss_weight * loss


def l2_loss(pred, target, valid_feat_mask, loss_weight=1.0, reduction="mean"):
    m = valid_feat_mask.bool()
    loss = ((pred[m] - target[m]) ** 2).sum(dim=1)
    if reduction == "mean":
        cnt = m.sum()
        loss = loss.sum() / cnt if cnt > 0 else loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss_weight * loss


def aggregated_contrastive_loss(pred, valid_feat_mask, segment, epoch_progress=None, temperature=0.2,
                                loss_weight=1.0, schedule="all", reduction="mean", rand_keys=None,
                                min_count=100):
    """rand_keys=None: draw torch.randperm per class from the global RNG in the
    reference's call order (misc.py:364-388) -> reproduces the reference bit for bit under
    the same torch.manual_seed.  rand_keys (N,) given: the permutation of a class is the
    ascending order of its rows' keys (ties by row index); the first n//2 rows form group
    a -- the explicit-randomness form the HIP head uses."""
    zero = torch.tensor(0.0)
    if "last_" in schedule and epoch_progress is not None:
        if epoch_progress <= 1 - float(schedule.split("_")[-1]) / 100:
            return zero
    elif schedule == "skip":
        return zero
    if segment is None:
        return zero
    valid = (valid_feat_mask > 0) & (segment != -1)
    if valid.sum() == 0:
        return zero
    feats = pred[valid]
    labels = segment[valid]
    keys = rand_keys[valid] if rand_keys is not None else None
    A, B = [], []
    for lab in torch.unique(labels):
        idx = (labels == lab).nonzero(as_tuple=True)[0]
        if idx.numel() < min_count:
            continue
        if keys is None:
            perm = idx[torch.randperm(idx.size(0))]
        else:
            perm = idx[torch.argsort(keys[idx], stable=True)]
        split = perm.size(0) // 2
        if split == 0 or perm.size(0) - split == 0:
            continue
        A.append(feats[perm[:split]].sum(0))
        B.append(feats[perm[split:]].sum(0))
    if not A:
        return zero
    A = F.normalize(torch.stack(A), p=2, dim=1)
    B = F.normalize(torch.stack(B), p=2, dim=1)
    logits = A @ B.t() / temperature
    tgt = torch.arange(logits.size(0))
    loss = (F.cross_entropy(logits, tgt) + F.cross_entropy(B @ A.t() / temperature, tgt)) / 2.0
    if reduction == "sum":
        loss = loss * logits.size(0)
    return loss_weight * loss


def lang_head(feat, lang_feat, valid_feat_mask, segment, epoch_progress, criteria, rand_keys=None):
    """F.normalize + Criteria sum (default.py:98-109, builder.py:20-27).
    criteria: list of dicts like the reference config (type, loss_weight, ...)."""
    pred = F.normalize(feat, p=2, dim=1)
    loss = 0
    for c in criteria:
        c = dict(c); t = c.pop("type")
        if t == "CosineSimilarity":
            loss = loss + cosine_similarity_loss(pred, lang_feat, valid_feat_mask, **c)
        elif t == "L2Loss":
            loss = loss + l2_loss(pred, lang_feat, valid_feat_mask, **c)
        elif t == "AggregatedContrastiveLoss":
            loss = loss + aggregated_contrastive_loss(pred, valid_feat_mask, segment, epoch_progress,
                                                      rand_keys=rand_keys, **c)
        else:
            raise KeyError(t)
    return loss
