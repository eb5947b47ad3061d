"""``Point``: the attribute-access dict PTv3 passes between modules (reference:
pointcept/models/utils/structure.py:14-45).  Here it is a thin carrier: the integer structure
lives in ``scenesplat_amd.plan.ScenePlan`` (attached as ``point.plan``); ``serialized_*`` and
``pad``-style keys are materialised from it on request so reference-shaped consumers keep
working."""
import torch


class Point(dict):
    # "batch" / "offset" are derived from each other like the reference (structure.py:41-45) but
    # lazily: offset2batch needs a device->host sync that the hot path never pays.
    def __missing__(self, k):
        if k == "batch" and dict.__contains__(self, "offset"):
            v = offset2batch(dict.__getitem__(self, "offset"))
        elif k == "offset" and dict.__contains__(self, "batch"):
            v = batch2offset(dict.__getitem__(self, "batch"))
        else:
            raise KeyError(k)
        self[k] = v
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]


@torch.inference_mode()
def offset2bincount(offset):
    return torch.diff(offset, prepend=torch.tensor([0], device=offset.device, dtype=torch.long))


@torch.inference_mode()
def offset2batch(offset):
    bincount = offset2bincount(offset)
    return torch.arange(len(bincount), device=offset.device, dtype=torch.long).repeat_interleave(bincount)


@torch.inference_mode()
def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).long()
