"""``Point``: the attribute-access dict PTv3 passes between modules (reference:
pointcept/models/utils/structure.py:14-45).  Here it is a thin carrier: the integer structure
lives in ``scenesplat_amd.plan.ScenePlan`` (attached as ``point.plan`` by the model, or built by
``Point.serialization``).  The reference-shaped keys a consumer outside the hot path may read are
materialised from the plan ON REQUEST (never on the hot path, which reads the plan directly):

  serialized_depth / serialized_code / serialized_order / serialized_inverse     structure.py:47-102
  sparse_shape / sparse_conv_feat (features, indices, spatial_shape, batch_size)   structure.py:104-140
  Point.padding(patch) -> (pad, unpad, cu_seqlens)                                 ptv3:114-170

all in the reference's dtypes (int64 codes / orders, int32 cu_seqlens) and in the level's CURRENT
curve order (the shuffle of structure.py:94-98 is the plan's curve permutation)."""
import torch


def _batch_from_offset(offset):
    """offset (B) cumulative counts -> batch (n) int64: row i belongs to the first element whose end exceeds i
    (pointcept/models/utils/misc.py:19-23 computes the same vector by repeat_interleave)."""
    offset = offset.long()
    n = int(offset[-1]) if offset.numel() else 0
    if offset.is_cuda:
        from .. import native as nv
        return nv.offsets_to_batch(offset.to(torch.int32).contiguous(), n).long()
    return torch.searchsorted(offset, torch.arange(n, dtype=torch.long), right=True)


def _offset_from_batch(batch):
    """batch (n) element ids -> offset (B) int64: the number of rows with id <= b (misc.py:26-28)."""
    batch = batch.long()
    nb = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.zeros(nb, dtype=torch.long, device=batch.device).scatter_add_(0, batch, torch.ones_like(batch))
    return counts.cumsum(0)


# the names the reference exports (pointcept/models/utils/__init__.py)
offset2batch, batch2offset = _batch_from_offset, _offset_from_batch


def offset2bincount(offset):
    offset = offset.long()
    return offset - torch.cat([offset.new_zeros(1), offset[:-1]])


class SparseConvFeat:
    """What the hot path's consumers read of a spconv.SparseConvTensor (structure.py:131-138; modules.py:64-91):
    features, indices [batch, x, y, z] int32, spatial_shape, batch_size, replace_feature()."""

    def __init__(self, features, indices, spatial_shape, batch_size):
        self.features, self.indices, self.spatial_shape, self.batch_size = features, indices, spatial_shape, batch_size

    def replace_feature(self, feature):
        return SparseConvFeat(feature, self.indices, self.spatial_shape, self.batch_size)


class Point(dict):
    # "batch" / "offset" are derived from each other like the reference (structure.py:41-45) but
    # lazily: offset2batch needs a device->host sync that the hot path never pays.
    def __missing__(self, k):
        if k == "batch" and dict.__contains__(self, "offset"):
            v = _batch_from_offset(dict.__getitem__(self, "offset"))
        elif k == "offset" and dict.__contains__(self, "batch"):
            v = _offset_from_batch(dict.__getitem__(self, "batch"))
        elif k in ("serialized_depth", "serialized_code", "serialized_order", "serialized_inverse") and self._level() is not None:
            lv = self._level()
            rows = torch.tensor(lv.curves, dtype=torch.long, device=lv.codes.device)
            v = {"serialized_depth": lambda: lv.depth,
                 "serialized_code": lambda: lv.codes.index_select(0, rows),
                 "serialized_order": lambda: lv.order.index_select(0, rows).long(),
                 "serialized_inverse": lambda: lv.inverse.index_select(0, rows).long()}[k]()
        elif k == "sparse_shape" and self._grid() is not None:
            v = (self._grid().max(0).values + 96).tolist()             # structure.py:124-128 (pad = 96)
        elif k == "sparse_conv_feat" and self._grid() is not None and dict.__contains__(self, "feat"):
            gc = self._grid()
            batch = self["batch"]
            v = SparseConvFeat(dict.__getitem__(self, "feat"), torch.cat([batch.unsqueeze(-1).int(), gc.int()], 1).contiguous(),
                               self["sparse_shape"], int(batch[-1]) + 1)
        else:
            raise KeyError(k)
        self[k] = v
        return v

    def _level(self):
        plan = dict.get(self, "plan")
        return None if plan is None else plan.levels[int(dict.get(self, "level", 0))]

    def _grid(self):
        if dict.__contains__(self, "grid_coord"):
            return dict.__getitem__(self, "grid_coord")
        lv = self._level()
        return None if lv is None else lv.grid_coord

    def serialization(self, order="z", depth=None, shuffle_orders=False):
        """structure.py:47-102 on the plan kernels (csrc/serialize.hip): builds the level-0 plan of this point cloud for the given
        curves and publishes serialized_depth / _code / _order / _inverse.  Needs grid_coord (or coord + grid_size) and offset on
        the GPU."""
        from ..plan import build_plan
        order = [order] if isinstance(order, str) else list(order)
        if not dict.__contains__(self, "grid_coord"):
            if not {"grid_size", "coord"} <= set(self.keys()):
                raise KeyError("need grid_coord, or coord + grid_size (structure.py:54-62)")
            c = self["coord"]
            self["grid_coord"] = torch.div(c - c.min(0)[0], self["grid_size"], rounding_mode="trunc").int()
        perms = [torch.randperm(len(order)).tolist()] if shuffle_orders else None
        self["plan"] = build_plan(self["grid_coord"], self["offset"], order, (), perms, depth=depth)
        self["level"] = 0
        for k in ("serialized_depth", "serialized_code", "serialized_order", "serialized_inverse"):
            dict.pop(self, k, None)
            self[k]                                                     # materialise now, like the reference
        return self

    def sparsify(self, pad=96):
        """structure.py:104-140: sparse_shape = max(grid_coord) + pad, sparse_conv_feat = the SparseConvTensor view."""
        if not dict.__contains__(self, "sparse_shape"):
            self["sparse_shape"] = (self._grid().max(0).values + pad).tolist()
        dict.pop(self, "sparse_conv_feat", None)
        return self["sparse_conv_feat"]

    @torch.no_grad()
    def padding(self, patch_size, order_index=0):
        """(pad, unpad, cu_seqlens) of SerializedAttention.get_padding_and_inverse (ptv3:114-170) for this level and patch size,
        read back from the plan's window index: the kernels use gidx = order[pad] and sidx (the canonical slot of every row), so
        pad = inverse[gidx] and unpad[inverse[row]] = slot."""
        lv = self._level()
        if lv is None:
            raise KeyError("padding() needs a plan (run the model or Point.serialization first)")
        w = lv.window(order_index, int(patch_size))
        inverse = lv.inverse_row(order_index).long()
        pad = inverse[w.gidx.long()]
        slots = torch.nonzero(w.sidx >= 0, as_tuple=True)[0]
        unpad = torch.empty(lv.n, dtype=torch.long, device=pad.device)
        unpad[inverse[w.sidx[slots].long()]] = slots
        return pad, unpad, w.win_start.to(torch.int32)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]
