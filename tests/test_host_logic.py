"""Host-side logic that needs no GPU: the plan-ahead thread (scenesplat_amd/plan.py:PlanAhead) and the stage labels of the gradient
exchange (scenesplat_amd/grad_exchange.py:default_stage_of).  (The GPU behaviour of both is covered in tests/test_hip_round2.py and
tests/test_engine_dist.py.)"""
import threading
import time

import pytest


def test_plan_ahead_hands_over_in_order_bounds_its_queue_and_stops():
    from scenesplat_amd.plan import PlanAhead
    built = []

    def build():
        built.append(len(built))
        return ("plan", built[-1])
    ahead = PlanAhead(build, depth=2, device=None)
    got = [ahead.get() for _ in range(5)]
    assert got == [("plan", i) for i in range(5)]                 # in build order
    time.sleep(0.2)
    assert len(built) <= 5 + 2 + 1                                 # at most `depth` queued + one being offered
    ahead.close()
    assert not ahead._t.is_alive()
    n = len(built)
    time.sleep(0.1)
    assert len(built) == n                                         # nothing is built after close()


def test_plan_ahead_retries_builds_that_hit_the_armed_sync_detector_and_surfaces_real_errors():
    from scenesplat_amd.plan import PlanAhead
    calls = []

    def flaky():
        calls.append(1)
        if len(calls) < 3:
            raise RuntimeError("called a synchronizing HIP operation")         # torch's process-wide detector, armed by another thread
        return "plan"
    ahead = PlanAhead(flaky, depth=1, device=None)
    assert ahead.get() == "plan" and len(calls) >= 3
    ahead.close()

    def broken():
        raise ValueError("no batch")
    bad = PlanAhead(broken, depth=1, device=None)
    with pytest.raises(RuntimeError, match="plan build thread failed") as ei:
        bad.get()
    assert isinstance(ei.value.__cause__, ValueError)
    bad.close()
    assert threading.active_count() < 20


def test_stage_labels_follow_the_backward_order_of_pt_v3m1_names():
    from scenesplat_amd.grad_exchange import default_stage_of
    assert default_stage_of("backbone.dec.dec0.block1.mlp.0.fc1.weight") == "dec.dec0"
    assert default_stage_of("module.backbone.enc.enc3.down.proj.weight") == "enc.enc3"
    assert default_stage_of("backbone.embedding.stem.conv.weight") == "embedding"
    assert default_stage_of("enc.enc0.block0.cpe.0.weight") == "enc.enc0"
    assert default_stage_of("criteria.0.temperature") == "other"


def test_point_batch_offset_helpers_on_the_cpu():
    """Point derives batch <-> offset lazily (structure.py:41-45 of the reference; values as models/utils/misc.py:11-28)."""
    import numpy as np
    import torch
    from scenesplat_amd.pointcept_api.structure import Point, batch2offset, offset2batch, offset2bincount
    off = torch.tensor([3, 3, 10, 11])                       # an empty element in the middle
    assert offset2bincount(off).tolist() == [3, 0, 7, 1]
    b = offset2batch(off)
    assert b.dtype == torch.int64 and b.tolist() == np.repeat([0, 1, 2, 3], [3, 0, 7, 1]).tolist()
    assert batch2offset(b).tolist() == [3, 3, 10, 11]
    p = Point(offset=off)
    assert p.batch.tolist() == b.tolist() and "batch" in p
    q = Point(batch=b)
    assert q["offset"].tolist() == [3, 3, 10, 11]
    with pytest.raises(AttributeError):
        Point(feat=torch.zeros(2, 2)).serialized_code        # no plan: nothing to materialise from


def test_plan_ahead_bounds_its_retries_and_waits_at_the_sync_gate(monkeypatch):
    """A detector somebody else left armed for good must end in an error at the consumer, not in a hang; and while the gate is
    held (steady_state.py holds it for the one eager step it runs under torch's process-wide sync detector) no build starts."""
    from scenesplat_amd import plan as P
    monkeypatch.setattr(P, "PLAN_BUILD_MAX_RETRIES", 5)
    n = []

    def always():
        n.append(1)
        raise RuntimeError("called a synchronizing HIP operation")
    bad = P.PlanAhead(always, depth=1, device=None)
    with pytest.raises(RuntimeError, match="plan build thread failed"):
        bad.get()
    assert len(n) == 6                                            # the first attempt + 5 retries
    bad.close()
    built = []
    with P.HOST_SYNC_GATE:
        ahead = P.PlanAhead(lambda: built.append(1) or "plan", depth=1, device=None)
        time.sleep(0.15)
        assert built == []                                        # parked at the gate
    assert ahead.get() == "plan"
    ahead.close()


def test_multi_dataset_loader_follows_the_reference_schedule():
    """MultiDatasetLoader = the batch schedule of pointcept/datasets/dataloader.py:23-102: ratios[i] consecutive batches from loader i per
    round, the FIRST loader ends the epoch, the others start over; len() as dataloader.py:97-102."""
    from scenesplat_amd.pointcept_api.engine import MultiDatasetLoader
    l = MultiDatasetLoader([list("abcde"), list("XY"), list("12345678")], [2, 1, 3])
    assert "".join(l) == "abX123cdY456e" and len(l) == 13 == 5 // 2 * 6 + 5 % 2
    assert "".join(l) == "abX123cdY456e"                        # a second epoch starts every loader afresh
    one = MultiDatasetLoader([list("ab")], [3])
    assert "".join(one) == "ab" and len(one) == 2
    exact = MultiDatasetLoader([list("abcd"), list("X")], [2, 2])
    assert "".join(exact) == "abXXcdXX" and len(exact) == 8     # the main loader runs out exactly at a round's end: the round completes
    with pytest.raises(ValueError):
        MultiDatasetLoader([list("ab")], [0])
