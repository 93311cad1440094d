"""rec_sgd_replay_static - the replay entry point that allocates nothing (tuples staged in a device array of the
library's code object) - against rec_sgd_replay on the same inputs: bit-identical parameters in the deterministic
mode, on two calls in a row, the second with fewer tuples.  Needs an MI355X."""
import numpy as np
import pytest

from recommendation_amd.capi import F32, F64, HOGWILD, SEQUENTIAL
from tests.engine_util import make_engine
from tests.golden_util import fixture_case, make_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,prec", [("BPRMF", F64), ("TransRec", F32), ("Fossil", F64), ("REBUS", F32)])
def test_static_replay_equals_replay_on_two_calls(model, prec):
    g, model, co, ci, labels = fixture_case(model, "toy")
    lr = float(g["lr"])
    first, second = g["tuples"][:1500], g["tuples"][1500:1900]   # the second call is the smaller one
    a = make_engine(g, model, co, ci, labels, prec)
    a.upload(g["W0"])
    a.sgd_replay(lr, first, SEQUENTIAL)
    wa1 = a.download()
    a.sgd_replay(lr, second, SEQUENTIAL)
    wa2 = a.download()
    a.close()
    b = make_engine(g, model, co, ci, labels, prec)
    b.upload(g["W0"])
    b.sgd_replay_static(lr, first, SEQUENTIAL)
    wb1 = b.download()
    b.sgd_replay_static(lr, second, SEQUENTIAL)
    wb2 = b.download()
    assert np.array_equal(wa1, wb1) and np.array_equal(wa2, wb2), model
    assert not np.array_equal(wb1, wb2) and not np.array_equal(g["W0"], wb1)
    if prec == F64:
        # and both are the reference's updateFactors on those 1900 tuples (restated by the CPU oracle)
        ref = make_oracle(g, model, co, ci, labels).replay(g["W0"], g["tuples"][:1900], lr)
        assert np.max(np.abs(wb2 - ref) / np.maximum(np.abs(ref), 1e-3)) < 1e-10
    b.close()


def test_static_replay_hogwild_and_limits():
    """Hogwild mode on tuples without shared rows (any order of the float atomics gives the same sums), and the
    size limit's error."""
    g, model, co, ci, labels = fixture_case("BPRMF", "toy")
    t = g["tuples"]
    keep, users, items = [], set(), set()
    for row in t:
        u, p, n = int(row[0]), int(row[1]), int(row[2])
        if u in users or p in items or n in items or p == n:
            continue
        users.add(u); items.update((p, n)); keep.append(row)
    keep = np.array(keep, np.int32)
    assert len(keep) >= 8
    outs = []
    for static in (False, True):
        e = make_engine(g, model, co, ci, labels, F64)
        e.upload(g["W0"])
        for part in (keep, keep[:3]):
            (e.sgd_replay_static if static else e.sgd_replay)(0.05, part, HOGWILD)
        outs.append(e.download())
        if static:
            with pytest.raises(RuntimeError, match="at most 65536"):
                e.sgd_replay_static(0.05, np.zeros((65537, 4), np.int32), SEQUENTIAL)
            with pytest.raises(RuntimeError, match="out of range"):
                e.sgd_replay_static(0.05, np.array([[10 ** 6, 0, 1, 0]], np.int32), SEQUENTIAL)
            with pytest.raises(RuntimeError, match="unknown replay mode"):
                e.sgd_replay_static(0.05, keep[:3], 7)
        e.close()
    assert np.array_equal(outs[0], outs[1])


def test_static_replay_leaves_the_device_allocator_alone():
    """Free device memory (hipMemGetInfo) is the same before and after calls of the allocation-free entry point, in both
    modes and for growing inputs - where rec_sgd_replay on the same fresh engine has to grow its tuple buffer."""
    import torch
    g, model, co, ci, labels = fixture_case("FPMC", "toy")
    e = make_engine(g, model, co, ci, labels, F32)
    e.upload(g["W0"])
    t = g["tuples"]
    e.sgd_replay_static(0.05, t[:8], SEQUENTIAL)       # first launches load the kernels' code
    e.sgd_replay_static(0.05, t[:8], HOGWILD)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for n in (16, 200, 2000):
        e.sgd_replay_static(0.05, t[:n], SEQUENTIAL)
        e.sgd_replay_static(0.05, t[:n], HOGWILD)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    e.close()
