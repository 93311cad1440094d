"""rec_shards_begin_round / rec_shards_combine with one shard per DEVICE: table = base + sum of the shards' deltas through
one RCCL all-reduce per table over xGMI (librccl loaded at the first such call).  Runs where the process sees at least two
GPUs and is skipped on a one-GPU box; the same-device branch of the two calls is covered by tests/test_gpu_shards.py and
tests/test_gpu_cli.py.  (Collected after every other GPU test but the allocation-free entry point's.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _devices():
    from recommendation_amd.capi import lib
    return int(lib().rec_device_count())


@pytest.mark.parametrize("model", ["BPRMF", "TransRec"])
def test_shards_on_separate_devices_are_combined_over_rccl(model):
    import recommendation_amd as ra
    from oracle import oracle_py as orc
    from recommendation_amd.capi import F32, SEQUENTIAL, lib
    from recommendation_amd.dist_util import shard_range
    from tests.golden_util import fixture_case
    n = min(_devices(), 4)
    if n < 2:
        pytest.skip("needs at least two GPUs in one process")
    g, _, co, ci, labels = fixture_case(model, "ml5")
    seq_off, seq_item, test, val = orc.split(co, ci)
    U, I = len(seq_off) - 1, int(g["n_items"])
    engs = [ra.Engine(model, I, seq_off, seq_item, test, val, K=10, lam=0.05, bias_reg=0.01, relation_reg=0.05, precision=F32,
                      device=r, user_begin=shard_range(U, r, n)[0], user_end=shard_range(U, r, n)[1]) for r in range(n)]
    for e in engs:
        e.init_params(0)
    W0 = engs[0].download()
    arr = (C.c_void_p * n)(*[e.handle for e in engs])
    for rnd in range(3):
        assert lib().rec_shards_begin_round(arr, n) == 0, lib().rec_last_error()
        before = [e.download() for e in engs]
        for r, e in enumerate(engs):
            # deterministic local steps: the round's tuples applied in order
            e.sgd_replay(0.05, e.sample_tuples(2000, seed=1, epoch=rnd * n + r), SEQUENTIAL)
        after = [e.download() for e in engs]
        assert lib().rec_shards_combine(arr, n) == 0, lib().rec_last_error()
        got = [e.download() for e in engs]
        # item-indexed part of the flat vector: everything but the user tables' rows
        mask = np.ones(len(W0), bool)
        off = 0
        for ptr, rows, cols, is_user in engs[0].tables():
            if is_user:
                mask[off:off + rows * cols] = False
            off += rows * cols
        base = before[0][mask].astype(np.float32)
        want = base.copy()
        for r in range(n):
            want += after[r][mask].astype(np.float32) - base
        for r in range(n):
            assert np.array_equal(got[r][mask], got[0][mask])                         # the replicas agree exactly
        if model == "TransRec":                                                        # the combined rows are projected onto the unit ball once
            H = got[0][I:I + I * 10].reshape(I, 10)
            assert np.linalg.norm(H, axis=1).max() <= 1.0 + 1e-5
        else:
            assert np.allclose(got[0][mask], want, rtol=0, atol=2e-6)                 # f32 sums in the ring's order
    sums = sum(np.asarray(e.eval_sums(50)) for e in engs)
    assert sums[7] == (test >= 0).sum() and 0.4 < sums[1] / sums[7] < 1.0
    for e in engs:
        e.close()
