"""Engines that hold a range of the users (rec_model_desc.user_begin / user_end): only those users' rows, sequences,
click sets and paths are resident on the device.  Whatever such an engine computes for its users must be what the
whole-corpus engine computes for them.  Needs an MI355X."""
import numpy as np
import pytest

import recommendation_amd as ra
from recommendation_amd.capi import F32, F64, HOGWILD, SEQUENTIAL
from recommendation_amd.dist_util import shard_range
from recommendation_amd.sharded import user_table_slices
from tests.engine_util import make_engine
from tests.golden_util import fixture_case

pytestmark = pytest.mark.gpu

MODELS = ["BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "TransRec", "FossilSimple", "Fossil", "REBUS"]


def _owned_mask(e):
    mask = np.ones(e.nw, bool)
    off = 0
    for ptr, rows, cols, is_user in e.tables():
        if is_user:
            mask[off:off + rows * cols] = False
            mask[off + e.user_begin * cols: off + e.user_end * cols] = True
        off += rows * cols
    return mask


@pytest.mark.parametrize("model", MODELS)
def test_shard_engine_agrees_with_the_whole_engine_on_its_users(model):
    g, model, co, ci, labels = fixture_case(model, "toy")
    U = len(co) - 1
    whole = make_engine(g, model, co, ci, labels, F64)
    whole.init_params(5)
    Wi = whole.download()
    counts_all = None
    for r in range(3):
        ub, ue = shard_range(U, r, 3)
        e = make_engine(g, model, co, ci, labels, F64, user_begin=ub, user_end=ue)
        own = _owned_mask(e)
        # the counter-based initialisation gives a shard the values the whole table has in its rows
        e.init_params(5)
        sentinel = np.full(e.nw, -777.0)
        W = sentinel.copy()
        import ctypes as C
        from recommendation_amd.capi import _check, _p, lib
        _check(lib().rec_params_download(e.handle, _p(W), C.c_int64(len(W))))
        assert np.array_equal(W[own], Wi[own]) and np.all(W[~own] == -777.0)     # rows of other users are left alone
        # same parameters, same tuples of its own users -> same result as the whole engine
        e.upload(g["W0"])
        t = g["tuples"]
        mine = t[(t[:, 0] >= ub) & (t[:, 0] < ue)] if model != "MC" else t
        e.sgd_replay(float(g["lr"]), mine[:300], SEQUENTIAL)
        whole.upload(g["W0"])
        whole.sgd_replay(float(g["lr"]), mine[:300], SEQUENTIAL)
        assert np.array_equal(e.download()[own], whole.download()[own]), (model, r)
        # evaluation of its users
        whole.upload(g["W1"]); e.upload(g["W1"])
        _, c_all, _ = whole.eval()
        _, c_mine, _ = e.eval()
        assert np.array_equal(c_mine[ub:ue], c_all[ub:ue]) and np.all(c_mine[:ub] == -1) and np.all(c_mine[ue:] == -1)
        # a tuple of somebody else's user is refused before any kernel runs
        if model != "MC" and ub > 0:
            foreign = t[t[:, 0] < ub][:1]
            with pytest.raises(RuntimeError, match="another engine"):
                e.sgd_replay(float(g["lr"]), foreign, SEQUENTIAL)
        # the sampler draws its own users only (MC: its share of the source items)
        s = e.sample_tuples(4000, seed=1, epoch=2)
        if model == "MC":
            I = int(g["n_items"])
            assert s[:, 0].min() >= I * ub // U and s[:, 0].max() < max(1, I * ue // U)
        else:
            assert s[:, 0].min() >= ub and s[:, 0].max() < ue
        e.sgd_epoch(0.01, 2000, seed=3, epoch=1)
        e.sync()
        assert np.isfinite(e.download()[own]).all()
        e.close()
    whole.close()
