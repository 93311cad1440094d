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


@pytest.mark.parametrize("world", [3, 4])
def test_sharded_transrec_projects_the_combined_table_once(world):
    """TransRec's projection of the item rows onto the unit ball (TransRec.cpp:245-258) is not additive.  Done by every
    shard after its launches it entered the sum of deltas once per shard: a row outside the ball was pulled back N times
    over, and on the bundled ML-5 corpus (K = 10, lr 0.05) the rows doubled every round from 3 shards on.  The shards'
    launches only add gradient steps; the combined table is projected once - by rec_shards_combine for shards inside one
    process, by ShardedTrainer / ShardGroup after their own all-reduce.  Both must train like one engine."""
    import ctypes as C
    from oracle import oracle_py as orc
    from recommendation_amd.capi import lib
    from recommendation_amd.sharded import ShardGroup
    g, model, co, ci, labels = fixture_case("TransRec", "ml5")
    seq_off, seq_item, test, val = orc.split(co, ci)
    U, I = len(seq_off) - 1, int(g["n_items"])

    def make(ub, ue):
        return ra.Engine("TransRec", I, seq_off, seq_item, test, val, K=10, lam=0.05, bias_reg=0.01, relation_reg=0.05,
                         precision=F32, user_begin=ub, user_end=ue)

    one = make(0, 0)
    one.init_params(0)
    for ep in range(40):
        one.sgd_epoch(0.05, one.steps_per_epoch, seed=0, epoch=ep)
    ref = one.eval()[0].auc_test
    one.close()
    assert ref > 0.7

    # (a) the C-ABI's in-process shards
    engs = [make(*shard_range(U, r, world)) for r in range(world)]
    for e in engs:
        e.init_params(0)
    arr = (C.c_void_p * world)(*[e.handle for e in engs])
    steps = engs[0].steps_per_epoch
    rounds = max(1, -(-steps // (2 * I)))
    per = steps // (world * rounds)
    for ep in range(40):
        for k in range(rounds):
            assert lib().rec_shards_begin_round(arr, world) == 0
            for r, e in enumerate(engs):
                e.sgd_epoch(0.05, per, seed=0, epoch=(ep * rounds + k) * world + r)
            assert lib().rec_shards_combine(arr, world) == 0
    Ws = [e.download() for e in engs]
    n_item = I + I * 10
    assert all(np.array_equal(Ws[0][:n_item], w[:n_item]) and np.array_equal(Ws[0][-10:], w[-10:]) for w in Ws[1:])   # replicas agree
    H = Ws[0][I:n_item].reshape(I, 10)
    assert np.isfinite(Ws[0]).all() and np.linalg.norm(H, axis=1).max() <= 1.0 + 1e-5
    sums = sum(np.asarray(e.eval_sums(50)) for e in engs)
    auc_c = sums[1] / sums[7]
    for e in engs:
        e.close()
    assert abs(auc_c - ref) < 0.012, (auc_c, ref)

    # (b) the Python trainer's emulation of N ranks on one device
    grp = ShardGroup(make, U, world, "cuda:0")
    grp.init_params(0)
    for ep in range(40):
        grp.epoch(0.05, seed=0, epoch=ep)
    auc_p = grp.metrics()[1]
    W = grp.gather_params()
    grp.close()
    assert np.isfinite(W).all() and abs(auc_p - ref) < 0.012, (auc_p, ref)


def test_launches_only_add_gradient_steps_when_the_projection_is_switched_off():
    """"project_after_launch" 0 (what sharded training sets): an update launch of TransRec must leave item rows that its
    steps have pushed out of the unit ball where they are - the projection of TransRec.cpp:245-258 is then the caller's,
    once, on the combined table.  With the option at 1 (the default) the same launch ends inside the ball."""
    from oracle import oracle_py as orc
    g, model, co, ci, labels = fixture_case("TransRec", "ml5")
    seq_off, seq_item, test, val = orc.split(co, ci)
    I = int(g["n_items"])
    norms = {}
    for opt in (1, 0):
        e = ra.Engine("TransRec", I, seq_off, seq_item, test, val, K=10, lam=0.0, bias_reg=0.0, relation_reg=0.0, precision=F32)
        e.set_option("project_after_launch", opt)
        e.init_params(0)                       # rows start ON the unit ball (TransRec.cpp:14-30): any step outward leaves it
        e.sgd_epoch(0.05, e.steps_per_epoch, seed=0, epoch=0)
        e.sync()
        W = e.download()
        norms[opt] = np.linalg.norm(W[I:I + I * 10].reshape(I, 10), axis=1).max()
        if opt == 0:
            e.project_rows()
            W = e.download()
            assert np.linalg.norm(W[I:I + I * 10].reshape(I, 10), axis=1).max() <= 1.0 + 1e-5
        e.close()
    assert norms[1] <= 1.0 + 1e-5, norms
    assert norms[0] > 1.0 + 1e-3, norms
