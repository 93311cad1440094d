"""The owner mode of the Hogwild update ("owner_neg": tuples partitioned by the slice of the catalog their negative
lies in, negatives' deltas summed in LDS accumulators): the partition itself, exactness where the reference defines
an exact answer, and conservation laws of the step on launches with heavily repeated negatives.  Needs an MI355X."""
import numpy as np
import pytest

import recommendation_amd as ra
from oracle import oracle_py as orc
from recommendation_amd.capi import F32, F64, HOGWILD
from tests.engine_util import make_engine, rel_err
from tests.golden_util import fixture_case, make_oracle

pytestmark = pytest.mark.gpu

OWNER_MODELS = ["BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "HRM_max", "TransRec", "TransRec_L1"]


def _corpus(U, I, rng):
    seq_off = np.arange(U + 1, dtype=np.int64) * 3
    seq_item = rng.integers(0, I, size=3 * U).astype(np.int32)
    return seq_off, seq_item, rng.integers(0, I, size=U).astype(np.int32), rng.integers(0, I, size=U).astype(np.int32)


@pytest.mark.parametrize("slices,tile", [(256, 0), (512, 0), (512, 4096), (1024, 0), (256, 7168)])
@pytest.mark.parametrize("model,n", [("BPRMF", 100_003), ("FPMC", 40_000), ("BPRMF", 17)])
def test_partition_orders_tuples_by_the_negatives_slice(model, n, slices, tile):
    rng = np.random.default_rng(3)
    U, I = 5000, 70_001
    seq_off, seq_item, test, val = _corpus(U, I, rng)
    e = ra.Engine(model, I, seq_off, seq_item, test, val, K=8, lam=0.01)
    t = np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)
    e.set_option("part_tile", tile)
    assert e.owner_slices == 256                                      # 274 items per slice fit the accumulators
    out, qoff = e.partition_tuples(t, slices)
    neg = out[:, 2] if model == "BPRMF" else out[:, 3]
    slice_rows = (I + slices - 1) // slices
    assert qoff[0] == 0 and qoff[slices] == n and np.all(np.diff(qoff) >= 0)
    sl = neg // slice_rows
    assert np.all(np.diff(sl) >= 0)                                   # grouped by slice, slices in order
    assert np.array_equal(np.searchsorted(sl, np.arange(slices + 1)), qoff)  # and the starts are where the groups begin
    # the same multiset of tuples
    key = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.array_equal(key(out), key(t))
    e.close()


@pytest.mark.parametrize("model", OWNER_MODELS)
@pytest.mark.parametrize("prec", [F32, F64])
def test_owner_mode_equals_sequential_without_collisions(model, prec):
    """Tuples that share no row: whatever the order the partition puts them in, the launch must equal the reference's
    in-order replay (TransRec: the sum of its single-step deltas, r being constant during a launch)."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    chosen, used_u, used_i = [], set(), set()
    for t in g["tuples"]:
        u = int(t[0]); items = [int(x) for x in (t[1:4] if model not in ("BPRMF", "MC") else t[1:3])]
        if model == "MC":
            items = [int(t[0])] + items; u = -1
        if len(set(items)) != len(items) or u in used_u or used_i & set(items):
            continue
        chosen.append(t); used_i |= set(items)
        if u >= 0:
            used_u.add(u)
    chosen = np.array(chosen, np.int32)
    assert len(chosen) >= 3
    o = make_oracle(g, model, co, ci, labels)
    W0 = g["W0"] if prec == F64 else g["W0"].astype(np.float32).astype(np.float64)
    if model.startswith("TransRec"):
        K, I = int(g["K"]), int(g["n_items"])
        W0 = W0.copy()
        H = W0[I:I + I * K].reshape(I, K)
        H /= np.maximum(1.0, np.sqrt((H * H).sum(1, keepdims=True)) * 1.0000001)
        if prec == F32:
            W0 = W0.astype(np.float32).astype(np.float64)
    lr = float(g["lr"])
    ref = o.replay(W0, chosen, lr)
    if model.startswith("TransRec"):
        ref = W0.copy()
        for t in chosen:
            ref += o.replay(W0, t[None, :], lr) - W0
    tol = 1e-10 if prec == F64 else 1e-5
    for flush_at in (1, 8):
        e = make_engine(g, model, co, ci, labels, prec)
        e.set_option("owner_neg", 1); e.set_option("owner_min", 1); e.set_option("owner_flush", flush_at)
        e.upload(W0)
        e.sgd_replay(lr, chosen, HOGWILD)
        W1 = e.download()
        assert rel_err(W1, ref, floor=1e-1) < tol, (model, flush_at, rel_err(W1, ref, floor=1e-1))
        e.close()


@pytest.mark.parametrize("model", ["BPRMF", "MC"])
@pytest.mark.parametrize("flush_at", [1, 8, 1000])
def test_owner_mode_conserves_what_the_step_conserves(model, flush_at):
    """Without regularisation a BPRMF / MC step moves the positive's and the negative's row by opposite amounts
    (BPRMF.cpp:97-103, MC.cpp:100-107), so the column sums of the item table do not change, whatever values the steps
    read - unless a delta is lost or applied twice.  200k tuples whose negatives come from 300 items only (every
    accumulator row is added to, flushed by count and at the end of the launch, hundreds of times), f64."""
    rng = np.random.default_rng(11)
    U, I, K, n = 200_000, 9_000, 16, 200_000
    seq_off, seq_item, test, val = _corpus(U, I, rng)
    e = ra.Engine(model, I, seq_off, seq_item, test, val, K=K, lam=0.0, bias_reg=0.0, precision=F64)
    e.set_option("update_policy", 0)          # float atomics on every row: no update is lost by a racing plain store
    e.set_option("owner_neg", 1); e.set_option("owner_min", 1); e.set_option("owner_flush", flush_at)
    o = orc.Oracle(model, U, I, K, seq_off, seq_item, test, val, lam=0.0, bias_reg=0.0)
    W0 = rng.uniform(-0.3, 0.3, o.nw)
    negs = rng.choice(I, 300, replace=False)
    first = rng.permutation(U)[:n] if model == "BPRMF" else rng.integers(0, I, n)
    t = np.stack([first, rng.integers(0, I, n), negs[rng.integers(0, 300, n)], np.zeros(n)], 1).astype(np.int32)
    e.upload(W0)
    e.sgd_replay(0.05, t, HOGWILD)
    W1 = e.download()
    assert np.isfinite(W1).all()
    if model == "BPRMF":
        items0, items1 = W0[I + U * K:].reshape(I, K), W1[I + U * K:].reshape(I, K)
        assert abs(W1[:I].sum() - W0[:I].sum()) < 1e-9                  # the biases move by +-lr*deri as well
    else:
        items0, items1 = W0[I * K:].reshape(I, K), W1[I * K:].reshape(I, K)   # eta; gamma holds the source items
    assert np.abs(items1 - items0).max() > 1e-3
    assert np.abs(items1.sum(0) - items0.sum(0)).max() < 1e-9, flush_at
    # only the 300 negatives and the drawn positives moved
    touched = np.zeros(I, bool); touched[t[:, 1]] = True; touched[t[:, 2]] = True
    assert np.array_equal(items1[~touched], items0[~touched])
    e.close()


def test_owner_mode_stays_close_to_the_sequential_replay():
    """Random tuples with repeated items on a small catalog: the owner-mode launch (stale reads bounded by the tuples in
    flight and by owner_flush) ends near the in-order replay, as the plain Hogwild launch does."""
    rng = np.random.default_rng(5)
    U, I, K, n = 4000, 3000, 16, 6000
    seq_off, seq_item, test, val = _corpus(U, I, rng)
    o = orc.Oracle("BPRMF", U, I, K, seq_off, seq_item, test, val, lam=0.01, bias_reg=0.01)
    W0 = rng.uniform(0, 0.5, o.nw)
    t = np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n), np.zeros(n)], 1).astype(np.int32)
    ref = o.replay(W0, t, 0.05)
    dist = {}
    for owner in (0, 1):
        e = ra.Engine("BPRMF", I, seq_off, seq_item, test, val, K=K, lam=0.01, bias_reg=0.01, precision=F64)
        e.set_option("update_policy", 0); e.set_option("stale_window", 0)
        e.set_option("owner_neg", owner); e.set_option("owner_min", 1)
        e.upload(W0)
        e.sgd_replay(0.05, t, HOGWILD)
        W1 = e.download()
        dist[owner] = float(np.abs(W1 - ref).max() / np.abs(ref - W0).max())
        e.close()
    # both differ from the replay only through stale reads of rows touched twice in flight (second order in lr)
    assert dist[0] < 0.2 and dist[1] < 0.2, dist


def _disjoint_tuples(model, U, I, n, rng):
    """n tuples that share no row: distinct users, every item named once over all roles."""
    items = rng.permutation(I)
    users = rng.permutation(U)[:n]
    if model in ("BPRMF",):
        return np.stack([users, items[:n], items[n:2 * n], np.zeros(n)], 1).astype(np.int32)
    if model == "MC":
        return np.stack([items[:n], items[n:2 * n], items[2 * n:3 * n], np.zeros(n)], 1).astype(np.int32)
    return np.stack([users, items[:n], items[n:2 * n], items[2 * n:3 * n]], 1).astype(np.int32)


@pytest.mark.parametrize("model", OWNER_MODELS)
@pytest.mark.parametrize("K", [8, 32, 64, 128])
def test_wide_kernel_equals_sequential_without_collisions(model, K):
    """The four-components-per-lane form of the update kernel (f32, K a multiple of 4), plain and owner mode, with
    every scatter policy: on tuples that share no row the launch must equal the reference's in-order replay within the
    f32 bound of criterion (a), 1e-5 (TransRec: the sum of its single-step deltas, r being constant during a launch)."""
    rng = np.random.default_rng(K)
    U, I = 3000, 7000
    n = 300 if model.startswith("TransRec") else 1500
    seq_off, seq_item, test, val = _corpus(U, I, rng)
    kw = dict(lam=0.03, bias_reg=0.02, relation_reg=0.04, alpha=0.3)
    o = orc.Oracle(model, U, I, K, seq_off, seq_item, test, val, L=1, **kw)
    W0 = rng.uniform(-0.4, 0.4, o.nw)
    if model.startswith("TransRec"):
        H = W0[I:I + I * K].reshape(I, K)
        H /= np.maximum(1.0, np.sqrt((H * H).sum(1, keepdims=True)) * 1.0001)
    W0 = W0.astype(np.float32).astype(np.float64)
    t = _disjoint_tuples(model, U, I, n, rng)
    ref = o.replay(W0, t, 0.05)
    if model.startswith("TransRec"):
        ref = W0.copy()
        for row in t:
            ref += o.replay(W0, row[None, :], 0.05) - W0
    for policy, owner in ((0, 0), (1, 0), (2, 0), (0, 1), (1, 1)):
        e = ra.Engine(model, I, seq_off, seq_item, test, val, K=K, L=1, precision=F32, **kw)
        e.set_option("update_policy", policy); e.set_option("stale_window", 0)
        e.set_option("global_cap", 1 << 20)   # TransRec: r takes the whole sum of the launch's gradients, as the expectation does
        e.set_option("owner_neg", owner); e.set_option("owner_min", 1); e.set_option("owner_flush", 3)
        e.upload(W0)
        e.sgd_replay(0.05, t, HOGWILD)
        W1 = e.download()
        err = rel_err(W1, ref, floor=1e-1)
        assert err < 1e-5, (model, K, policy, owner, err)
        e.close()


@pytest.mark.parametrize("model,K,I,slices", [("BPRMF", 64, 9_000, 256), ("MC", 32, 9_000, 256), ("BPRMF", 8, 9_000, 256),
                                              ("BPRMF", 64, 200_000, 512),      # two workgroups per CU, one after the other
                                              ("BPRMF", 64, 600_000, 1024)])    # 586 items per slice, accumulators for 520
def test_wide_owner_mode_conserves_what_the_step_conserves(model, K, I, slices):
    """The conservation law of test_owner_mode_conserves_what_the_step_conserves through the wide kernel (f32 tables:
    the column sums of the item table move by rounding only, a lost or doubled delta would move them by ~1e-2)."""
    rng = np.random.default_rng(13)
    U, n = 200_000, 200_000
    seq_off, seq_item, test, val = _corpus(U, I, rng)
    o = orc.Oracle(model, U, I, K, seq_off, seq_item, test, val, lam=0.0, bias_reg=0.0)
    W0 = rng.uniform(-0.3, 0.3, o.nw).astype(np.float32).astype(np.float64)
    negs = rng.choice(I, 300, replace=False)
    first = rng.permutation(U)[:n] if model == "BPRMF" else rng.integers(0, I, n)
    t = np.stack([first, rng.integers(0, I, n), negs[rng.integers(0, 300, n)], np.zeros(n)], 1).astype(np.int32)
    for flush_at in (1, 8, 1000):
        e = ra.Engine(model, I, seq_off, seq_item, test, val, K=K, lam=0.0, bias_reg=0.0, precision=F32)
        e.set_option("update_policy", 0)
        e.set_option("owner_neg", 1); e.set_option("owner_min", 1); e.set_option("owner_flush", flush_at)
        assert e.owner_slices == slices
        e.upload(W0)
        e.sgd_replay(0.05, t, HOGWILD)
        W1 = e.download()
        off = I + U * K if model == "BPRMF" else I * K
        items0, items1 = W0[off:].reshape(I, K), W1[off:].reshape(I, K)
        assert np.abs(items1 - items0).max() > 1e-3
        # f32 sums of ~700 additions per negative row: rounding ~1e-4 in a column sum
        assert np.abs(items1.sum(0) - items0.sum(0)).max() < 2e-3, (flush_at, np.abs(items1.sum(0) - items0.sum(0)).max())
        touched = np.zeros(I, bool); touched[t[:, 1]] = True; touched[t[:, 2]] = True
        assert np.array_equal(items1[~touched], items0[~touched])
        e.close()
