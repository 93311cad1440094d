"""Parity of the HIP path (through the C ABI) with the reference's outputs (tests/golden) and with
the CPU oracle.  Needs an MI355X."""
import numpy as np
import pytest

import recommendation_amd as ra
from oracle import oracle_py as orc
from recommendation_amd.capi import F32, F64, HOGWILD, SEQUENTIAL
from tests.engine_util import make_engine, rel_err
from tests.golden_util import ML5_TAGS, TOY_MODELS, fixture_case, make_oracle

pytestmark = pytest.mark.gpu

HOGWILD_MODELS = ["BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "HRM_max", "TransRec", "TransRec_L1"]
EVAL_MODELS = ["MP"] + TOY_MODELS


@pytest.mark.parametrize("model", TOY_MODELS)
def test_sequential_replay_f64_matches_reference(model):
    """Criterion (a): a fixed tuple sequence through the deterministic batch-1 mode; bound 1e-5
    relative per parameter, observed ~1e-13 in f64."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    e = make_engine(g, model, co, ci, labels, F64)
    assert e.nw == int(g["NW"])
    e.upload(g["W0"])
    assert np.array_equal(e.download(), g["W0"])
    e.sgd_replay(float(g["lr"]), g["tuples"], SEQUENTIAL)
    W1 = e.download()
    assert rel_err(W1, g["W1"]) < 1e-10, (model, rel_err(W1, g["W1"]))
    e.close()


@pytest.mark.parametrize("model", TOY_MODELS)
def test_sequential_replay_f32_bounds(model):
    """Criterion (a) with f32 tables: all 2000 fixture tuples (every row is rewritten ~100 times in place, each write
    rounded to f32) against the fp64 restatement started from the same f32-rounded values.  The 1e-5 relative bound per
    parameter is met by the f64 mode (test above, 1e-10).  With f32 storage it holds relative to max(|w|, 0.1) - an
    absolute error below 1e-6 on parameters of size 0.1 .. 1 (measured <= 2.7e-6 of that scale) - while the pure
    relative error of small parameters (1e-3 <= |w|, left over from cancellations) reaches 1e-4 after 2000 steps and
    stays below 2e-5 after 200: both are asserted."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    o = make_oracle(g, model, co, ci, labels)
    W0 = g["W0"].astype(np.float32).astype(np.float64)
    for n, rel_bound in ((200, 2e-5), (2000, 2e-4)):
        e = make_engine(g, model, co, ci, labels, F32)
        e.upload(g["W0"])
        e.sgd_replay(float(g["lr"]), g["tuples"][:n], SEQUENTIAL)
        W1 = e.download()
        ref = o.replay(W0, g["tuples"][:n], float(g["lr"]))
        big = np.abs(ref) >= 1e-3
        rel = float(np.max(np.abs(W1[big] - ref[big]) / np.abs(ref[big]))) if big.any() else 0.0
        small = float(np.max(np.abs(W1[~big] - ref[~big]))) if (~big).any() else 0.0
        assert rel_err(W1, ref, floor=1e-1) < 5e-6, (model, n, rel_err(W1, ref, floor=1e-1))
        assert rel < rel_bound and small < 2e-7, (model, n, rel, small)
        e.close()


@pytest.mark.parametrize("model", EVAL_MODELS)
def test_eval_toy_matches_reference(model):
    g, model, co, ci, labels = fixture_case(model, "toy")
    for prec in (F64,):
        e = make_engine(g, model, co, ci, labels, prec)
        if model != "MP":
            e.upload(g["W1"])
        m, counts, xpos = e.eval(want_xpos=True)
        assert np.array_equal(counts, g["counts"]), model
        assert np.allclose(xpos, g["xpos"], rtol=1e-12, atol=1e-13)
        assert np.allclose(m.as_array(), g["metrics"], rtol=0, atol=1e-15)
        e.close()


@pytest.mark.parametrize("tag", ["MP"] + ML5_TAGS)
def test_eval_saved_model_matches_reference(tag):
    """Criterion (b): a model file saved by the reference CLI, scored to AUC / HIT@50 within 1e-6
    (observed: identical per-user counts)."""
    g, model, co, ci, labels = fixture_case(tag, "ml5")
    e = make_engine(g, model, co, ci, labels, F64)
    if "W_micro" in g:
        e.upload(g["W_micro"].astype(np.float64) / 1e6)
    m, counts, _ = e.eval()
    assert np.abs(m.as_array() - g["metrics"]).max() < 1e-6
    assert np.array_equal(counts, g["counts"]), tag
    assert int(m.n_users) == int((g["counts"][:, 2] >= 0).sum())
    e.close()


def _check_topk(row, items, scores, k):
    """Same list as the oracle's up to permutations among (near-)equal scores."""
    order = np.lexsort((np.arange(len(row)), -row))[:k]
    n_real = int((row[order] > -np.inf).sum())
    assert (items[:n_real] >= 0).all() and (items[n_real:] == -1).all()
    got = items[:n_real]
    assert len(set(got.tolist())) == n_real
    tol = 1e-9 * max(1.0, float(np.abs(row[order[:n_real]]).max()))
    assert np.abs(row[got] - scores[:n_real]).max() <= tol          # the score reported is the item's score
    assert (np.diff(scores[:n_real]) <= 0).all()                      # best first
    kth = row[order[n_real - 1]]
    must = np.nonzero(row > kth + tol)[0]
    assert set(must.tolist()) <= set(got.tolist())
    assert (row[got] >= kth - tol).all()
    return np.array_equal(got, order[:n_real])


@pytest.mark.parametrize("tag", ["MP"] + ML5_TAGS)
def test_top50_lists_of_saved_model_match_oracle(tag):
    """Criterion (b), second half: the 50 best-ranked items per user and context are the oracle's,
    modulo ties."""
    g, model, co, ci, labels = fixture_case(tag, "ml5")
    W = g["W_micro"].astype(np.float64) / 1e6 if "W_micro" in g else None
    e = make_engine(g, model, co, ci, labels, F64)
    if W is not None:
        e.upload(W)
    o = make_oracle(g, model, co, ci, labels)
    valid = np.nonzero(np.asarray(g["counts"])[:, 2] >= 0)[0]
    users = valid[:: max(1, len(valid) // 24)][:24].astype(np.int32)
    items, scores = e.eval_topk(users, 50)
    _, counts, _ = e.eval()
    exact = 0
    for a, u in enumerate(users):
        for ctx in (0, 1):
            row = o.score_row(W, int(u), ctx)
            exact += _check_topk(row, items[a, ctx], scores[a, ctx], 50)
            # HIT@50 of model.cpp:52-58 is membership of the positive in this list
            pos = int(o.val_item[u] if ctx == 0 else o.test_item[u])
            if model != "MP":
                rank = counts[u, 2] - counts[u, ctx]
                assert (pos in items[a, ctx].tolist()) == (rank < 50)
    if model != "MP":   # MostPopular scores are integers: ties everywhere
        assert exact >= 0.9 * 2 * len(users)
    e.close()


@pytest.mark.parametrize("model", EVAL_MODELS)
def test_topk_toy_all_models_and_short_lists(model):
    g, model, co, ci, labels = fixture_case(model, "toy")
    e = make_engine(g, model, co, ci, labels, F64)
    W = g["W1"] if model != "MP" else None
    if W is not None:
        e.upload(W)
    o = make_oracle(g, model, co, ci, labels)
    users = np.nonzero(np.asarray(g["counts"])[:, 2] >= 0)[0][:12].astype(np.int32)
    for k in (1, 7, 256):   # 256 exceeds the toy catalog's candidates for some users
        items, scores = e.eval_topk(users, k)
        for a, u in enumerate(users):
            for ctx in (0, 1):
                _check_topk(o.score_row(W, int(u), ctx), items[a, ctx], scores[a, ctx], k)
    with pytest.raises(RuntimeError):
        e.eval_topk(users, 257)
    bad = np.nonzero(np.asarray(g["counts"])[:, 2] < 0)[0]
    if len(bad):
        with pytest.raises(RuntimeError):
            e.eval_topk(bad[:1].astype(np.int32), 5)
    e.close()


def _f32(W):
    return None if W is None else np.asarray(W, np.float64).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("tag", ML5_TAGS)
def test_fused_top50_of_saved_model_match_oracle(tag):
    """Criterion (b), second half, on the path that serves whole user populations: with f32 tables the lists come out of
    the f16 sweep itself (running k-th best per query, survivors rescored in fp64; no score matrix).  Every user of the
    reference-saved ML-5 model, both contexts, against the oracle's ranking of the same f32-rounded parameters, modulo
    ties; HIT@50 of model.cpp:52-58 is membership of the positive in the list."""
    g, model, co, ci, labels = fixture_case(tag, "ml5")
    W = _f32(g["W_micro"].astype(np.float64) / 1e6)
    e = make_engine(g, model, co, ci, labels, F32)
    e.upload(W)
    o = make_oracle(g, model, co, ci, labels)
    valid = np.nonzero(np.asarray(g["counts"])[:, 2] >= 0)[0].astype(np.int32)
    items, scores = e.eval_topk(len(valid), 50)            # users = NULL: everybody that has a test item
    _, lists = e.kernel_stats(5)
    assert lists == 2 * len(valid), (tag, lists)            # all of them through the fused sweep
    _, counts, _ = e.eval()
    rank = counts[valid, 2:3] - counts[valid, 0:2]          # [user][ctx]: candidates scored at least as high as the positive
    pos = np.stack([o.val_item[valid], o.test_item[valid]], 1)
    assert np.array_equal((items == pos[:, :, None]).any(2), rank < 50)
    exact = 0
    picked = valid[:: max(1, len(valid) // 96)][:96]
    for u in picked:
        a = int(np.searchsorted(valid, u))
        for ctx in (0, 1):
            exact += _check_topk(o.score_row(W, int(u), ctx), items[a, ctx], scores[a, ctx], 50)
    assert exact >= 0.9 * 2 * len(picked)
    # an explicit list of users gives the same lists, and so does the score-matrix path up to the order of ties
    sub = valid[5:400:7]
    i2, s2 = e.eval_topk(sub, 50)
    assert np.array_equal(i2, items[5:400:7]) and np.array_equal(s2, scores[5:400:7])
    e.set_option("topk_fused", 0)
    try:
        i3, s3 = e.eval_topk(sub, 50)
        assert e.kernel_stats(5)[1] == 0
    finally:
        e.set_option("topk_fused", 1)
    assert np.allclose(s3, s2, rtol=1e-9, atol=1e-12)
    assert (i3 == i2).mean() > 0.99
    e.close()


@pytest.mark.parametrize("model", TOY_MODELS)
def test_fused_topk_toy_all_models(model):
    g, model, co, ci, labels = fixture_case(model, "toy")
    W = _f32(g["W1"])
    e = make_engine(g, model, co, ci, labels, F32)
    e.upload(W)
    o = make_oracle(g, model, co, ci, labels)
    users = np.nonzero(np.asarray(g["counts"])[:, 2] >= 0)[0].astype(np.int32)
    for k in (1, 7, 50, 64):
        items, scores = e.eval_topk(users, k)
        fused = e.kernel_stats(5)[1]
        assert fused == 2 * len(users), (model, k, fused)   # TransRec_L1 too: its lists come out of the |.|_1 sweep
        for a, u in enumerate(users):
            for ctx in (0, 1):
                _check_topk(o.score_row(W, int(u), ctx), items[a, ctx], scores[a, ctx], k)
    e.close()


@pytest.mark.parametrize("prec", [F32, F64])
def test_l1_top50_out_of_the_sweep_match_oracle_and_score_matrix(prec):
    """TransRec_L1 on the ML-5 corpus (parameters: the reference-saved TransRec model, scored with |.|_1): the lists of
    every user come out of the |.|_1 sweep itself, equal the oracle's modulo ties and the score-matrix path's, and HIT@50
    of model.cpp:52-58 is membership of the positive."""
    g, model, co, ci, labels = fixture_case("TransRec", "ml5")
    W = g["W_micro"].astype(np.float64) / 1e6
    if prec == F32:
        W = _f32(W)
    e = make_engine(g, "TransRec_L1", co, ci, labels, prec)
    e.upload(W)
    o = make_oracle(g, "TransRec_L1", co, ci, labels)
    valid = np.nonzero(np.asarray(g["counts"])[:, 2] >= 0)[0].astype(np.int32)
    items, scores = e.eval_topk(len(valid), 50)
    assert e.kernel_stats(5)[1] == 2 * len(valid)
    # the counts behind AUC / HIT@50: the sweep on fixed-point operands (one v_sad_u32 per term, undecided columns in fp64),
    # the fp64 sweep, and the oracle's model::AUC give the same (count_val, count_test, max) for every user - also where
    # a candidate's score EQUALS the positive's (sums of |differences| of six-decimal parameters are multiples of 1e-6:
    # with f64 tables some queries have such a tie, and every score is summed in the reference's order, from beta, k ascending)
    _, counts, _ = e.eval()
    assert e.kernel_stats(4)[1] == 1
    e.set_option("eval_screen", 0)
    try:
        _, counts64, _ = e.eval()
        assert e.kernel_stats(4)[1] == 0
    finally:
        e.set_option("eval_screen", 1)
    assert np.array_equal(counts, counts64)
    _, cref = o.eval(W)
    assert np.array_equal(counts[valid], cref[valid])
    rank = counts[valid, 2:3] - counts[valid, 0:2]
    pos = np.stack([o.val_item[valid], o.test_item[valid]], 1)
    assert np.array_equal((items == pos[:, :, None]).any(2), rank < 50)
    exact = 0
    picked = valid[:: max(1, len(valid) // 96)][:96]
    for u in picked:
        a = int(np.searchsorted(valid, u))
        for ctx in (0, 1):
            exact += _check_topk(o.score_row(W, int(u), ctx), items[a, ctx], scores[a, ctx], 50)
    assert exact >= 0.9 * 2 * len(picked)
    sub = valid[3:900:5]
    i2, s2 = e.eval_topk(sub, 50)
    assert np.array_equal(i2, items[3:900:5]) and np.array_equal(s2, scores[3:900:5])
    # the lists above came out of the sweep on fixed-point operands (rescored in fp64); out of the fp64 sweep they are the same
    e.set_option("eval_screen", 0)
    try:
        i4, s4 = e.eval_topk(sub, 50)
        assert e.kernel_stats(5)[1] == 2 * len(sub)
    finally:
        e.set_option("eval_screen", 1)
    assert np.array_equal(i4, i2) and np.array_equal(s4, s2)
    e.set_option("topk_fused", 0)
    try:
        i3, s3 = e.eval_topk(sub, 50)
        assert e.kernel_stats(5)[1] == 0
    finally:
        e.set_option("topk_fused", 1)
    assert np.allclose(s3, s2, rtol=1e-12, atol=1e-12)
    assert (i3 == i2).mean() > 0.99
    # one item row far outside the range of the others: the fixed-point scale is too coarse to tell the others apart - counts
    # come from the fp64 sweep (the screen's lists overflow) and the lists from the fp64 sweep's exact lists, not the score matrix
    Wo = W.copy()
    Wo[o.n_items + 7 * int(g["K"]): o.n_items + 8 * int(g["K"])] = 3.0e7 if prec == F32 else 3.0e9      # row 7 of H
    e.upload(Wo)
    _, c_o, _ = e.eval()
    assert e.kernel_stats(4)[1] == 2
    _, cref = o.eval(Wo)
    assert np.array_equal(c_o[valid], cref[valid])
    few = valid[:40]
    i5, s5 = e.eval_topk(few, 50)
    assert e.kernel_stats(5)[1] == 2 * len(few)
    for a, u in enumerate(few[:8]):
        for ctx in (0, 1):
            _check_topk(o.score_row(Wo, int(u), ctx), i5[a, ctx], s5[a, ctx], 50)
    e.upload(W)
    # all-zero tables: every candidate ties, no list can hold them, the score-matrix path answers (lowest item ids first)
    Z = np.zeros(e.nw)
    e.upload(Z)
    users = valid[:48]
    items, scores = e.eval_topk(users, 50)
    assert e.kernel_stats(5)[1] == 0
    for a, u in enumerate(users):
        for ctx in (0, 1):
            assert _check_topk(o.score_row(Z, int(u), ctx), items[a, ctx], scores[a, ctx], 50)
    # ... and every column is undecided for the integer screen: the fp64 sweep answers, with the oracle's counts
    _, counts, _ = e.eval()
    assert e.kernel_stats(4)[1] == 2
    _, cref = o.eval(Z)
    assert np.array_equal(counts[valid], cref[valid])
    # a table entry that is not finite: the screen declines, the fp64 sweep counts as the oracle does
    Wn = W.copy(); Wn[-3] = np.inf
    e.upload(Wn)
    _, counts, _ = e.eval()
    assert e.kernel_stats(4)[1] == 2
    e.close()


def test_fused_topk_hands_ties_it_cannot_hold_to_the_score_matrix():
    """All-zero tables: every candidate ties with every other one, no list can hold them, and every user comes back from
    the score-matrix path (lowest item ids first) - with trained item biases on top, the lists hold again."""
    g, model, co, ci, labels = fixture_case("BPRMF", "ml5")
    e = make_engine(g, model, co, ci, labels, F32)
    W = np.zeros(e.nw)
    e.upload(W)
    o = make_oracle(g, model, co, ci, labels)
    users = np.nonzero(np.asarray(g["counts"])[:, 2] >= 0)[0][:64].astype(np.int32)
    items, scores = e.eval_topk(users, 50)
    assert e.kernel_stats(5)[1] == 0
    for a, u in enumerate(users):
        for ctx in (0, 1):
            assert _check_topk(o.score_row(W, int(u), ctx), items[a, ctx], scores[a, ctx], 50)   # ties in item order
    W[:o.n_items] = np.linspace(-1, 1, o.n_items)    # beta
    e.upload(W)
    items, scores = e.eval_topk(users, 50)
    assert e.kernel_stats(5)[1] == 2 * len(users)
    for a, u in enumerate(users):
        for ctx in (0, 1):
            assert _check_topk(o.score_row(_f32(W), int(u), ctx), items[a, ctx], scores[a, ctx], 50)
    e.close()


@pytest.mark.parametrize("model", EVAL_MODELS)
def test_sampled_eval_counts_match_oracle_on_the_same_draws(model):
    """rec_eval_sampled (the sampleAUC estimator on the device): for the item draws it reports, the per-user
    counts must be what model.cpp:119-140 computes on those draws - skipped when clicked / held out, not replaced."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    e = make_engine(g, model, co, ci, labels, F64)
    W = g["W1"] if model != "MP" else None
    if W is not None:
        e.upload(W)
    o = make_oracle(g, model, co, ci, labels)
    m, counts, sampled = e.eval_sampled(n_sampled=150, seed=9)
    assert len(sampled) == 150 and sampled.min() >= 0 and sampled.max() < int(g["n_items"])
    full = np.asarray(g["counts"])
    n_ok = 0
    for u in range(len(full)):
        if full[u, 2] < 0:
            assert (counts[u] == -1).all()
            continue
        rows = [o.score_row(W, u, 0), o.score_row(W, u, 1)]
        pos = [int(o.val_item[u]), int(o.test_item[u])]
        cand = [j for j in sampled.tolist() if np.isfinite(rows[0][j]) and j not in pos]
        if not cand:
            assert (counts[u] == -1).all()
            continue
        want = [sum(1 for j in cand if rows[c][pos[c]] > rows[c][j]) for c in (0, 1)]
        assert counts[u].tolist() == [want[0], want[1], len(cand)], (model, u)
        n_ok += 1
    assert n_ok > 10 and int(m.n_users) == n_ok
    # the estimator is unbiased for the mean: 40 draws of 150 against the exact value
    exact = e.eval()[0]
    est = np.mean([e.eval_sampled(150, seed=s)[0].auc_test for s in range(40)])
    assert abs(est - exact.auc_test) < 0.03, (est, exact.auc_test)
    e.close()


@pytest.mark.parametrize("prec", [F64, F32])
@pytest.mark.parametrize("model", EVAL_MODELS)
def test_per_user_sampled_eval_counts_match_oracle_on_the_same_draws(model, prec):
    """rec_eval_sampled_per_user (model::sampleAUC with the reference's sampling, model.cpp:119-140): every user has
    draws of its own; for the draws the engine reports, (count_val, count_test, max) must be what the reference's loop
    computes on them - a draw that hits a clicked or held-out item is skipped and not replaced."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    e = make_engine(g, model, co, ci, labels, prec)
    W = g["W1"] if model != "MP" else None
    if W is not None:
        e.upload(W)
        W = e.download()     # f32 tables: the oracle scores the rounded values
    o = make_oracle(g, model, co, ci, labels)
    n_draws = 90
    m, counts, draws = e.eval_sampled_per_user(n_sampled=n_draws, seed=11, want_draws=True)
    I = int(g["n_items"])
    full = np.asarray(g["counts"])
    n_ok, rows_seen = 0, set()
    for u in range(len(full)):
        if full[u, 2] < 0:
            assert (counts[u] == -1).all() and (draws[u] == -1).all()
            continue
        assert draws[u].min() >= 0 and draws[u].max() < I
        rows_seen.add(tuple(draws[u].tolist()))
        rows = [o.score_row(W, u, 0), o.score_row(W, u, 1)]
        pos = [int(o.val_item[u]), int(o.test_item[u])]
        cand = [j for j in draws[u].tolist() if np.isfinite(rows[0][j]) and j not in pos]
        if not cand:
            assert (counts[u] == -1).all()
            continue
        want = [sum(1 for j in cand if rows[c][pos[c]] > rows[c][j]) for c in (0, 1)]
        assert counts[u].tolist() == [want[0], want[1], len(cand)], (model, u)
        n_ok += 1
    assert n_ok > 10 and int(m.n_users) == n_ok
    assert len(rows_seen) == n_ok            # no two users share their draws
    # the same seed gives the same draws, another seed others
    again = e.eval_sampled_per_user(n_draws, seed=11, want_draws=True)
    other = e.eval_sampled_per_user(n_draws, seed=12, want_draws=True)
    assert np.array_equal(again[2], draws) and np.array_equal(again[1], counts) and not np.array_equal(other[2], draws)
    # draws are uniform over the catalog: every item is drawn, none far more often than the others
    hist = np.bincount(draws[draws >= 0], minlength=I)
    assert hist.min() > 0 and hist.max() < 2.5 * hist.mean()
    e.close()


@pytest.mark.parametrize("tag", ["BPRMF", "TransRec", "Fossil", "REBUS_mc2_L3"])
def test_per_user_sampled_validation_has_the_spread_of_the_reference(tag):
    """The reference's own sampleAUC on its saved ML-5 models, 24 consecutive calls (tests/golden/ml5_sampleauc.json,
    oracle/make_sampleauc.py): its item-sampling error averages out over the 6039 users (standard deviation about 1e-4).
    The per-user estimator here must show the same spread around the same exact value; one draw shared by all users
    (rec_eval_sampled) does not average out and is an order of magnitude noisier."""
    import json
    import os
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ml5_sampleauc.json")))[tag]
    g, model, co, ci, labels = fixture_case(tag, "ml5")
    e = make_engine(g, model, co, ci, labels, F64)
    e.upload(g["W_micro"].astype(np.float64) / 1e6)
    exact = e.eval()[0]
    assert abs(exact.auc_val - ref["exact_auc_val"]) < 1e-9 and abs(exact.auc_test - ref["exact_auc_test"]) < 1e-9
    per_user = np.array([[m.auc_val, m.auc_test, m.var_test] for m in (e.eval_sampled_per_user(2000, seed=s)[0] for s in range(24))])
    shared = np.array([[m.auc_val, m.auc_test] for m in (e.eval_sampled(2000, seed=s)[0] for s in range(24))])
    for c, key, ex in ((0, "auc_val", exact.auc_val), (1, "auc_test", exact.auc_test)):
        r = np.array(ref[key])
        sd_ref, sd = r.std(ddof=1), per_user[:, c].std(ddof=1)
        assert 0.5 * sd_ref < sd < 2.0 * sd_ref, (tag, key, sd, sd_ref)
        # both are unbiased for the exact value: means of 24 calls within 4 standard errors of it
        assert abs(per_user[:, c].mean() - ex) < 4 * sd_ref / np.sqrt(24) + 1e-5, (tag, key, per_user[:, c].mean(), ex)
        assert abs(r.mean() - ex) < 4 * sd_ref / np.sqrt(24) + 1e-5
        assert shared[:, c].std(ddof=1) > 3 * sd, (tag, key, shared[:, c].std(ddof=1), sd)
    # "Test var" of the printed validation line: the variance over users of the per-user estimates
    assert abs(per_user[:, 2].mean() - np.mean(ref["var_test"])) < 2e-4
    e.close()


@pytest.mark.parametrize("tag", ["BPRMF", "TransRec", "REBUS_mc2_L3"])
def test_eval_f32_storage_matches_oracle_on_same_values(tag):
    """f32 tables are scored in f64 after an exact widening, so the oracle fed the rounded values must agree."""
    g, model, co, ci, labels = fixture_case(tag, "ml5")
    W = (g["W_micro"].astype(np.float64) / 1e6).astype(np.float32).astype(np.float64)
    e = make_engine(g, model, co, ci, labels, F32)
    e.upload(W)
    m, counts, _ = e.eval()
    o = make_oracle(g, model, co, ci, labels)
    _, ref = o.eval(W)
    assert np.array_equal(counts, ref)
    e.close()


@pytest.mark.parametrize("model", HOGWILD_MODELS)
@pytest.mark.parametrize("prec", [F32, F64])
def test_hogwild_equals_sequential_without_collisions(model, prec):
    """Tuples that share no row: the Hogwild launch must equal the in-order replay."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    tuples = g["tuples"]
    chosen, used_u, used_i = [], set(), set()
    for t in tuples:
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
        # item rows start inside the unit ball, as they do after TransRec::init and after every step
        K, I = int(g["K"]), int(g["n_items"])
        W0 = W0.copy()
        H = W0[I:I + I * K].reshape(I, K)
        H /= np.maximum(1.0, np.sqrt((H * H).sum(1, keepdims=True)) * 1.0000001)
        if prec == F32:
            W0 = W0.astype(np.float32).astype(np.float64)
    ref = o.replay(W0, chosen, float(g["lr"]))
    e = make_engine(g, model, co, ci, labels, prec)
    e.upload(W0)
    e.sgd_replay(float(g["lr"]), chosen, HOGWILD)
    W1 = e.download()
    if model.startswith("TransRec"):
        # the global translation r is read-only during a launch and receives the summed gradient
        # afterwards, i.e. every tuple sees the initial r: the exact expectation is the sum of the
        # reference's single-step deltas, each taken from W0 (the rows are disjoint)
        ref = W0.copy()
        for t in chosen:
            ref += o.replay(W0, t[None, :], float(g["lr"])) - W0
    tol = 1e-10 if prec == F64 else 1e-5
    assert rel_err(W1, ref, floor=1e-1) < tol, (model, rel_err(W1, ref, floor=1e-1))
    e.close()


RUN_MODELS = ["BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "HRM_max"]


def _synthetic_corpus(U, I, rng):
    """Every user: 3 training items + validation + test item (contents are irrelevant to a replay)."""
    seq_off = np.arange(U + 1, dtype=np.int64) * 3
    seq_item = rng.integers(0, I, size=3 * U).astype(np.int32)
    return seq_off, seq_item, rng.integers(0, I, size=U).astype(np.int32), rng.integers(0, I, size=U).astype(np.int32)


@pytest.mark.parametrize("model", RUN_MODELS)
@pytest.mark.parametrize("prec", [F32, F64])
def test_run_combining_update_equals_sequential(model, prec):
    """Tuples grouped by positive item, every other row named once: the run-combining kernel keeps the
    positive's rows in registers across a run, so the launch must equal the reference's in-order replay.
    400 runs of 128 tuples: several lane-group segments, runs crossing batch boundaries."""
    rng = np.random.default_rng(5)
    n_runs, run = 400, 128
    n = n_runs * run
    U, I, K = n, n_runs + 2 * n, 8
    seq_off, seq_item, test, val = _synthetic_corpus(U, I, rng)
    pos = np.repeat(np.arange(n_runs), run)
    neg = n_runs + np.arange(n)
    prev = n_runs + n + np.arange(n)
    users = rng.permutation(U)
    if model == "BPRMF":
        tuples = np.stack([users, pos, neg, np.zeros(n)], 1)
    elif model == "MC":
        tuples = np.stack([prev, pos, neg, np.zeros(n)], 1)
    else:
        tuples = np.stack([users, prev, pos, neg], 1)
    tuples = tuples.astype(np.int32)
    kw = dict(K=K, L=1, lam=0.05, bias_reg=0.01, relation_reg=0.0, alpha=0.3)
    o = orc.Oracle(model, U, I, K, seq_off, seq_item, test, val, L=1, lam=0.05, bias_reg=0.01, alpha=0.3)
    W0 = rng.uniform(-0.5, 0.5, o.nw)
    if prec == F32:
        W0 = W0.astype(np.float32).astype(np.float64)
    ref = o.replay(W0, tuples, 0.05)
    e = ra.Engine(model, I, seq_off, seq_item, test, val, precision=prec, **kw)
    e.set_option("update_policy", 0)
    e.set_option("group_by_item", 1)
    e.set_option("stale_window", 0)   # one launch: segments stay aligned with the runs
    e.set_option("run_segment", 64)   # 64 batches of 4 (or 2) tuples per lane group: whole runs
    e.upload(W0)
    e.sgd_replay(0.05, tuples, HOGWILD)
    W1 = e.download()
    tol = 1e-10 if prec == F64 else 2e-5   # 128 dependent f32 steps per positive row
    assert rel_err(W1, ref, floor=1e-1) < tol, (model, rel_err(W1, ref, floor=1e-1))
    # the same launch on shuffled tuples (runs of length one, stale reads) stays close but is not the replay
    e.upload(W0)
    e.sgd_replay(0.05, tuples[rng.permutation(n)], HOGWILD)
    assert np.isfinite(e.download()).all()
    e.close()


@pytest.mark.parametrize("model", RUN_MODELS)
def test_event_major_sampler_matches_tuple_major_distribution(model):
    """The event-major sampler (Poisson count per sampling event, output grouped by positive item) must
    draw from the distribution of the tuple-major sampler: same support, grouped output, Poisson total,
    and matching frequencies of owners, positives and (owner, previous, positive) events."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    e = make_engine(g, model, co, ci, labels, F32)
    n = 400000
    a = e.sample_tuples(n, seed=4, epoch=2)
    b = e.sample_events(n, seed=4, epoch=2)
    assert abs(len(b) - n) < 6 * n ** 0.5
    col = 1 if model in ("BPRMF", "MC") else 2
    assert (np.diff(b[:, col]) >= 0).all()                     # grouped by positive item
    again = e.sample_events(n, seed=4, epoch=2)
    assert np.array_equal(b, again)                            # a function of (seed, epoch) only
    assert not np.array_equal(b[:1000], e.sample_events(n, seed=5, epoch=2)[:1000])
    ncol = 2 if model in ("BPRMF", "MC") else 3                # the columns that identify the event
    def freq(x):
        keys, cnt = np.unique(x[:, :ncol], axis=0, return_counts=True)
        return {tuple(k): c for k, c in zip(keys.tolist(), cnt.tolist())}
    fa, fb = freq(a), freq(b)
    assert set(fb) <= set(fa) or len(set(fb) - set(fa)) < 0.01 * len(fb)   # rare events may be missing from one draw
    ratio = len(a) / len(b)                                     # the event-major total is Poisson, not fixed
    z = []
    for k in set(fa) | set(fb):
        ca, cb = fa.get(k, 0), fb.get(k, 0) * ratio
        z.append((ca - cb) / max(1.0, (ca + cb)) ** 0.5)      # difference of two counts with equal means
    z = np.array(z)
    assert abs(z.mean()) < 5 / len(z) ** 0.5 and 0.8 < z.std() < 1.2 and np.abs(z).max() < 6, (z.mean(), z.std(), np.abs(z).max(), len(z))
    # negatives: same support rule and the same marginal
    neg_a, neg_b = a[:, ncol], b[:, ncol]
    I = int(g["n_items"])
    ha, hb = np.bincount(neg_a, minlength=I), np.bincount(neg_b, minlength=I)
    zz = (ha - hb) / np.maximum(1.0, ha + hb) ** 0.5
    assert np.abs(zz).max() < 6
    e.close()


@pytest.mark.parametrize("model", RUN_MODELS)
def test_event_major_epochs_train_like_tuple_major_epochs(model):
    """Epochs through the event-major sampler and the run-combining kernel (the large-problem path,
    forced here on the toy corpus) reach the validation AUC of the default path."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    res = {}
    for grp in (0, 2):
        e = make_engine(g, model, co, ci, labels, F32)
        e.set_option("group_by_item", grp)
        if grp == 2:
            e.set_option("stale_window", 0)
        e.init_params(3)
        for ep in range(40):
            e.sgd_epoch(0.05, e.steps_per_epoch, seed=11, epoch=ep)
        e.sync()
        assert np.isfinite(e.download()).all()
        if grp == 2:
            drawn, want = e.tuples_drawn, 40 * e.steps_per_epoch
            assert abs(drawn - want) < 6 * want ** 0.5
        res[grp] = e.eval()[0].auc_val
        e.close()
    assert abs(res[0] - res[2]) < 0.03, res

@pytest.mark.parametrize("model", ["FossilSimple", "Fossil", "REBUS"])
def test_history_hogwild_single_tuples(model):
    """One tuple per launch: the Hogwild kernels of the history models against updateFactors.
    When the tuple's history holds no repeated item and its short-term items do not alias the
    positive / negative the two are the same arithmetic; otherwise the reference's second in-place
    update of a row sees the first one (an O(lr^2) difference the parallel form does not reproduce)."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    o = make_oracle(g, model, co, ci, labels)
    seq_off, seq_item, test, val = orc.split(co, ci)
    L = int(g["L"])
    e = make_engine(g, model, co, ci, labels, F64)
    e.set_option("update_policy", 0)
    n_clean = n_alias = 0
    for t in g["tuples"][:120]:
        u = int(t[0]); hist = seq_item[seq_off[u]:seq_off[u + 1]].tolist()
        if model == "FossilSimple":
            used, pos, neg, prev = hist, int(t[2]), int(t[3]), [int(t[1])]
        elif model == "Fossil":
            tp = int(t[1]); used, pos, neg, prev = hist, hist[tp], int(t[2]), hist[max(0, tp - L):tp]
        else:
            tp = int(t[1]) + 1; used, pos, neg, prev = hist[:tp], hist[tp], int(t[2]), []
        clean = (len(set(used)) == len(used) or model == "REBUS") and pos not in prev and neg not in prev
        ref = o.replay(g["W0"], t[None, :], float(g["lr"]))
        e.upload(g["W0"])
        e.sgd_replay(float(g["lr"]), t[None, :], HOGWILD)
        W1 = e.download()
        err = rel_err(W1, ref, floor=1e-1)
        if clean:
            n_clean += 1
            assert err < 1e-9, (model, t, err)
        else:
            n_alias += 1
            assert err < 6e-3, (model, t, err)   # two rounds of eight history rows are in flight: a repeated item is read before its first update
    assert n_clean >= 10 and (n_alias >= 1 or model == "REBUS")
    e.close()


@pytest.mark.parametrize("G", [2, 4, 8])
@pytest.mark.parametrize("model", ["FossilSimple", "Fossil", "REBUS"])
def test_user_batched_history_kernel_sums_the_single_tuple_steps(model, G):
    """Option user_batch: the sampler gives every aligned group of G steps one user, and the batched kernel
    applies to the parameters it found the sum of the G single-tuple updateFactors deltas (each taken from
    those same parameters), walking the shared history once."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    o = make_oracle(g, model, co, ci, labels)
    seq_off, seq_item, test, val = orc.split(co, ci)
    L = int(g["L"]); lr = float(g["lr"]); W0 = g["W0"]
    e = make_engine(g, model, co, ci, labels, F64)
    e.set_option("update_policy", 0)
    T1 = e.sample_tuples(4096, seed=5, epoch=2)
    e.set_option("user_batch", G)
    e.set_option("pair_min_history", 0)     # the toy users are short: shared walks for them too (see the test below)
    T = e.sample_tuples(4096, seed=5, epoch=2)
    assert (T[:, 0].reshape(-1, G) == T[::G, 0][:, None]).all()          # groups share their user
    R = T.reshape(-1, G, 4)
    assert (R[:, 1:, 1:] != R[:, :1, 1:]).any(axis=(1, 2)).mean() > 0.5   # ... but not the rest of the draw
    assert len(np.unique(T[::G, 0])) > 0.5 * len(np.unique(T1[:, 0]))       # the group users are spread like single draws

    def aliased(t):
        u = int(t[0]); hist = seq_item[seq_off[u]:seq_off[u + 1]].tolist()
        if model == "FossilSimple":
            used, pos, neg, prev = hist, int(t[2]), int(t[3]), [int(t[1])]
        elif model == "Fossil":
            tp = int(t[1]); used, pos, neg, prev = hist, hist[tp], int(t[2]), hist[max(0, tp - L):tp]
        else:
            return False
        return len(set(used)) != len(used) or pos in prev or neg in prev

    n_clean = n_alias = 0
    for k in range(0, 60 * G, G):
        grp = T[k:k + G]
        if k == 0:   # a short group (the tail of a launch)
            grp = grp[:G - 1]
        ref = W0.copy()
        for t in grp:
            ref += o.replay(W0, t[None, :], lr) - W0
        e.upload(W0)
        e.sgd_replay(lr, grp, HOGWILD)
        err = rel_err(e.download(), ref, floor=1e-1)
        if any(aliased(t) for t in grp):
            n_alias += 1
            assert err < 6e-3 * G, (model, G, k, err)   # the O(lr^2) effect of repeated items, once per tuple of the group
        else:
            n_clean += 1
            assert err < 1e-9, (model, G, k, err)
    assert n_clean >= 5
    # tuples that are not grouped by user are walked one by one: two tuples of users without a common item
    a = T[0]; hist_a = set(seq_item[seq_off[a[0]]:seq_off[a[0] + 1]].tolist()) | {int(x) for x in a[1:]}
    for b in T:
        hist_b = set(seq_item[seq_off[b[0]]:seq_off[b[0] + 1]].tolist()) | {int(x) for x in b[1:]}
        if b[0] != a[0] and not (hist_a & hist_b) and not aliased(a) and not aliased(b):
            mixed = np.stack([a, b])
            ref = W0 + (o.replay(W0, a[None, :], lr) - W0) + (o.replay(W0, b[None, :], lr) - W0)
            e.upload(W0)
            e.sgd_replay(lr, mixed, HOGWILD)
            assert rel_err(e.download(), ref, floor=1e-1) < 1e-9
            break
    e.close()


@pytest.mark.parametrize("G", [2, 4])
@pytest.mark.parametrize("model", ["FossilSimple", "Fossil", "REBUS"])
def test_groups_of_a_short_user_are_applied_one_after_the_other(model, G):
    """A user with fewer than "pair_min_history" (default 8) training clicks never shares a walk: stepping G nearly
    identical tuples from the same parameters multiplies the step.  The kernel applies such a group tuple by tuple, each
    seeing the one before it: the group equals G launches of one tuple each (REBUS: to rounding; Fossil / FossilSimple fold
    the gradient of their global weights once per launch, which moves the second tuple by O(lr^2))."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    o = make_oracle(g, model, co, ci, labels)
    seq_off, _, _, _ = orc.split(co, ci)
    lr = float(g["lr"]); W0 = g["W0"]
    e = make_engine(g, model, co, ci, labels, F64)
    e.set_option("update_policy", 0)
    e.set_option("user_batch", G)
    T = e.sample_tuples(96 * G, seed=8, epoch=1)
    worst_seq = worst_sum = 0.0
    n = 0
    for k in range(0, 96 * G, G):
        grp = T[k:k + G]
        if seq_off[grp[0, 0] + 1] - seq_off[grp[0, 0]] >= 8:
            continue
        n += 1
        e.upload(W0)
        e.sgd_replay(lr, grp, HOGWILD)
        got = e.download()
        e.upload(W0)
        for t in grp:
            e.sgd_replay(lr, t[None, :], HOGWILD)
        one_by_one = e.download()
        summed = W0 + sum(o.replay(W0, t[None, :], lr) - W0 for t in grp)
        worst_seq = max(worst_seq, rel_err(got, one_by_one, floor=1e-1))
        worst_sum = max(worst_sum, rel_err(got, summed, floor=1e-1))
    assert n >= 20
    # measured: REBUS identical; FossilSimple / Fossil 5e-4 / 3e-4 (G = 2), 2.8e-3 / 1.5e-3 (G = 4); against the summed step 0.05 .. 0.33
    assert worst_seq < (1e-9 if model == "REBUS" else 1e-3 * G), (model, G, worst_seq)
    assert worst_sum > 1e-2 and worst_sum > 10 * worst_seq, (model, G, worst_sum, worst_seq)    # ... and not the sum of steps taken from the same parameters
    e.close()


@pytest.mark.parametrize("model", TOY_MODELS)
def test_hogwild_epoch_moves_towards_the_data(model):
    """A few device epochs (sampler + Hogwild update, both precisions) keep the parameters finite
    and improve the validation AUC of the toy corpus over its initial value."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    for prec in (F32, F64):
        e = make_engine(g, model, co, ci, labels, prec)
        e.init_params(3)
        m0, _, _ = e.eval()
        for ep in range(30):
            e.sgd_epoch(0.05, e.steps_per_epoch, seed=11, epoch=ep)
        e.sync()
        W = e.download()
        assert np.isfinite(W).all()
        m1, counts, _ = e.eval()
        assert m1.auc_val > m0.auc_val, (model, prec, m0.auc_val, m1.auc_val)
        _, ref = make_oracle(g, model, co, ci, labels).eval(W)
        assert np.array_equal(counts, ref)
        e.close()


@pytest.mark.parametrize("G", [2, 8])
@pytest.mark.parametrize("model", ["FossilSimple", "Fossil", "REBUS"])
def test_user_batched_epochs_train(model, G):
    """Device epochs with user_batch G (chunks and global-row launches cut on group boundaries, the last
    group of a launch short) keep the parameters finite and improve the validation AUC of the toy corpus."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    e = make_engine(g, model, co, ci, labels, F32)
    e.set_option("user_batch", G)
    e.set_option("chunk", 1001)          # not a multiple of G: rounded up inside
    e.set_option("global_window", 333)
    e.init_params(3)
    m0, _, _ = e.eval()
    n = e.steps_per_epoch + 3
    for ep in range(30):
        e.sgd_epoch(0.02, n, seed=11, epoch=ep)   # a small step: on histories this short the tuples of a group mostly coincide
    e.sync()
    assert e.tuples_drawn == 30 * n
    W = e.download()
    assert np.isfinite(W).all()
    m1, counts, _ = e.eval()
    assert m1.auc_val > m0.auc_val, (model, G, m0.auc_val, m1.auc_val)
    _, ref = make_oracle(g, model, co, ci, labels).eval(W)
    assert np.array_equal(counts, ref)
    e.close()


@pytest.mark.parametrize("share", [2, 4])
@pytest.mark.parametrize("policy", ["atomic", "hybrid"])
@pytest.mark.parametrize("model", ["BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "HRM_max"])
def test_steps_sharing_a_negative_are_scattered_once_with_the_sum(model, policy, share):
    """Option share_negative: 2 or 4 consecutive steps take the first candidate of one stream that their own
    tuple allows (nearly always the same item), and the update kernel folds the rows (and biases) the tuples of a
    batch share into one scatter: the result is the sum of the single-tuple updateFactors deltas."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    o = make_oracle(g, model, co, ci, labels)
    lr = float(g["lr"]); W0 = g["W0"]
    e = make_engine(g, model, co, ci, labels, F64)
    if policy == "atomic":
        e.set_option("update_policy", 0)
    else:   # user rows and biases stored plainly, item rows atomic, no item exempt
        e.set_option("update_policy", 1)
        e.set_option("hot_count", 0)
    four = model in ("FPMC", "PRME", "HRM_avg", "HRM_max")
    nf = 3 if four else 2            # item fields of a tuple
    T1 = e.sample_tuples(8192, seed=5, epoch=2)
    e.set_option("share_negative", share)
    T = e.sample_tuples(8192, seed=5, epoch=2)
    assert np.array_equal(T[:, :nf], T1[:, :nf])             # only the negatives changed
    neg = T[:, nf]
    # groups share their negative (always, but for candidates one of them has clicked: many on this tiny catalog) ...
    assert (neg.reshape(-1, share) == neg[::share, None]).all(axis=1).mean() > (0.6 if share == 2 else 0.3)
    assert len(np.unique(neg[::share])) > 0.2 * len(np.unique(neg))   # ... and different groups draw different ones
    done = 0
    for k in range(0, len(T), share):
        grp = T[k:k + share]
        if len(set(grp[:, nf].tolist())) != 1 or len(set(grp[:, 0].tolist())) != share:
            continue   # wanted: one negative, different owners (their rows are plain stores in the hybrid policy)
        if any(len(set(t[1:1 + nf].tolist())) < nf for t in grp):
            continue   # a tuple that names one item twice: the reference's in-place second update sees the first
        if policy == "hybrid" and model == "BPRMF" and set(grp[:, 1].tolist()) & set(grp[:, 2].tolist()):
            continue   # one tuple's positive is another's negative: plain bias stores of different roles to one address
        ref = W0.copy()
        for t in grp:
            ref += o.replay(W0, t[None, :], lr) - W0
        e.upload(W0)
        e.sgd_replay(lr, grp, HOGWILD)
        err = rel_err(e.download(), ref, floor=1e-1)
        assert err < 1e-9, (model, policy, grp, err)
        done += 1
        if done == 40:
            break
    assert done >= 10
    e.close()


@pytest.mark.parametrize("model", ["BPRMF", "MC", "FPMC", "PRME", "HRM_avg"])
def test_pairs_sharing_their_positive_keep_the_reference_distribution(model):
    """Option share_positive 2: step 2p+1 is another event of step 2p's positive item, drawn with the events' own
    weights.  Every tuple lies in the support of the reference's sampler, the two steps of a pair share the positive
    (and nearly always the negative), and the second steps visit the (owner, event) pairs as often as the first."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    e = make_engine(g, model, co, ci, labels, F32)
    seq_off, seq_item, test, val = orc.split(co, ci)
    tr = [seq_item[seq_off[u]:seq_off[u + 1]].tolist() for u in range(len(co) - 1)]
    succ = {}
    for s_ in tr:
        for a, b in zip(s_[:-1], s_[1:]):
            succ.setdefault(a, set()).add(b)
    e.set_option("share_positive", 2)
    n = 400000
    T = e.sample_tuples(n, seed=7, epoch=3)
    assert np.array_equal(T, e.sample_tuples(n, seed=7, epoch=3))
    three = model in ("BPRMF", "MC")
    pos = T[:, 1] if three else T[:, 2]
    neg = T[:, 2] if three else T[:, 3]
    assert (pos[0::2] == pos[1::2]).all()
    assert (neg[0::2] == neg[1::2]).mean() > 0.5
    for a, b, c, d in T[:6000].tolist():
        if model == "BPRMF":
            assert b in tr[a] and c not in tr[a]
        elif model == "MC":
            assert b in succ[a] and c not in succ[a]
        else:
            assert any(x == b and y == c for x, y in zip(tr[a][:-1], tr[a][1:])) and d != c
    # the same events at the same rates in both halves of the pairs: compare the histograms of (owner, aux, positive)
    key = T[:, 0].astype(np.int64) * 1000003 + (T[:, 1].astype(np.int64) if three else T[:, 1].astype(np.int64) * 1009 + T[:, 2])
    ka, ca = np.unique(key[0::2], return_counts=True)
    kb, cb = np.unique(key[1::2], return_counts=True)
    allk = np.union1d(ka, kb)
    fa = np.zeros(len(allk)); fa[np.searchsorted(allk, ka)] = ca
    fb = np.zeros(len(allk)); fb[np.searchsorted(allk, kb)] = cb
    z = (fa - fb) / np.sqrt(np.maximum(fa + fb, 1.0))
    assert np.abs(z).max() < 6.0 and abs(z.mean()) < 0.2, (np.abs(z).max(), z.mean())
    e.close()


@pytest.mark.parametrize("model", TOY_MODELS)
def test_sampler_support(model):
    """Every drawn tuple lies in the support of the reference's sampler for that model."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    e = make_engine(g, model, co, ci, labels, F32)
    seq_off, seq_item, test, val = orc.split(co, ci)
    I = int(g["n_items"]); L = int(g["L"])
    tr = [seq_item[seq_off[u]:seq_off[u + 1]].tolist() for u in range(len(co) - 1)]
    T = e.sample_tuples(20000, seed=7, epoch=3)
    assert np.array_equal(T, e.sample_tuples(20000, seed=7, epoch=3))       # counter-based: reproducible
    assert not np.array_equal(T, e.sample_tuples(20000, seed=7, epoch=4))
    succ = {}
    for s in tr:
        for a, b in zip(s[:-1], s[1:]):
            succ.setdefault(a, set()).add(b)
    for a, b, c, d in T[:4000].tolist():
        if model == "BPRMF":
            assert b in tr[a] and c not in tr[a]
        elif model == "MC":
            assert b in succ[a] and c not in succ[a]
        elif model in ("FPMC", "PRME", "HRM_avg", "HRM_max", "FossilSimple"):
            assert any(x == b and y == c for x, y in zip(tr[a][:-1], tr[a][1:])) and d != c
            if model == "FossilSimple":
                assert d != b
            if model.startswith("HRM"):
                assert len(set(tr[a])) >= 2
        elif model.startswith("TransRec"):
            assert any(x == b and y == c for x, y in zip(tr[a][:-1], tr[a][1:])) and d not in tr[a]
        elif model == "Fossil":
            assert 1 <= b < len(tr[a]) and c != tr[a][b] and c not in tr[a][max(0, b - L):b]
        elif model == "REBUS":
            assert 0 <= b < len(tr[a]) - 1 and c != tr[a][b + 1]
    # users are drawn uniformly among the eligible ones
    if model != "MC":
        users, cnt = np.unique(T[:, 0], return_counts=True)
        assert cnt.max() < 2.0 * cnt.mean() and cnt.min() > 0.5 * cnt.mean()
    e.close()


def test_errors_are_reported_not_thrown():
    g, model, co, ci, labels = fixture_case("BPRMF", "toy")
    e = make_engine(g, model, co, ci, labels, F32)
    with pytest.raises(RuntimeError, match="NW not match"):
        e.upload(np.zeros(3))
    bad = np.array([[0, 10 ** 6, 0, 0]], np.int32)
    with pytest.raises(RuntimeError, match="out of range"):
        e.sgd_replay(0.05, bad, SEQUENTIAL)
    e.close()


def _traj_groups():
    import json
    import os
    from tests.golden_util import GOLD
    rows = json.load(open(os.path.join(GOLD, "ml5_trajectories.json")))
    groups = {}
    for r in rows:
        if r["seed"] == 1:
            continue  # glibc's srand(1) is the same stream as srand(0)
        groups.setdefault((r["model"], r["L"], r["minCount"]), []).append(r)
    return groups


def _check_hit50(key, refs, got):
    """HIT@50 on the test item at the best-validation checkpoint: the mean of the engine's runs within max(0.015, the
    spread of the reference's own srand seeds) of the mean of the reference's distinct runs (srand(0) and srand(1) give
    the same run in glibc).  6040 users: one evaluation has a standard error of 0.005, the checkpoint validation picks
    moves it by more (the reference's seeds differ by up to 0.018, Fossil)."""
    ref_hits = sorted({round(r["hit50_test_at_best"], 6) for r in refs})
    hit, ref_hit = np.mean([g["hit50_test_at_best"] for g in got]), np.mean(ref_hits)
    print(key, "hit50 engine %.4f reference %.4f (%s)" % (hit, ref_hit, ref_hits))
    assert abs(hit - ref_hit) <= max(0.015, ref_hits[-1] - ref_hits[0]), (key, hit, ref_hits)


@pytest.mark.parametrize("key", sorted(_traj_groups()))
def test_converged_auc_within_0_005_of_reference(key):
    """Criterion (c).  Reference side: its own oneiteration + AUC on ML-5 (K=10), best-validation
    checkpoint of the trajectory, five distinct srand seeds.  Engine side: device sampler + Hogwild update with the
    same epochs and evaluation cadence, four seeds.  Bound: |mean test AUC difference| <= 0.005.
    (Single runs of one configuration spread by up to 0.008 on either side - TransRec_L1, whose step is a sign: engine
    0.7356 .. 0.7428, reference 0.7384 .. 0.7446; HRM_max: reference 0.7463 .. 0.7542 - so with two runs a side, as this
    test had it first, the means of the noisiest configurations sat up to 0.005 apart by chance.)"""
    from tests.converge_util import best_of_trajectory
    refs = _traj_groups()[key]
    ref_test = np.mean([r["auc_test_at_best"] for r in refs])
    ref_val = np.mean([r["best_auc_val"] for r in refs])
    got = [best_of_trajectory(refs[0], seed=s) for s in (0, 1, 2, 3)]
    test_auc = np.mean([g["auc_test_at_best"] for g in got])
    val_auc = np.mean([g["best_auc_val"] for g in got])
    assert abs(test_auc - ref_test) <= 0.005, (key, test_auc, ref_test)
    assert abs(val_auc - ref_val) <= 0.005, (key, val_auc, ref_val)
    _check_hit50(key, refs, got)


_SHARED_NEG_KEYS = [k for k in sorted(_traj_groups()) if k[0] in ("BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "HRM_max")]


@pytest.mark.parametrize("key", _SHARED_NEG_KEYS)
def test_converged_auc_with_shared_negatives_within_0_005_of_reference(key):
    """Criterion (c) again with the option the engine switches on by itself once an epoch has 4M steps (ML-5 has
    far fewer): pairs of steps share their negative.  (The history models' pairs of tuples per walk are not forced
    here: on ML-5's users of at most three training clicks the two tuples mostly coincide, which doubles the step -
    the engine only pairs them where histories are long.)"""
    from tests.converge_util import best_of_trajectory
    refs = _traj_groups()[key]
    ref_test = np.mean([r["auc_test_at_best"] for r in refs])
    ref_val = np.mean([r["best_auc_val"] for r in refs])
    got = [best_of_trajectory(refs[0], seed=s, options={"share_negative": 2}) for s in (0, 1, 2, 3)]
    test_auc = np.mean([g["auc_test_at_best"] for g in got])
    val_auc = np.mean([g["best_auc_val"] for g in got])
    assert abs(test_auc - ref_test) <= 0.005, (key, test_auc, ref_test)
    assert abs(val_auc - ref_val) <= 0.005, (key, val_auc, ref_val)
    _check_hit50(key, refs, got)


@pytest.mark.parametrize("model", ["BPRMF", "FPMC", "TransRec", "REBUS"])
def test_user_shards_partition_sampling_and_evaluation(model):
    """Engines restricted to user ranges (one per rank): they sample only their own users, and their
    evaluation sums add up to the sums of an unrestricted engine."""
    g, model, co, ci, labels = fixture_case(model, "toy")
    U = len(co) - 1
    full = make_engine(g, model, co, ci, labels, F64)
    full.upload(g["W1"])
    want = np.array(full.eval_sums(50))
    got = np.zeros(8)
    cuts = [0, U // 3, U // 3, U]       # an empty shard in the middle
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == b:
            continue
        e = make_engine(g, model, co, ci, labels, F64, user_begin=a, user_end=b)
        e.upload(g["W1"])
        got += np.array(e.eval_sums(50))
        t = e.sample_tuples(2000, seed=1, epoch=0)
        assert t[:, 0].min() >= a and t[:, 0].max() < b
        m, counts, _ = e.eval()
        assert (counts[:a] == -1).all() and (counts[b:] == -1).all()
        e.close()
    assert got[7] == want[7] and np.allclose(got, want, rtol=1e-12, atol=1e-12)
    full.close()



@pytest.mark.parametrize("model", ["FossilSimple", "Fossil", "REBUS"])
def test_history_kernels_on_long_histories_match_updateFactors(model):
    """The toy users above have a handful of clicks: one round of the history walk.  Here 2000 users x 1500 items x 200k
    clicks (mean history 100, up to 400: many rounds, both passes streaming, the packed list of first occurrences for REBUS):
    64 single tuples through the Hogwild kernel against updateFactors (REBUS: the same arithmetic, 1e-9; Fossil /
    FossilSimple: a repeated item of the history is read before its first update, an O(lr^2) difference, 1e-5), then groups
    of 2 and 8 tuples of one user against the sum of their single steps from the same parameters (users with fewer than 8
    clicks, whose groups are applied one after the other, left out)."""
    import torch
    from bench import make_synthetic
    U, I, K, L = 2000, 1500, 64, 3
    seq_off, seq_item, test, val, _ = make_synthetic(U, I, 200_000, 7, torch.device("cuda:0"))
    po = pi = None
    kw = dict(L=L, lam=0.01, bias_reg=0.01, relation_reg=0.01, alpha=-0.6 if model == "REBUS" else 0.5)
    if model == "REBUS":
        lo, li = ra.fsub_mine(I, seq_off, seq_item, L, 2)
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
    e = ra.Engine(model, I, seq_off, seq_item, test, val, K=K, precision=F64, path_off=po, path_item=pi, **kw)
    o = orc.Oracle(model, U, I, K, seq_off, seq_item, test, val, path_off=po, path_item=pi, **kw)
    e.set_option("update_policy", 0)
    e.init_params(3)
    W0 = e.download()
    single = 1e-9 if model == "REBUS" else 1e-5
    for G in (1, 2, 8):
        e.set_option("user_batch", G)
        T = e.sample_tuples(64 * G, seed=5, epoch=1)
        n = 0
        longest = 0
        for k in range(0, 64 * G, G):
            grp = T[k:k + G]
            nu = int(seq_off[grp[0, 0] + 1] - seq_off[grp[0, 0]])
            if G > 1 and nu < 8:
                continue
            ref = W0.copy()
            for t in grp:
                ref += o.replay(W0, t[None, :], 0.05) - W0
            e.upload(W0)
            e.sgd_replay(0.05, grp, HOGWILD)
            err = rel_err(e.download(), ref, floor=1e-1)
            assert err < single * G, (model, G, grp.tolist(), nu, err)
            n += 1
            longest = max(longest, nu)
        assert n >= 40 and longest >= 100, (model, G, n, longest)
    e.close()
