"""The CPU oracle against the real reference's outputs (tests/golden).  No GPU."""
import numpy as np
import pytest

from oracle import oracle_py as orc
from tests.golden_util import ML5_TAGS, TOY_MODELS, fixture_case, load, make_oracle


def test_split_matches_reference():
    for name in ("toy_MP", "ml5_MP"):
        g = load(name)
        seq_off, seq_item, test, val = orc.split(g["click_off"], g["click_item"])
        sp = g["split"]  # test, test_prev(=val), val, val_prev
        assert np.array_equal(test, sp[:, 0])
        assert np.array_equal(val, sp[:, 2])
        valid = test >= 0
        assert np.array_equal(sp[valid, 1], val[valid])
        last_train = seq_item[seq_off[1:][valid] - 1]
        assert np.array_equal(sp[valid, 3], last_train)
        assert int(g["num_pos_events"]) == int(seq_off[-1])


@pytest.mark.parametrize("model", TOY_MODELS)
def test_replay_matches_reference(model):
    g, model, co, ci, labels = fixture_case(model, "toy")
    o = make_oracle(g, model, co, ci, labels)
    assert o.nw == int(g["NW"])
    W1 = o.replay(g["W0"], g["tuples"], float(g["lr"]))
    err = np.abs(W1 - g["W1"]) / np.maximum(np.abs(g["W1"]), 1e-3)
    assert err.max() < 1e-12, (model, err.max())


@pytest.mark.parametrize("model", ["MP"] + TOY_MODELS)
def test_eval_matches_reference_toy(model):
    g, model, co, ci, labels = fixture_case(model, "toy")
    o = make_oracle(g, model, co, ci, labels)
    xpos, counts = o.eval(g.get("W1"))
    assert np.array_equal(counts, g["counts"]), model
    assert np.allclose(xpos, g["xpos"], rtol=1e-12, atol=1e-13)
    m = orc.metrics(counts)
    assert np.allclose(m, g["metrics"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("tag", ["MP"] + ML5_TAGS)
def test_eval_matches_reference_saved_model(tag):
    g, model, co, ci, labels = fixture_case(tag, "ml5")
    o = make_oracle(g, model, co, ci, labels)
    W = g["W_micro"].astype(np.float64) / 1e6 if "W_micro" in g else None
    xpos, counts = o.eval(W)
    assert np.array_equal(counts, g["counts"]), tag
    m = orc.metrics(counts)
    assert np.allclose(m, g["metrics"], rtol=0, atol=1e-15)
    if tag == "MP":  # the paper's Table 4 "Pop / ML-5" entry, reproduced by the reference CLI
        assert f"{m[1]:.6f}" == "0.735201" and f"{m[3]:.6f}" == "0.126511"


def test_rebus_paths_and_eta_match_reference():
    for tag, fam in (("REBUS", "toy"), ("REBUS_mc1_L1", "ml5"), ("REBUS_mc2_L3", "ml5")):
        g, model, co, ci, labels = fixture_case(tag, fam)
        o = make_oracle(g, model, co, ci, labels)
        L = int(g["L"])
        eta_ref = g["eta_flat"]  # rows n = 0..L, row n has n entries
        p = 0
        for n in range(L + 1):
            assert np.allclose(o.eta[n * L:n * L + n], eta_ref[p:p + n], rtol=1e-15, atol=0)
            p += n
        # reference order: all training positions of all users, then (val, test) per valid user
        flat = []
        U = o.n_users
        for u in range(U):
            n = int(o.seq_off[u + 1] - o.seq_off[u])
            for c in range(1, n):
                ctx = int(o.seq_off[u]) + 2 * u + c - 1
                path = o.path_item[o.path_off[ctx]:o.path_off[ctx + 1]]
                flat.append(len(path)); flat.extend(path.tolist())
        for u in range(U):
            if o.test_item[u] < 0:
                continue
            n = int(o.seq_off[u + 1] - o.seq_off[u])
            for c in (n, n + 1):
                ctx = int(o.seq_off[u]) + 2 * u + c - 1
                path = o.path_item[o.path_off[ctx]:o.path_off[ctx + 1]]
                flat.append(len(path)); flat.extend(path.tolist())
        assert np.array_equal(np.array(flat, np.int32), g["paths_flat"]), tag
