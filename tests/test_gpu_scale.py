"""The hot path at the size the throughput is quoted on (BPRMF K=64, 1M users x 200k items x 100M
interactions): the parity checks that do not depend on the size, run against the CPU restatement on a
sample (replay of sampled tuples, per-user counts of sampled users, sampler support), and an
idempotence check of a whole Hogwild epoch."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu

U, I, CLICKS, K = 1_000_000, 200_000, 100_000_000, 64


@pytest.fixture(scope="module")
def corpus():
    import torch
    from bench import make_synthetic
    seq_off, seq_item, test, val, _ = make_synthetic(U, I, CLICKS, 12345, torch.device("cuda:0"))
    torch.cuda.empty_cache()
    return seq_off, seq_item, test, val


@pytest.fixture(scope="module")
def big(corpus):
    import recommendation_amd as ra
    from recommendation_amd.capi import F32
    seq_off, seq_item, test, val = corpus
    eng = ra.Engine("BPRMF", I, seq_off, seq_item, test, val, K=K, lam=0.01, bias_reg=0.01, precision=F32)
    eng.init_params(seed=2024)
    yield eng, seq_off, seq_item, test, val
    eng.close()


def test_full_shape_replay_within_1e5_of_updateFactors(big):
    from oracle import oracle_py as orc
    from recommendation_amd.capi import SEQUENTIAL
    eng, seq_off, seq_item, test, val = big
    W0 = eng.download()
    T = eng.sample_tuples(4096, seed=3, epoch=1)
    o = orc.Oracle("BPRMF", U, I, K, seq_off, seq_item, test, val, lam=0.01, bias_reg=0.01)
    ref = o.replay(W0, T, 0.05)
    eng.sgd_replay(0.05, T, SEQUENTIAL)
    W1 = eng.download()
    touched = ref != W0
    assert 4096 * K < touched.sum() <= 4096 * (3 * K + 2)
    err = np.abs(W1[touched] - ref[touched]) / np.maximum(np.abs(ref[touched]), 1e-1)   # same floor as the toy-size replay tests
    assert err.max() < 1e-5, err.max()                  # f32 storage against the fp64 reference arithmetic
    assert np.array_equal(W1[~touched], W0[~touched])   # nothing else was written
    eng.upload(W0)


def test_full_shape_counts_of_sampled_users_match_oracle(big):
    from oracle import oracle_py as orc
    eng, seq_off, seq_item, test, val = big
    W = eng.download()
    m, counts, _ = eng.eval()
    sample = np.linspace(0, U - 1, 2048).astype(np.int64)
    t_s = np.full(U, -1, np.int32); t_s[sample] = test[sample]
    o = orc.Oracle("BPRMF", U, I, K, seq_off, seq_item, t_s, val, lam=0.01, bias_reg=0.01)
    _, ref = o.eval(W, nthreads=min(16, os.cpu_count() or 1))
    assert (ref[sample, 2] >= 0).all()
    assert np.array_equal(counts[sample], ref[sample])  # (count_val, count_test, max) per user, bit-exact
    assert 0.4 < m.auc_test < 0.6                       # random parameters


def test_full_shape_sampler_support(big):
    eng, seq_off, seq_item, test, val = big
    T = eng.sample_tuples(1 << 20, seed=9, epoch=4)
    assert T[:, 0].min() >= 0 and T[:, 0].max() < U and T[:, 1:3].min() >= 0 and T[:, 1:3].max() < I
    for u, p, n, _ in T[:20000].tolist():
        s = seq_item[seq_off[u]:seq_off[u + 1]]
        assert (s == p).any() and not (s == n).any()     # BPRMF.cpp:74-83, 135-143
    assert (T[0::2, 2] == T[1::2, 2]).mean() < 1e-3       # the default: every step draws its own negative, as the reference does
    eng.set_option("share_negative", 2)                   # option: pairs of steps take their negative from one candidate stream
    T2 = eng.sample_tuples(1 << 16, seed=9, epoch=4)
    eng.set_option("share_negative", 1)
    assert (T2[0::2, 2] == T2[1::2, 2]).mean() > 0.99
    # users uniform: the mean user id of 1M draws is within 5 sigma of (U - 1) / 2
    assert abs(T[:, 0].mean() - (U - 1) / 2) < 5 * U / np.sqrt(12 * len(T))


def test_full_shape_epoch_with_zero_learning_rate_changes_nothing(big):
    eng, seq_off, seq_item, test, val = big
    W0 = eng.download()
    before = eng.tuples_drawn
    eng.sgd_epoch(0.0, eng.steps_per_epoch, seed=5, epoch=0)
    eng.sync()
    assert eng.tuples_drawn - before == eng.steps_per_epoch
    assert np.array_equal(eng.download(), W0)           # every scatter of the epoch added (or stored) a zero delta


@pytest.mark.parametrize("model", ["FPMC", "TransRec", "HRM_avg", "Fossil"])
def test_full_shape_other_models_replay_and_counts(corpus, model):
    """The first-order and history models at the same size: replay of sampled tuples against updateFactors,
    per-user counts of sampled users against model::AUC."""
    import recommendation_amd as ra
    from oracle import oracle_py as orc
    from recommendation_amd.capi import F32, SEQUENTIAL
    seq_off, seq_item, test, val = corpus
    kw = dict(K=K, L=3, lam=0.01, bias_reg=0.01, relation_reg=0.01, alpha=0.5)
    eng = ra.Engine(model, I, seq_off, seq_item, test, val, precision=F32, **kw)
    eng.init_params(seed=7)
    W0 = eng.download()
    T = eng.sample_tuples(512, seed=3, epoch=1)
    sample = np.linspace(0, U - 1, 512).astype(np.int64)
    t_s = np.full(U, -1, np.int32); t_s[sample] = test[sample]
    o = orc.Oracle(model, U, I, K, seq_off, seq_item, t_s, val, **{k: v for k, v in kw.items() if k != "K"})
    ref = o.replay(W0, T, 0.05)
    eng.sgd_replay(0.05, T, SEQUENTIAL)
    W1 = eng.download()
    touched = ref != W0
    assert touched.sum() > 512 * K
    err = np.abs(W1[touched] - ref[touched]) / np.maximum(np.abs(ref[touched]), 1e-1)
    assert err.max() < 1e-5, (model, err.max())
    assert np.array_equal(W1[~touched], W0[~touched])
    _, counts, _ = eng.eval()
    _, cref = o.eval(W1, nthreads=min(16, os.cpu_count() or 1))
    assert np.array_equal(counts[sample], cref[sample]), model
    eng.close()
