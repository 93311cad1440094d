"""The reference's train() policy (validation cadence, best snapshot, patience: e.g. BPRMF.cpp:152-203,
REBUS.cpp:559-615) driven through the engine's Python binding."""
import numpy as np

import recommendation_amd as ra
from oracle import oracle_py as orc
from recommendation_amd.capi import F32
from tests.golden_util import ml5_corpus, read_labels

CADENCE = {"REBUS": 100}
PATIENCE = {"MC": 1000, "FossilSimple": 500}


def train_like_reference(r, precision=F32, seed=0, max_epochs=None):
    co, ci = ml5_corpus()
    seq_off, seq_item, test, val = orc.split(co, ci)
    I = int(ci.max()) + 1
    model = r["model"]
    po = pi = None
    if model == "REBUS":
        labels = read_labels(f"ML1M-atmost-5_root_fsub_minCount_{r['minCount']}_L_{r['L']}.txt.gz")
        lo, li = orc.parse_labels(labels, I)
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
    e = ra.Engine(model, I, seq_off, seq_item, test, val, K=r["K"], L=r["L"], lam=r["lam"], bias_reg=r["bias_reg"],
                  relation_reg=r["relation_reg"], alpha=r["alpha"], precision=precision, path_off=po, path_item=pi)
    e.init_params(seed)
    cadence, patience = CADENCE.get(model, 50), PATIENCE.get(model, 300)
    best, best_iter, it = -1.0, 0, 0
    max_epochs = max_epochs or r["max_iterations"]
    for it in range(1, max_epochs + 1):
        e.sgd_epoch(0.05, e.steps_per_epoch, seed=seed, epoch=it)
        if it % cadence == 0 and (model != "REBUS" or it > 49):
            m, _, _ = e.eval(want_counts=False)
            if best < m.auc_val:
                best, best_iter = m.auc_val, it
                e.snapshot_best()
            elif it > best_iter + patience:
                break
    e.restore_best()
    m, _, _ = e.eval(want_counts=False)
    e.close()
    return dict(auc_test=m.auc_test, auc_val=m.auc_val, hit50_test=m.hit_test, epochs=it, best_iter=best_iter)


def best_of_trajectory(r, precision=F32, seed=0, policy=None, options=None):
    """Engine counterpart of oracle/make_trajectories.py: r["epochs"] epochs, full evaluation every
    r["every"] epochs, returns the checkpoint with the best validation AUC."""
    co, ci = ml5_corpus()
    seq_off, seq_item, test, val = orc.split(co, ci)
    I = int(ci.max()) + 1
    model = r["model"]
    po = pi = None
    if model == "REBUS":
        labels = read_labels(f"ML1M-atmost-5_root_fsub_minCount_{r['minCount']}_L_{r['L']}.txt.gz")
        lo, li = orc.parse_labels(labels, I)
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
    e = ra.Engine(model, I, seq_off, seq_item, test, val, K=r["K"], L=r["L"], lam=r["lam"], bias_reg=r["bias_reg"],
                  relation_reg=r["relation_reg"], alpha=r["alpha"], precision=precision, path_off=po, path_item=pi)
    if policy is not None:
        e.set_option("update_policy", policy)
    for k, v in (options or {}).items():
        e.set_option(k, v)
    e.init_params(seed)
    best = None
    traj = []
    for it in range(1, r["epochs"] + 1):
        e.sgd_epoch(0.05, e.steps_per_epoch, seed=seed, epoch=it)
        if it % r["every"] == 0:
            m, _, _ = e.eval(want_counts=False)
            traj.append((it, m.auc_val, m.auc_test))
            if best is None or m.auc_val > best[1]:
                best = (it, m.auc_val, m.auc_test, m.hit_test)
    e.close()
    return dict(best_iter=best[0], best_auc_val=best[1], auc_test_at_best=best[2], hit50_test_at_best=best[3], trajectory=traj)


def mid_corpus():
    """The mid-size long-history click file of oracle/make_mid_corpus.py as the reference's loader leaves it: ids in
    order of first appearance (src/corpus.cpp:176-210), every user's clicks in time order; returns (click_off, click_item)."""
    import gzip
    import os
    from tests.golden_util import GOLD
    with gzip.open(os.path.join(GOLD, "mid_clicks.txt.gz"), "rt") as f:
        rows = [ln.split() for ln in f]
    users = np.array([r[0] for r in rows]); items = np.array([r[1] for r in rows]); times = np.array([int(r[3]) for r in rows])

    def first_appearance_ids(tokens):
        uniq, first, inv = np.unique(tokens, return_index=True, return_inverse=True)
        rank = np.empty(len(uniq), np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(len(uniq))
        return rank[inv], len(uniq)

    u, U = first_appearance_ids(users)
    i, _ = first_appearance_ids(items)
    order = np.lexsort((times, u))   # by user, then by time (time stamps are distinct inside a user)
    click_off = np.zeros(U + 1, np.int64)
    np.add.at(click_off, u + 1, 1)
    return np.cumsum(click_off), i[order].astype(np.int32)


def best_of_trajectory_mid(r, options, precision=F32, seed=0):
    """Engine counterpart of oracle/make_mid_trajectories.py on the mid-size corpus with the given engine options."""
    co, ci = mid_corpus()
    seq_off, seq_item, test, val = orc.split(co, ci)
    I = int(ci.max()) + 1
    model = r["model"]
    po = pi = None
    if model == "REBUS":
        labels = read_labels(f"mid_clicks_root_fsub_minCount_{r['minCount']}_L_{r['L']}.txt.gz")
        lo, li = orc.parse_labels(labels, I)
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
    e = ra.Engine(model, I, seq_off, seq_item, test, val, K=r["K"], L=r["L"], lam=r["lam"], bias_reg=r["bias_reg"],
                  relation_reg=r["relation_reg"], alpha=r["alpha"], precision=precision, path_off=po, path_item=pi)
    for k, v in options:
        e.set_option(k, v)
    e.init_params(seed)
    best = None
    traj = []
    for it in range(1, r["epochs"] + 1):
        e.sgd_epoch(0.05, e.steps_per_epoch, seed=seed, epoch=it)
        if it % r["every"] == 0:
            m, _, _ = e.eval(want_counts=False)
            traj.append([it, m.auc_val, m.auc_test, m.hit_val, m.hit_test])   # the layout of the reference's trajectories
            if best is None or m.auc_val > best[1]:
                best = (it, m.auc_val, m.auc_test, m.hit_test)
    e.close()
    return dict(best_iter=best[0], best_auc_val=best[1], auc_test_at_best=best[2], hit50_test_at_best=best[3], trajectory=traj)
