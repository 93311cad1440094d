"""Finds the epoch at which an ML-5 REBUS (L=1, minCount=1) training with user_batch 2 turns non-finite and replays it."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendation_amd as ra  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402
from recommendation_amd.capi import F64, HOGWILD  # noqa: E402
from tests.golden_util import ml5_corpus, read_labels  # noqa: E402
from tests.test_gpu_parity import _traj_groups  # noqa: E402

r = _traj_groups()[("REBUS", 1, 1)][0]
co, ci = ml5_corpus()
seq_off, seq_item, test, val = orc.split(co, ci)
I = int(ci.max()) + 1
labels = read_labels(f"ML1M-atmost-5_root_fsub_minCount_{r['minCount']}_L_{r['L']}.txt.gz")
lo, li = orc.parse_labels(labels, I)
po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
lr = 0.05
for rep in range(30):
    e = ra.Engine("REBUS", I, seq_off, seq_item, test, val, K=r["K"], L=r["L"], lam=r["lam"], bias_reg=r["bias_reg"],
                  relation_reg=r["relation_reg"], alpha=r["alpha"], precision=F64, path_off=po, path_item=pi)
    e.set_option("user_batch", 2)
    e.init_params(rep)
    prev = e.download()
    found = False
    for it in range(1, 1001):
        e.sgd_epoch(lr, e.steps_per_epoch, seed=rep, epoch=it)
        if it % 10 == 0:
            W = e.download()
            if not np.isfinite(W).all():
                found = True
                break
            prev = W
    if not found:
        e.close(); continue
    print("rep", rep, "non-finite within epochs", it - 9, "..", it, "|prev|max", np.abs(prev).max(), flush=True)
    # redo the ten epochs one by one from prev
    e.upload(prev)
    for it2 in range(it - 9, it + 1):
        before = e.download()
        e.sgd_epoch(lr, e.steps_per_epoch, seed=rep, epoch=it2)
        W = e.download()
        print("  epoch", it2, "finite", bool(np.isfinite(W).all()), "|W|max", float(np.abs(W[np.isfinite(W)]).max()), flush=True)
        if not np.isfinite(W).all():
            T = e.sample_tuples(e.steps_per_epoch, seed=rep, epoch=it2)
            nbad = 0
            for k in range(10):
                e.upload(before); e.sgd_replay(lr, T, HOGWILD)
                nbad += 0 if np.isfinite(e.download()).all() else 1
            print("  replay of that epoch's tuples from the state before it: non-finite in", nbad, "of 10", flush=True)
            e.set_option("user_batch", 1)
            e.upload(before); e.sgd_replay(lr, T, HOGWILD)
            print("  same tuples walked one by one: finite", bool(np.isfinite(e.download()).all()), flush=True)
            bad = np.nonzero(~np.isfinite(W))[0]
            print("  non-finite parameters:", len(bad), "first", bad[:8], "rows", sorted(set(((bad[bad >= I + 1] - (I + 1)) // r["K"]).tolist()))[:10], flush=True)
            break
    e.close()
    break
