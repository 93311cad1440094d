"""The schedule of tests/test_gpu_seq_anchor.py (60k users x 200k items x 6M clicks, six epochs, two evaluations per epoch,
four seeds) for any list of engine option sets, next to the sequential restatement's trajectory of the fixture.
usage: exp_anchor_opts.py MODEL[,MODEL] "opt=val,opt=val;opt=val;" (an empty set = the defaults)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from bench import make_synthetic  # noqa: E402
from recommendation_amd.capi import F32  # noqa: E402

REF = {r["model"]: r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "seq_60k_trajectories.json")))}


def best(traj):
    t = max(traj, key=lambda row: row[1])
    return t[1], t[2], t[4]


models = sys.argv[1].split(",")
variants = [[(kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv] for v in (sys.argv[2] if len(sys.argv) > 2 else "").split(";")]
r0 = REF[models[0]]
seq_off, seq_item, test, val, _ = make_synthetic(r0["users"], r0["items"], r0["clicks"], r0["corpus_seed"], torch.device("cuda:0"))
torch.cuda.empty_cache()
for model in models:
    r = REF[model]
    I = r["items"]
    po = pi = None
    if model == "REBUS":
        lo, li = ra.fsub_mine(I, seq_off, seq_item, r["L"], 2)
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
    ref = np.array([best(x["trajectory"]) for x in r["runs"]]).mean(0)
    for opts in variants:
        e = ra.Engine(model, I, seq_off, seq_item, test, val, K=r["K"], L=r["L"], lam=0.01, bias_reg=0.01, relation_reg=0.01,
                      alpha=-0.6 if model == "REBUS" else 0.5, precision=F32, path_off=po, path_item=pi)
        for k, v in opts:
            e.set_option(k, v)
        got = []
        for seed in range(4):
            e.init_params(seed)
            traj = []
            for ck in range(1, r["epochs"] * r["evals_per_epoch"] + 1):
                e.sgd_epoch(r["lr"], e.steps_per_epoch // r["evals_per_epoch"], seed=seed, epoch=ck)
                m, _, _ = e.eval(want_counts=False)
                traj.append([ck, m.auc_val, m.auc_test, m.hit_val, m.hit_test])
            got.append(best(traj))
        e.close()
        got = np.array(got)
        print(model, opts, "engine - sequential (val AUC, test AUC, HIT@50)", (got.mean(0) - ref).round(4).tolist(), "per seed test AUC", got[:, 1].round(4).tolist(), flush=True)
