"""N user shards in one process on one GPU (recommendation_amd.sharded.ShardGroup) on the mid-size corpus: best-validation
test AUC / HIT@50 by number of shards, rounds per epoch and combination rule, next to the reference's trajectories."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import recommendation_amd as ra
from oracle import oracle_py as orc
from recommendation_amd.capi import F32
from recommendation_amd.sharded import ShardGroup
from tests.converge_util import mid_corpus
from tests.golden_util import GOLD, read_labels

REF = json.load(open(os.path.join(GOLD, "mid_trajectories.json")))
models = sys.argv[1].split(",") if len(sys.argv) > 1 else ["BPRMF", "FPMC"]
configs = [(1, 1, "sum")] + [(n, r, rule) for n in (2, 4, 8) for r in (1, 4, 16, 64) for rule in ("sum", "mean")]
if len(sys.argv) > 2:
    configs = [tuple(int(x) if x.isdigit() else x for x in c.split(":")) for c in sys.argv[2].split(",")]
co, ci = mid_corpus()
seq_off, seq_item, test, val = orc.split(co, ci)
I = int(ci.max()) + 1
U = len(seq_off) - 1
dev = torch.device("cuda", 0)
for model in models:
    ref = [r for r in REF if r["model"] == model]
    r0 = ref[0]
    print(model, "reference: auc_test %.4f hit %.4f val %.4f best_iter %s" % (np.mean([r["auc_test_at_best"] for r in ref]), np.mean([r["hit50_test_at_best"] for r in ref]), np.mean([r["best_auc_val"] for r in ref]), [r["best_iter"] for r in ref]), flush=True)
    po = pi = None
    if model == "REBUS":
        lo, li = orc.parse_labels(read_labels(f"mid_clicks_root_fsub_minCount_{r0['minCount']}_L_{r0['L']}.txt.gz"), I)
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
    for world, rounds, rule in configs:
        def make(ub, ue):
            return ra.Engine(model, I, seq_off, seq_item, test, val, K=r0["K"], L=r0["L"], lam=r0["lam"], bias_reg=r0["bias_reg"],
                             relation_reg=r0["relation_reg"], alpha=r0["alpha"], precision=F32, user_begin=ub, user_end=ue, path_off=po, path_item=pi)
        g = ShardGroup(make, U, world, dev, rounds=rounds, rule=rule)
        g.init_params(0)
        best = None
        for it in range(1, r0["epochs"] + 1):
            g.epoch(0.05, 0, it)
            if it % r0["every"] == 0:
                m = g.metrics()
                if not np.isfinite(m).all():
                    best = (it, float("nan"), float("nan"), float("nan")); break
                if best is None or m[0] > best[1]:
                    best = (it, m[0], m[1], m[3])
        g.close()
        print("  shards %d rounds %3d %-4s -> best_iter %2d val %.4f auc_test %.4f hit50 %.4f" % (world, rounds, rule, *best), flush=True)
