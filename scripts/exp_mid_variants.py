"""Engine options on the mid-size corpus next to the reference's trajectories: best-validation checkpoint and trajectory
means (test AUC / HIT@50), 2 engine seeds.  usage: exp_mid_variants.py MODEL "opt=val,opt=val;opt=val" ..."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401
from tests.converge_util import best_of_trajectory_mid
from tests.golden_util import GOLD

REF = json.load(open(os.path.join(GOLD, "mid_trajectories.json")))
model = sys.argv[1]
variants = [[(kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv] for v in sys.argv[2].split(";")]
ref = [r for r in REF if r["model"] == model]
tr = np.array([[np.mean([t[2] for t in r["trajectory"]]), np.mean([t[4] for t in r["trajectory"]])] for r in ref])
print("%s reference: best auc %.4f val %.4f hit %.4f iters %s | traj auc %.4f hit %.4f" % (
    model, np.mean([r["auc_test_at_best"] for r in ref]), np.mean([r["best_auc_val"] for r in ref]),
    np.mean([r["hit50_test_at_best"] for r in ref]), [r["best_iter"] for r in ref], tr[:, 0].mean(), tr[:, 1].mean()), flush=True)
for opts in variants:
    got = [best_of_trajectory_mid(ref[0], opts, seed=s) for s in (0, 1)]
    tm = np.array([[np.mean([t[2] for t in g["trajectory"]]), np.mean([t[4] for t in g["trajectory"]])] for g in got]).mean(0)
    print("  %-60s best auc %.4f val %.4f hit %.4f iters %s | traj auc %.4f hit %.4f" % (
        opts, np.mean([g["auc_test_at_best"] for g in got]), np.mean([g["best_auc_val"] for g in got]),
        np.mean([g["hit50_test_at_best"] for g in got]), [g["best_iter"] for g in got], tm[0], tm[1]), flush=True)
