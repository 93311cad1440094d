"""Run-to-run spread of one configuration of tests/test_gpu_parity.py::test_converged_auc_within_0_005_of_reference:
exp_converged_key.py MODEL [REPEATS [option=value,...]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from tests.converge_util import best_of_trajectory  # noqa: E402
from tests.test_gpu_parity import _traj_groups  # noqa: E402

model = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[3].split(",")} if len(sys.argv) > 3 else None
for key, refs in sorted(_traj_groups().items()):
    if model != "all" and key[0] != model:
        continue
    ref_test = np.mean([r["auc_test_at_best"] for r in refs]); ref_val = np.mean([r["best_auc_val"] for r in refs])
    for rep in range(reps):
        got = [best_of_trajectory(refs[0], seed=4 * rep + s, options=opts) for s in (0, 1, 2, 3)]
        t = [g["auc_test_at_best"] for g in got]; v = [g["best_auc_val"] for g in got]
        print(key, "rep", rep, "test auc", np.round(t, 4), "mean - ref %.4f" % (np.mean(t) - ref_test), "val mean - ref %.4f" % (np.mean(v) - ref_val), "ref runs", [round(r["auc_test_at_best"], 4) for r in refs], flush=True)
