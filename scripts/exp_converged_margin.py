import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import _traj_groups
from tests.converge_util import best_of_trajectory
G = _traj_groups()
for rep in range(2):
    for i, key in enumerate(sorted(G)):
        refs = G[key]
        rt = np.mean([r["auc_test_at_best"] for r in refs]); rv = np.mean([r["best_auc_val"] for r in refs])
        out = []
        for options in (None, {"user_batch": 2} if key[0] in ("Fossil", "FossilSimple", "REBUS") else {"share_negative": 2}):
            got = [best_of_trajectory(refs[0], seed=s, options=options) for s in (0, 1)]
            out.append((round(np.mean([g["auc_test_at_best"] for g in got]) - rt, 4), round(np.mean([g["best_auc_val"] for g in got]) - rv, 4)))
        print(rep, i, key, out, flush=True)
