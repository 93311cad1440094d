"""Run-to-run spread of one ML-5 configuration of the converged-AUC test (Hogwild updates are not deterministic)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import _traj_groups  # noqa: E402
from tests.converge_util import best_of_trajectory  # noqa: E402

model, L, mc, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
options = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[5:])}
refs = _traj_groups()[(model, L, mc)]
rt = np.mean([r["auc_test_at_best"] for r in refs]); rv = np.mean([r["best_auc_val"] for r in refs])
d = []
for rep in range(reps):
    got = [best_of_trajectory(refs[0], seed=s, options=options) for s in (0, 1)]
    d.append((np.mean([g["auc_test_at_best"] for g in got]) - rt, np.mean([g["best_auc_val"] for g in got]) - rv))
    print(rep, [round(x, 4) for x in d[-1]], [g["best_iter"] for g in got], flush=True)
d = np.array(d)
print("test: mean %.4f min %.4f max %.4f | val: mean %.4f min %.4f max %.4f" % (d[:, 0].mean(), d[:, 0].min(), d[:, 0].max(), d[:, 1].mean(), d[:, 1].min(), d[:, 1].max()))
