import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.converge_util import best_of_trajectory
ref = json.load(open("tests/golden/ml5_trajectories.json"))
for r in ref:
    if r["model"] == "HRM_max":
        for win in (0, 512, 128, 32):
            for seed in range(2):
                res = best_of_trajectory(r, seed=seed, options={"stale_window": win} if win else None)
                print("HRM_max seed", seed, "win", win, "best@%d val %.4f test %.4f" % (res["best_iter"], res["best_auc_val"], res["auc_test_at_best"]), "| ref val %.4f test %.4f" % (r["best_auc_val"], r["auc_test_at_best"]), flush=True)
