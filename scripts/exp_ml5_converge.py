#!/usr/bin/env python3
"""Best-validation checkpoint of device training on the ML-5 fixture next to the reference's own
trajectory (tests/golden/ml5_trajectories.json, same epochs and evaluation cadence)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.converge_util import best_of_trajectory  # noqa: E402

ref = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ml5_trajectories.json")))
only = sys.argv[1:] or None
for r in ref:
    if only and r["model"] not in only:
        continue
    t0 = time.time()
    res = best_of_trajectory(r)
    print(f"{r['model']:13s} L={r['L']} mc={r['minCount']}: engine best@{res['best_iter']:4d} val {res['best_auc_val']:.4f} test "
          f"{res['auc_test_at_best']:.4f} HIT {res['hit50_test_at_best']:.4f} | reference best@{r['best_iter']:4d} val {r['best_auc_val']:.4f} "
          f"test {r['auc_test_at_best']:.4f} HIT {r['hit50_test_at_best']:.4f} | d_test {res['auc_test_at_best'] - r['auc_test_at_best']:+.4f} "
          f"d_val {res['best_auc_val'] - r['best_auc_val']:+.4f} [{time.time() - t0:.1f}s]", flush=True)
