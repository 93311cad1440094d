#!/usr/bin/env python3
"""Trains every model on the ML-5 fixture with the reference's train() policy and prints the
early-stopped test AUC next to the reference CLI's (tests/golden/ml5_converged.json)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.converge_util import train_like_reference  # noqa: E402

ref = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ml5_converged.json")))
only = sys.argv[1:] or None
for r in ref:
    if only and r["model"] not in only:
        continue
    t0 = time.time()
    res = train_like_reference(r)
    print(f"{r['model']:13s} L={r['L']} mc={r['minCount']}: engine AUC_test {res['auc_test']:.6f} (val {res['auc_val']:.6f}, "
          f"HIT50 {res['hit50_test']:.4f}, {res['epochs']} epochs, best {res['best_iter']}) | reference {r['auc_test']:.6f} "
          f"(val {r['auc_val']:.6f}, HIT50 {r['hit50_test']:.4f}) | delta {res['auc_test'] - r['auc_test']:+.4f}  [{time.time() - t0:.1f}s]",
          flush=True)
