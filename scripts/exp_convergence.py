#!/usr/bin/env python3
"""AUC trajectory of device training under the two update policies (float atomics vs plain racy
stores) on the synthetic workload of bench.py.  Experiment script, not part of the product."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendation_amd as ra  # noqa: E402
from bench import make_synthetic  # noqa: E402
from recommendation_amd.capi import F32  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=200_000)
ap.add_argument("--clicks", type=int, default=100_000_000)
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--epochs", type=int, default=8)
ap.add_argument("--model", default="BPRMF")
ap.add_argument("--modes", default="0,1,2")
args = ap.parse_args()
dev = torch.device("cuda", 0)
seq_off, seq_item, test_item, val_item, n = make_synthetic(args.users, args.items, args.clicks, 12345, dev)
for atomic in [int(x) for x in args.modes.split(",")]:
    e = ra.Engine(args.model, args.items, seq_off, seq_item, test_item, val_item, K=args.k, lam=0.01, bias_reg=0.01,
                  relation_reg=0.01, alpha=0.5, precision=F32)
    e.init_params(2024)
    e.set_option("update_policy", atomic)
    e.set_option("chunk", 1 << 24)
    for ep in range(args.epochs):
        t0 = time.perf_counter()
        e.sgd_epoch(0.05, e.steps_per_epoch, seed=5, epoch=ep)
        e.sync()
        dt = time.perf_counter() - t0
        s = e.eval_sums(50)
        print(f"policy={atomic} epoch {ep + 1}: {dt * 1e3:.1f} ms  AUC_val {s[0] / s[7]:.5f} AUC_test {s[1] / s[7]:.5f} "
              f"HIT50_test {s[3] / s[7]:.5f}", flush=True)
    e.close()
