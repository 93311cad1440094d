#!/usr/bin/env python3
"""Time of the Hogwild update kernel under different per-row scatter policies (experiment)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendation_amd as ra  # noqa: E402
from bench import make_synthetic  # noqa: E402
from recommendation_amd.capi import F32  # noqa: E402

dev = torch.device("cuda", 0)
U, I = 1_000_000, 200_000
seq_off, seq_item, test_item, val_item, n = make_synthetic(U, I, 100_000_000, 12345, dev)
e = ra.Engine("BPRMF", I, seq_off, seq_item, test_item, val_item, K=64, lam=0.01, bias_reg=0.01, precision=F32)
e.init_params(2024)
e.set_option("chunk", 1 << 24)
e.set_option("profile", 1)
N = 1 << 25
import numpy as np
cnt = np.bincount(seq_item, minlength=I)
print("train count percentiles 50/90/99/99.9/max:", [int(np.percentile(cnt, q)) for q in (50, 90, 99, 99.9, 100)])
for name, mask, hot in [("plain", 0, 0), ("items+bias", 0x106, 100000), ("items", 0x006, 100000), ("neg", 0x004, 100000),
                        ("pos", 0x002, 100000), ("bias", 0x100, 100000), ("neg+bias", 0x104, 100000), ("pos, hot 10k", 0x002, 10000),
                        ("items, hot 10k", 0x006, 10000)]:
    e.set_option("atomic", mask)
    e.set_option("hot_count", hot)
    nh = int((cnt >= hot).sum()) if hot else 0
    name = f"{name} ({nh} hot, {cnt[cnt >= hot].sum() / cnt.sum() * 100 if hot else 0:.1f}% of pos)"
    e.sgd_epoch(0.05, N, seed=1, epoch=0)
    e.sync(); e.kernel_stats(0)
    e.sgd_epoch(0.05, N, seed=1, epoch=1)
    e.sync()
    ms, k = e.kernel_stats(0)
    print(f"{name:45s} mask {mask:#05x}: {ms / k:.3f} ms per {N // k} tuples -> {N / ms / 1e6:.2f} G triples/s", flush=True)
e.close()
