"""Where the time of rec_eval goes outside its kernels: end-to-end time against the kernels' HIP-event time for a few sizes
of the undecided lists (the largest allocation of a call), profile on and off."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recommendation_amd as ra  # noqa: E402
from recommendation_amd.capi import F32  # noqa: E402
from bench import make_synthetic  # noqa: E402

import torch
seq_off, seq_item, test, val, _ = make_synthetic(1_000_000, 200_000, 100_000_000, 12345, torch.device("cuda:0"))
eng = ra.Engine("BPRMF", 200_000, seq_off, seq_item, test, val, K=64, lam=0.01, bias_reg=0.01, precision=F32)
eng.init_params(seed=1)
for prof in (1, 0):
    eng.set_option("profile", prof)
    for cap in (0, 256, 4096):
        eng.set_option("eval_screen_cap", cap)
        eng.eval_sums(50)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); eng.eval_sums(50); best = min(best, time.perf_counter() - t0)
        print(f"profile {prof} cap {cap}: eval {best * 1e3:.1f} ms", flush=True)
eng.set_option("eval_screen_cap", 0)
