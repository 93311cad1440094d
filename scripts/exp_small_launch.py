"""Update-kernel time per launch as a function of the launch length (BPRMF K = 64 at the 1M x 200k shape): what a round of the
sharded trainer costs."""
import sys
import time
sys.path.insert(0, ".")
import torch
import recommendation_amd as ra
from bench import make_synthetic
from recommendation_amd.capi import F32
U, I, K = 1_000_000, 200_000, 64
seq_off, seq_item, test, val, _ = make_synthetic(U, I, 100_000_000, 12345, torch.device("cuda:0"))
e = ra.Engine("BPRMF", I, seq_off, seq_item, test, val, K=K, lam=0.01, bias_reg=0.01, precision=F32)
e.init_params(seed=1); e.set_option("profile", 1)
for n in (50_000, 100_000, 200_000, 400_000, 1_600_000, 6_400_000):
    for rep in range(3):
        e.sgd_epoch(0.05, n, seed=1, epoch=rep)
    e.sync(); e.kernel_stats(0); e.kernel_stats(2)
    t0 = time.perf_counter()
    for rep in range(20):
        e.sgd_epoch(0.05, n, seed=1, epoch=10 + rep)
    e.sync()
    dt = (time.perf_counter() - t0) / 20
    um, un = e.kernel_stats(0); sm, sn = e.kernel_stats(2)
    print("n %8d: %.3f ms per call (update %.3f ms, sampler %.3f ms) -> %.2f G tuples/s" % (n, dt * 1e3, um / un, sm / sn, n / dt / 1e9), flush=True)
