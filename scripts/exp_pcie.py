import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendation_amd as ra
from recommendation_amd.capi import F32
from bench import make_synthetic
so, si, t, v, _ = make_synthetic(1_000_000, 200_000, 100_000_000, 12345, torch.device('cuda:0'))
t0 = time.perf_counter()
e = ra.Engine('BPRMF', 200_000, so, si, t, v, K=64, lam=0.01, bias_reg=0.01, precision=F32)
print('rec_create (host index build + upload of the corpus)', round(time.perf_counter() - t0, 2), 's')
e.init_params(1)
for _ in range(2):
    t0 = time.perf_counter(); W = e.download(); t1 = time.perf_counter(); e.upload(W); t2 = time.perf_counter()
    print('download', round(t1 - t0, 3), 's upload', round(t2 - t1, 3), 's for', W.nbytes / 1e6, 'MB')
