import sys, os
import numpy as np
sys.path.insert(0, ".")
import torch
import recommendation_amd as ra
from bench import make_synthetic
from oracle import oracle_py as orc
from recommendation_amd.capi import F32, F64, SEQUENTIAL
U, I, K = 1_000_000, 200_000, 64
seq_off, seq_item, test, val, _ = make_synthetic(U, I, 100_000_000, 12345, torch.device("cuda:0"))
L, MC = 10, 2
lo, li = ra.fsub_mine(I, seq_off, seq_item, L, MC)
po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
print("labels", len(lo) - 1, "max path", int(np.diff(po).max()), "mean path %.2f" % np.diff(po).mean(), flush=True)
kw = dict(K=K, L=L, lam=0.01, bias_reg=0.0, relation_reg=0.0, alpha=-0.6)
for prec in (F64, F32):
    eng = ra.Engine("REBUS", I, seq_off, seq_item, test, val, precision=prec, path_off=po, path_item=pi, **kw)
    eng.init_params(seed=7)
    W0 = eng.download()
    T = eng.sample_tuples(512, seed=3, epoch=1)
    o = orc.Oracle("REBUS", U, I, K, seq_off, seq_item, test, val, path_off=po, path_item=pi, **{k: v for k, v in kw.items() if k != "K"})
    ref = o.replay(W0, T, 0.05)
    eng.sgd_replay(0.05, T, SEQUENTIAL)
    W1 = eng.download()
    touched = ref != W0
    err = np.abs(W1 - ref) / np.maximum(np.abs(ref), 1e-1)
    idx = np.argsort(-err)[:8]
    print("precision", prec, "max err", err.max(), "n > 1e-5:", int((err > 1e-5).sum()), "of", int(touched.sum()))
    for i in idx:
        print("   idx", i, "row", (i - 0) // K, "ref", ref[i], "got", W1[i], "w0", W0[i], "d_ref", ref[i] - W0[i], "d_got", W1[i] - W0[i])
    # how many of the 512 tuples touch the worst row
    eng.close()
