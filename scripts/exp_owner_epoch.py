"""One whole epoch at the 1M x 200k shape from the same parameters, owner mode against the plain kernel: who moved, by how much."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import recommendation_amd as ra
from bench import make_synthetic
from recommendation_amd.capi import F32

U, I, CLICKS, K = 1_000_000, 200_000, 100_000_000, 64
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda:0")
seq_off, seq_item, test, val, _ = make_synthetic(U, I, CLICKS, 12345, dev)
torch.cuda.empty_cache()
cnt = np.bincount(seq_item, minlength=I)
order = np.argsort(-cnt)
for name, opts in (("plain", [("update_policy", 0), ("owner_neg", 0)]), ("owner512", [("update_policy", 0), ("owner_neg", 1), ("owner_min", 1)]),
                   ("owner256", [("update_policy", 0), ("owner_neg", 1), ("owner_min", 1), ("owner_slices", 256)])):
    e = ra.Engine("BPRMF", I, seq_off, seq_item, test, val, K=K, lam=0.01, bias_reg=0.01, precision=F32)
    for k, v in opts:
        e.set_option(k, v)
    e.set_option("hot_count", 1000000000)
    e.init_params(seed=2024)
    W0 = e.download()
    e.sgd_epoch(0.05, int(e.steps_per_epoch * frac), seed=99, epoch=0)
    e.sync()
    W1 = e.download()
    s = e.eval_sums(50)
    e.close()
    d = W1 - W0
    beta, gu, gi = d[:I], d[I:I + U * K].reshape(U, K), d[I + U * K:].reshape(I, K)
    print(name, "auc %.4f hit %.4f" % (s[1] / s[7], s[3] / s[7]))
    print("   users moved %.4f mean |d user row| %.4f  items moved %.4f" % ((np.abs(gu).sum(1) > 0).mean(), np.abs(gu).sum(1).mean(), (np.abs(gi).sum(1) > 0).mean()))
    dn = np.sqrt((gi * gi).sum(1))
    print("   |d item row| by popularity decile of rank:", np.round([dn[order[a:b]].mean() for a, b in zip([0, 10, 100, 1000, 10000, 100000], [10, 100, 1000, 10000, 100000, 200000])], 4))
    print("   |d item row| by 16ths of the catalog:", np.round([dn[a * I // 16:(a + 1) * I // 16].mean() for a in range(16)], 4))
    print("   d beta by rank:", np.round([beta[order[a:b]].mean() for a, b in zip([0, 10, 100, 1000, 10000, 100000], [10, 100, 1000, 10000, 100000, 200000])], 4))
    w1i = W1[I + U * K:].reshape(I, K); w1u = W1[I:I + U * K].reshape(U, K)
    print("   |item row| by rank:", np.round([np.sqrt((w1i[order[a:b]] ** 2).sum(1)).mean() for a, b in zip([0, 10, 100, 1000, 10000, 100000], [10, 100, 1000, 10000, 100000, 200000])], 4),
          " |user row| mean %.4f" % np.sqrt((w1u ** 2).sum(1)).mean(), flush=True)
