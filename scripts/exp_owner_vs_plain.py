"""One launch of n tuples at the 1M x 200k shape from the same parameters: owner mode against the plain kernel, table by table."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import recommendation_amd as ra
from bench import make_synthetic
from recommendation_amd.capi import F32, HOGWILD

U, I, CLICKS, K = 1_000_000, 200_000, 100_000_000, 64
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
LR = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
dev = torch.device("cuda:0")
seq_off, seq_item, test, val, _ = make_synthetic(U, I, CLICKS, 12345, dev)
torch.cuda.empty_cache()
res = {}
from recommendation_amd.capi import SEQUENTIAL
for name, opts in (("plain", None), ("plainwide", [("update_policy", 0), ("owner_neg", 0)]), ("plainv1", [("update_policy", 0), ("owner_neg", 0), ("wide", 0)]),
                   ("owner", [("update_policy", 0), ("owner_neg", 1), ("owner_min", 1)]),
                   ("ownerv1", [("update_policy", 0), ("owner_neg", 1), ("owner_min", 1), ("wide", 0)]),
                   ("owner256", [("update_policy", 0), ("owner_neg", 1), ("owner_min", 1), ("owner_slices", 256)])):
    e = ra.Engine("BPRMF", I, seq_off, seq_item, test, val, K=K, lam=0.01, bias_reg=0.01, precision=F32)
    for k, v in (opts or []):
        e.set_option(k, v)
    e.set_option("hot_count", 1000000000)
    e.init_params(seed=2024)
    W0 = e.download()
    T = e.sample_tuples(n, seed=3, epoch=1)
    e.sgd_replay(LR, T, HOGWILD if opts is not None else SEQUENTIAL)
    res[name] = e.download() - W0
    e.close()
off = [0, I, I + U * K, I + U * K + I * K]
for name in ("plainwide", "plainv1", "owner", "ownerv1", "owner256"):
    for t, tn in enumerate(("beta", "gamma_user", "gamma_item")):
        a, b = res["plain"][off[t]:off[t + 1]], res[name][off[t]:off[t + 1]]
        print(name, tn, "plain |d| %.4e  %s |d| %.4e  |diff| %.4e  corr %.4f  sum plain %.4e sum %.4e" % (
            np.abs(a).sum(), name, np.abs(b).sum(), np.abs(a - b).sum(), np.corrcoef(a, b)[0, 1], a.sum(), b.sum()), flush=True)
    # negatives only: items that were drawn as negative but never as positive
    neg_only = np.setdiff1d(T[:, 2], T[:, 1])
    a = res["plain"][off[2]:off[3]].reshape(I, K)[neg_only]; b = res[name][off[2]:off[3]].reshape(I, K)[neg_only]
    print(name, "rows of items drawn as negative only (%d): |plain| %.4e |%s| %.4e |diff| %.4e" % (len(neg_only), np.abs(a).sum(), name, np.abs(b).sum(), np.abs(a - b).sum()))
