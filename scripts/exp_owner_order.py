"""Is it the order or the kernel?  One epoch's tuples at the 1M x 200k shape, applied (a) by the plain kernel in sampled order,
(b) by the plain kernel in the partitioned order, (c) by the owner kernel."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import recommendation_amd as ra
from bench import make_synthetic
from recommendation_amd.capi import F32, HOGWILD

U, I, CLICKS, K = 1_000_000, 200_000, 100_000_000, 64
dev = torch.device("cuda:0")
seq_off, seq_item, test, val, _ = make_synthetic(U, I, CLICKS, 12345, dev)
torch.cuda.empty_cache()
which = sys.argv[1].split(",")
for name in which:
    e = ra.Engine("BPRMF", I, seq_off, seq_item, test, val, K=K, lam=0.01, bias_reg=0.01, precision=F32)
    e.set_option("update_policy", 0); e.set_option("hot_count", 1000000000); e.set_option("chunk", 1 << 30)
    e.init_params(seed=2024)
    T = e.sample_tuples(e.steps_per_epoch, seed=99, epoch=0)
    if name == "plain":
        e.set_option("owner_neg", 0)
    elif name == "plain_partitioned":
        T, _ = e.partition_tuples(T, 256); e.set_option("owner_neg", 0)
    elif name == "plain_shuffled_slices":
        # partitioned, then interleaved: tuple k of every slice, slice after slice (what 256 workgroups running side by side see)
        P, q = e.partition_tuples(T, 256)
        m = int(np.diff(q).min())
        idx = (q[:-1][None, :] + np.arange(m)[:, None]).reshape(-1)
        T = np.ascontiguousarray(P[idx]); e.set_option("owner_neg", 0)
    elif name == "plain_75":
        T = T[:73815808]; e.set_option("owner_neg", 0)
    elif name == "owner_flush1":
        e.set_option("owner_neg", 1); e.set_option("owner_min", 1); e.set_option("owner_slices", 256); e.set_option("owner_flush", 1)
    elif name == "owner_chunks":
        # the epoch in 64 launches
        e.set_option("owner_neg", 1); e.set_option("owner_min", 1); e.set_option("owner_slices", 256); e.set_option("chunk", 1 << 21)
    elif name == "plain_chunks":
        e.set_option("owner_neg", 0); e.set_option("chunk", 1 << 21)
    elif name == "owner":
        e.set_option("owner_neg", 1); e.set_option("owner_min", 1); e.set_option("owner_slices", 256)
    elif name == "owner_v1":
        e.set_option("owner_neg", 1); e.set_option("owner_min", 1); e.set_option("wide", 0)
    e.sgd_replay(0.05, T, HOGWILD)
    s = e.eval_sums(50)
    print(name, len(T), "auc %.4f hit %.4f" % (s[1] / s[7], s[3] / s[7]), flush=True)
    e.close()
