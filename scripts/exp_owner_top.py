"""Which items fill the top of the ranking after one epoch in the interleaved order / in owner mode?"""
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
cnt = np.bincount(seq_item, minlength=I)
rank_of = np.empty(I, np.int64); rank_of[np.argsort(-cnt)] = np.arange(I)
for name in sys.argv[1].split(","):
    e = ra.Engine("BPRMF", I, seq_off, seq_item, test, val, K=K, lam=0.01, bias_reg=0.01, precision=F32)
    e.set_option("update_policy", 0); e.set_option("hot_count", 1000000000); e.set_option("chunk", 1 << 30)
    e.init_params(seed=2024)
    W0 = e.download()
    T = e.sample_tuples(e.steps_per_epoch, seed=99, epoch=0)
    if name == "plain":
        e.set_option("owner_neg", 0)
    elif name == "interleaved":
        P, q = e.partition_tuples(T, 256)
        m = int(np.diff(q).min())
        idx = (q[:-1][None, :] + np.arange(m)[:, None]).reshape(-1)
        T = np.ascontiguousarray(P[idx]); e.set_option("owner_neg", 0)
    elif name == "owner":
        e.set_option("owner_neg", 1); e.set_option("owner_min", 1); e.set_option("owner_slices", 256)
    e.sgd_replay(0.05, T, HOGWILD)
    s = e.eval_sums(50)
    W1 = e.download()
    users = np.arange(0, U, U // 200)[:200].astype(np.int32)
    items, scores = e.eval_topk(users, 50)
    e.close()
    top = items[:, 1, :].reshape(-1)
    uniq, c = np.unique(top, return_counts=True)
    o = np.argsort(-c)[:12]
    beta1 = W1[:I]; g1 = W1[I + U * K:].reshape(I, K); g0 = W0[I + U * K:].reshape(I, K)
    print(name, "auc %.4f hit %.4f; distinct items in 200 users' top-50: %d" % (s[1] / s[7], s[3] / s[7], len(uniq)))
    print("   most frequent in the top lists: item, times, popularity rank, clicks, beta, |row|, |row0|, slice")
    for k in o:
        j = int(uniq[k])
        print("     %7d %4d %7d %8d %7.3f %6.2f %6.2f %4d" % (j, c[k], rank_of[j], cnt[j], beta1[j], np.linalg.norm(g1[j]), np.linalg.norm(g0[j]), j // 782))
    print("   popularity rank of the items in the top lists: median %d, 90%% %d" % (np.median(rank_of[top]), np.percentile(rank_of[top], 90)))
    print("   beta: mean of top-100 popular %.3f, of all %.3f; min %.3f max %.3f" % (beta1[np.argsort(-cnt)[:100]].mean(), beta1.mean(), beta1.min(), beta1.max()))
    nrm = np.linalg.norm(g1, axis=1)
    print("   |row| quantiles 1%% %.2f 50%% %.2f 99%% %.2f max %.2f (item %d, rank %d)" % (np.percentile(nrm, 1), np.median(nrm), np.percentile(nrm, 99), nrm.max(), nrm.argmax(), rank_of[nrm.argmax()]), flush=True)
