"""Stability of the update modes on the 1M-user x 200k-item x 100M-interaction corpus (BPRMF K = 64): test AUC / HIT@50
after every epoch and the largest |value| of the item tables, for the scatter modes side by side.
usage: exp_c2_stability.py EPOCHS "opt=val,opt=val;opt=val" """
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch
import recommendation_amd as ra
from bench import make_synthetic
from recommendation_amd.capi import F32
from recommendation_amd.sharded import item_table_tensors

U, I, CLICKS, K = 1_000_000, 200_000, 100_000_000, 64
epochs = int(sys.argv[1])
variants = [[(kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv] for v in sys.argv[2].split(";")]
dev = torch.device("cuda:0")
seq_off, seq_item, test, val, _ = make_synthetic(U, I, CLICKS, 12345, dev)
torch.cuda.empty_cache()
cnt = np.bincount(seq_item, minlength=I)
top = np.argsort(-cnt)[:4]
print("most clicked items", top, cnt[top], "of", len(seq_item), flush=True)
for opts in variants:
    e = ra.Engine("BPRMF", I, seq_off, seq_item, test, val, K=K, lam=0.01, bias_reg=0.01, precision=F32)
    for k, v in opts:
        e.set_option(k, v)
    e.init_params(seed=2024)
    tabs = item_table_tensors(e, dev)
    line = []
    for ep in range(epochs):
        t0 = time.perf_counter()
        e.sgd_epoch(0.05, e.steps_per_epoch, seed=99, epoch=ep)
        e.sync()
        dt = time.perf_counter() - t0
        s = e.eval_sums(50)
        mx = max(float(t.abs().max()) for t in tabs)
        gam = tabs[-1].view(I, K); beta = tabs[0]
        line.append("ep %d %.1f ms auc %.4f hit %.4f max|w| %.2f |top rows| %s beta %s max|beta| %.2f" % (ep, dt * 1e3, s[1] / s[7], s[3] / s[7], mx,
                    np.round(gam[torch.as_tensor(top, device=dev)].norm(dim=1).cpu().numpy(), 2),
                    np.round(beta[torch.as_tensor(top, device=dev)].cpu().numpy(), 2), float(beta.abs().max())))
    bj = int(beta.abs().argmax())
    print(opts, "item of max|beta|: %d (slice of 782: %d, of 391: %d, clicks %d), beta %.2f" % (bj, bj // 782, bj // 391, cnt[bj], float(beta[bj])),
          "items with |beta| > 2.5:", int((beta.abs() > 2.5).sum()), "their clicks (first 10):", cnt[(beta.abs() > 2.5).cpu().numpy()[:I]][:10])
    for l in line:
        print("   ", l, flush=True)
    e.close()
