"""Does the length of a launch (the stretch over which Fossil's global weights / FossilSimple's alpha stay constant and collect
their gradient) change what they learn at the 1M-user x 200k-item x 100M-interaction shape?  Test AUC / HIT@50 after each epoch.
usage: exp_fossil_window.py MODEL EPOCHS WINDOW[,WINDOW...] [SEEDS]"""
import sys
import time

sys.path.insert(0, ".")
import torch
import recommendation_amd as ra
from bench import make_synthetic
from recommendation_amd.capi import F32

U, I, CLICKS, K = 1_000_000, 200_000, 100_000_000, 64
model, epochs = sys.argv[1], int(sys.argv[2])
windows = [int(w) for w in sys.argv[3].split(",")]
seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda:0")
seq_off, seq_item, test, val, _ = make_synthetic(U, I, CLICKS, 12345, dev)
torch.cuda.empty_cache()
e = ra.Engine(model, I, seq_off, seq_item, test, val, K=K, L=3, lam=0.01, bias_reg=0.01, relation_reg=0.01, alpha=0.5, precision=F32)
for w in windows:
    e.set_option("global_window", w)
    for seed in range(seeds):
        e.init_params(seed=2024 + seed)
        line = []
        for ep in range(epochs):
            t0 = time.perf_counter()
            e.sgd_epoch(0.05, e.steps_per_epoch, seed=seed, epoch=ep)
            e.sync()
            dt = time.perf_counter() - t0
            s = e.eval_sums(50)
            line.append("%.0f ms auc %.4f hit %.4f" % (dt * 1e3, s[1] / s[7], s[3] / s[7]))
        print(model, "window", w, "seed", seed, " | ".join(line), flush=True)
e.close()
