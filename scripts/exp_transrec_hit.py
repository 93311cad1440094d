"""TransRec at the 1M-user shape: what the hybrid scatter's plain stores (user rows, biases, hot items) cost in HIT@50 after two
epochs against float atomics everywhere; means over seeds, time per epoch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recommendation_amd as ra  # noqa: E402
from recommendation_amd.capi import F32  # noqa: E402
from bench import make_synthetic  # noqa: E402


def main():
    import torch
    model = sys.argv[1] if len(sys.argv) > 1 else "TransRec"
    U, I, K = 1_000_000, 200_000, 64
    seq_off, seq_item, test, val, _ = make_synthetic(U, I, 100_000_000, 12345, torch.device("cuda:0"))
    torch.cuda.empty_cache()
    e = ra.Engine(model, I, seq_off, seq_item, test, val, K=K, L=3, lam=0.01, bias_reg=0.01, relation_reg=0.01, alpha=0.5, precision=F32)
    spe = e.steps_per_epoch
    variants = [("hot_x4", [("update_policy", 1), ("hot_count", spe // 256)]), ("hot_x16", [("update_policy", 1), ("hot_count", spe // 64)]),
                ("hot_x64", [("update_policy", 1), ("hot_count", spe // 16)])] if len(sys.argv) > 2 else [("default", [("update_policy", 1)]), ("biases_atomic", [("update_policy", 1), ("atomic", 0x1fe)]),
                ("no_hot", [("update_policy", 1), ("hot_count", 0)]), ("no_hot_biases_atomic", [("update_policy", 1), ("hot_count", 0), ("atomic", 0x1fe)]),
                ("user_rows_atomic", [("update_policy", 1), ("atomic", 0x0ff)]), ("atomics", [("update_policy", 0)])]
    for name, opts in variants:
        for k, v in opts:
            e.set_option(k, v)
        res = []
        t_ep = []
        for seed in (99, 7, 31):
            e.init_params(seed=2024 + seed)
            for ep in range(2):
                e.sync(); t0 = time.perf_counter()
                e.sgd_epoch(0.05, e.steps_per_epoch, seed=seed, epoch=ep)
                e.sync(); t_ep.append(time.perf_counter() - t0)
            s = e.eval_sums(50)
            res.append((s[1] / s[7], s[3] / s[7]))
        r = np.array(res)
        print("%-22s auc %.5f hit %.5f  (hits %s)  %.1f ms/epoch" % (name, r[:, 0].mean(), r[:, 1].mean(), r[:, 1].round(4).tolist(), 1e3 * min(t_ep)), flush=True)


if __name__ == "__main__":
    main()
