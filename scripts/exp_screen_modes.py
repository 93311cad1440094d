"""Which sweep a trained model's evaluation takes at the full shape: for every screen setting the mode the sweep ended in
(1 screen held, 2 a query overflowed its undecided list and fp64 took over), the undecided columns and the time."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recommendation_amd as ra  # noqa: E402
from recommendation_amd.capi import F32  # noqa: E402
from bench import make_synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=200_000)
    ap.add_argument("--clicks", type=int, default=100_000_000)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--model", default="Fossil")
    ap.add_argument("--epochs", type=int, default=3)
    args = ap.parse_args()
    import torch
    seq_off, seq_item, test, val, _ = make_synthetic(args.users, args.items, args.clicks, 12345, torch.device("cuda:0"))
    eng = ra.Engine(args.model, args.items, seq_off, seq_item, test, val, K=args.k, L=3, lam=0.01, bias_reg=0.01, relation_reg=0.01,
                    alpha=0.5, precision=F32)
    eng.init_params(seed=2024)
    eng.set_option("profile", 1)
    for ep in range(args.epochs + 1):
        if ep:
            eng.sgd_epoch(0.05, eng.steps_per_epoch, seed=99, epoch=ep)
            eng.sync()
        for screen, cap in ((1, 0), (1, 4096), (2, 0), (2, 4096)):
            eng.set_option("eval_screen", screen); eng.set_option("eval_screen_cap", cap)
            t0 = time.perf_counter()
            s = eng.eval_sums(50)
            dt = time.perf_counter() - t0
            open_cols, mode = eng.kernel_stats(4)
            print(f"epoch {ep} screen {screen} cap {cap}: mode {mode} undecided/query {open_cols / (2 * s[7]):.1f} eval {dt * 1e3:.0f} ms auc {s[1] / s[7]:.4f}", flush=True)
    W = eng.download()
    import numpy as np
    off = 0
    for ptr, rows, cols, is_user in eng.tables():
        t = W[off:off + rows * cols]
        print("table", rows, cols, "max|x|", float(np.abs(t).max()), "rms", float(np.sqrt((t * t).mean())), flush=True)
        off += rows * cols
    eng.set_option("eval_screen", 1); eng.set_option("eval_screen_cap", 0)


if __name__ == "__main__":
    main()
