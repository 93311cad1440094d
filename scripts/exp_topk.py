"""Top-50 lists of every user out of the fused sweep on a synthetic corpus (for profiling topk_sweep16_kernel / topk_finish_kernel)."""
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
    ap.add_argument("--model", default="BPRMF")
    ap.add_argument("--epochs", type=int, default=1, help="epochs of training before the lists are taken")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--lib", default="", help="another build of the library to load (A/B runs of kernel variants)")
    args = ap.parse_args()
    if args.lib:
        ra.capi.LIB_PATH = os.path.abspath(args.lib)
    import torch
    seq_off, seq_item, test, val, _ = make_synthetic(args.users, args.items, args.clicks, 12345, torch.device("cuda:0"))
    torch.cuda.empty_cache()
    eng = ra.Engine(args.model, args.items, seq_off, seq_item, test, val, K=args.k, L=3, lam=0.01, bias_reg=0.01, relation_reg=0.01, alpha=0.5, precision=F32)
    eng.init_params(seed=1)
    for ep in range(args.epochs):
        eng.sgd_epoch(0.05, eng.steps_per_epoch, seed=3, epoch=ep)
    eng.sync()
    n = eng.n_eval_users
    for _ in range(args.reps):
        t0 = time.perf_counter()
        eng.eval_topk(n, 50, want_scores=False)
        dt = time.perf_counter() - t0
        ms, lists = eng.kernel_stats(5)
        print(f"{args.model}: top-50 of {n} users in {dt:.3f} s, device {ms:.1f} ms, {lists} lists fused, {n / (ms * 1e-3) / 1e6:.2f} M users/s", flush=True)


if __name__ == "__main__":
    main()
