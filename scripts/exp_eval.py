"""Full-catalog evaluation alone on a synthetic corpus (for profiling the scoring kernel)."""
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
    ap.add_argument("--users", type=int, default=200_000)
    ap.add_argument("--items", type=int, default=200_000)
    ap.add_argument("--clicks", type=int, default=4_000_000)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--model", default="BPRMF")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--ranges", type=int, default=0, help="eval_ranges: ranges the screen cuts the catalog into (0 = automatic)")
    ap.add_argument("--screen", type=int, default=1, help="eval_screen: 0 fp64 sweep, 1 f16 screen, 2 f32 screen")
    ap.add_argument("--lib", default="", help="another build of the library to load (A/B runs of kernel variants)")
    args = ap.parse_args()
    if args.lib:
        ra.capi.LIB_PATH = os.path.abspath(args.lib)
    import torch
    seq_off, seq_item, test, val, _ = make_synthetic(args.users, args.items, args.clicks, 7, torch.device("cuda:0"))
    eng = ra.Engine(args.model, args.items, seq_off, seq_item, test, val, K=args.k, lam=0.01, bias_reg=0.01, alpha=0.5, precision=F32)
    eng.init_params(seed=1)
    eng.set_option("profile", 1)
    eng.set_option("eval_ranges", args.ranges)
    eng.set_option("eval_screen", args.screen)
    for _ in range(args.reps):
        t0 = time.perf_counter()
        eng.eval_sums(50)
        dt = time.perf_counter() - t0
        ms, n = eng.kernel_stats(1)
        ctx = 1 if args.model in ("BPRMF", "MP") else 2
        kq = args.k if args.model not in ("FPMC", "PRME") else 2 * args.k
        flop = 2.0 * ctx * args.users * args.items * kq
        print(f"eval {dt:.3f} s  score kernel {ms:.2f} ms  {flop / (ms * 1e-3) / 1e12:.2f} TFLOP/s f64", flush=True)


if __name__ == "__main__":
    main()
