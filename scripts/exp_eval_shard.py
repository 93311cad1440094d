"""How evaluation time falls with the size of a rank's user range: one engine per shard count N holding users
[0, U / N) of the 1M x 200k x 100M corpus (what rank 0 of an N-rank job holds), rec_eval_sums timed end to end."""
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
    args = ap.parse_args()
    import torch
    seq_off, seq_item, test, val, _ = make_synthetic(args.users, args.items, args.clicks, 12345, torch.device("cuda:0"))
    torch.cuda.empty_cache()
    base = None
    for n in (1, 2, 4, 8):
        ue = args.users // n
        eng = ra.Engine(args.model, args.items, seq_off, seq_item, test, val, K=args.k, lam=0.01, bias_reg=0.01, alpha=0.5,
                        precision=F32, user_begin=0, user_end=ue)
        eng.init_params(seed=1)
        eng.set_option("profile", 1)
        eng.eval_sums(50)
        eng.kernel_stats(1)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            eng.eval_sums(50)
            best = min(best, time.perf_counter() - t0)
        ms, cnt = eng.kernel_stats(1)
        base = base or best
        print(f"shards {n}: {ue} users  eval {best * 1e3:.1f} ms  sweep {ms / max(cnt, 1):.1f} ms  users/s x N = {ue / best * n / 1e6:.2f} M  "
              f"efficiency {base / (best * n):.3f}", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
