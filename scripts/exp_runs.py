"""Isolated timing of the Hogwild update kernels on one chunk of the bench workload:
ungrouped vs grouped-by-positive tuples, and which scatters cost what (atomic mask variants)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recommendation_amd as ra  # noqa: E402
from recommendation_amd.capi import F32, HOGWILD  # noqa: E402
from bench import make_synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=200_000)
    ap.add_argument("--clicks", type=int, default=100_000_000)
    ap.add_argument("--n", type=int, default=16_313_777)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    import torch
    seq_off, seq_item, test, val, _ = make_synthetic(args.users, args.items, args.clicks, 7, torch.device("cuda:0"))
    eng = ra.Engine("BPRMF", args.items, seq_off, seq_item, test, val, K=args.k, lam=0.01, bias_reg=0.01, precision=F32)
    eng.init_params(seed=1)
    eng.set_option("profile", 1)
    eng.set_option("chunk", args.n)
    eng.set_option("group_by_item", 0)
    plain = eng.sample_tuples(args.n, seed=3, epoch=0)
    t0 = time.perf_counter()
    grouped = eng.sample_events(args.n, seed=3, epoch=0)
    print("event-major sample + copy s", time.perf_counter() - t0, "tuples", len(grouped), flush=True)
    by_user = plain[np.argsort(plain[:, 0], kind="stable")]

    def run(name, tuples, group, mask, hot):
        eng.set_option("group_by_item", group)
        eng.set_option("atomic", mask)
        eng.set_option("hot_count", hot)
        eng.kernel_stats(0)
        for _ in range(args.reps):
            eng.sgd_replay(0.05, tuples, HOGWILD)
        ms, n = eng.kernel_stats(0)
        print(f"{name:58s} {ms / n:8.3f} ms  {1552 * len(tuples) / (ms / n * 1e-3) / 1e12:6.2f} TB/s alg", flush=True)

    hot = max(1, eng.steps_per_epoch // 1024)
    if args.quick:
        run("hogwild ungrouped, all plain", plain, 0, 0, 0)
        run("runs ungrouped, all plain", plain, 1, 0, 0)
        run("runs grouped, all plain", grouped, 1, 0, 0)
        run("runs grouped, pos+neg+bias atomic", grouped, 1, 0x1fe, 0)
        return
    run("hogwild ungrouped, hybrid (items+bias atomic, hot exempt)", plain, 0, 0x1fe, hot)
    run("hogwild ungrouped, item rows atomic, bias plain", plain, 0, 0x0fe, hot)
    run("hogwild ungrouped, all plain", plain, 0, 0, 0)
    run("runs grouped, pos+neg+bias atomic", grouped, 1, 0x1fe, 0)
    run("runs grouped, pos+neg atomic, bias plain", grouped, 1, 0x0fe, 0)
    run("runs grouped, pos atomic only (neg plain, bias plain)", grouped, 1, 0x002, 0)
    run("runs grouped, pos + bias atomic (neg rows plain)", grouped, 1, 0x102, 0)
    run("runs grouped, all plain", grouped, 1, 0, 0)
    run("runs ungrouped, all plain", plain, 1, 0, 0)
    run("runs sorted by user, all plain", by_user, 1, 0, 0)
    shuffled_runs = grouped.reshape(-1, 4)[np.argsort(np.repeat(np.random.default_rng(0).permutation((len(grouped) + 511) // 512), 512)[:len(grouped)], kind="stable")]
    run("runs grouped, 512-tuple pieces shuffled, all plain", shuffled_runs, 1, 0, 0)
    run("hogwild grouped, 512-tuple pieces shuffled, all plain", shuffled_runs, 0, 0, 0)
    run("hogwild sorted by user, hybrid", by_user, 0, 0x1fe, hot)
    run("hogwild sorted by user, all plain", by_user, 0, 0, 0)
    run("hogwild grouped by pos, all plain", grouped, 0, 0, 0)


if __name__ == "__main__":
    main()
