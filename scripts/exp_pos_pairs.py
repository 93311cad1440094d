"""What pairs of steps that share their positive item (on top of the shared negative) would buy and cost: tuples are
drawn on the host (numpy) and fed through rec_sgd_replay in Hogwild mode, whose update kernel folds the rows that
neighbouring tuples share.  Variant 0: independent positives; variant 1: the second step of a pair takes another event
of the first step's positive item, drawn with weight 1/n_u of the event's user (so that the second step's (user,
position) has the reference's distribution: user uniform, then position uniform); variant 2: uniform among the item's
events instead (favours users with long histories)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendation_amd as ra  # noqa: E402
from recommendation_amd.capi import F32, HOGWILD  # noqa: E402
from bench import make_synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=200_000)
    ap.add_argument("--items", type=int, default=50_000)
    ap.add_argument("--clicks", type=int, default=20_000_000)
    ap.add_argument("--epochs", type=int, default=12)
    args = ap.parse_args()
    import torch
    seq_off, seq_item, test, val, _ = make_synthetic(args.users, args.items, args.clicks, 7, torch.device("cuda:0"))
    U, I = args.users, args.items
    n_u = np.diff(seq_off)
    ev_user = np.repeat(np.arange(U, dtype=np.int32), n_u)
    order = np.argsort(seq_item, kind="stable")
    by_item_user = ev_user[order]
    item_start = np.searchsorted(seq_item[order], np.arange(I + 1))
    cw = np.concatenate([[0.0], np.cumsum(1.0 / n_u[by_item_user])])   # cumulative event weights in item order
    rng = np.random.default_rng(1)
    for variant in (0, 1, 2):
        eng = ra.Engine("BPRMF", I, seq_off, seq_item, test, val, K=64, lam=0.01, bias_reg=0.01, precision=F32)
        eng.set_option("share_negative", 2)
        eng.set_option("chunk", 1 << 27)
        eng.set_option("profile", 1)
        eng.init_params(seed=1)
        n = int(eng.steps_per_epoch) // 2 * 2
        line = []
        for ep in range(args.epochs):
            u = rng.integers(0, U, n).astype(np.int32)
            pos = seq_item[seq_off[u] + (rng.random(n) * n_u[u]).astype(np.int64)]
            if variant:
                p0 = pos[0::2]
                if variant == 1:
                    lo, hi = cw[item_start[p0]], cw[item_start[p0 + 1]]
                    k = np.searchsorted(cw, lo + rng.random(n // 2) * (hi - lo), side="right") - 1
                    k = np.clip(k, item_start[p0], item_start[p0 + 1] - 1)
                else:
                    k = item_start[p0] + (rng.random(n // 2) * (item_start[p0 + 1] - item_start[p0])).astype(np.int64)
                u[1::2] = by_item_user[k]; pos[1::2] = p0
            neg = np.repeat(rng.integers(0, I, n // 2).astype(np.int32), 2)
            T = np.stack([u, pos.astype(np.int32), neg, np.zeros(n, np.int32)], axis=1)
            eng.sgd_replay(0.05, T, HOGWILD)
            print(f"[variant {variant} epoch {ep + 1}]", file=sys.stderr, flush=True)
            if (ep + 1) % 4 == 0:
                m = eng.eval()[0]
                line.append(f"ep{ep + 1}: val {m.auc_val:.4f} test {m.auc_test:.4f} hit {m.hit_test:.4f}")
        um, un = eng.kernel_stats(0)
        print(f"variant {variant}: " + " | ".join(line) + f"  update {um / max(un, 1):.3f} ms x {un}", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
