"""Accuracy per epoch of the scatter policies on a mid-size synthetic corpus (no CPU reference at
this scale: policy 0 = float atomics on every row, tuple-major sampling, is the yardstick)."""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=200_000)
    ap.add_argument("--items", type=int, default=50_000)
    ap.add_argument("--clicks", type=int, default=20_000_000)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--every", type=int, default=5)
    ap.add_argument("--model", default="BPRMF")
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--l", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=1 << 22)
    ap.add_argument("--variants", default="0:0,1:0,1:2,2:0")
    args = ap.parse_args()
    import torch
    seq_off, seq_item, test, val, _ = make_synthetic(args.users, args.items, args.clicks, 7, torch.device("cuda:0"))
    for var in args.variants.split(","):
        # a variant is a '+'-joined list of: "<policy>:<group>[:run_hot:run_hot_atomic:run_shared]", p<policy>, s<stale_window>,
        # g<global_window>, m<hex atomic mask on top of the hybrid policy>, u<user_batch>, h<hot_count>, e<event_phases>, n<share_negative>, q<share_positive>
        mask = stale = gwin = ub = hotc = phases = negshare = posshare = None
        policy, group, rest = -1, 0, []
        for tok in var.split("+"):
            if tok[0] == "s":
                stale = int(tok[1:])
            elif tok[0] == "g":
                gwin = int(tok[1:])
            elif tok[0] == "m":
                mask, policy = int(tok[1:], 16), 1
            elif tok[0] == "u":
                ub = int(tok[1:])
            elif tok[0] == "h":
                hotc = int(tok[1:])
            elif tok[0] == "e":
                phases = int(tok[1:])
            elif tok[0] == "n":
                negshare = int(tok[1:])
            elif tok[0] == "q":
                posshare = int(tok[1:])
            elif tok[0] == "p":
                policy = int(tok[1:])
            else:
                policy, group, *rest = (int(x) for x in tok.split(":"))
        po = pi = None
        if args.model == "REBUS":
            lo, li = ra.fsub_mine(args.items, seq_off, seq_item, args.l, 2)
            po, pi = ra.build_rebus_paths(args.items, seq_off, seq_item, val, lo, li)
        eng = ra.Engine(args.model, args.items, seq_off, seq_item, test, val, K=args.k, L=args.l, lam=0.01, bias_reg=0.01,
                        alpha=-0.6 if args.model == "REBUS" else 0.5, precision=F32, path_off=po, path_item=pi)
        if policy >= 0:
            eng.set_option("update_policy", policy)
        eng.set_option("group_by_item", group)
        if stale is not None:
            eng.set_option("stale_window", stale)
        if gwin is not None:
            eng.set_option("global_window", gwin)
        if mask is not None:
            eng.set_option("atomic", mask)
        if ub is not None:
            eng.set_option("user_batch", ub)
        if hotc is not None:
            eng.set_option("hot_count", hotc)
        if phases is not None:
            eng.set_option("event_phases", phases)
        if negshare is not None:
            eng.set_option("share_negative", negshare)
        if posshare is not None:
            eng.set_option("share_positive", posshare)
        eng.set_option("chunk", args.chunk)
        for name, v in zip(("run_hot", "run_hot_atomic", "run_shared"), rest):
            eng.set_option(name, v)
        eng.set_option("profile", 1)
        eng.init_params(seed=1)
        t0 = time.perf_counter()
        line = []
        best = (0.0, 0.0, 0.0, 0)   # validation AUC, test AUC and HIT@50 at that epoch, epoch: what the reference's early stopping keeps
        for ep in range(args.epochs):
            eng.sgd_epoch(args.lr, eng.steps_per_epoch, seed=5, epoch=ep)
            if (ep + 1) % args.every == 0:
                eng.sync()
                m = eng.eval()[0]
                if m.auc_val > best[0]:
                    best = (m.auc_val, m.auc_test, m.hit_test, ep + 1)
                if args.every > 1 or (ep + 1) % 5 == 0:
                    line.append(f"ep{ep + 1}: val {m.auc_val:.4f} test {m.auc_test:.4f} hit {m.hit_test:.4f}")
        eng.sync()
        t_train = time.perf_counter() - t0
        um, un = eng.kernel_stats(0)
        sm, sn = eng.kernel_stats(2)
        W = eng.download()
        print(f"{var}: best val {best[0]:.4f} -> test {best[1]:.4f} hit {best[2]:.4f} @ep{best[3]} | " + " | ".join(line) + f"  finite={np.isfinite(W).all()} |W|max={np.abs(W).max():.2f} "
              f"{t_train:.2f}s update {um / max(un, 1):.3f} ms x {un} sampler {sm / max(sn, 1):.3f} ms x {sn}", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
