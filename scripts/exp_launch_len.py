"""Does the length of an update launch change what a history model learns?  REBUS / Fossil on the 60k-user corpus, two epochs
applied in 1, 4 or 16 calls per epoch; test AUC after each epoch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recommendation_amd as ra  # noqa: E402
from recommendation_amd.capi import F32  # noqa: E402
from bench import make_synthetic  # noqa: E402


def main():
    import torch
    model = sys.argv[1] if len(sys.argv) > 1 else "REBUS"
    opts = [kv.split("=") for kv in sys.argv[2:]]
    U, I = 60_000, 20_000
    seq_off, seq_item, test, val, _ = make_synthetic(U, I, 6_000_000, 4242, torch.device("cuda:0"))
    po = pi = None
    if model == "REBUS":
        lo, li = ra.fsub_mine(I, seq_off, seq_item, 3, 2)
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
    e = ra.Engine(model, I, seq_off, seq_item, test, val, K=64, L=3, lam=0.01, bias_reg=0.01, relation_reg=0.01,
                  alpha=-0.6 if model == "REBUS" else 0.5, precision=F32, path_off=po, path_item=pi)
    for k, v in opts:
        e.set_option(k, int(v))
    for calls in (1, 4, 16, 1, 4):
        e.init_params(0)
        out = []
        for ep in range(2):
            for c in range(calls):
                e.sgd_epoch(0.05, e.steps_per_epoch // calls, seed=0, epoch=ep * calls + c + 1)
            m, _, _ = e.eval(want_counts=False)
            out.append(round(m.auc_test, 4))
        print(model, opts, "calls per epoch", calls, out, flush=True)


if __name__ == "__main__":
    main()
