"""Sequential CPU restatement against the engine's large-corpus defaults on a corpus big enough for those defaults to
switch on by themselves (>= 4M steps per epoch): the same tuple distribution (the engine's sampler, independent tuples)
applied one by one in fp64 by oracle/ (updateFactors restated) and Hogwild-style by the engine; every checkpoint is
scored by the engine's evaluation.  One process per model.

  python scripts/exp_seq_vs_default.py --models Fossil,FossilSimple,REBUS --epochs 4 --out gpurun_out/seq_vs_default.json
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def corpus_and_kw(args, model):
    import torch
    import recommendation_amd as ra
    from bench import make_synthetic
    seq_off, seq_item, test, val, _ = make_synthetic(args.users, args.items, args.clicks, args.corpus_seed, torch.device("cuda:0"))
    torch.cuda.empty_cache()
    kw = dict(K=args.k, L=args.l, lam=0.01, bias_reg=0.01, relation_reg=0.01, alpha=-0.6 if model == "REBUS" else 0.5)
    po = pi = None
    if model == "REBUS":
        lo, li = ra.fsub_mine(args.items, seq_off, seq_item, args.l, 2)
        po, pi = ra.build_rebus_paths(args.items, seq_off, seq_item, val, lo, li)
    return seq_off, seq_item, test, val, kw, po, pi


def metrics_of(e):
    m, _, _ = e.eval(want_counts=False)
    return [round(float(x), 6) for x in (m.auc_val, m.auc_test, m.hit_val, m.hit_test)]


def run_model(args, model, q, proc=0):
    import recommendation_amd as ra
    from oracle import oracle_py as orc
    from recommendation_amd.capi import F32
    seq_off, seq_item, test, val, kw, po, pi = corpus_and_kw(args, model)
    I, U = args.items, args.users
    out = {"model": model, "users": U, "items": I, "clicks": args.clicks, "corpus_seed": args.corpus_seed, "K": args.k, "L": args.l,
           "lr": args.lr, "runs": []}

    def engine(opts=()):
        e = ra.Engine(model, I, seq_off, seq_item, test, val, precision=F32, path_off=po, path_item=pi, **kw)
        for k_, v_ in opts:
            e.set_option(k_, v_)
        return e

    # the engine as shipped, and with float atomics on every row
    variants = [("default", ())] + ([("atomics_everywhere", (("update_policy", 0),))] if args.with_atomics else [])
    for spec in filter(None, args.variants.split(";")):      # name:opt=val,opt=val
        name, _, body = spec.partition(":")
        variants.append((name, tuple((kv.split("=")[0], int(kv.split("=")[1])) for kv in body.split(",") if kv)))
    if args.only_variants:
        variants = variants[1 + int(bool(args.with_atomics)):]
    for name, opts in variants:
        for seed in range(args.engine_seeds):
            e = engine(opts)
            e.init_params(seed)
            traj = []
            t0 = time.time()
            for ep in range(1, args.epochs * args.evals_per_epoch + 1):
                e.sgd_epoch(args.lr, e.steps_per_epoch // args.evals_per_epoch, seed=seed, epoch=ep)
                traj.append([ep] + metrics_of(e))
            out["steps_per_epoch"] = int(e.steps_per_epoch)
            e.close()
            out["runs"].append({"kind": name, "seed": seed, "trajectory": traj, "seconds": round(time.time() - t0, 1)})
            print(model, name, seed, traj[-1], flush=True)
    # the sequential restatement on tuples the engine's sampler draws (independent tuples: user_batch 1)
    o = orc.Oracle(model, U, I, args.k, seq_off, seq_item, test, val, path_off=po, path_item=pi, **{k_: v_ for k_, v_ in kw.items() if k_ != "K"})
    for seed in range(proc * args.oracle_seeds, (proc + 1) * args.oracle_seeds):
        e = engine((("user_batch", 1),))
        e.init_params(100 + seed)
        W = e.download()
        traj = []
        t0 = time.time()
        for ep in range(1, args.epochs * args.evals_per_epoch + 1):
            T = e.sample_tuples(e.steps_per_epoch // args.evals_per_epoch, seed=100 + seed, epoch=ep)
            W = o.replay(W, T, args.lr)
            e.upload(W)
            traj.append([ep] + metrics_of(e))
            print(model, "sequential", seed, traj[-1], "%.0f s" % (time.time() - t0), flush=True)
        e.close()
        out["runs"].append({"kind": "sequential_fp64_restatement", "seed": 100 + seed, "trajectory": traj, "seconds": round(time.time() - t0, 1)})
    q.put(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="Fossil,FossilSimple,REBUS")
    ap.add_argument("--users", type=int, default=60_000)
    ap.add_argument("--items", type=int, default=20_000)
    ap.add_argument("--clicks", type=int, default=6_000_000)
    ap.add_argument("--corpus-seed", type=int, default=4242)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--l", type=int, default=3)
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--engine-seeds", type=int, default=2)
    ap.add_argument("--oracle-seeds", type=int, default=2)
    ap.add_argument("--with-atomics", type=int, default=1)
    ap.add_argument("--evals-per-epoch", type=int, default=1, help="checkpoints per epoch (each after steps_per_epoch / this many steps)")
    ap.add_argument("--variants", default="", help="more engine runs: name:opt=val,opt=val;name2:...")
    ap.add_argument("--only-variants", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    # a model named more than once gets one process per mention, each with its own oracle seeds
    procs = [ctx.Process(target=run_model, args=(args, m, q, i)) for i, m in enumerate(args.models.split(","))]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join()
    res.sort(key=lambda r: r["model"])
    merged = {}
    for r in res:
        if r["model"] in merged:
            merged[r["model"]]["runs"] += r["runs"]
        else:
            merged[r["model"]] = r
    res = list(merged.values())
    for r in res:
        print(r["model"], "steps per epoch", r.get("steps_per_epoch"))
        for run in r["runs"]:
            print("  %-28s seed %3d  " % (run["kind"], run["seed"]) + " ".join("%.4f/%.3f" % (t[2], t[4]) for t in run["trajectory"]) + "  | best-val test auc %.4f" % max(run["trajectory"], key=lambda t: t[1])[2])
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
