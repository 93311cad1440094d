#!/usr/bin/env python3
"""Throughput of the hot path on synthetic data: BPR-SGD triples/s (headline) and full-catalog
evaluation users/s, for BPRMF K=64 on 1M users x 200k items x 100M interactions.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A step is one SGD epoch: num_pos_events updateFactors calls (model.hpp:47-51) drawn by the device
sampler and applied by the Hogwild update kernel.  With N ranks the users are sharded, every rank
applies 1/N of the epoch's tuples to its own users, and the item tables are combined once per
step with an RCCL all-reduce of the per-rank deltas (total work fixed: strong scaling).
Evaluation is sharded by user; the metric sums are all-reduced.

One JSON line on stdout (rank 0).  `roofline` prices the update kernel of the engine's defaults (the
reference's sampling: every step draws its own negative) against HBM with the algorithmic bytes per
triple (each distinct row and bias a step names read once and written once: 3 K-float rows + 2 biases
= 1552 B for BPRMF K=64 f32) and the kernel's average duration from HIP events on the engine's stream;
`variants` (N=1 only) holds the same measurement for the other scatter modes and for f64 storage;
`cpu_baseline` times BPRMF::oneiteration of the reference binary built under oracle/_ref on a user sample
of the same corpus (`cpu_port`: the single-thread CPU restatement on the whole corpus); `eval.cpu_baseline`
times the reference binary's own model::AUC on a user sample over the full catalog.
`roofline.traffic` is null: HBM bytes come from rocprofv3 --pmc passes (profiles/), never from this run.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, spec
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix, spec (v_mfma_f64_16x16x4_f64)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X FP32 matrix, spec (v_mfma_f32_16x16x4_f32)
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense f16 / bf16 matrix (v_mfma_f32_16x16x32_f16)


def make_synthetic(n_users, n_items, n_clicks, seed, device):
    """Per-user lengths 3 + geometric (mean n_clicks / n_users), items Zipf(0.9) over popularity
    ranks, half of a user's clicks shifted into one of 64 taste clusters, ranks mapped to item ids
    by a fixed permutation.  Returns the post-split training CSR and the test / validation items
    (every user has >= 3 clicks, so every user is evaluated)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    mean_extra = n_clicks / n_users - 3.0
    p = 1.0 / (mean_extra + 1.0)
    u = torch.rand(n_users, generator=g, device=device, dtype=torch.float64).clamp_min(1e-300)
    n = 3 + torch.floor(torch.log(u) / math.log(1.0 - p)).long().clamp_max(4000)
    off = torch.zeros(n_users + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(n, 0)
    total = int(off[-1])
    w = torch.arange(1, n_items + 1, device=device, dtype=torch.float64).pow(-0.9)
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1].clone()
    perm = torch.randperm(n_items, generator=g, device=device)
    cluster = torch.randint(0, 64, (n_users,), generator=g, device=device)
    items = torch.empty(total, dtype=torch.int32, device=device)
    user_of = torch.repeat_interleave(torch.arange(n_users, device=device), n)
    step = 1 << 24
    for a in range(0, total, step):
        b = min(total, a + step)
        r = torch.rand(b - a, generator=g, device=device, dtype=torch.float64)
        z = torch.searchsorted(cdf, r).clamp_max(n_items - 1)
        shifted = torch.rand(b - a, generator=g, device=device) < 0.5
        shift = cluster[user_of[a:b]] * (n_items // 64)
        z = torch.where(shifted, (z + shift) % n_items, z)
        items[a:b] = perm[z].int()
    pos = torch.arange(total, device=device) - off[:-1][user_of]
    keep = pos < (n[user_of] - 2)
    seq_item = items[keep]
    seq_off = torch.zeros(n_users + 1, dtype=torch.int64, device=device)
    seq_off[1:] = torch.cumsum(n - 2, 0)
    test_item = items[off[1:] - 1]
    val_item = items[off[1:] - 2]
    out = [t.cpu().numpy() for t in (seq_off, seq_item, test_item, val_item)]
    del items, user_of, pos, keep, seq_item
    return out + [total]


class DevArray:
    """Engine-owned device memory seen through the CUDA array interface (for torch.distributed)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def _time_reference(args, seq_off, seq_item, test_item, val_item, K):
    """BPRMF::oneiteration of the reference binary oracle/_ref/ref_driver (when it was built) on a click file holding the
    first --cpu-ref-users users of the corpus: seconds per epoch as the binary reports them -> triples/s."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if args.cpu_ref_users <= 0 or not os.path.exists(exe):
        return None
    nu = min(args.cpu_ref_users, len(seq_off) - 1)
    work = tempfile.mkdtemp(prefix="recsys_ref_")
    try:
        os.makedirs(os.path.join(work, "01-Data"))
        path = os.path.join(work, "01-Data", "sample.txt")
        lens = np.diff(seq_off[:nu + 1])
        users = np.repeat(np.arange(nu), lens + 2)
        # training sequence, then the validation and the test item, with increasing time stamps
        items = np.empty(len(users), np.int64); times = np.empty(len(users), np.int64)
        pos = 0
        starts = np.concatenate([[0], np.cumsum(lens + 2)])
        for u in range(nu):
            a, b = int(seq_off[u]), int(seq_off[u + 1])
            n = b - a
            items[starts[u]:starts[u] + n] = seq_item[a:b]
            items[starts[u] + n] = val_item[u]; items[starts[u] + n + 1] = test_item[u]
            times[starts[u]:starts[u + 1]] = np.arange(n + 2)
        np.savetxt(path, np.stack([users, items, np.ones_like(users), times], 1), fmt="%d")
        steps = int(lens.sum())
        r = subprocess.run([exe, "BPRMF", "01-Data/sample.txt", str(K), "1", "1", "0.01", "0.01", "0", "0", "-", "traj:2:1000000",
                            "0.05", os.path.join(work, "out")], cwd=work, capture_output=True, text=True, timeout=600)
        secs = [float(x) for x in re.findall(r"EPOCH \d+ ([0-9.]+)", r.stderr + r.stdout)]
        if r.returncode != 0 or not secs:
            return None
        best = min(secs)
        return {"value": round(steps / best, 1), "unit": "triples/s", "cores": 1, "kind": "reference",
                "sample": f"BPRMF::oneiteration of the reference binary on the first {nu} users of the same corpus "
                          f"(all {args.items} items, {steps} steps per epoch, fp64), best of {len(secs)} epochs, {best:.1f} s"}
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _time_reference_eval(args, seq_off, seq_item, test_item, val_item, K, I):
    """model::AUC of the reference binary (oracle/_ref/ref_driver, BPRMF) on the first --cpu-ref-eval-users users of the
    corpus ranked against the whole catalog: the click file holds those users plus two-click filler users (never
    evaluated: model.hpp:22-31 needs three clicks) that bring every item of the catalog into the corpus."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if args.cpu_ref_eval_users <= 0 or not os.path.exists(exe):
        return None
    nu = min(args.cpu_ref_eval_users, len(seq_off) - 1)
    work = tempfile.mkdtemp(prefix="recsys_refeval_")
    try:
        os.makedirs(os.path.join(work, "01-Data"))
        path = os.path.join(work, "01-Data", "sample.txt")
        rows = []
        for u in range(nu):
            a, b = int(seq_off[u]), int(seq_off[u + 1])
            its = list(seq_item[a:b]) + [int(val_item[u]), int(test_item[u])]
            rows += [(u, int(i), 1, t) for t, i in enumerate(its)]
        filler = np.arange(I + (I & 1))
        filler[-1] = min(filler[-1], I - 1)
        for k in range(len(filler) // 2):
            rows += [(nu + k, int(filler[2 * k]), 1, 0), (nu + k, int(filler[2 * k + 1]), 1, 1)]
        np.savetxt(path, np.array(rows, np.int64), fmt="%d")
        r = subprocess.run([exe, "BPRMF", "01-Data/sample.txt", str(K), "1", "1", "0.01", "0.01", "0", "0", "-", "traj:1:1",
                            "0.05", os.path.join(work, "out")], cwd=work, capture_output=True, text=True, timeout=900)
        secs = [float(x) for x in re.findall(r"EVALSEC \d+ ([0-9.]+)", r.stderr + r.stdout)]
        if r.returncode != 0 or not secs:
            return None
        return {"value": round(nu / secs[0], 2), "unit": "users/s", "cores": 1, "kind": "reference",
                "sample": f"model::AUC of the reference binary on the first {nu} users of the same corpus, all {I} items, "
                          f"K={K}, fp64, {secs[0]:.1f} s"}
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=200_000)
    ap.add_argument("--clicks", type=int, default=100_000_000)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--model", default="BPRMF")
    ap.add_argument("--l", type=int, default=3, help="L of Fossil / REBUS")
    ap.add_argument("--min-count", type=int, default=2, help="minCount of the REBUS label set")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--group", type=int, default=-1, help="group_by_item: 0 off, 1 run-combining kernel only, 2 event-major sampling (chunks grouped by positive item); -1 = engine default")
    ap.add_argument("--policy", type=int, default=-1, help="update_policy: 0 exact atomics, 1 hybrid, 2 plain stores; -1 = engine default")
    ap.add_argument("--hot-count", type=int, default=-1, help="hot_count: items with at least this many training occurrences get plain stores under the hybrid policy; -1 = engine default")
    ap.add_argument("--chunk", type=int, default=1 << 27)
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("--share-positive", type=int, default=0, help="share_positive of the group models (1, 2); 0 = engine default")
    ap.add_argument("--share-negative", type=int, default=0, help="share_negative of the group models (1, 2); 0 = engine default")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="any other engine option (rec_set_option), applied last")
    ap.add_argument("--user-batch", type=int, default=0, help="user_batch of the history models (1, 2, 4, 8); 0 = engine default")
    ap.add_argument("--owner", type=int, default=-1, help="owner_neg: negatives' deltas summed in LDS by the workgroup that owns their slice of the catalog (0, 1); -1 = engine default")
    ap.add_argument("--part-tile", type=int, default=0, help="tuples the owner mode's partition sorts in LDS at a time (0 = engine default)")
    ap.add_argument("--owner-flush", type=int, default=0, help="owner_flush (additions per atomic row); 0 = engine default")
    ap.add_argument("--cpu-ref-eval-users", type=int, default=1000, help="users the reference binary's model::AUC is timed on (0 = skip)")
    ap.add_argument("--wide", type=int, default=-1, help="wide: the four-components-per-lane form of the update kernel (0, 1); -1 = engine default")
    ap.add_argument("--atomic-mask", type=int, default=-1, help="experiment: the engine's 'atomic' row mask (bit r = row r of the tuple, bit 8 = biases)")
    ap.add_argument("--rounds", type=int, default=0, help="N > 1: item-table all-reduces per epoch (0 = the accuracy-preserving default, ceil(steps / (2 items)))")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra measurements of the other scatter modes and of f64 storage")
    ap.add_argument("--cpu-eval-users", type=int, default=16384, help="users of the CPU evaluation baseline (0 = skip)")
    ap.add_argument("--cpu-ref-users", type=int, default=60000, help="users of the click file the reference binary is timed on (0 = skip)")
    ap.add_argument("--cpu-steps", type=int, default=12_000_000, help="tuples of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--eval-calls", type=int, default=5, help="timed full-catalog evaluations (after one untimed call)")
    ap.add_argument("--no-top50", action="store_true", help="skip the timing of the top-50 lists of all users")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: collectives staged through host memory (rehearsal of the N-rank path on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="every rank uses device 0 (only with --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import recommendation_amd as ra
    from recommendation_amd.capi import F32, F64

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    def all_reduce(t, op):
        """In-place all-reduce of a device tensor over RCCL (or, rehearsing, over gloo via the host)."""
        if args.backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    seq_off, seq_item, test_item, val_item, n_clicks = make_synthetic(args.users, args.items, args.clicks, 12345, dev)
    torch.cuda.empty_cache()
    U, I, K = args.users, args.items, args.k
    from recommendation_amd.dist_util import combine_item_tables, shard_range
    ub, ue = shard_range(U, rank, world)
    prec = F32 if args.precision == "f32" else F64
    esz = 4 if prec == F32 else 8
    def stage(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    stage(f"corpus ready: {U} users, {I} items, {n_clicks} interactions")
    po = pi = None
    L = args.l
    if args.model == "REBUS":
        # labels mined from the training sequences (the reference reads them from a precomputed 96-FSUB file)
        lo, li = ra.fsub_mine(I, seq_off, seq_item, L, args.min_count)
        stage(f"{len(lo) - 1} labels mined")
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val_item, lo, li)
        stage("matched paths built")
    eng = ra.Engine(args.model, I, seq_off, seq_item, test_item, val_item, K=K, L=L, lam=0.01, bias_reg=0.01,
                    relation_reg=0.01, alpha=-0.6 if args.model == "REBUS" else 0.5, precision=prec, device=local_rank,
                    user_begin=ub, user_end=ue, path_off=po, path_item=pi)
    stage("engine created")
    eng.init_params(seed=2024)
    GROUP_MODELS = ("BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "HRM_max")
    HISTORY_MODELS = ("Fossil", "FossilSimple", "REBUS")
    big = eng.steps_per_epoch >= (1 << 22)
    # the engine's defaults (rec_create), overridden by the flags
    mode = {"policy": 1 if big else 0, "owner": 0, "owner_flush": 8,
            "neg_share": 1, "pos_share": 1, "group": 0}
    if args.policy >= 0:
        eng.set_option("update_policy", args.policy); mode["policy"] = args.policy
    if args.hot_count >= 0:
        eng.set_option("hot_count", args.hot_count)
    if args.group >= 0:
        eng.set_option("group_by_item", args.group); mode["group"] = args.group
    eng.set_option("chunk", args.chunk)
    eng.set_option("overlap", args.overlap)
    if args.user_batch > 0:
        eng.set_option("user_batch", args.user_batch)
    if args.share_negative > 0:
        eng.set_option("share_negative", args.share_negative); mode["neg_share"] = args.share_negative
    if args.share_positive > 0:
        eng.set_option("share_positive", args.share_positive); mode["pos_share"] = args.share_positive
    if args.atomic_mask >= 0:
        eng.set_option("atomic", args.atomic_mask)
    if args.wide >= 0:
        eng.set_option("wide", args.wide)
    if args.owner >= 0:
        eng.set_option("owner_neg", args.owner); mode["owner"] = args.owner
    if args.owner_flush > 0:
        eng.set_option("owner_flush", args.owner_flush); mode["owner_flush"] = args.owner_flush
    if args.part_tile > 0:
        eng.set_option("part_tile", args.part_tile)
    if mode["group"] or mode["pos_share"] == 2 or args.overlap:
        mode["owner"] = 0   # the engine falls back to the plain kernel in these modes
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v)); mode[k] = int(v)
    eng.set_option("profile", 1)
    steps_per_epoch = eng.steps_per_epoch
    my_steps = steps_per_epoch // world
    if my_steps < (1 << 20) or 8 * I < (1 << 20):
        mode["owner"] = 0   # launches below "owner_min" are not partitioned

    def barrier(e=None):
        (e or eng).sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # exact mean number of distinct history rows a REBUS step touches: |S| = distinct items of the prefix before the
    # target position, averaged over the (user, position) pairs the sampler draws (user uniform, position uniform)
    mean_hist = float(seq_off[-1]) / U
    rebus_rows = None
    rebus_group_rows = {}
    if args.model == "REBUS":
        lens = np.diff(seq_off)
        elig = np.nonzero(lens >= 2)[0]
        sample_u = elig[:: max(1, len(elig) // 20000)]
        acc = 0.0
        gacc = {1: 0.0, 2: 0.0, 4: 0.0, 8: 0.0}
        for u in sample_u:
            seq = seq_item[seq_off[u]:seq_off[u + 1]]
            seen, sizes = set(), []
            for x in seq[:-1]:
                seen.add(int(x)); sizes.append(len(seen))
            acc += float(np.mean(sizes))   # target positions 1 .. n-1: prefix of 1 .. n-1 items
            # G tuples of one user walk the longest of their G prefixes once: E[|S| of the max of G uniform positions]
            n = len(sizes)
            sz = np.asarray(sizes, dtype=np.float64)
            for G in gacc:
                p_max = ((np.arange(1, n + 1) / n) ** G) - ((np.arange(0, n) / n) ** G)
                gacc[G] += float((p_max * sz).sum())
        rebus_rows = acc / len(sample_u)
        rebus_group_rows = {G: v / len(sample_u) for G, v in gacc.items()}

    def measure(e, md, prec_name, steps, warmup, trainer=None, epoch0=0):
        """steps timed SGD epochs on engine e -> the fields of one bench entry."""
        esz_ = 4 if prec_name == "f32" else 8

        def one_step(epoch):
            if trainer is not None:
                trainer.epoch(0.05, seed=99, epoch=epoch)      # rounds of local steps + item-table all-reduce (sharded.py)
                torch.cuda.synchronize()
            else:
                e.sgd_epoch(0.05, my_steps, seed=99, epoch=epoch * world + rank)
                e.sync()

        for w in range(warmup):
            one_step(epoch0 + w)
        barrier(e)
        for which in (0, 2, 3):
            e.kernel_stats(which)   # reset the HIP-event counters
        t0 = time.perf_counter()
        for s_ in range(steps):
            one_step(epoch0 + warmup + s_)
        barrier(e)
        dt = time.perf_counter() - t0
        upd_ms, upd_n = e.kernel_stats(0)
        smp_ms, smp_n = e.kernel_stats(2)
        part_ms, part_n = e.kernel_stats(3)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            all_reduce(tt, dist.ReduceOp.MAX)
            dt = float(tt.item())
        grouped = md["group"] == 2 and args.model in GROUP_MODELS
        # algorithmic bytes per triple: every distinct row / scalar a step names, read once + written once.  Steps that
        # share an item by construction (pairs drawing one negative, or one positive and one negative) name it once per pair.
        # The history models walk a user's history once for a group of G tuples of that user ("user_batch"): the group
        # names each history row once (REBUS: the rows of the longest of its G prefixes), every tuple its own two items.
        walk_g = 1
        if args.model in HISTORY_MODELS:
            walk_g = max(1, int(e.kernel_stats(6)[0]))
        rows = {"BPRMF": 3, "MC": 3, "HRM_avg": 4, "HRM_max": 4, "TransRec": 4, "TransRec_L1": 4, "FPMC": 6, "PRME": 6,
                "Fossil": mean_hist / walk_g + 2, "FossilSimple": mean_hist / walk_g + 2,
                "REBUS": rebus_group_rows.get(walk_g, rebus_rows or 0) / walk_g + 2}[args.model]
        scalars = {"BPRMF": 2, "TransRec": 2, "TransRec_L1": 2, "Fossil": 2 + 2 * L, "FossilSimple": 4, "REBUS": 2}.get(args.model, 0)
        item_rows_per_role = {"FPMC": 2, "PRME": 2}.get(args.model, 1)
        if args.model in GROUP_MODELS:
            shared = 0.0
            if md["neg_share"] > 1 or md["pos_share"] == 2:
                shared += 1.0 - 1.0 / max(md["neg_share"], 2)
            if md["pos_share"] == 2:
                shared += 0.5
            rows -= shared * item_rows_per_role
            if args.model == "BPRMF":
                scalars -= shared
        bytes_per_triple = int(round(2 * rows * K * esz_ + 2 * scalars * esz_))
        launches = max(upd_n, 1)
        tuples_per_launch = my_steps * steps / launches
        kern_ms = (upd_ms - part_ms) / launches          # the update kernel alone (owner mode: without its partition pass)
        achieved = bytes_per_triple * tuples_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        kernel = "sgd_history_batch_kernel" if args.model in HISTORY_MODELS else "sgd_runs_kernel" if grouped else \
            ("sgd_wide_kernel" if (prec_name == "f32" and K % 4 == 0 and K >= 8 and md["neg_share"] == 1 and md["pos_share"] == 1 and args.wide != 0) else "sgd_hogwild_kernel") + ("<OWNER>" if md["owner"] else "")
        roof = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_triple": bytes_per_triple,
                "tuples_per_launch": int(tuples_per_launch), "avg_launch_ms": round(kern_ms, 4), "launches": int(upd_n),
                **({"tuples_per_walk": walk_g, "bytes_per_triple_unshared": int(round(2 * ({"REBUS": rebus_rows or 0}.get(args.model, mean_hist) + 2) * K * esz_ + 2 * scalars * esz_))}
                   if args.model in HISTORY_MODELS else {}),
                "sampler_ms_per_launch": round(smp_ms / max(smp_n, 1), 4),
                "partition_ms_per_launch": round(part_ms / launches, 4)}
        # what bounds the scatter: float atomics execute at the memory side at about 1.3 TB/s chip-wide
        atomic_rows = {"BPRMF": 2, "MC": 3, "HRM_avg": 3, "HRM_max": 3, "TransRec": 3, "TransRec_L1": 3, "FPMC": 5, "PRME": 5}.get(args.model)
        if atomic_rows and md["policy"] == 1 and kern_ms > 0:
            neg_rows = item_rows_per_role
            ar = float(atomic_rows)
            if md["owner"]:
                # negatives whose item has an LDS accumulator cost one atomic row per owner_flush additions
                slices = e.owner_slices
                slice_rows = (I + slices - 1) // slices
                acc_w = K * (2 if args.model in ("FPMC", "PRME") else 1)
                hot_w = K * {"MC": 2, "FPMC": 3, "PRME": 2}.get(args.model, 1)
                lds = 160 * 1024 - 1280 - 16 * 1024 - 32 * (hot_w * esz_ + 12)   # less the staging areas and the hot items' accumulators
                res = min(slice_rows, lds // (acc_w * esz_ + 4 + (4 if args.model == "BPRMF" else 0)))
                frac_res = res / slice_rows
                roof["owner_slices"] = slices; roof["negatives_resident"] = round(frac_res, 4)
                ar -= neg_rows * frac_res * (1.0 - 1.0 / md["owner_flush"])
            elif args.model in GROUP_MODELS:
                if md["neg_share"] > 1 or md["pos_share"] == 2:
                    ar -= neg_rows * (1.0 - 1.0 / max(md["neg_share"], 2))
                if md["pos_share"] == 2:
                    ar -= neg_rows * 0.5
            ab = int(round(ar * K * esz_))
            rate = ab * tuples_per_launch / (kern_ms * 1e-3) / 1e9
            roof["atomic"] = {"bytes_per_triple": ab, "achieved": round(rate, 1), "peak": 1300.0, "unit": "GB/s",
                              "frac": round(rate / 1300.0, 4)}
        if args.model in HISTORY_MODELS and md["policy"] == 1 and kern_ms > 0:
            # one atomic row per history row and group, two per tuple for its own items (upper bound: the hottest items are plain stores)
            ar = (mean_hist if args.model != "REBUS" else rebus_group_rows.get(walk_g, rebus_rows or 0)) / walk_g + 2
            ab = int(round(ar * K * esz_))
            rate = ab * tuples_per_launch / (kern_ms * 1e-3) / 1e9
            roof["atomic"] = {"bytes_per_triple": ab, "achieved": round(rate, 1), "peak": 1300.0, "unit": "GB/s",
                              "frac": round(rate / 1300.0, 4)}
        update = ["float atomics on every row", "float atomics on item rows (plain stores for user rows, biases and items holding >= 1/1024 of the positives)", "plain stores"][md["policy"]]
        if grouped:
            update += ", chunks drawn by event and grouped by positive item, one atomic row per run"
        if md["owner"]:
            update += f"; launch partitioned by the negative's slice of the catalog, negatives' deltas summed in LDS, one atomic row per {md['owner_flush']} additions"
        if md["pos_share"] == 2 and md["policy"] != 2:
            update += "; pairs of steps share their positive and their negative, each scattered once for both"
        elif md["neg_share"] >= 2 and md["policy"] != 2:
            update += f"; groups of {md['neg_share']} steps share their negative, scattered once for all"
        else:
            update += "; every step draws its own negative (the reference's sampling)"
        total_triples = my_steps * world * steps   # expected; the event-major sampler draws Poisson(this) tuples
        return {"value": round(total_triples / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "dtype": prec_name,
                "update": update, "roofline": roof}

    # N ranks: every rank trains its own users; the replicated item tables are combined after every round
    trainer = None
    if world > 1:
        from recommendation_amd.sharded import ShardedTrainer
        trainer = ShardedTrainer(eng, rank, world, lambda t: all_reduce(t, dist.ReduceOp.SUM), dev,
                                 rounds=args.rounds if args.rounds > 0 else None)
        my_steps = trainer.steps_per_round() * trainer.rounds
        if trainer.steps_per_round() < (1 << 20):
            mode["owner"] = 0

    main_res = measure(eng, mode, args.precision, args.steps, args.warmup, trainer)
    roofline = main_res["roofline"]
    roofline["tuples_drawn_over_expected"] = round(eng.tuples_drawn / max(1, my_steps * (args.steps + args.warmup)), 6)
    # (roofline.traffic stays null: the HBM-side counters need rocprofv3 around the process; what they read for this
    # command is in profiles/, collected by scripts/collect_profiles.sh)
    # practical ceiling: a device-to-device copy of 1 GiB (read + write counted)
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev); dst = torch.empty_like(src)
    dst.copy_(src); torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(5):
        dst.copy_(src)
    torch.cuda.synchronize()
    roofline["measured_copy_gbs"] = round(5 * 2 * src.numel() * 4 / (time.perf_counter() - c0) / 1e9, 1)
    del src, dst
    value = main_res["value"]

    out = {"metric": "bpr_sgd_triples_per_sec", "value": value, "unit": "triples/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.precision,
           "data": "synthetic",
           "config": {"workload": f"{args.model} K={K} oneiteration (BPR-SGD epoch) on synthetic {U} users x {I} items x "
                                  f"{n_clicks} interactions", "triples_per_step": my_steps * world,
                      "update": main_res["update"],
                      "process_group": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else None,
                                        "device": torch.cuda.get_device_name(local_rank), "devices_visible": torch.cuda.device_count()},
                      "parallelism": f"user-sharded x{world}" + (
                          f", item tables all-reduced (sum of deltas) {trainer.rounds} times per epoch, {sum(t.numel() * t.element_size() for t in trainer.tabs)} bytes each" if trainer else "")},
           "roofline": roofline}

    # ---------------- evaluation: users/s over the full catalog
    if not args.no_eval:
        def one_eval():
            s_ = eng.eval_sums(50)
            torch.cuda.synchronize()
            if world > 1:
                ts = torch.tensor(s_, dtype=torch.float64, device=dev)
                all_reduce(ts, dist.ReduceOp.SUM)
                s_ = ts.cpu().numpy()
            return s_

        one_eval()                      # untimed: first-touch allocations
        barrier()
        eng.kernel_stats(1)
        ecalls = max(1, args.eval_calls)
        t0 = time.perf_counter()
        for _ in range(ecalls):
            sums = one_eval()
        barrier()
        et = (time.perf_counter() - t0) / ecalls
        if world > 1:
            tt = torch.tensor([et], dtype=torch.float64, device=dev)
            all_reduce(tt, dist.ReduceOp.MAX)
            et = float(tt.item())
        sc_ms, sc_n = eng.kernel_stats(1)
        sc_ms /= max(sc_n, 1)
        n_eval = float(sums[7])
        nctx = 1 if args.model in ("BPRMF", "MP") else 2
        kq = K * (2 if args.model in ("FPMC", "PRME") else 1)
        flops = 2.0 * nctx * kq * I * (n_eval / world)   # per rank
        tf = flops / (sc_ms * 1e-3) / 1e12 if sc_ms > 0 else 0.0
        open_cols, sweep_mode = eng.kernel_stats(4)
        screened = sweep_mode == 1
        # The default screen (f32 tables) runs on the 16-bit matrix pipe with every operand split into two f16 halves: three
        # MFMAs per contraction step, so it executes 3x the algorithmic flops.  `achieved` is algorithmic (2 C K' I per
        # user) against the dense f16 peak; `executed` counts the MFMAs that ran.
        peak = F16_MFMA_PEAK_TFLOPS if screened else FP64_MFMA_PEAK_TFLOPS
        l1 = args.model == "TransRec_L1"
        out["eval"] = {"users_per_sec": round(n_eval / et, 1), "seconds": round(et, 4), "calls": ecalls, "users": int(n_eval),
                       "auc_test": round(float(sums[1] / n_eval), 6), "hit50_test": round(float(sums[3] / n_eval), 6),
                       "roofline": {"bound": "mfma", "kernel": "score_screen16_kernel" if screened else "score_count_kernel",
                                    "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                                    "dtype": "f16 halves of f32 operands (three MFMAs per step), proven rounding margin, undecided columns in f64" if screened else "f64",
                                    "launch_ms": round(sc_ms, 3)}}
        if l1:
            # |.|_1 is no contraction: the sweep runs on the vector ALU, two operations (|a - b|, +) per term - one v_sad_u32 on
            # fixed-point operands when the integer screen holds, two fp64 additions otherwise - against the vector peak
            # (256 CUs x 4 SIMDs x 16 lanes x 2 operations x 2.4 GHz; the same for 32-bit integers and fp64)
            out["eval"]["roofline"].update({"bound": "valu", "kernel": "score_count_l1_sad_kernel" if screened else "score_count_l1_kernel",
                                            "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "Top/s", "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                                            "dtype": "u32 fixed point (one v_sad_u32 per term), proven quantisation margin, undecided columns in f64" if screened else "f64"})
            if screened:
                out["eval"]["roofline"]["undecided_columns_per_query"] = round(open_cols / max(1.0, n_eval * nctx), 2)
        elif screened:
            out["eval"]["roofline"]["executed"] = round(3 * tf, 2)
            out["eval"]["roofline"]["executed_frac"] = round(3 * tf / peak, 4)
            out["eval"]["roofline"]["undecided_columns_per_query"] = round(open_cols / max(1.0, n_eval * nctx), 2)

    # ---------------- the ranking behind HIT@50 (model.cpp:32-58): top-50 lists of every user of this rank, both contexts
    if not args.no_eval and not args.no_top50 and args.model not in ("MP",):
        n_mine = int(round(n_eval / world)) if world > 1 else int(n_eval)
        n_mine = min(n_mine, eng.n_eval_users)
        eng.eval_topk(min(n_mine, 4096), 50, want_scores=False)      # untimed
        barrier()
        t0 = time.perf_counter()
        items50, _ = eng.eval_topk(n_mine, 50, want_scores=False)
        barrier()
        tt_ = time.perf_counter() - t0
        dev_ms, lists = eng.kernel_stats(5)
        if world > 1:
            t3 = torch.tensor([tt_, dev_ms], dtype=torch.float64, device=dev)
            all_reduce(t3, dist.ReduceOp.MAX)
            tt_, dev_ms = float(t3[0]), float(t3[1])
        out["eval"]["top50"] = {"users_per_sec_device": round(n_eval / (dev_ms * 1e-3), 1) if dev_ms > 0 else None,
                                "users_per_sec_with_copy_to_host": round(n_eval / tt_, 1), "device_ms": round(dev_ms, 2),
                                "seconds_with_copy_to_host": round(tt_, 4), "lists_from_the_fused_sweep": int(lists),
                                "lists": 2 * n_mine, "k": 50,
                                "what": ("queries, |.|_1 sweep on fixed-point operands (one v_sad_u32 per term) with a running 50th-best bar per (user, context), fp64 rescoring and exact ordering of the listed; the host copy is 400 bytes of item ids per user"
                                         if args.model == "TransRec_L1" else
                                         "queries, f16 sweep with a running 50th-best bar per (user, context), fp64 rescoring and exact ordering of the survivors; the host copy is 400 bytes of item ids per user")}
        del items50

    # ---------------- the other scatter modes and f64 storage, same corpus, same measurement (N=1 only)
    if world == 1 and not args.no_variants and args.model in GROUP_MODELS and args.policy < 0 and args.owner < 0 and \
            args.share_negative == 0 and args.share_positive == 0 and args.group < 0:
        variants = {}
        vsteps = max(2, min(args.steps, 3))

        def variant(name, opts, md):
            for k, v in opts.items():
                eng.set_option(k, v)
            r = measure(eng, md, args.precision, vsteps, 1, epoch0=1000 * (len(variants) + 1))
            variants[name] = {"triples_per_sec": r["value"], "ms_per_step": r["ms_per_step"], "update": r["update"], "roofline": r["roofline"]}

        if mode["policy"] == 1 and not mode["owner"]:
            # options that trade fidelity for fewer float atomics (what each costs in accuracy: tests/test_gpu_mid.py)
            variant("pairs_share_negative", {"share_negative": 2}, dict(mode, neg_share=2))
            variant("plain_stores", {"share_negative": 1, "update_policy": 2}, dict(mode, policy=2))
            eng.set_option("update_policy", 1)
            if eng.owner_slices > 0 and 8 * I >= (1 << 20):
                variant("owner_neg_8_per_item", {"owner_neg": 1}, dict(mode, owner=1))
                variant("owner_neg_one_launch_per_epoch", {"owner_per_item": 1 << 20}, dict(mode, owner=1))
                eng.set_option("owner_neg", 0); eng.set_option("owner_per_item", 8)
        if args.precision == "f32":
            e64 = ra.Engine(args.model, I, seq_off, seq_item, test_item, val_item, K=K, L=L, lam=0.01, bias_reg=0.01,
                            relation_reg=0.01, alpha=0.5, precision=F64, device=local_rank, user_begin=ub, user_end=ue)
            e64.init_params(seed=2024)
            e64.set_option("chunk", args.chunk); e64.set_option("profile", 1)
            r = measure(e64, mode, "f64", vsteps, 1)
            variants["f64_storage"] = {"triples_per_sec": r["value"], "ms_per_step": r["ms_per_step"], "update": r["update"], "roofline": r["roofline"]}
            e64.close()
        out["variants"] = variants
        if "f64_storage" in variants:
            v = variants["f64_storage"]
            out["f64_storage"] = {"value": v["triples_per_sec"], "unit": "triples/s", "ms_per_step": v["ms_per_step"],
                                  "roofline_frac": v["roofline"]["frac"], "bytes_per_triple": v["roofline"]["bytes_per_triple"],
                                  "atomic_rate_frac": v["roofline"].get("atomic", {}).get("frac")}
    out["f32_storage"] = {"value": value, "unit": "triples/s", "ms_per_step": main_res["ms_per_step"], "roofline_frac": roofline["frac"],
                          "bytes_per_triple": roofline["bytes_per_triple"], "atomic_rate_frac": roofline.get("atomic", {}).get("frac")}

    # ---------------- CPU baseline: the single-thread restatement of BPRMF::oneiteration, rank 0, N=1 only
    if rank == 0 and world == 1 and args.cpu_steps > 0 and args.model == "BPRMF":
        from oracle import oracle_py as orc
        o = orc.Oracle("BPRMF", U, I, K, seq_off, seq_item, test_item, val_item, lam=0.01, bias_reg=0.01)
        W = np.random.default_rng(0).random(o.nw)
        W[:I] = 0
        o.bprmf_epoch(W, 0.05, 200_000, 1)  # touch the pages
        t0 = time.perf_counter()
        done = o.bprmf_epoch(W, 0.05, args.cpu_steps, 2)
        ct = time.perf_counter() - t0
        port = {"value": round(done / ct, 1), "unit": "triples/s", "cores": 1, "kind": "port",
                "sample": f"{done} consecutive updateFactors steps of one epoch on the same corpus, fp64, {ct:.1f} s"}
        out["cpu_baseline"] = port
        # the real reference (compiled from its own sources into oracle/_ref) on the first users of the same corpus
        ref = _time_reference(args, seq_off, seq_item, test_item, val_item, K)
        if ref is not None:
            out["cpu_baseline"] = ref
            out["cpu_port"] = port
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)

    # CPU baseline of the evaluation: the restatement of model::AUC on a user sample, all host cores (OpenMP over users)
    if rank == 0 and world == 1 and args.cpu_eval_users > 0 and "eval" in out and args.model == "BPRMF":
        from oracle import oracle_py as orc
        sample = np.linspace(0, U - 1, args.cpu_eval_users).astype(np.int64)
        t_s = np.full(U, -1, np.int32); t_s[sample] = test_item[sample]
        o = orc.Oracle("BPRMF", U, I, K, seq_off, seq_item, t_s, val_item, lam=0.01, bias_reg=0.01)
        W = eng.download()
        cores = min(16, os.cpu_count() or 1)
        t0 = time.perf_counter()
        o.eval(W, nthreads=cores)
        ct = time.perf_counter() - t0
        out["eval"]["cpu_baseline"] = {"value": round(len(sample) / ct, 2), "unit": "users/s", "cores": cores, "kind": "port",
                                       "sample": f"{len(sample)} evenly spaced users, full catalog, fp64, {ct:.1f} s"}
        out["eval"]["cpu_port"] = out["eval"]["cpu_baseline"]
        ref = _time_reference_eval(args, seq_off, seq_item, test_item, val_item, K, I)
        if ref is not None:
            out["eval"]["cpu_baseline"] = ref
        out["eval"]["gpu_over_cpu"] = round(out["eval"]["users_per_sec"] / out["eval"]["cpu_baseline"]["value"], 1)

    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
