#!/usr/bin/env python3
"""Validation/test AUC trajectories of the REAL reference on ML-5 (its own oneiteration + AUC, driven
through oracle/_ref/ref_driver every C epochs).  Gives the best-validation checkpoint at a finer grain
than the CLI's 50-epoch cadence.  Test infrastructure; needs /root/reference."""
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference"
WORK = "/tmp/refwork_traj"
DRIVER = os.path.join(HERE, "_ref", "ref_driver")
# model, K, L, minCount, lambda, bias_reg, relation_reg, alpha, epochs, every
RUNS = [
    ("BPRMF", 10, 1, 1, 0.01, 0.01, 0.0, 0.0, 300, 10),
    ("MC", 10, 1, 1, 0.01, 0.0, 0.0, 0.0, 300, 10),
    ("FPMC", 10, 1, 1, 0.01, 0.0, 0.0, 0.0, 300, 10),
    ("HRM_avg", 10, 1, 1, 0.01, 0.0, 0.0, 0.0, 300, 10),
    ("HRM_max", 10, 1, 1, 0.01, 0.0, 0.0, 0.0, 300, 10),
    ("PRME", 10, 1, 1, 0.01, 0.0, 0.0, 0.4, 800, 25),
    ("TransRec", 10, 1, 1, 0.05, 0.05, 0.05, 0.0, 2000, 50),
    ("TransRec_L1", 10, 1, 1, 0.05, 0.05, 0.05, 0.0, 2000, 50),
    ("FossilSimple", 10, 1, 1, 0.01, 0.01, 0.0, 0.0, 300, 10),
    ("Fossil", 10, 3, 1, 0.01, 0.01, 0.0, 0.0, 300, 10),
    ("REBUS", 10, 1, 1, 0.01, 0.0, 0.0, -0.6, 1500, 50),
    ("REBUS", 10, 3, 2, 0.01, 0.0, 0.0, -0.6, 1500, 50),
]


SEEDS = (0, 1, 2, 3, 4, 5)   # srand() seed of the run; the CLI uses 0 (main.cpp:144); glibc gives seeds 0 and 1 the same stream


def run(job):
    cfg, seed = job
    name, K, L, mc, lam, breg, rreg, alpha, n, c = cfg
    out = os.path.join(WORK, "out", f"{name}_{L}_{mc}_{seed}")
    cmd = [DRIVER, name, "01-Data/ML1M-atmost-5.txt", str(K), str(L), str(mc), repr(lam), repr(breg), repr(rreg), repr(alpha),
           "-", f"traj:{n}:{c}", "0.05", out]
    r = subprocess.run(cmd, cwd=WORK, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, REF_SEED=str(seed)))
    traj = [[float(x) for x in ln.split()[1:]] for ln in r.stderr.split("\n") if ln.startswith("TRAJ")]
    best = max(traj, key=lambda t: t[1])
    res = dict(seed=seed, model=name, K=K, L=L, minCount=mc, lam=lam, bias_reg=breg, relation_reg=rreg, alpha=alpha, epochs=n, every=c,
               best_iter=int(best[0]), best_auc_val=best[1], auc_test_at_best=best[2], hit50_test_at_best=best[4])
    print(name, L, mc, "seed", seed, "best", best, flush=True)
    return res


if __name__ == "__main__":
    shutil.rmtree(WORK, ignore_errors=True)
    for d in ("01-Data", "96-FSUB", "out"):
        os.makedirs(os.path.join(WORK, d))
    shutil.copy(os.path.join(REF, "01-Data", "ML1M-atmost-5.txt"), os.path.join(WORK, "01-Data"))
    for f in os.listdir(os.path.join(REF, "96-FSUB")):
        if f.startswith("ML1M-atmost-5_"):
            shutil.copy(os.path.join(REF, "96-FSUB", f), os.path.join(WORK, "96-FSUB"))
    with ThreadPoolExecutor(7) as ex:
        results = list(ex.map(run, [(cfg, seed) for cfg in RUNS for seed in SEEDS]))
    with open(os.path.join(GOLD, "ml5_trajectories.json"), "w") as f:
        json.dump(results, f, indent=0)
