#!/usr/bin/env python3
"""Converged accuracy of the REAL reference CLI on the bundled ML-5 data (criterion: the engine's
converged AUC must be within 0.005).  Runs oracle/_ref/train once per model with early stopping as
shipped and records the final lines it prints.  Test infrastructure; needs /root/reference."""
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference"
WORK = "/tmp/refwork_conv"
TRAIN = os.path.join(HERE, "_ref", "train")
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
# model, K, L, minCount, lambda, bias_reg, relation_reg, alpha
RUNS = [
    ("BPRMF", 10, 1, 1, 0.01, 0.01, 0.0, 0.0),
    ("MC", 10, 1, 1, 0.01, 0.0, 0.0, 0.0),
    ("FPMC", 10, 1, 1, 0.01, 0.0, 0.0, 0.0),
    ("HRM_avg", 10, 1, 1, 0.01, 0.0, 0.0, 0.0),
    ("HRM_max", 10, 1, 1, 0.01, 0.0, 0.0, 0.0),
    ("PRME", 10, 1, 1, 0.01, 0.0, 0.0, 0.4),
    ("TransRec", 10, 1, 1, 0.05, 0.05, 0.05, 0.0),
    ("TransRec_L1", 10, 1, 1, 0.05, 0.05, 0.05, 0.0),
    ("FossilSimple", 10, 1, 1, 0.01, 0.01, 0.0, 0.0),
    ("Fossil", 10, 3, 1, 0.01, 0.01, 0.0, 0.0),
    ("REBUS", 10, 1, 1, 0.01, 0.0, 0.0, -0.6),
    ("REBUS", 10, 3, 2, 0.01, 0.0, 0.0, -0.6),
]


def run(cfg):
    name, K, L, mc, lam, breg, rreg, alpha = cfg
    tag = f"{name}_mc{mc}_L{L}"
    cmd = [TRAIN, "01-Data/ML1M-atmost-5.txt", "0", "0", str(mc), str(L), str(K), repr(lam), repr(breg), repr(rreg),
           repr(alpha), str(ITERS), "03-Models/" + tag, name]
    r = subprocess.run(cmd, cwd=WORK, capture_output=True, text=True)
    out = r.stdout
    m = re.search(r"Test AUC = ([0-9.]+), Val AUC = ([0-9.]+)", out)
    h = re.search(r"Test HIT50 = ([0-9.]+), Val HIT50 = ([0-9.]+)", out)
    iters = len(re.findall(r"^Iter: ", out, flags=re.M))
    res = dict(model=name, K=K, L=L, minCount=mc, lam=lam, bias_reg=breg, relation_reg=rreg, alpha=alpha,
               max_iterations=ITERS, iterations_run=iters, auc_test=float(m.group(1)), auc_val=float(m.group(2)),
               hit50_test=float(h.group(1)), hit50_val=float(h.group(2)))
    print(res, flush=True)
    return res


if __name__ == "__main__":
    shutil.rmtree(WORK, ignore_errors=True)
    for d in ("01-Data", "96-FSUB", "03-Models"):
        os.makedirs(os.path.join(WORK, d))
    shutil.copy(os.path.join(REF, "01-Data", "ML1M-atmost-5.txt"), os.path.join(WORK, "01-Data"))
    for f in os.listdir(os.path.join(REF, "96-FSUB")):
        if f.startswith("ML1M-atmost-5_"):
            shutil.copy(os.path.join(REF, "96-FSUB", f), os.path.join(WORK, "96-FSUB"))
    with ThreadPoolExecutor(6) as ex:
        results = list(ex.map(run, RUNS))
    with open(os.path.join(GOLD, "ml5_converged.json"), "w") as f:
        json.dump(results, f, indent=1)
