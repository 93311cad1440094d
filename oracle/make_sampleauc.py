#!/usr/bin/env python3
"""The spread of the reference's own model::sampleAUC (2000 rand() draws per user, model.cpp:100-167; Fossil.cpp:480-564,
REBUS.cpp:625-730) on the ML-5 models the reference saved (tests/golden/ml5_<tag>.npz): N consecutive calls on fixed
parameters, the global rand() stream running on -> tests/golden/ml5_sampleauc.json.

Test infrastructure only; needs /root/reference and `make -C oracle ref`."""
import json
import os
import re
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
WORK = "/tmp/refwork_sauc"
DRIVER = os.path.join(HERE, "_ref", "ref_driver")
ML5 = "ML1M-atmost-5"
TAGS = {"BPRMF": "BPRMF", "TransRec": "TransRec", "Fossil": "Fossil", "REBUS_mc2_L3": "REBUS"}
CALLS = 24

if __name__ == "__main__":
    if not os.path.exists(DRIVER):
        sys.exit("run `make -C oracle ref` first")
    shutil.rmtree(WORK, ignore_errors=True)
    for d in ("01-Data", "96-FSUB", "out"):
        os.makedirs(os.path.join(WORK, d))
    shutil.copy(os.path.join(REF, "01-Data", ML5 + ".txt"), os.path.join(WORK, "01-Data"))
    for f in os.listdir(os.path.join(REF, "96-FSUB")):
        if f.startswith(ML5 + "_"):
            shutil.copy(os.path.join(REF, "96-FSUB", f), os.path.join(WORK, "96-FSUB"))
    res = {}
    for tag, name in TAGS.items():
        g = np.load(os.path.join(GOLD, f"ml5_{tag}.npz"))
        w0p = os.path.join(WORK, "out", tag + ".W0.f64")
        (g["W_micro"].astype(np.float64) / 1e6).astype("<f8").tofile(w0p)
        cmd = [DRIVER, name, f"01-Data/{ML5}.txt", str(int(g["K"])), str(int(g["L"])), str(int(g["minCount"])),
               repr(float(g["lam"])), repr(float(g["bias_reg"])), repr(float(g["relation_reg"])), repr(float(g["alpha"])),
               w0p, f"sauc:{CALLS}", "0.05", os.path.join(WORK, "out", tag)]
        r = subprocess.run(cmd, cwd=WORK, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        rows = [[float(x) for x in m] for m in re.findall(r"SAUC \d+ (\S+) (\S+) (\S+)", r.stderr)]
        assert len(rows) == CALLS, r.stderr[-400:]
        a = np.array(rows)
        res[tag] = {"calls": CALLS, "draws_per_user": 2000, "auc_val": a[:, 0].tolist(), "auc_test": a[:, 1].tolist(),
                    "var_test": a[:, 2].tolist(), "exact_auc_val": float(g["metrics"][0]), "exact_auc_test": float(g["metrics"][1])}
        print(tag, "val mean %.6f std %.2e (exact %.6f)  test mean %.6f std %.2e (exact %.6f)" % (
            a[:, 0].mean(), a[:, 0].std(ddof=1), g["metrics"][0], a[:, 1].mean(), a[:, 1].std(ddof=1), g["metrics"][1]))
    with open(os.path.join(GOLD, "ml5_sampleauc.json"), "w") as f:
        json.dump(res, f, indent=1)
