"""Loading of tests/golden fixtures (outputs of the real reference, see oracle/make_golden.py)."""
import gzip
import os

import numpy as np

from oracle import oracle_py as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOY_MODELS = ["BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "HRM_max", "TransRec", "TransRec_L1",
              "FossilSimple", "Fossil", "REBUS"]
ML5_TAGS = ["BPRMF", "FPMC", "TransRec", "Fossil", "REBUS_mc1_L1", "REBUS_mc2_L3"]


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))


def read_labels(fname):
    path = os.path.join(GOLD, fname)
    if path.endswith(".gz"):
        with gzip.open(path, "rt") as f:
            return f.read().split("\n")
    with open(path) as f:
        return f.read().split("\n")


def ml5_corpus():
    d = load("ml5_MP")
    return d["click_off"], d["click_item"]


def fixture_case(name, family):
    """Returns (golden dict, model name, click_off, click_item, label lines or None)."""
    g = load(f"{family}_{name}")
    model = name.split("_mc")[0]
    if family == "toy":
        co, ci = g["click_off"], g["click_item"]
        labels = read_labels("toy_root_fsub_minCount_2_L_3.txt") if model == "REBUS" else None
    else:
        co, ci = ml5_corpus()
        labels = None
        if model == "REBUS":
            labels = read_labels(f"ML1M-atmost-5_root_fsub_minCount_{int(g['minCount'])}_L_{int(g['L'])}.txt.gz")
    return g, model, co, ci, labels


def make_oracle(g, model, co, ci, labels):
    seq_off, seq_item, test, val = orc.split(co, ci)
    U, I = len(co) - 1, int(g["n_items"])
    po = pi = None
    if model == "REBUS":
        lo, li = orc.parse_labels(labels, I)
        po, pi = orc.rebus_paths(I, seq_off, seq_item, val, lo, li)
    o = orc.Oracle(model, U, I, int(g["K"]), seq_off, seq_item, test, val, L=int(g["L"]),
                   lam=float(g["lam"]), bias_reg=float(g["bias_reg"]), relation_reg=float(g["relation_reg"]),
                   alpha=float(g["alpha"]), path_off=po, path_item=pi)
    return o
