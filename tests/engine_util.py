"""Builds a recommendation_amd.Engine for a golden fixture (same inputs as tests/golden_util.make_oracle)."""
import numpy as np

import recommendation_amd as ra
from oracle import oracle_py as orc


def make_engine(g, model, co, ci, labels, precision, **kw):
    seq_off, seq_item, test, val = orc.split(co, ci)  # input preparation only (model.hpp:22-45)
    I = int(g["n_items"])
    po = pi = None
    if model == "REBUS":
        lo, li = orc.parse_labels(labels, I)
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
    return ra.Engine(model, I, seq_off, seq_item, test, val, K=int(g["K"]), L=int(g["L"]), lam=float(g["lam"]),
                     bias_reg=float(g["bias_reg"]), relation_reg=float(g["relation_reg"]), alpha=float(g["alpha"]),
                     precision=precision, path_off=po, path_item=pi, **kw)


def rel_err(a, b, floor=1e-3):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
