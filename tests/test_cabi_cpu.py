"""CPU-side checks of the C-ABI library: it loads, exports what include/recsys_engine.h declares,
and its host-only REBUS path builder agrees with the reference's findPathStars output."""
import os
import re

import numpy as np

import recommendation_amd as ra
from oracle import oracle_py as orc
from tests.golden_util import fixture_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "recsys_engine.h")).read()
    declared = set(re.findall(r"\b(rec_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"rec_engine", "rec_corpus", "rec_model_desc", "rec_metrics"}
    assert len(declared) >= 20
    lib = ra.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_rebus_path_builder_matches_reference():
    for tag, fam in (("REBUS", "toy"), ("REBUS_mc1_L1", "ml5"), ("REBUS_mc2_L3", "ml5")):
        g, model, co, ci, labels = fixture_case(tag, fam)
        seq_off, seq_item, test, val = orc.split(co, ci)
        I = int(g["n_items"])
        lo, li = orc.parse_labels(labels, I)
        po, pi = ra.build_rebus_paths(I, seq_off, seq_item, val, lo, li)
        U = len(seq_off) - 1
        flat = []
        for u in range(U):
            n = int(seq_off[u + 1] - seq_off[u])
            for c in range(1, n):
                ctx = int(seq_off[u]) + 2 * u + c - 1
                path = pi[po[ctx]:po[ctx + 1]]
                flat.append(len(path)); flat.extend(path.tolist())
        for u in range(U):
            if test[u] < 0:
                continue
            n = int(seq_off[u + 1] - seq_off[u])
            for c in (n, n + 1):
                ctx = int(seq_off[u]) + 2 * u + c - 1
                path = pi[po[ctx]:po[ctx + 1]]
                flat.append(len(path)); flat.extend(path.tolist())
        assert np.array_equal(np.array(flat, np.int32), g["paths_flat"]), tag
