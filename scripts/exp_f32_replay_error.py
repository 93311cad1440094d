"""f32 replay error of the deterministic mode against the fp64 restatement, by number of replayed tuples (GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from recommendation_amd.capi import F32, SEQUENTIAL
from tests.engine_util import make_engine
from tests.golden_util import TOY_MODELS, fixture_case, make_oracle
for model in TOY_MODELS:
    g, model, co, ci, labels = fixture_case(model, "toy")
    o = make_oracle(g, model, co, ci, labels)
    out = []
    for n in (200, 500, 1000, 2000):
        e = make_engine(g, model, co, ci, labels, F32)
        e.upload(g["W0"])
        e.sgd_replay(float(g["lr"]), g["tuples"][:n], SEQUENTIAL)
        W1 = e.download()
        ref = o.replay(g["W0"].astype(np.float32).astype(np.float64), g["tuples"][:n], float(g["lr"]))
        big = np.abs(ref) >= 1e-3
        rel = float(np.max(np.abs(W1[big] - ref[big]) / np.abs(ref[big]))) if big.any() else 0.0
        small = float(np.max(np.abs(W1[~big] - ref[~big]))) if (~big).any() else 0.0
        floor1 = float(np.max(np.abs(W1 - ref) / np.maximum(np.abs(ref), 1e-1)))
        out.append((n, "%.2e" % rel, "%.2e" % small, "%.2e" % floor1))
        e.close()
    print(model, out, flush=True)
