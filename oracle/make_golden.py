#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Test infrastructure only.  Needs /root/reference and `make -C oracle ref` (which compiles the
reference sources in place into oracle/_ref/); the fixtures it writes are plain data (inputs and the
reference's outputs) so that the tests can run where the reference is absent.

Two families of fixtures:
  toy_<MODEL>.npz   a 64-user x 40-item click file with repeats, short users and shuffled lines;
                    W0 -> 2000 captured tuples through <MODEL>::updateFactors -> W1; positives'
                    scores, per-user (count_val, count_test, max) and the 7 outputs of AUC() at W1.
  ml5_<MODEL>.npz   the bundled 01-Data/ML1M-atmost-5.txt trained by the reference CLI for 100
                    iterations, its SAVED model file (6-decimal text, model.cpp:176-192) re-read,
                    then counts / metrics of that saved model.
"""
import gzip
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
WORK = "/tmp/refwork"
DRIVER = os.path.join(HERE, "_ref", "ref_driver")
TRAIN = os.path.join(HERE, "_ref", "train")

LR = 0.05
HYPER = dict(K=6, L=3, minCount=2, lam=0.05, bias_reg=0.02, relation_reg=0.03)


# ----------------------------------------------------------------------------- toy click file
def make_toy_clicks(rng, n_users=64, n_items=40):
    """Lines 'user item rating time' (corpus.cpp:91).  Users/items get string names that do not
    sort like their ids; per-user timestamps strictly increase (no unstable-sort ties,
    corpus.cpp:219); items repeat inside a user; some users have 1 or 2 clicks (no val/test)."""
    lines = []
    popular = rng.permutation(n_items)
    for u in range(n_users):
        if u % 13 == 0:
            n = 1
        elif u % 13 == 1:
            n = 2
        else:
            n = int(rng.integers(3, 14))
        # zipf-ish item choice with a per-user favourite so that repeats are common
        fav = popular[int(rng.integers(0, 8))]
        t = int(rng.integers(1000, 2000))
        for _ in range(n):
            r = rng.random()
            if r < 0.25:
                it = fav
            else:
                it = popular[min(n_items - 1, int(rng.zipf(1.3)) - 1)] if r < 0.7 else int(rng.integers(0, n_items))
            t += int(rng.integers(1, 50))
            lines.append((f"u{(u * 37) % 101:03d}", f"i{(int(it) * 29) % 97:02d}", int(rng.integers(1, 6)), t))
    order = rng.permutation(len(lines))
    return [lines[i] for i in order]


def load_clicks(lines):
    """First-appearance id assignment + per-user time sort, as corpus.cpp:176-224 with
    userMin = itemMin = 0."""
    uid, iid, seqs = {}, {}, []
    for un, inn, _, t in lines:
        if inn not in iid:
            iid[inn] = len(iid)
        if un not in uid:
            uid[un] = len(uid)
            seqs.append([])
        seqs[uid[un]].append((iid[inn], t))
    for s in seqs:
        s.sort(key=lambda p: p[1])
    return seqs, len(iid)


def train_seqs(seqs):
    """Leave-last-two split, model.hpp:22-45."""
    return [[i for i, _ in (s[:-2] if len(s) >= 3 else s)] for s in seqs]


def mine_fsub(tr, min_count, L):
    """Label set of a 96-FSUB file: 'Root' plus every contiguous substring (length 1..L) of the
    training sequences that occurs at least min_count times, items joined by '-' oldest first."""
    cnt = {}
    for s in tr:
        for a in range(len(s)):
            for ln in range(1, L + 1):
                if a + ln > len(s):
                    break
                key = tuple(s[a:a + ln])
                cnt[key] = cnt.get(key, 0) + 1
    labels = ["Root"] + ["-".join(map(str, k)) for k, c in cnt.items() if c >= min_count]
    return labels


# ----------------------------------------------------------------------------- model sizes
def model_nw(name, U, I, K, L):
    return {
        "BPRMF": K * (U + I) + I,
        "MC": 2 * K * I,
        "FPMC": K * U + 3 * K * I,
        "PRME": K * (U + I) + K * I,
        "HRM_avg": (U + I) * K,
        "HRM_max": (U + I) * K,
        "TransRec": K + I * K + K * U + I,
        "TransRec_L1": K + I * K + K * U + I,
        "FossilSimple": 1 + U + I + 2 * I * K,
        "Fossil": 2 * I * K + L + I + L * U,
        "REBUS": (K + 1) * (I + 1),
    }[name]


def make_w0(name, nw, rng):
    if name in ("BPRMF", "MC", "FPMC", "PRME", "REBUS"):
        return rng.random(nw)                      # rand()/RAND_MAX, e.g. BPRMF.cpp:13
    if name.startswith("HRM"):
        return (2.0 * rng.random(nw) - 1.0) / HYPER["K"]   # HRM_avg.cpp:16
    return rng.random(nw) - 0.5                    # generic, exercises the unit-ball projection


def make_tuples(name, tr, I, L, labels, n, rng):
    """Tuples drawn from the support of each model's sampler (see the oneiteration functions),
    4 ints per tuple; meaning per model documented in ref_driver.cpp."""
    out = []
    elig2 = [u for u, s in enumerate(tr) if len(s) >= 2]
    eligd = [u for u, s in enumerate(tr) if len(set(s)) >= 2]
    elig1 = [u for u, s in enumerate(tr) if 0 < len(set(s)) < I]
    succ = {}
    for s in tr:
        for a, b in zip(s[:-1], s[1:]):
            succ.setdefault(a, []).append(b)
    src_items = sorted(succ)
    lab = set(labels)
    while len(out) < n:
        if name == "BPRMF":
            u = elig1[rng.integers(len(elig1))]
            pos = tr[u][rng.integers(len(tr[u]))]
            neg = int(rng.integers(I))
            if neg in set(tr[u]):
                continue
            out.append((u, pos, neg, 0))
        elif name == "MC":
            i = src_items[rng.integers(len(src_items))]
            pos = succ[i][rng.integers(len(succ[i]))]
            neg = int(rng.integers(I))
            if neg in set(succ[i]):
                continue
            out.append((i, pos, neg, 0))
        elif name in ("FPMC", "PRME", "FossilSimple"):
            u = elig2[rng.integers(len(elig2))]
            t = int(rng.integers(len(tr[u]) - 1))
            prev, pos = tr[u][t], tr[u][t + 1]
            neg = int(rng.integers(I))
            if neg == pos or (name == "FossilSimple" and neg == prev):
                continue
            out.append((u, prev, pos, neg))
        elif name.startswith("HRM"):
            u = eligd[rng.integers(len(eligd))]
            t = int(rng.integers(len(tr[u]) - 1))
            prev, pos = tr[u][t], tr[u][t + 1]
            neg = int(rng.integers(I))
            if neg == pos:
                continue
            out.append((u, prev, pos, neg))
        elif name.startswith("TransRec"):
            u = eligd[rng.integers(len(eligd))]
            t = int(rng.integers(len(tr[u]) - 1))
            x, y = tr[u][t], tr[u][t + 1]
            yn = int(rng.integers(I))
            if yn in set(tr[u]):
                continue
            out.append((u, x, y, yn))
        elif name == "Fossil":
            u = elig2[rng.integers(len(elig2))]
            t = int(rng.integers(1, len(tr[u])))
            prev = tr[u][max(0, t - L):t]
            neg = int(rng.integers(I))
            if neg == tr[u][t] or neg in prev:
                continue
            out.append((u, t, neg, 0))
        elif name == "REBUS":
            u = elig2[rng.integers(len(elig2))]
            t = int(rng.integers(1, len(tr[u])))
            path = find_path_stars(tr[u][:t], lab, I)
            neg = int(rng.integers(I))
            if neg == tr[u][t] or neg in path:
                continue
            out.append((u, t - 1, neg, 0))
        else:
            raise ValueError(name)
    return np.array(out, dtype=np.int32)


def find_path_stars(prefix, lab, I):
    """REBUS::findPathStars (REBUS.cpp:484-524) on string labels, for tuple generation only."""
    path, seq = [], ""
    for it in reversed(prefix):
        cand = str(it) if not seq else f"{it}-{seq}"
        if cand in lab:
            seq = cand
            path.append(it)
    return path or [I]


# ----------------------------------------------------------------------------- driver plumbing
def run_driver(name, data_rel, K, L, mc, lam, breg, rreg, alpha, w0, tuples, out):
    w0p = "-"
    if w0 is not None:
        w0p = out + ".W0.f64"
        np.asarray(w0, dtype="<f8").tofile(w0p)
    tp = "-"
    if tuples is not None and len(tuples):
        tp = out + ".tuples.txt"
        np.savetxt(tp, tuples, fmt="%d")
    cmd = [DRIVER, name, data_rel, str(K), str(L), str(mc), repr(lam), repr(breg), repr(rreg), repr(alpha),
           w0p, tp, repr(LR), out]
    r = subprocess.run(cmd, cwd=WORK, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed: {r.stderr}")


def read_outputs(out, name):
    d = {}
    c = np.fromfile(out + ".corpus.i32", dtype="<i4")
    U, I, NC = int(c[0]), int(c[1]), int(c[2])
    p, off, items, times = 3, [0], [], []
    for _ in range(U):
        n = int(c[p]); p += 1
        pairs = c[p:p + 2 * n].reshape(n, 2); p += 2 * n
        items.append(pairs[:, 0]); times.append(pairs[:, 1]); off.append(off[-1] + n)
    d["n_users"], d["n_items"], d["n_clicks"] = np.int64(U), np.int64(I), np.int64(NC)
    d["click_off"] = np.array(off, dtype=np.int64)
    d["click_item"] = np.concatenate(items).astype(np.int32) if items else np.zeros(0, np.int32)
    d["click_time"] = np.concatenate(times).astype(np.int32) if times else np.zeros(0, np.int32)
    d["split"] = np.fromfile(out + ".split.i32", dtype="<i4").reshape(U, 4)
    meta = open(out + ".meta.txt").read().split("\n")
    d["NW"], d["num_pos_events"], d["to_string"] = np.int64(meta[0]), np.int64(meta[1]), np.array(meta[2])
    d["xpos"] = np.fromfile(out + ".xpos.f64", dtype="<f8").reshape(U, 2)
    d["counts"] = np.fromfile(out + ".counts.i32", dtype="<i4").reshape(U, 3)
    d["metrics"] = np.fromfile(out + ".metrics.f64", dtype="<f8")
    if name != "MP":
        d["W1"] = np.fromfile(out + ".W1.f64", dtype="<f8")
        wi = np.fromfile(out + ".Winit.f64", dtype="<f8")
        d["Winit_minmax"] = np.array([wi.min(), wi.max()])
    if name == "REBUS":
        d["paths_flat"] = np.fromfile(out + ".paths.i32", dtype="<i4")
        d["eta_flat"] = np.fromfile(out + ".eta.f64", dtype="<f8")
    return d


def setup_work():
    shutil.rmtree(WORK, ignore_errors=True)
    os.makedirs(os.path.join(WORK, "01-Data"))
    os.makedirs(os.path.join(WORK, "96-FSUB"))
    os.makedirs(os.path.join(WORK, "03-Models"))
    os.makedirs(os.path.join(WORK, "out"))


# ----------------------------------------------------------------------------- fixtures
TOY_MODELS = ["MP", "BPRMF", "MC", "FPMC", "PRME", "HRM_avg", "HRM_max", "TransRec", "TransRec_L1",
              "FossilSimple", "Fossil", "REBUS"]


def make_toy():
    rng = np.random.default_rng(20240607)
    lines = make_toy_clicks(rng)
    with open(os.path.join(WORK, "01-Data", "toy.txt"), "w") as f:
        for un, inn, r, t in lines:
            f.write(f"{un} {inn} {r} {t}\n")
    shutil.copy(os.path.join(WORK, "01-Data", "toy.txt"), os.path.join(GOLD, "toy_clicks.txt"))
    seqs, I = load_clicks(lines)
    tr = train_seqs(seqs)
    U = len(seqs)
    K, L, mc = HYPER["K"], HYPER["L"], HYPER["minCount"]
    labels = mine_fsub(tr, mc, L)
    fsub = os.path.join(WORK, "96-FSUB", f"toy_root_fsub_minCount_{mc}_L_{L}.txt")
    with open(fsub, "w") as f:
        f.write("\n".join(labels) + "\n")
    shutil.copy(fsub, os.path.join(GOLD, os.path.basename(fsub)))
    for name in TOY_MODELS:
        alpha = -0.6 if name == "REBUS" else 0.3
        out = os.path.join(WORK, "out", "toy_" + name)
        if name == "MP":
            run_driver(name, "01-Data/toy.txt", K, L, mc, HYPER["lam"], HYPER["bias_reg"],
                       HYPER["relation_reg"], alpha, None, None, out)
            d = read_outputs(out, name)
        else:
            nw = model_nw(name, U, I, K, L)
            w0 = make_w0(name, nw, rng)
            tup = make_tuples(name, tr, I, L, labels, 2000, rng)
            run_driver(name, "01-Data/toy.txt", K, L, mc, HYPER["lam"], HYPER["bias_reg"],
                       HYPER["relation_reg"], alpha, w0, tup, out)
            d = read_outputs(out, name)
            assert int(d["NW"]) == nw, (name, d["NW"], nw)
            d["W0"], d["tuples"] = w0, tup
        d.update(K=np.int64(K), L=np.int64(L), minCount=np.int64(mc), lam=np.float64(HYPER["lam"]),
                 bias_reg=np.float64(HYPER["bias_reg"]), relation_reg=np.float64(HYPER["relation_reg"]),
                 alpha=np.float64(alpha), lr=np.float64(LR))
        # the python restatement of loader + split must agree with the reference's dump
        assert int(d["n_users"]) == U and int(d["n_items"]) == I
        got = [list(d["click_item"][d["click_off"][u]:d["click_off"][u + 1]]) for u in range(U)]
        assert got == [[i for i, _ in s] for s in seqs], "loader restatement differs"
        np.savez_compressed(os.path.join(GOLD, f"toy_{name}.npz"), **d)
        print("toy", name, "NW", int(d["NW"]), "metrics", d["metrics"][:4])


ML5 = "ML1M-atmost-5"
# (fixture tag, model, K, L, minCount, lambda, bias_reg, relation_reg, alpha, iterations)
ML5_RUNS = [
    ("MP", "MP", 10, 1, 1, 0.0, 0.0, 0.0, 0.0, 0),
    ("BPRMF", "BPRMF", 10, 1, 1, 0.01, 0.01, 0.0, 0.0, 100),
    ("FPMC", "FPMC", 10, 1, 1, 0.01, 0.0, 0.0, 0.0, 100),
    ("TransRec", "TransRec", 10, 1, 1, 0.05, 0.01, 0.05, 0.0, 100),
    ("Fossil", "Fossil", 10, 3, 1, 0.01, 0.01, 0.0, 0.0, 100),
    ("REBUS_mc1_L1", "REBUS", 10, 1, 1, 0.01, 0.0, 0.0, -0.6, 100),
    ("REBUS_mc2_L3", "REBUS", 10, 3, 2, 0.01, 0.0, 0.0, -0.6, 100),
]


def parse_saved_model(path):
    """Inverse of model::saveModel (model.cpp:176-192): returns W as integer micro-units."""
    txt = open(path).read()
    nw = int(txt.split('"NW":')[1].split(",")[0])
    body = txt.split('"W": [')[1].split("]")[0]
    vals = body.split(",")
    assert len(vals) == nw
    micro = np.array([int(round(float(v) * 1e6)) for v in vals], dtype=np.int64)
    w = micro.astype(np.float64) / 1e6
    assert all(float(v) == x for v, x in zip(vals[:2000], w[:2000]))
    assert np.abs(micro).max() < 2 ** 31
    return micro.astype(np.int32)


def make_ml5():
    shutil.copy(os.path.join(REF, "01-Data", ML5 + ".txt"), os.path.join(WORK, "01-Data"))
    for f in os.listdir(os.path.join(REF, "96-FSUB")):
        if f.startswith(ML5 + "_"):
            shutil.copy(os.path.join(REF, "96-FSUB", f), os.path.join(WORK, "96-FSUB"))
    # data files the tests need where /root/reference is absent
    with open(os.path.join(REF, "01-Data", ML5 + ".txt"), "rb") as f, \
            gzip.GzipFile(os.path.join(GOLD, ML5 + ".txt.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    for tag in ("minCount_1_L_1", "minCount_2_L_3"):
        fn = f"{ML5}_root_fsub_{tag}.txt"
        with open(os.path.join(REF, "96-FSUB", fn), "rb") as f, \
                gzip.GzipFile(os.path.join(GOLD, fn + ".gz"), "wb", mtime=0) as g:
            g.write(f.read())
    data_rel = f"01-Data/{ML5}.txt"
    for tag, name, K, L, mc, lam, breg, rreg, alpha, iters in ML5_RUNS:
        out = os.path.join(WORK, "out", "ml5_" + tag)
        w0 = None
        extra = {}
        if name != "MP":
            cmd = [TRAIN, data_rel, "0", "0", str(mc), str(L), str(K), repr(lam), repr(breg), repr(rreg),
                   repr(alpha), str(iters), "03-Models/" + tag, name]
            r = subprocess.run(cmd, cwd=WORK, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            saved = [f for f in os.listdir(os.path.join(WORK, "03-Models")) if f.startswith(tag + "__")]
            assert len(saved) == 1, saved
            micro = parse_saved_model(os.path.join(WORK, "03-Models", saved[0]))
            extra["W_micro"] = micro
            extra["saved_name"] = np.array(saved[0])
            extra["cli_tail"] = np.array("\n".join(r.stdout.strip().split("\n")[-12:]))
            w0 = micro.astype(np.float64) / 1e6
        else:
            r = subprocess.run([TRAIN, data_rel, "0", "0", "1", "1", "10", "0", "0", "0", "0", "0", "03-Models/x", "MP"],
                               cwd=WORK, capture_output=True, text=True)
            extra["cli_tail"] = np.array("\n".join(r.stdout.strip().split("\n")[-4:]))
        run_driver(name, data_rel, K, L, mc, lam, breg, rreg, alpha, w0, None, out)
        d = read_outputs(out, name)
        d.pop("W1", None)
        d.update(extra)
        d.update(K=np.int64(K), L=np.int64(L), minCount=np.int64(mc), lam=np.float64(lam),
                 bias_reg=np.float64(breg), relation_reg=np.float64(rreg), alpha=np.float64(alpha))
        if tag != "MP":
            # the corpus is identical for every run; keep it once (in ml5_MP.npz)
            for k in ("click_off", "click_item", "click_time"):
                d.pop(k)
        np.savez_compressed(os.path.join(GOLD, f"ml5_{tag}.npz"), **d)
        print("ml5", tag, "NW", int(d["NW"]), "metrics", d["metrics"])


if __name__ == "__main__":
    if not (os.path.exists(DRIVER) and os.path.exists(TRAIN)):
        sys.exit("run `make -C oracle ref` first")
    os.makedirs(GOLD, exist_ok=True)
    setup_work()
    what = sys.argv[1:] or ["toy", "ml5"]
    if "toy" in what:
        make_toy()
    if "ml5" in what:
        make_ml5()
