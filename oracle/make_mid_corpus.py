#!/usr/bin/env python3
"""A mid-size synthetic click file with long histories, of the same distribution as bench.py's corpus
(per-user lengths 3 + geometric, items Zipf(0.9) over popularity ranks, half of a user's clicks shifted
into one of 64 taste clusters), small enough for the reference binary to train on it in minutes:

    tests/golden/mid_clicks.txt.gz                         user item rating time   (src/corpus.cpp:91)
    tests/golden/mid_clicks_root_fsub_minCount_2_L_3.txt.gz   label set mined by recommendation_amd/fsub_mine

Time stamps increase strictly inside a user, so the reference's unstable per-user sort (src/corpus.cpp:219)
cannot reorder anything.  Test infrastructure."""
import gzip
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

N_USERS, N_ITEMS, MEAN_CLICKS, SEED = 12000, 4000, 50.0, 20261111


def generate(n_users=N_USERS, n_items=N_ITEMS, mean_clicks=MEAN_CLICKS, seed=SEED):
    rng = np.random.Generator(np.random.PCG64(seed))
    p = 1.0 / (mean_clicks - 3.0 + 1.0)
    n = 3 + np.minimum(rng.geometric(p, n_users) - 1, 600)
    w = np.arange(1, n_items + 1, dtype=np.float64) ** -0.9
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    perm = rng.permutation(n_items)
    cluster = rng.integers(0, 64, n_users)
    total = int(n.sum())
    user_of = np.repeat(np.arange(n_users), n)
    z = np.minimum(np.searchsorted(cdf, rng.random(total)), n_items - 1)
    shifted = rng.random(total) < 0.5
    z = np.where(shifted, (z + cluster[user_of] * (n_items // 64)) % n_items, z)
    items = perm[z]
    start = np.concatenate([[0], np.cumsum(n)[:-1]])
    times = np.arange(total) - start[user_of] + 1
    return user_of, items, times


def main():
    users, items, times = generate()
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, "mid_clicks.txt.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write("".join(f"u{u} i{i} 1 {t}\n" for u, i, t in zip(users.tolist(), items.tolist(), times.tolist())).encode())
    print(len(users), "clicks ->", path)
    # the label set of REBUS minCount 2 / L 3, mined by this repository's own tool (the reference ships no miner)
    work = "/tmp/mid_fsub"
    os.makedirs(os.path.join(work, "01-Data"), exist_ok=True)
    os.makedirs(os.path.join(work, "96-FSUB"), exist_ok=True)
    with gzip.open(path, "rb") as f, open(os.path.join(work, "01-Data", "mid_clicks.txt"), "wb") as g:
        g.write(f.read())
    miner = os.path.join(ROOT, "recommendation_amd", "fsub_mine")
    subprocess.check_call([miner, "01-Data/mid_clicks.txt", "0", "0", "2", "3"], cwd=work)
    lab = os.path.join(work, "96-FSUB", "mid_clicks_root_fsub_minCount_2_L_3.txt")
    with open(lab, "rb") as f, gzip.GzipFile(os.path.join(GOLD, "mid_clicks_root_fsub_minCount_2_L_3.txt.gz"), "wb", mtime=0) as g:
        g.write(f.read())


if __name__ == "__main__":
    sys.exit(main())
