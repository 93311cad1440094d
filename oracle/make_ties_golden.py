#!/usr/bin/env python3
"""Loader fixtures with tied time stamps and long sequences: the reference sorts every user's clicks with an
unstable std::sort (src/corpus.cpp:219), so the order of tied clicks is whatever libstdc++'s introsort leaves,
which beyond 16 elements differs from a stable sort.  Writes a click file (whitespace form and the .csv form of
src/corpus.cpp:43-81) and the corpus the REAL reference loads from each (oracle/_ref/ref_driver's dump):

    tests/golden/ties_clicks.txt, tests/golden/ties_clicks.csv, tests/golden/ties_corpus.npz

Test infrastructure; needs oracle/_ref (built from /root/reference by oracle/Makefile)."""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
DRIVER = os.path.join(HERE, "_ref", "ref_driver")
WORK = "/tmp/refwork_ties"


def main():
    rng = np.random.Generator(np.random.PCG64(7))
    lines = []
    for u in range(60):
        n = int(rng.integers(3, 70))
        # few distinct time stamps: many ties, also in sequences far longer than introsort's 16-element threshold
        t = rng.integers(0, max(2, n // 4), n) + 1000
        it = rng.integers(0, 50, n)
        lines += [(f"u{u}", f"i{i}", "4.0", int(tt)) for i, tt in zip(it.tolist(), t.tolist())]
    order = rng.permutation(len(lines))   # users interleaved, as in a real log
    lines = [lines[k] for k in order]
    os.makedirs(os.path.join(WORK, "01-Data"), exist_ok=True)
    out = {}
    for ext, sep in ((".txt", " "), (".csv", ",")):
        text = "".join(sep.join(str(x) for x in ln) + "\n" for ln in lines)
        open(os.path.join(GOLD, "ties_clicks" + ext), "w").write(text)
        open(os.path.join(WORK, "01-Data", "ties" + ext), "w").write(text)
        subprocess.check_call([DRIVER, "MP", "01-Data/ties" + ext, "1", "1", "1", "0", "0", "0", "0", "-", "-", "0.05",
                               os.path.join(WORK, "o" + ext[1:])], cwd=WORK, stdout=subprocess.DEVNULL)
        out["corpus_" + ext[1:]] = np.fromfile(os.path.join(WORK, "o" + ext[1:] + ".corpus.i32"), np.int32)
    np.savez_compressed(os.path.join(GOLD, "ties_corpus.npz"), **out)
    print({k: v[:3] for k, v in out.items()})


if __name__ == "__main__":
    main()
