"""The ./train CLI (reference's 13 positional arguments, src/main.cpp:142-231) on the HIP engine:
known answer of MostPopular, progress / result lines, saved-model format, REBUS label loading."""
import gzip
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from tests.golden_util import GOLD, fixture_case, make_oracle
from oracle import oracle_py as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAIN = os.path.join(ROOT, "recommendation_amd", "train")
ML5 = "ML1M-atmost-5"


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    w = tmp_path_factory.mktemp("cli")
    for d in ("01-Data", "96-FSUB", "03-Models"):
        os.makedirs(w / d)
    with gzip.open(os.path.join(GOLD, ML5 + ".txt.gz"), "rb") as f, open(w / "01-Data" / (ML5 + ".txt"), "wb") as g:
        shutil.copyfileobj(f, g)
    # the loader reads gzip as well (the reference goes through igzstream)
    shutil.copy(os.path.join(GOLD, ML5 + ".txt.gz"), w / "01-Data" / (ML5 + "z.txt"))
    for tag in ("minCount_1_L_1", "minCount_2_L_3"):
        fn = f"{ML5}_root_fsub_{tag}.txt"
        with gzip.open(os.path.join(GOLD, fn + ".gz"), "rb") as f, open(w / "96-FSUB" / fn, "wb") as g:
            shutil.copyfileobj(f, g)
    return w


def run(workdir, *args, expect=0, env=None):
    r = subprocess.run([TRAIN] + [str(a) for a in args], cwd=workdir, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **env) if env else None)
    assert r.returncode == expect, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def parse_saved(path):
    txt = open(path).read()
    nw = int(txt.split('"NW":')[1].split(",")[0])
    vals = np.array([float(v) for v in txt.split('"W": [')[1].split("]")[0].split(",")])
    assert len(vals) == nw
    assert txt.startswith('{\n  "NW": ') and txt.endswith("]\n}\n")
    return vals


def test_usage_exits_with_1_without_a_gpu_call():
    r = subprocess.run([TRAIN, "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "13. Model name" in r.stdout


@pytest.mark.gpu
def test_most_popular_known_answer(workdir):
    out = run(workdir, f"01-Data/{ML5}.txt", 0, 0, 1, 1, 10, 0, 0, 0, 0, 0, "03-Models/x", "MP")
    assert '"nUsers": 6040, "nItems": 2848, "nClicks": 30175' in out
    assert "num_pos_events :18097" in out
    # the reference prints exactly these two lines for this command (paper Table 4, Pop / ML-5)
    assert "#Users = 6039, VALID: AUC = 0.739072, MRR = 0.019851, HR@50 = 0.131148" in out
    assert "#Users = 6039, TEST:  AUC = 0.735201, MRR = 0.018224, HR@50 = 0.126511" in out
    out2 = run(workdir, f"01-Data/{ML5}z.txt", 0, 0, 1, 1, 10, 0, 0, 0, 0, 0, "03-Models/x", "MP")
    assert "TEST:  AUC = 0.735201" in out2


@pytest.mark.gpu
def test_bprmf_run_and_saved_model(workdir):
    out = run(workdir, f"01-Data/{ML5}.txt", 0, 0, 1, 1, 10, 0.01, 0.01, 0, 0, 100, "03-Models/m", "BPRMF")
    assert "<<< BPR-MF__K_10_lambda_0.010000_biasReg_0.010000 >>>" in out
    assert len(re.findall(r"^Iter: \d+, took ", out, flags=re.M)) == 100
    assert len(re.findall(r"^\[Valid AUC = ", out, flags=re.M)) == 2
    m = re.search(r"Test AUC = ([0-9.]+), Val AUC = ([0-9.]+)", out)
    auc_test, auc_val = float(m.group(1)), float(m.group(2))
    assert 0.68 < auc_test < 0.76
    path = workdir / "03-Models" / "m__BPR-MF__K_10_lambda_0.010000_biasReg_0.010000.txt"
    W = parse_saved(path)
    assert len(W) == 91728
    # the saved file, scored by the CPU oracle, reproduces what the CLI printed (6-decimal text)
    g, model, co, ci, labels = fixture_case("BPRMF", "ml5")
    _, counts = make_oracle(g, model, co, ci, labels).eval(W)
    mm = orc.metrics(counts)
    assert abs(mm[1] - auc_test) < 2e-4 and abs(mm[0] - auc_val) < 2e-4
    # 0 iterations + the existing model file: evaluation only, criterion (b) through the command line
    out = run(workdir, f"01-Data/{ML5}.txt", 0, 0, 1, 1, 10, 0.01, 0.01, 0, 0, 0, "03-Models/m", "BPRMF")
    assert "loading parameters from 03-Models/m__BPR-MF__K_10_lambda_0.010000_biasReg_0.010000.txt" in out
    assert "Iter: " not in out and "Model saved" not in out
    m = re.search(r"Test AUC = ([0-9.]+), Val AUC = ([0-9.]+)", out)
    assert abs(float(m.group(1)) - mm[1]) < 1e-6 and abs(float(m.group(2)) - mm[0]) < 1e-6
    m = re.search(r"Test HIT50 = ([0-9.]+), Val HIT50 = ([0-9.]+)", out)
    assert abs(float(m.group(1)) - mm[3]) < 1e-6 and abs(float(m.group(2)) - mm[2]) < 1e-6


@pytest.mark.gpu
def test_rebus_run(workdir):
    out = run(workdir, f"01-Data/{ML5}.txt", 0, 0, 2, 3, 10, 0.01, 0, 0, -0.6, 100, "03-Models/r", "REBUS")
    assert "ss : 96-FSUB/ML1M-atmost-5_root_fsub_minCount_2_L_3.txt" in out
    assert " --NW : 31339 -- " in out
    assert "Validation ..." in out
    m = re.search(r"Test AUC = ([0-9.]+), Val AUC = ([0-9.]+)", out)
    assert 0.70 < float(m.group(1)) < 0.82
    saved = [f for f in os.listdir(workdir / "03-Models") if f.startswith("r__REBUS_K_10_minCount_2_L_3")]
    assert len(saved) == 1 and len(parse_saved(workdir / "03-Models" / saved[0])) == 31339
    out = run(workdir, f"01-Data/{ML5}.txt", 0, 0, 7, 7, 10, 0.01, 0, 0, -0.6, 1, "03-Models/r", "REBUS")
    assert "No substrings found for the dataset" in out


@pytest.mark.gpu
def test_run_shorter_than_the_first_validation_keeps_the_trained_model(workdir):
    """10 iterations: no validation ever runs (cadence 50), so there is no best snapshot to copy back
    (the reference copies an uninitialised bestW here, BPRMF.cpp:189-191).  The trained parameters must be
    what is evaluated and saved."""
    out = run(workdir, f"01-Data/{ML5}.txt", 0, 0, 1, 1, 10, 0.01, 0.01, 0, 0, 10, "03-Models/s", "BPRMF")
    assert "[Valid AUC" not in out
    W = parse_saved(workdir / "03-Models" / "s__BPR-MF__K_10_lambda_0.010000_biasReg_0.010000.txt")
    assert np.count_nonzero(W) > 0.9 * len(W)
    g, model, co, ci, labels = fixture_case("BPRMF", "ml5")
    _, counts = make_oracle(g, model, co, ci, labels).eval(W)
    mm = orc.metrics(counts)
    m = re.search(r"Test AUC = ([0-9.]+), Val AUC = ([0-9.]+)", out)
    assert abs(float(m.group(1)) - mm[1]) < 2e-4 and abs(float(m.group(2)) - mm[0]) < 2e-4
    assert float(m.group(1)) > 0.55


@pytest.mark.gpu
def test_sampled_validation_modes_of_train(workdir):
    """train() validates on the exact full-catalog AUC by default; RECSYS_SAMPLED_VALIDATION=1 brings back the reference's
    estimate (model::sampleAUC, model.cpp:100-167: 2000 draws per user), =2 one draw shared by all users.  The epochs are
    the same in all three runs (the device sampler does not share a random stream with the validation), so the printed
    validation values are three estimates of the same numbers."""
    args = (f"01-Data/{ML5}.txt", 0, 0, 1, 1, 10, 0.01, 0.01, 0, 0, 100, "03-Models/v", "BPRMF")

    def valid_lines(out):
        return [(float(a), float(b)) for a, b in re.findall(r"^\[Valid AUC = ([0-9.]+)\], Test AUC = ([0-9.]+)", out, flags=re.M)]

    exact = valid_lines(run(workdir, *args))
    per_user = valid_lines(run(workdir, *args, env={"RECSYS_SAMPLED_VALIDATION": "1"}))
    shared = valid_lines(run(workdir, *args, env={"RECSYS_SAMPLED_VALIDATION": "2"}))
    assert len(exact) == len(per_user) == len(shared) == 2
    for x, p, s_ in zip(exact, per_user, shared):
        # the reference's estimator has a standard deviation of about 1.2e-4 on this corpus (tests/golden/ml5_sampleauc.json)
        assert p != x and abs(p[0] - x[0]) < 7e-4 and abs(p[1] - x[1]) < 7e-4, (x, p)
        assert abs(s_[0] - x[0]) < 0.03 and abs(s_[1] - x[1]) < 0.03, (x, s_)


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra", [("BPRMF", (1, 1, 10, 0.01, 0.01, 0, 0)), ("TransRec", (1, 1, 10, 0.05, 0.01, 0.05, 0)),
                                         ("REBUS", (2, 3, 10, 0.01, 0, 0, -0.6))])
def test_train_with_user_shards_in_one_process(workdir, name, extra):
    """RECSYS_SHARDS=N: train() runs N engines with consecutive user ranges (here all on the one GPU) in rounds of local
    steps, the item tables combined after every round (rec_shards_begin_round / rec_shards_combine).  The run must train
    like the single-engine run, validate and evaluate over all users, and save a model file whose re-scoring by the CPU
    oracle reproduces the printed metrics (user rows gathered from their shards)."""
    args = (f"01-Data/{ML5}.txt", 0, 0) + extra + (100, "03-Models/sh", name)
    single = run(workdir, *args)
    aucs = {}
    for n in (2, 3):
        out = run(workdir, *args, env={"RECSYS_SHARDS": str(n)})
        assert f'"shards": {n}, "devices": 1, "rounds per epoch": ' in out
        assert len(re.findall(r"^Iter: \d+, took ", out, flags=re.M)) == 100
        assert len(re.findall(r"^\[Valid AUC = ", out, flags=re.M)) == (1 if name == "REBUS" else 2)
        m = re.search(r"Test AUC = ([0-9.]+), Val AUC = ([0-9.]+)", out)
        aucs[n] = (float(m.group(1)), float(m.group(2)))
        h = re.search(r"Test HIT50 = ([0-9.]+), Val HIT50 = ([0-9.]+)", out)
        saved = [f for f in os.listdir(workdir / "03-Models") if f.startswith("sh__")]
        assert len(saved) == 1, saved
        W = parse_saved(workdir / "03-Models" / saved[0])
        os.remove(workdir / "03-Models" / saved[0])
        tag = {"BPRMF": "BPRMF", "TransRec": "TransRec", "REBUS": "REBUS_mc2_L3"}[name]
        g, model, co, ci, labels = fixture_case(tag, "ml5")
        _, counts = make_oracle(g, model, co, ci, labels).eval(W)
        mm = orc.metrics(counts)
        assert abs(mm[1] - aucs[n][0]) < 3e-4 and abs(mm[0] - aucs[n][1]) < 3e-4, (n, mm[:2], aucs[n])
        assert abs(mm[3] - float(h.group(1))) < 2e-3 and abs(mm[2] - float(h.group(2))) < 2e-3
    m = re.search(r"Test AUC = ([0-9.]+), Val AUC = ([0-9.]+)", single)
    for n in (2, 3):
        assert abs(aucs[n][0] - float(m.group(1))) < 0.012 and abs(aucs[n][1] - float(m.group(2))) < 0.012, (n, aucs[n], m.groups())


@pytest.mark.gpu
def test_first_benchmark_configuration_as_shipped(workdir):
    """`./train 01-Data/ML1M-atmost-5.txt 0 0 1 1 10 0.01 0 0 -0.6 100 03-Models/my_model REBUS`, the command the reference
    ships with (minCount 1, L 1, alpha -0.6, 100 iterations), end to end through the CLI on the engine; then criterion (b)
    through the command line: the model file the REFERENCE binary saved for this very command (tests/golden/ml5_REBUS_mc1_L1:
    its 6-decimal parameters and the metrics its own AUC() printed for them) is put where the CLI looks for it and scored
    with 0 iterations - AUC and HIT@50 within 1e-6 of the reference's."""
    args = (f"01-Data/{ML5}.txt", 0, 0, 1, 1, 10, 0.01, 0, 0, -0.6)
    out = run(workdir, *args, 100, "03-Models/my_model", "REBUS")
    assert "ss : 96-FSUB/ML1M-atmost-5_root_fsub_minCount_1_L_1.txt" in out
    assert " --NW : 31339 -- " in out
    assert len(re.findall(r"^Iter: \d+, took ", out, flags=re.M)) == 100
    assert len(re.findall(r"^\[Valid AUC = ", out, flags=re.M)) == 1          # REBUS validates every 100 iterations (REBUS.cpp:578)
    m = re.search(r"Test AUC = ([0-9.]+), Val AUC = ([0-9.]+)", out)
    g, model, co, ci, labels = fixture_case("REBUS_mc1_L1", "ml5")
    ref = g["metrics"]                                                          # AUC_val AUC_test HIT_val HIT_test MRR_val MRR_test var
    # 100 iterations of the reference end at test AUC ref[1]; a parallel run of the same length lands near it
    assert abs(float(m.group(1)) - ref[1]) < 0.02, (m.groups(), ref[:2])
    saved = [f for f in os.listdir(workdir / "03-Models") if f.startswith("my_model__REBUS_K_10_minCount_1_L_1")]
    assert len(saved) == 1, saved
    assert len(parse_saved(workdir / "03-Models" / saved[0])) == 31339
    # the reference's own saved parameters in its own text format (model.cpp:176-192), under the same name
    W = g["W_micro"].astype(np.int64)
    body = ", ".join(("-" if v < 0 else "") + "%d.%06d" % divmod(abs(int(v)), 1000000) for v in W)
    with open(workdir / "03-Models" / saved[0], "w") as f:
        f.write('{\n  "NW": %d,\n  "W": [%s]\n}\n' % (len(W), body))
    out = run(workdir, *args, 0, "03-Models/my_model", "REBUS")
    assert "loading parameters from 03-Models/" + saved[0] in out and "Iter: " not in out
    m = re.search(r"Test AUC = ([0-9.]+), Val AUC = ([0-9.]+)", out)
    h = re.search(r"Test HIT50 = ([0-9.]+), Val HIT50 = ([0-9.]+)", out)
    assert abs(float(m.group(1)) - ref[1]) < 1e-6 and abs(float(m.group(2)) - ref[0]) < 1e-6, (m.groups(), ref)
    assert abs(float(h.group(1)) - ref[3]) < 1e-6 and abs(float(h.group(2)) - ref[2]) < 1e-6, (h.groups(), ref)
