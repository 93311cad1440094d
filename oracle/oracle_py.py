"""ctypes access to oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY: imported
by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg - never by the product
package recommendation_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_ID = {"MP": 0, "BPRMF": 1, "MC": 2, "FPMC": 3, "PRME": 4, "HRM_avg": 5, "HRM_max": 6,
            "TransRec": 7, "TransRec_L1": 8, "FossilSimple": 9, "Fossil": 10, "REBUS": 11}


class OrcCtx(C.Structure):
    _fields_ = [("model", C.c_int32), ("n_users", C.c_int32), ("n_items", C.c_int32),
                ("K", C.c_int32), ("KK", C.c_int32), ("L", C.c_int32),
                ("lam", C.c_double), ("bias_reg", C.c_double), ("relation_reg", C.c_double),
                ("alpha", C.c_double), ("dataset_factor", C.c_int32),
                ("seq_off", C.c_void_p), ("seq_item", C.c_void_p), ("test_item", C.c_void_p),
                ("val_item", C.c_void_p), ("path_off", C.c_void_p), ("path_item", C.c_void_p),
                ("eta", C.c_void_p)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.orc_nw.restype = C.c_int64
        _lib.orc_rebus_paths.restype = C.c_int64
        _lib.orc_bprmf_epoch.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def split(click_off, click_item):
    """model.hpp:22-45 on a click CSR sorted by time."""
    U = len(click_off) - 1
    click_off = np.ascontiguousarray(click_off, np.int64)
    click_item = np.ascontiguousarray(click_item, np.int32)
    seq_off = np.zeros(U + 1, np.int64)
    seq_item = np.zeros(max(1, len(click_item)), np.int32)
    test = np.zeros(U, np.int32)
    val = np.zeros(U, np.int32)
    lib().orc_split(C.c_int32(U), _p(click_off), _p(click_item), _p(seq_off), _p(seq_item), _p(test), _p(val))
    return seq_off, seq_item[:seq_off[-1]].copy(), test, val


def eta_table(L):
    eta = np.zeros((L + 1) * max(L, 1), np.float64)
    lib().orc_eta_table(C.c_int(L), _p(eta))
    return eta


def parse_labels(lines, n_items):
    """96-FSUB label lines -> (lab_off, lab_item); 'Root' is dropped (it is never a candidate
    of findPathStars, REBUS.cpp:498,507, since candidates are made of item ids only)."""
    off, items = [0], []
    for ln in lines:
        ln = ln.strip()
        if not ln or ln == "Root":
            continue
        items.extend(int(x) for x in ln.split("-"))
        off.append(len(items))
    return np.array(off, np.int64), np.array(items, np.int32)


def rebus_paths(n_items, seq_off, seq_item, val_item, lab_off, lab_item):
    U = len(seq_off) - 1
    n_ctx = int(seq_off[-1]) + 2 * U
    path_off = np.zeros(n_ctx + 1, np.int64)
    args = (C.c_int32(U), C.c_int32(n_items), _p(seq_off), _p(seq_item), _p(val_item),
            C.c_int64(len(lab_off) - 1), _p(lab_off), _p(lab_item))
    n = lib().orc_rebus_paths(*args, _p(path_off), None)
    path_item = np.zeros(max(1, n), np.int32)
    lib().orc_rebus_paths(*args, _p(path_off), _p(path_item))
    return path_off, path_item


class Oracle:
    """Holds the arrays an orc_ctx points to."""

    def __init__(self, model, n_users, n_items, K, seq_off, seq_item, test_item, val_item, L=1, KK=None,
                 lam=0.0, bias_reg=0.0, relation_reg=0.0, alpha=0.0, dataset_factor=10,
                 path_off=None, path_item=None):
        self.model = model
        self.seq_off = np.ascontiguousarray(seq_off, np.int64)
        self.seq_item = np.ascontiguousarray(seq_item, np.int32)
        self.test_item = np.ascontiguousarray(test_item, np.int32)
        self.val_item = np.ascontiguousarray(val_item, np.int32)
        self.path_off = None if path_off is None else np.ascontiguousarray(path_off, np.int64)
        self.path_item = None if path_item is None else np.ascontiguousarray(path_item, np.int32)
        self.eta = eta_table(L)
        self.ctx = OrcCtx(MODEL_ID[model], n_users, n_items, K, K if KK is None else KK, L, lam, bias_reg,
                          relation_reg, alpha, dataset_factor, _p(self.seq_off), _p(self.seq_item),
                          _p(self.test_item), _p(self.val_item), _p(self.path_off), _p(self.path_item),
                          _p(self.eta))
        self.n_users, self.n_items = n_users, n_items

    @property
    def nw(self):
        return int(lib().orc_nw(C.byref(self.ctx)))

    def replay(self, W, tuples, lr):
        W = np.ascontiguousarray(W, np.float64).copy()
        tuples = np.ascontiguousarray(tuples, np.int32)
        rc = lib().orc_replay(C.byref(self.ctx), _p(W), _p(tuples), C.c_int64(len(tuples)), C.c_double(lr))
        assert rc == 0
        return W

    def eval(self, W, nthreads=0):
        W = np.ascontiguousarray(W if W is not None else np.zeros(1), np.float64)
        xpos = np.zeros((self.n_users, 2), np.float64)
        counts = np.zeros((self.n_users, 3), np.int32)
        rc = lib().orc_eval(C.byref(self.ctx), _p(W), _p(xpos), _p(counts), C.c_int(nthreads))
        assert rc == 0
        return xpos, counts

    def score_row(self, W, u, is_test):
        W = np.ascontiguousarray(W if W is not None else np.zeros(1), np.float64)
        row = np.zeros(self.n_items, np.float64)
        rc = lib().orc_score_row(C.byref(self.ctx), _p(W), C.c_int(u), C.c_int(is_test), _p(row))
        assert rc == 0
        return row

    def bprmf_epoch(self, W, lr, n_steps, seed):
        """In place on W (float64, contiguous)."""
        assert W.dtype == np.float64 and W.flags.c_contiguous
        return int(lib().orc_bprmf_epoch(C.byref(self.ctx), _p(W), C.c_double(lr), C.c_int64(n_steps),
                                         C.c_uint64(seed)))


def metrics(counts, topn=50):
    counts = np.ascontiguousarray(counts, np.int32)
    out = np.zeros(7, np.float64)
    lib().orc_metrics(C.c_int32(len(counts)), _p(counts), C.c_int(topn), _p(out))
    return out
