"""ctypes binding of include/recsys_engine.h (same names, same argument meaning)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librecsys_amd.so")

MODEL_ID = {"MP": 0, "BPRMF": 1, "MC": 2, "FPMC": 3, "PRME": 4, "HRM_avg": 5, "HRM_max": 6,
            "TransRec": 7, "TransRec_L1": 8, "FossilSimple": 9, "Fossil": 10, "REBUS": 11}
F32, F64 = 0, 1
SEQUENTIAL, HOGWILD = 0, 1

EXPORTS = ["rec_last_error", "rec_create", "rec_destroy", "rec_sync", "rec_nw", "rec_params_upload",
           "rec_params_download", "rec_params_read", "rec_params_init", "rec_snapshot_best", "rec_restore_best", "rec_sgd_replay", "rec_sgd_replay_static", "rec_partition_tuples", "rec_owner_slices",
           "rec_sgd_epoch", "rec_sample_tuples", "rec_sample_events", "rec_tuples_drawn", "rec_steps_per_epoch", "rec_eval", "rec_eval_sampled", "rec_eval_sampled_per_user", "rec_eval_topk", "rec_eval_user_count", "rec_eval_sums",
           "rec_device_count", "rec_shards_begin_round", "rec_shards_combine", "rec_project_rows", "rec_rebus_build_paths", "rec_fsub_mine", "rec_num_tables", "rec_table_info", "rec_set_option", "rec_kernel_stats"]


class RecCorpus(C.Structure):
    _fields_ = [("n_users", C.c_int32), ("n_items", C.c_int32), ("seq_off", C.c_void_p), ("seq_item", C.c_void_p),
                ("test_item", C.c_void_p), ("val_item", C.c_void_p), ("path_off", C.c_void_p), ("path_item", C.c_void_p)]


class RecModelDesc(C.Structure):
    _fields_ = [("model", C.c_int32), ("K", C.c_int32), ("KK", C.c_int32), ("L", C.c_int32),
                ("lam", C.c_double), ("bias_reg", C.c_double), ("relation_reg", C.c_double), ("alpha", C.c_double),
                ("dataset_factor", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32),
                ("user_begin", C.c_int32), ("user_end", C.c_int32)]


class Metrics(C.Structure):
    _fields_ = [("auc_val", C.c_double), ("auc_test", C.c_double), ("hit_val", C.c_double), ("hit_test", C.c_double),
                ("mrr_val", C.c_double), ("mrr_test", C.c_double), ("var_test", C.c_double), ("n_users", C.c_int64)]

    def as_array(self):
        """Order of the reference's AUC() outputs: AUC_val AUC_test HIT_val HIT_test MRR_val MRR_test var."""
        return np.array([self.auc_val, self.auc_test, self.hit_val, self.hit_test, self.mrr_val, self.mrr_test,
                         self.var_test])


_lib = None


def lib():
    """The HIP library; raises if it has not been built (there is no CPU path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(LIB_PATH)
        _lib.rec_last_error.restype = C.c_char_p
        _lib.rec_nw.restype = C.c_int64
        _lib.rec_nw.argtypes = [C.c_void_p]
        _lib.rec_steps_per_epoch.restype = C.c_int64
        _lib.rec_steps_per_epoch.argtypes = [C.c_void_p]
        _lib.rec_fsub_mine.restype = C.c_int64
        _lib.rec_tuples_drawn.restype = C.c_int64
        _lib.rec_tuples_drawn.argtypes = [C.c_void_p]
        _lib.rec_rebus_build_paths.restype = C.c_int64
        _lib.rec_eval_user_count.restype = C.c_int64
        _lib.rec_eval_user_count.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc):
    if rc != 0:
        raise RuntimeError("recsys engine: " + lib().rec_last_error().decode())


def build_rebus_paths(n_items, seq_off, seq_item, val_item, lab_off, lab_item):
    seq_off = np.ascontiguousarray(seq_off, np.int64)
    seq_item = np.ascontiguousarray(seq_item, np.int32)
    val_item = np.ascontiguousarray(val_item, np.int32)
    lab_off = np.ascontiguousarray(lab_off, np.int64)
    lab_item = np.ascontiguousarray(lab_item, np.int32)
    U = len(seq_off) - 1
    path_off = np.zeros(int(seq_off[-1]) + 2 * U + 1, np.int64)
    args = (C.c_int32(U), C.c_int32(n_items), _p(seq_off), _p(seq_item), _p(val_item), C.c_int64(len(lab_off) - 1),
            _p(lab_off), _p(lab_item))
    n = lib().rec_rebus_build_paths(*args, _p(path_off), None)
    if n < 0:
        _check(-1)
    path_item = np.zeros(max(1, n), np.int32)
    lib().rec_rebus_build_paths(*args, _p(path_off), _p(path_item))
    return path_off, path_item


def fsub_mine(n_items, seq_off, seq_item, L, min_count):
    """Labels of a 96-FSUB file (without 'Root') as (lab_off, lab_item)."""
    seq_off = np.ascontiguousarray(seq_off, np.int64)
    seq_item = np.ascontiguousarray(seq_item, np.int32)
    U = len(seq_off) - 1
    total = C.c_int64(0)
    args = (C.c_int32(U), C.c_int32(n_items), _p(seq_off), _p(seq_item), C.c_int32(L), C.c_int64(min_count))
    n = lib().rec_fsub_mine(*args, None, None, C.byref(total))
    if n < 0:
        _check(-1)
    lab_off = np.zeros(n + 1, np.int64)
    lab_item = np.zeros(max(1, total.value), np.int32)
    lib().rec_fsub_mine(*args, _p(lab_off), _p(lab_item), C.byref(total))
    return lab_off, lab_item[:total.value]


class Engine:
    def __init__(self, model, n_items, seq_off, seq_item, test_item, val_item, K, KK=0, L=1, lam=0.0, bias_reg=0.0,
                 relation_reg=0.0, alpha=0.0, dataset_factor=0, precision=F32, device=0, user_begin=0, user_end=0,
                 path_off=None, path_item=None):
        self.handle = C.c_void_p()
        seq_off = np.ascontiguousarray(seq_off, np.int64)
        seq_item = np.ascontiguousarray(seq_item, np.int32)
        test_item = np.ascontiguousarray(test_item, np.int32)
        val_item = np.ascontiguousarray(val_item, np.int32)
        if path_off is not None:
            path_off = np.ascontiguousarray(path_off, np.int64)
            path_item = np.ascontiguousarray(path_item, np.int32)
        self.n_users, self.n_items = len(seq_off) - 1, n_items
        corpus = RecCorpus(self.n_users, n_items, _p(seq_off), _p(seq_item), _p(test_item), _p(val_item),
                           _p(path_off), _p(path_item))
        desc = RecModelDesc(MODEL_ID[model], K, KK, L, lam, bias_reg, relation_reg, alpha, dataset_factor, precision,
                            device, user_begin, user_end)
        _check(lib().rec_create(C.byref(corpus), C.byref(desc), C.byref(self.handle)))
        self.precision = precision
        self.model = model
        self.user_begin, self.user_end = (0, self.n_users) if (user_begin == 0 and user_end == 0) else (user_begin, user_end)

    def close(self):
        if self.handle:
            lib().rec_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def nw(self):
        return int(lib().rec_nw(self.handle))

    @property
    def steps_per_epoch(self):
        return int(lib().rec_steps_per_epoch(self.handle))

    def sync(self):
        _check(lib().rec_sync(self.handle))

    def upload(self, W):
        W = np.ascontiguousarray(W, np.float64)
        _check(lib().rec_params_upload(self.handle, _p(W), C.c_int64(len(W))))

    def download(self):
        W = np.zeros(self.nw, np.float64)
        _check(lib().rec_params_download(self.handle, _p(W), C.c_int64(len(W))))
        return W

    def read_params(self, first, n):
        out = np.zeros(n, np.float64)
        _check(lib().rec_params_read(self.handle, C.c_int64(first), C.c_int64(n), _p(out)))
        return out

    def init_params(self, seed):
        _check(lib().rec_params_init(self.handle, C.c_uint64(seed)))

    def snapshot_best(self):
        _check(lib().rec_snapshot_best(self.handle))

    def restore_best(self):
        _check(lib().rec_restore_best(self.handle))

    def sgd_replay(self, lr, tuples, mode=SEQUENTIAL):
        tuples = np.ascontiguousarray(tuples, np.int32)
        assert tuples.ndim == 2 and tuples.shape[1] == 4
        _check(lib().rec_sgd_replay(self.handle, C.c_double(lr), _p(tuples), C.c_int64(len(tuples)), C.c_int(mode)))

    def sgd_replay_static(self, lr, tuples, mode=SEQUENTIAL):
        """rec_sgd_replay without any device allocation (at most 65536 tuples per call)."""
        tuples = np.ascontiguousarray(tuples, np.int32)
        assert tuples.ndim == 2 and tuples.shape[1] == 4
        _check(lib().rec_sgd_replay_static(self.handle, C.c_double(lr), _p(tuples), C.c_int64(len(tuples)), C.c_int(mode)))

    def sgd_epoch(self, lr, n_steps, seed, epoch):
        _check(lib().rec_sgd_epoch(self.handle, C.c_double(lr), C.c_int64(n_steps), C.c_uint64(seed), C.c_uint64(epoch)))

    def sample_tuples(self, n, seed, epoch):
        out = np.zeros((n, 4), np.int32)
        _check(lib().rec_sample_tuples(self.handle, C.c_int64(n), C.c_uint64(seed), C.c_uint64(epoch), _p(out)))
        return out

    @property
    def owner_slices(self):
        return int(lib().rec_owner_slices(self.handle))

    def partition_tuples(self, tuples, slices=256):
        """The owner mode's partition alone: (tuples ordered by the negative's slice, slice starts [slices + 1])."""
        tuples = np.ascontiguousarray(tuples, np.int32)
        out = np.zeros_like(tuples)
        qoff = np.zeros(slices + 1, np.int64)
        _check(lib().rec_partition_tuples(self.handle, _p(tuples), C.c_int64(len(tuples)), C.c_int32(slices), _p(out), _p(qoff)))
        return out, qoff

    def sample_events(self, n_expected, seed, epoch, cap=None):
        cap = int(cap if cap is not None else n_expected + 8 * n_expected ** 0.5 + 1024)
        out = np.zeros((cap, 4), np.int32)
        n = C.c_int64(0)
        _check(lib().rec_sample_events(self.handle, C.c_int64(n_expected), C.c_uint64(seed), C.c_uint64(epoch), C.c_int64(cap),
                                       _p(out), C.byref(n)))
        return out[:n.value]

    @property
    def tuples_drawn(self):
        return int(lib().rec_tuples_drawn(self.handle))

    def eval(self, topn=50, want_counts=True, want_xpos=False):
        m = Metrics()
        counts = np.zeros((self.n_users, 3), np.int32) if want_counts else None
        xpos = np.zeros((self.n_users, 2), np.float64) if want_xpos else None
        _check(lib().rec_eval(self.handle, C.c_int(topn), C.byref(m), _p(counts), _p(xpos)))
        return m, counts, xpos

    def eval_sampled(self, n_sampled=2000, seed=0, topn=50):
        m = Metrics()
        counts = np.zeros((self.n_users, 3), np.int32)
        sampled = np.zeros(n_sampled, np.int32)
        _check(lib().rec_eval_sampled(self.handle, C.c_int(topn), C.c_int32(n_sampled), C.c_uint64(seed), C.byref(m), _p(counts),
                                      _p(sampled)))
        return m, counts, sampled

    def eval_sampled_per_user(self, n_sampled=2000, seed=0, topn=50, want_draws=False):
        """model::sampleAUC with the reference's sampling: n_sampled candidate draws per user."""
        m = Metrics()
        counts = np.zeros((self.n_users, 3), np.int32)
        draws = np.zeros((self.n_users, n_sampled), np.int32) if want_draws else None
        _check(lib().rec_eval_sampled_per_user(self.handle, C.c_int(topn), C.c_int32(n_sampled), C.c_uint64(seed), C.byref(m),
                                               _p(counts), _p(draws)))
        return m, counts, draws

    def project_rows(self):
        """TransRec: every item row back onto the unit ball (what an update launch does unless "project_after_launch" is 0)."""
        _check(lib().rec_project_rows(self.handle))

    @property
    def n_eval_users(self):
        return int(lib().rec_eval_user_count(self.handle))

    def eval_topk(self, users, k=50, want_scores=True):
        """users: an array of user ids, or an int n = the first n users that have a test item."""
        if isinstance(users, (int, np.integer)):
            n, users = int(users), None
        else:
            users = np.ascontiguousarray(users, np.int32)
            n = len(users)
        items = np.zeros((n, 2, k), np.int32)
        scores = np.zeros((n, 2, k), np.float64) if want_scores else None
        _check(lib().rec_eval_topk(self.handle, _p(users), C.c_int32(n), C.c_int32(k), _p(items), _p(scores)))
        return items, scores

    def eval_sums(self, topn=50):
        s = np.zeros(8, np.float64)
        _check(lib().rec_eval_sums(self.handle, C.c_int(topn), _p(s)))
        return s

    def tables(self):
        n = lib().rec_num_tables(self.handle)
        out = []
        for t in range(n):
            ptr, rows, cols, isu = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int()
            _check(lib().rec_table_info(self.handle, C.c_int(t), C.byref(ptr), C.byref(rows), C.byref(cols), C.byref(isu)))
            out.append((ptr.value, rows.value, cols.value, bool(isu.value)))
        return out

    def set_option(self, name, value):
        _check(lib().rec_set_option(self.handle, name.encode(), C.c_int64(value)))

    def kernel_stats(self, which):
        ms, n = C.c_double(), C.c_int64()
        _check(lib().rec_kernel_stats(self.handle, C.c_int(which), C.byref(ms), C.byref(n)))
        return ms.value, n.value
