"""Host logic of the multi-rank path (one process per GPU): user sharding, the combination of
per-rank item tables after a training step, and metrics from all-reduced evaluation sums.
Pure functions over torch / numpy objects so that they can be exercised with the gloo backend."""
import numpy as np


def shard_range(n_users, rank, world):
    """Contiguous user shard of `rank`: evaluation and sampling only touch users in [begin, end)."""
    return n_users * rank // world, n_users * (rank + 1) // world


def combine_item_tables(tables, bases, world, all_reduce_sum):
    """Every rank started the step from `bases` and applied its own tuples: the combined table is
    base + sum over ranks of (table_r - base) = allreduce_sum(table) - (world - 1) * base, in place."""
    for t, base in zip(tables, bases):
        all_reduce_sum(t)
        t.sub_(base, alpha=float(world - 1))


def sums_from_counts(counts, topn=50):
    """Per-user (count_val, count_test, max) -> the 8 sums rec_eval_sums returns (model.cpp:49-97)."""
    c = np.asarray(counts)
    ok = c[:, 2] >= 0
    cv, ct, mx = c[ok, 0].astype(np.float64), c[ok, 1].astype(np.float64), c[ok, 2].astype(np.float64)
    at = ct / mx
    rank_v, rank_t = mx - cv + 1, mx - ct + 1
    return np.array([(cv / mx).sum(), at.sum(), (rank_v <= topn).sum(), (rank_t <= topn).sum(), (1.0 / rank_v).sum(),
                     (1.0 / rank_t).sum(), (at * at).sum(), float(ok.sum())])


def metrics_from_sums(s):
    """All-reduced sums -> AUC_val AUC_test HIT_val HIT_test MRR_val MRR_test var (var as E[x^2] - mean^2;
    the reference's two-pass form, model.cpp:91-97, differs from it only by rounding)."""
    n = s[7]
    m = np.array(s[:6]) / n
    return np.concatenate([m, [s[6] / n - m[1] * m[1]]])
