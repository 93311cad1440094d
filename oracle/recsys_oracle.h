/* CPU oracle: a plain-C, double-precision restatement of the reference's BPR-SGD step and
 * full-catalog evaluation.  TEST INFRASTRUCTURE ONLY: it may be loaded by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product path.
 * Pinned against the real reference by tests/golden/ (see oracle/make_golden.py). */
#ifndef RECSYS_ORACLE_H
#define RECSYS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum orc_model {
    ORC_MP = 0, ORC_BPRMF = 1, ORC_MC = 2, ORC_FPMC = 3, ORC_PRME = 4, ORC_HRM_AVG = 5,
    ORC_HRM_MAX = 6, ORC_TRANSREC = 7, ORC_TRANSREC_L1 = 8, ORC_FOSSIL_SIMPLE = 9,
    ORC_FOSSIL = 10, ORC_REBUS = 11
};

typedef struct {
    int32_t model;
    int32_t n_users, n_items;
    int32_t K, KK, L;
    double lambda, bias_reg, relation_reg, alpha;
    int32_t dataset_factor;       /* Fossil.hpp:22, 10 */
    /* post-split training sequences (chronological) */
    const int64_t* seq_off;       /* [U+1] */
    const int32_t* seq_item;      /* [E]   */
    const int32_t* test_item;     /* [U], -1 = user has no val/test */
    const int32_t* val_item;      /* [U] */
    /* REBUS only: matched path per context.  Context of user u with prefix length c
     * (c = 1 .. n_u+1) has id seq_off[u] + 2*u + (c-1); c = n_u is the validation context,
     * c = n_u+1 the test context (prefix = train sequence + validation item). */
    const int64_t* path_off;      /* [E + 2U + 1] */
    const int32_t* path_item;
    const double* eta;            /* (L+1) x L, eta[n*L + r], REBUS.cpp:112-120 */
} orc_ctx;

int64_t orc_nw(const orc_ctx* c);

/* model.hpp:22-45.  In: per-user click CSR sorted by time.  Out: training CSR + test/val items. */
void orc_split(int32_t n_users, const int64_t* click_off, const int32_t* click_item,
               int64_t* seq_off, int32_t* seq_item, int32_t* test_item, int32_t* val_item);

/* One updateFactors call per tuple, in order.  Tuple meaning per model:
 *   BPRMF (u,pos,neg,-)  MC (i,pos,neg,-)  FPMC/PRME/FossilSimple/HRM (u,prev,pos,neg)
 *   TransRec[_L1] (u,x,y,yn)  Fossil (u,t,neg,-) target = seq[u][t]
 *   REBUS (u,t-1,neg,-) target = seq[u][t]. */
int orc_replay(const orc_ctx* c, double* W, const int32_t* tuples, int64_t n, double lr);

/* Scores of the positives (x_test, x_val) and (count_val, count_test, max) per user, model.cpp:16-59
 * and the Fossil / REBUS overrides.  Invalid users get zeros / -1. nthreads <= 0: all cores. */
int orc_eval(const orc_ctx* c, const double* W, double* xpos, int32_t* counts, int nthreads);

/* The scores model.cpp:36-47 compares for user u in the validation (is_test = 0) or test context:
 * row[j] for every candidate j and for the context's positive, -inf for the excluded items. */
int orc_score_row(const orc_ctx* c, const double* W, int u, int is_test, double* row);

/* counts -> AUC_val AUC_test HIT_val HIT_test MRR_val MRR_test var, model.cpp:49-97 */
void orc_metrics(int32_t n_users, const int32_t* counts, int topn, double* out7);

/* REBUS.cpp:112-120 with common.hpp:92-110.  eta must hold (L+1)*L doubles. */
void orc_eta_table(int L, double* eta);

/* REBUS::findPathStars (REBUS.cpp:484-524) for every context of every user.  Labels are given as
 * integer sequences (oldest item first): lab_off[n_labels+1], lab_item[].  "Root" is not a label
 * here.  Returns the number of path items written (call with path_item = NULL to size). */
int64_t orc_rebus_paths(int32_t n_users, int32_t n_items, const int64_t* seq_off,
                        const int32_t* seq_item, const int32_t* val_item,
                        int64_t n_labels, const int64_t* lab_off, const int32_t* lab_item,
                        int64_t* path_off, int32_t* path_item);

/* Single-thread BPRMF epoch with the reference's sampler shape (BPRMF.cpp:107-150): swap-pop
 * positives, rejection-sampled negatives.  Used only as bench.py's cpu_baseline. Returns steps. */
int64_t orc_bprmf_epoch(const orc_ctx* c, double* W, double lr, int64_t n_steps, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
