// Full-catalog evaluation (AUC / HIT@N / MRR) as batched scoring.
//
// For every ranked candidate j (an item the user has not clicked, model.cpp:33-37) each model's
// prediction() is   score(q, j) = a_q + b_j + <q', T_j>   with a per-(user, context) query q'
// that does not depend on j, a per-item constant b_j and a static item table T (one or two table
// segments side by side):
//     dot models   a_q = 0,            b_j = beta_j,                   q' = q
//     L2 models    a_q = -sum w q^2,   b_j = -(beta_j + sum w T_j^2),  q' = 2 w q     (||q-T||^2 expanded)
// so the 2*nItems predictions per user become one FP64 MFMA GEMM tile sweep whose epilogue compares
// with the positive's score and counts (strict >, as model.cpp:40,45), instead of nItems*n_u*K work
// per user in Fossil / REBUS.  The positive's own score follows the reference formula literally,
// including its "skip the scored item" rule (Fossil.cpp:90, REBUS.cpp:223).  Clicked items are
// counted by the GEMM like any other column and taken out again by a small correction kernel that
// walks the user's (short) click list.
#include "common.h"
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

struct EvalDev {
    int nq, nq_pad, nctx, KQ, K1, K2;
    int seg0, seg1;            // table indices of the two item-table segments (-1 = none)
    double* Q;                 // [nq_pad][KQ] scaled queries q'
    double* thr;               // [nq_pad] positive's score minus a_q (validation positive when nctx == 1)
    double* thr2;              // [nq_pad] test positive's score minus a_q (nctx == 1 only)
    int32_t* cnt; int32_t* cnt2;  // [nq_pad]
    double* bconst;            // [I]
    int32_t* counts;           // [U][3]
    double* xpos;              // [U][2]
    double* aq;                // [nq_pad] a_q of every query (top-k output only)
    double* S;                 // [nq_pad][I] relative scores b_j + <q', T_j> (top-k only)
    int force2;                // 1: one query per (user, context) even for models whose two contexts share q
    // sampled evaluation: the scored columns are a multiset of items instead of the whole catalog
    const int32_t* col_item;   // [columns] item id behind every scored column (null: column j is item j)
    const double* col_b;       // [columns] b of every scored column (bconst gathered; = bconst when all items are scored)
    const int32_t* item_mult;  // [I] how often an item occurs among the columns (null: once)
    int n_cols;                // number of scored columns
    // f32 screen (score_screen_kernel): per-query margin, and per query the columns it could not decide
    float* qmar;               // [nq_pad] |f32 score - exact score| is below this for every column of the query
    int32_t* amb;              // [nq_pad][amb_cap] undecided columns: item | which thresholds (bits 28, 29)
    int32_t* amb_cnt;          // [nq_pad] how many (may exceed amb_cap: then *amb_overflow is set and the lists are not used)
    int amb_cap;
    int32_t* amb_overflow;
    int heldout_counted;       // 1: the sweep counted the test / validation items like any column (the screen does): taken out afterwards
    // the screen's catalog split: blockIdx.y sweeps items [y * range_items, (y + 1) * range_items); with more than one range
    // the counts and the undecided lists of a query are shared by its blocks (integer atomics on zeroed arrays)
    int ranges, range_items;
};

#define E REC_MAX_EPL

template <class T>
struct QB {
    const DevModel& m; int lane;
    __device__ QB(const DevModel& m_) : m(m_), lane(threadIdx.x & 63) {}
    __device__ double ld(int t, int64_t row, int K, int k) const { return (double)((const T*)m.tab[t])[row * K + k]; }
    __device__ double sc(int t, int64_t i) const { return (double)((const T*)m.tab[t])[i]; }
    __device__ static double wsum(double v) { return group_sum<64>(v); }
};

// one wavefront per query
template <class T>
__global__ void __launch_bounds__(256) build_queries_kernel(DevModel m, EvalDev ev)
{
    const QB<T> B(m);
    const int lane = B.lane;
    const int w = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= ev.nq) return;
    const int u = m.eval_users[w / ev.nctx];
    const int ctx = ev.nctx == 2 ? (w & 1) : 0;     // 0 = validation, 1 = test
    const bool shared_q = m.model == REC_BPRMF || m.model == REC_MP;
    const int64_t so = m.seq_off[u]; const int n = (int)(m.seq_off[u + 1] - so);
    const int32_t* seq = m.seq_item + so;
    const int val = m.val_item[u], test = m.test_item[u];
    const int pos = ctx ? test : val;
    const int prev = ctx ? val : seq[n - 1];         // model.hpp:37-39
    const int K = m.K, KK = m.KK;
    double* q = ev.Q + (int64_t)w * ev.KQ;
    double aq = 0, xa = 0, xb = 0;                   // xa: this context's positive; xb: test positive when nctx == 1

    switch (m.model) {
    case REC_MP: {  // MostPopular.cpp:4-7
        xa = (double)m.train_count[val]; xb = (double)m.train_count[test];
        break;
    }
    case REC_BPRMF: {  // BPRMF.cpp:69-72
        double pv = 0, pt = 0;
        for (int k = lane; k < K; k += 64) {
            const double g = B.ld(BPRMF_GU, u, K, k);
            q[k] = g; pv += g * B.ld(BPRMF_GI, val, K, k); pt += g * B.ld(BPRMF_GI, test, K, k);
        }
        xa = B.sc(BPRMF_BETA, val) + B.wsum(pv); xb = B.sc(BPRMF_BETA, test) + B.wsum(pt);
        break;
    }
    case REC_MC: {  // MC.cpp:75-78
        double p = 0;
        for (int k = lane; k < K; k += 64) { const double g = B.ld(MC_GI, prev, K, k); q[k] = g; p += g * B.ld(MC_ETA, pos, K, k); }
        xa = B.wsum(p);
        break;
    }
    case REC_FPMC: {  // FPMC.cpp:71-75
        double p1 = 0, p2 = 0;
        for (int k = lane; k < K; k += 64) { const double g = B.ld(FPMC_GU, u, K, k); q[k] = g; p1 += g * B.ld(FPMC_GI, pos, K, k); }
        for (int k = lane; k < KK; k += 64) { const double g = B.ld(FPMC_KAPPA, prev, KK, k); q[K + k] = g; p2 += g * B.ld(FPMC_ETA, pos, KK, k); }
        xa = B.wsum(p1) + B.wsum(p2);
        break;
    }
    case REC_PRME: {  // PRME.cpp:64-69
        const double al = m.alpha;
        double n1 = 0, n2 = 0, d1 = 0, d2 = 0;
        for (int k = lane; k < K; k += 64) {
            const double g = B.ld(PRME_GU, u, K, k), t = B.ld(PRME_GI, pos, K, k);
            q[k] = 2 * al * g; n1 += g * g; d1 += (g - t) * (g - t);
        }
        for (int k = lane; k < KK; k += 64) {
            const double g = B.ld(PRME_ETA, prev, KK, k), t = B.ld(PRME_ETA, pos, KK, k);
            q[K + k] = 2 * (1 - al) * g; n2 += g * g; d2 += (g - t) * (g - t);
        }
        aq = -(al * B.wsum(n1) + (1 - al) * B.wsum(n2));
        xa = -(al * B.wsum(d1) + (1 - al) * B.wsum(d2));
        break;
    }
    case REC_HRM_AVG: case REC_HRM_MAX: {  // HRM_avg.cpp:64-71, HRM_max.cpp:64-71
        double p = 0;
        for (int k = lane; k < K; k += 64) {
            const double a = B.ld(HRM_U, u, K, k), b = B.ld(HRM_V, prev, K, k);
            const double h = m.model == REC_HRM_MAX ? (a > b ? a : b) : (a + b);
            q[k] = h; p += h * B.ld(HRM_V, pos, K, k);
        }
        xa = B.wsum(p);
        break;
    }
    case REC_TRANSREC: {  // TransRec.cpp:85-92
        double nq = 0, d = 0;
        for (int k = lane; k < K; k += 64) {
            const double g = B.ld(TR_H, prev, K, k) + B.sc(TR_r, k) + B.ld(TR_R, u, K, k);
            const double e = g - B.ld(TR_H, pos, K, k);
            q[k] = 2 * g; nq += g * g; d += e * e;
        }
        aq = -B.wsum(nq);
        xa = -(B.sc(TR_BETA, pos) + B.wsum(d));
        break;
    }
    case REC_TRANSREC_L1: {  // TransRec_L1.cpp:85-92; scored by the |.|_1 kernel, so q is kept unscaled
        // The positive's score is summed exactly as the reference sums it (from beta, k ascending), and so is every score
        // it is compared with (l1_seq_score, score_count_l1_kernel): sums of |differences| of parameters read from a
        // six-decimal model file are multiples of 1e-6 and TIE with the positive's a few times per thousand queries, and a
        // tie must compare equal, as it does in the reference, not by the last bit of two summation orders.
        double pred = B.sc(TR_BETA, pos);
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            double term = 0;
            if (k < K) {
                const double g = B.ld(TR_H, prev, K, k) + B.sc(TR_r, k) + B.ld(TR_R, u, K, k);
                q[k] = g; term = fabs(g - B.ld(TR_H, pos, K, k));
            }
            const int n = min(64, K - k0);
            for (int l = 0; l < n; l++) pred += __shfl(term, l, 64);
        }
        xa = -pred;
        break;
    }
    case REC_FOSSIL_SIMPLE: case REC_FOSSIL: {
        // FossilSimple.cpp:82-103 through model::AUC; Fossil.cpp:81-117 with the contexts of Fossil.cpp:389-404
        const bool simple = m.model == REC_FOSSIL_SIMPLE;
        const int tUi = simple ? FS_UI : FO_UI, tV = simple ? FS_V : FO_V, tB = simple ? FS_BETA : FO_BETA;
        double sall[E], sex[E];
#pragma unroll
        for (int j = 0; j < E; j++) sall[j] = sex[j] = 0;
        int cex = 0;
        // eight positions per round: their ids in one load (lane q fetches position t0 + q), their rows requested together,
        // then added in the order of the sequence - one position at a time cost two memory round trips per row
        const T* tabUi = (const T*)m.tab[tUi];
        for (int t0 = 0; t0 < n; t0 += 8) {
            const int idv = seq[t0 + lane < n ? t0 + lane : n - 1];
            int its[8]; T x[8][E];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                its[q] = __builtin_amdgcn_readlane(idv, q);
#pragma unroll
                for (int j = 0; j < E; j++) { const int k = lane + 64 * j; x[q][j] = tabUi[(int64_t)its[q] * K + (k < K ? k : K - 1)]; }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (t0 + q >= n) break;
                const int it = its[q];
                if (it != pos) cex++;
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const int k = lane + 64 * j;
                    if (k < K) { const double xv = (double)x[q][j]; sall[j] += xv; if (it != pos) sex[j] += xv; }
                }
            }
        }
        const double wall = n > 0 ? pow((double)n, -0.2) : 0, wex = cex > 0 ? pow((double)cex, -0.2) : 0;
        if (simple) {
            const double a0 = B.sc(FS_ALPHA, 0) + B.sc(FS_ALPHA_U, u);
            double p1 = 0, p2 = 0;
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int k = lane + 64 * j;
                if (k < K) {
                    const double up = B.ld(tUi, prev, K, k), vp = B.ld(tV, pos, K, k);
                    q[k] = wall * sall[j] + a0 * up;
                    p1 += sex[j] * vp; p2 += up * vp;
                }
            }
            xa = B.sc(tB, pos); xa += wex * B.wsum(p1); xa += a0 * B.wsum(p2);
        } else {
            const int L = m.L;
            const int np = ctx ? 1 + (L - 1 < n ? L - 1 : n) : (L < n ? L : n);
            double sq[E];
#pragma unroll
            for (int j = 0; j < E; j++) { sq[j] = wall * sall[j]; sex[j] *= wex; }
            for (int ind = 0; ind < np; ind++) {
                const int c = ctx ? (ind == 0 ? val : seq[n - ind]) : seq[n - 1 - ind];
                const double wgt = B.sc(FO_WT, ind) + B.sc(FO_WTU, (int64_t)u * L + ind);
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const int k = lane + 64 * j;
                    if (k < K) { const double x = wgt * B.ld(tUi, c, K, k); sq[j] += x; if (c != pos) sex[j] += x; }
                }
            }
            double p = 0;
#pragma unroll
            for (int j = 0; j < E; j++) { const int k = lane + 64 * j; if (k < K) { q[k] = sq[j]; p += sex[j] * B.ld(tV, pos, K, k); } }
            xa = B.sc(tB, pos) + B.wsum(p);
        }
        break;
    }
    case REC_REBUS: {  // REBUS.cpp:205-269 with the contexts of REBUS.cpp:758-782
        const int64_t cid = so + 2 * (int64_t)u + (n - 1) + ctx;
        const int32_t* path = m.path_item + m.path_off[cid]; const int np = (int)(m.path_off[cid + 1] - m.path_off[cid]);
        const double* etas = m.eta + (int64_t)np * m.L;
        // the test prefix also holds the validation item; it joins the set when it is new
        const bool extra = ctx && !sorted_contains(m.clk_item + m.clk_off[u], (int)(m.clk_off[u + 1] - m.clk_off[u]), val);
        double sall[E], sex[E], ss[E];
#pragma unroll
        for (int j = 0; j < E; j++) sall[j] = sex[j] = ss[j] = 0;
        int call = 0, cex = 0;
        // the first occurrences of the sequence from the packed list the training kernel walks ({item, position} in the order
        // of the sequence, {0, INT_MAX} after the last one), eight per round: ids in one load (lane q fetches entry t0 + q),
        // rows requested together, added in the order of the sequence; then the validation item where it joins the set
        const T* tabP = (const T*)m.tab[RB_P];
        const int2* kept = (const int2*)(m.kept_pair + 2 * so);
        bool more = true;
        for (int t0 = 0; more && t0 < n; t0 += 8) {
            const int2 ev2 = kept[t0 + lane < n ? t0 + lane : n - 1];
            const int idv = (t0 + lane < n && ev2.y != 0x7fffffff) ? ev2.x : -1;
            int its[8]; T x[8][E];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                its[q] = __builtin_amdgcn_readlane(idv, q);
#pragma unroll
                for (int j = 0; j < E; j++) { const int k = lane + 64 * j; x[q][j] = tabP[(int64_t)(its[q] > 0 ? its[q] : 0) * K + (k < K ? k : K - 1)]; }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int it = its[q];
                if (it < 0) { more = false; break; }
                call++; if (it != pos) cex++;
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const int k = lane + 64 * j;
                    if (k < K) { const double xv = (double)x[q][j]; sall[j] += xv; if (it != pos) sex[j] += xv; }
                }
            }
        }
        if (extra) {
            call++; if (val != pos) cex++;
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int k = lane + 64 * j;
                if (k < K) { const double xv = B.ld(RB_P, val, K, k); sall[j] += xv; if (val != pos) sex[j] += xv; }
            }
        }
        const double wall = call > 0 ? pow((double)call, m.alpha) : 0, wex = cex > 0 ? pow((double)cex, m.alpha) : 0;
        for (int r = 0; r < np; r++) {
#pragma unroll
            for (int j = 0; j < E; j++) { const int k = lane + 64 * j; if (k < K) ss[j] += etas[r] * B.ld(RB_P, path[r], K, k); }
        }
        double nq = 0, d2 = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int k = lane + 64 * j;
            if (k < K) {
                const double g = wall * sall[j] + ss[j];
                const double d = wex * sex[j] + ss[j] - B.ld(RB_P, pos, K, k);
                q[k] = 2 * g; nq += g * g; d2 += d * d;
            }
        }
        aq = -B.wsum(nq);
        xa = -(B.sc(RB_BETA, pos) + B.wsum(d2));
        break;
    }
    default: break;
    }
    if (shared_q && ev.nctx == 2 && ctx == 1) xa = xb;
    if (lane == 0) {
        if (ev.aq) ev.aq[w] = aq;
        ev.thr[w] = xa - aq;
        if (ev.nctx == 1) {
            ev.thr2[w] = xb - aq;
            ev.xpos[2 * (int64_t)u] = xb; ev.xpos[2 * (int64_t)u + 1] = xa;
        } else {
            ev.xpos[2 * (int64_t)u + (ctx ? 0 : 1)] = xa;
        }
    }
}

// b_j, one wavefront per item
template <class T>
__global__ void __launch_bounds__(256) item_consts_kernel(DevModel m, EvalDev ev)
{
    const QB<T> B(m);
    const int lane = B.lane;
    const int j = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (j >= m.I) return;
    const int K = m.K, KK = m.KK;
    double b = 0;
    switch (m.model) {
    case REC_MP: b = (double)m.train_count[j]; break;
    case REC_BPRMF: b = B.sc(BPRMF_BETA, j); break;
    case REC_TRANSREC_L1: b = B.sc(TR_BETA, j); break;
    case REC_FOSSIL_SIMPLE: b = B.sc(FS_BETA, j); break;
    case REC_FOSSIL: b = B.sc(FO_BETA, j); break;
    case REC_PRME: {
        double s1 = 0, s2 = 0;
        for (int k = lane; k < K; k += 64) { const double t = B.ld(PRME_GI, j, K, k); s1 += t * t; }
        for (int k = lane; k < KK; k += 64) { const double t = B.ld(PRME_ETA, j, KK, k); s2 += t * t; }
        b = -(m.alpha * B.wsum(s1) + (1 - m.alpha) * B.wsum(s2));
        break;
    }
    case REC_TRANSREC: {
        double s = 0;
        for (int k = lane; k < K; k += 64) { const double t = B.ld(TR_H, j, K, k); s += t * t; }
        b = -(B.sc(TR_BETA, j) + B.wsum(s));
        break;
    }
    case REC_REBUS: {
        double s = 0;
        for (int k = lane; k < K; k += 64) { const double t = B.ld(RB_P, j, K, k); s += t * t; }
        b = -(B.sc(RB_BETA, j) + B.wsum(s));
        break;
    }
    default: break;
    }
    if (lane == 0) ev.bconst[j] = b;
}

// ------------------------------------------------------------------------------------------------
// Scoring + counting.  A block of 4 wavefronts owns 4*MT*16 queries and sweeps the whole item
// table once: tiles of NI items are staged through LDS (converted to f64, rows padded by two
// doubles so that the ds_read_b64 fragment reads of a 32-lane group fall on 64 distinct banks), the
// next tile's global loads are issued before the current tile's MFMAs, and every wavefront keeps its
// A fragments (the queries) in registers for the whole sweep.
// v_mfma_f64_16x16x4_f64 operand maps: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[i = (lane >> 4) + 4 r][j = lane & 15] for result register r.
template <int KQ> struct TileShape {
    static constexpr int NI = KQ <= 64 ? 64 : KQ <= 128 ? 32 : 16;
    static constexpr int MT = KQ <= 32 ? 4 : KQ <= 64 ? 2 : 1;
};

template <class T, int KQ, bool TWO, bool STORE>
__global__ void __launch_bounds__(256, 2) score_count_kernel(EvalDev ev, const T* __restrict__ seg0, const T* __restrict__ seg1,
                                                             const int32_t* __restrict__ qtest, const int32_t* __restrict__ qval, int I)
{
    constexpr int KC = KQ / 4, NI = TileShape<KQ>::NI, MT = TileShape<KQ>::MT, STR = KQ + 2;
    constexpr int PER = NI * KQ / 256;  // tile elements per thread
    constexpr int QB = 4 * MT * 16;     // queries per block
    __shared__ double tile[NI * STR];
    __shared__ double tile_b[NI];       // b_j of the tile's items
    __shared__ int tile_id[NI];         // their item ids (the column index unless a subset is scored)
    // per-query constants live in LDS, not in registers: the epilogue that reads them runs once per 16 * MT MFMAs,
    // and the registers they would occupy are what allows two wavefronts per SIMD
    __shared__ double2 q_thr[QB];       // (threshold, second threshold)
    __shared__ int2 q_ex[QB];           // (test item, validation item): never counted
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    const int64_t qblock = (int64_t)blockIdx.x * QB;
    const int qw = wave * (MT * 16);    // this wavefront's first query inside the block
    const int K1 = ev.K1, K2 = ev.K2;

    for (int i = tid; i < QB; i += 256) {
        q_thr[i] = make_double2(ev.thr[qblock + i], TWO ? ev.thr2[qblock + i] : 0.0);
        q_ex[i] = make_int2(qtest[qblock + i], qval[qblock + i]);
    }

    double a[MT][KC];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int kc = 0; kc < KC; kc++) a[mt][kc] = ev.Q[(qblock + qw + mt * 16 + col) * KQ + 4 * kc + quad];

    int cnt[MT][4], cnt2[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) { cnt[mt][r] = 0; cnt2[mt][r] = 0; }

    const bool dense = K1 == KQ;        // one table segment that fills the query width: a tile is one contiguous run
    auto fetch = [&](int item0, T (&pre)[PER], double& preb, int& preid) {
        preid = (tid < NI && item0 + tid < I) ? (ev.col_item ? ev.col_item[item0 + tid] : item0 + tid) : -2;
        if (dense && item0 + NI <= I) {
            const T* src = seg0 + (int64_t)item0 * KQ + tid;
#pragma unroll
            for (int p = 0; p < PER; p++) pre[p] = src[256 * p];
            preb = tid < NI ? ev.col_b[item0 + tid] : 0.0;
            return;
        }
        preb = (tid < NI && item0 + tid < I) ? ev.col_b[item0 + tid] : 0.0;
        if (K1 + K2 == 0) {   // no vectors at all (MostPopular): nothing to clamp into
#pragma unroll
            for (int p = 0; p < PER; p++) pre[p] = (T)0;
            return;
        }
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const int idx = tid + 256 * p, it = idx / KQ, k = idx % KQ;
            const int64_t j = (int64_t)item0 + it;
            // one unconditional load per element from an address clamped into the tables, the value selected afterwards
            // (a load under a branch would be waited for before the next one is issued)
            const bool ok = j < I && k < K1 + K2;
            const int64_t jc = j < I ? j : (int64_t)I - 1;
            const int kk = k < K1 + K2 ? k : 0;
            const T* src = kk < K1 ? seg0 + jc * K1 + kk : seg1 + jc * K2 + (kk - K1);
            const T v = *src;
            pre[p] = ok ? v : (T)0;
        }
    };

    T pre[PER]; double preb; int preid;
    fetch(0, pre, preb, preid);
    for (int item0 = 0; item0 < I; item0 += NI) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const int idx = tid + 256 * p, it = idx / KQ, k = idx % KQ;
            tile[it * STR + k] = (double)pre[p];
        }
        if (tid < NI) { tile_b[tid] = preb; tile_id[tid] = preid; }
        __syncthreads();
        if (item0 + NI < I) fetch(item0 + NI, pre, preb, preid);
#pragma unroll 1
        for (int sub = 0; sub < NI / 16; sub++) {
            const int jj = sub * 16 + col;
            const int j = item0 + jj;
            const bool live = j < I;
            double4_t acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kc = 0; kc < KC; kc++) {
                const double b = tile[jj * STR + 4 * kc + quad];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt][kc], b, acc[mt], 0, 0, 0);
            }
            const double bj = tile_b[jj];
            const int jid = tile_id[jj];
            if (STORE) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (live) ev.S[(qblock + qw + mt * 16 + quad + 4 * r) * (int64_t)I + j] = acc[mt][r] + bj;
                continue;
            }
            // all constants first, then straight-line predicate arithmetic: no LDS round trip per element
            double2 th[MT][4]; int2 ex[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) { const int qi = qw + mt * 16 + quad + 4 * r; th[mt][r] = q_thr[qi]; ex[mt][r] = q_ex[qi]; }
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double s = acc[mt][r] + bj;
                    const int ok = (int)live & (int)(jid != ex[mt][r].x) & (int)(jid != ex[mt][r].y);
                    cnt[mt][r] += ok & (int)(th[mt][r].x > s);
                    if (TWO) cnt2[mt][r] += ok & (int)(th[mt][r].y > s);
                }
        }
    }
    if (STORE) return;
    // the 16 lanes that share a quad hold partial counts of the same 4 rows
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int c = cnt[mt][r], c2 = cnt2[mt][r];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); if (TWO) c2 += __shfl_xor(c2, o, 64); }
            if (col == 0) {
                const int64_t qi = qblock + qw + mt * 16 + quad + 4 * r;
                ev.cnt[qi] = c;
                if (TWO) ev.cnt2[qi] = c2;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// f32 screen.  When the tables are f32, the sweep above spends its time in FP64 MFMAs on numbers that were f32 to begin
// with.  This kernel runs the same sweep on the FP32 matrix pipe (twice the rate, half the LDS bytes) and decides every
// comparison it can PROVE: with u = 2^-24, q32 = fl(q'), and any order of accumulation,
//     |fl32(b_j + <q32, T_j>) - (b_j + <q', T_j>)|  <=  (KQ + 2) u ||q'|| ||T_j||  +  u (|b_j| + |score|)
// (the FP32 MFMA accumulates like a chain of fmaf, one rounding to nearest per term), so a column whose f32 score
// differs from the f32 threshold by more than the query's margin
//     M_q = 1.05 (KQ + 2) u ||q'|| Tmax + 4 u (Bmax + |thr| + ||q'|| Tmax)        (Tmax = max ||T_j||, Bmax = max |b_j|)
// is counted, or not, for certain.  The others - a few of every 100k columns per query - go to the query's list and are
// decided in fp64 by correct_clicked_kernel with the very summation it uses for the clicked items, so the counts are
// those of the fp64 sweep.
// v_mfma_f32_16x16x4_f32 operand maps: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[i = 4 (lane >> 4) + r][j = lane & 15] for result register r.  The contraction index is permuted so that quad
// (lane >> 4) owns k = quad * KC .. quad * KC + KC - 1: its B values of an item are contiguous and come as ds_read_b128.
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int KQ, bool DENSE = false> struct ScreenShape {
    // items per tile; twice as many when the table segments fill the query width in 16-byte pieces (fetched without the
    // per-element address arithmetic of the general case, whose registers the larger tile needs)
    static constexpr int NI = (KQ <= 64 ? 64 : KQ <= 128 ? 32 : 16) * (DENSE ? 2 : 1);
    static constexpr int MT = KQ <= 16 ? 4 : KQ <= 128 ? 2 : 1;       // 16-query tiles per wavefront
};

template <int KQ, bool TWO, bool DENSE>
__global__ void __launch_bounds__(256, DENSE && KQ <= 64 ? 3 : 2) score_screen_kernel(EvalDev ev, const float* __restrict__ seg0, const float* __restrict__ seg1,
                                                              const int32_t* __restrict__ qtest, const int32_t* __restrict__ qval, int I)
{
    constexpr int KC = KQ / 4, NI = ScreenShape<KQ, DENSE>::NI, MT = ScreenShape<KQ, DENSE>::MT, STR = KQ + 4;
    constexpr int PER4 = NI * KQ / 4 / 256;   // float4 tile elements per thread
    constexpr int QB = 4 * MT * 16;           // queries per block
    static_assert(PER4 >= 1 && KC % 4 == 0, "tile shape");
    __shared__ __align__(16) float tile[NI * STR];
    __shared__ float tile_b[NI];
    __shared__ int q_amb[QB];                 // undecided columns of the query so far
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    const int64_t qblock = (int64_t)blockIdx.x * QB;
    const int qw = wave * (MT * 16);
    const int K1 = ev.K1, K2 = ev.K2;

    for (int i = tid; i < QB; i += 256) q_amb[i] = 0;
    // thresholds and margin of this lane's 4 MT result rows (held-out items are counted like any column here and taken
    // out by correct_clicked_kernel: no per-element exclusion test)
    // (one margin for the wavefront, the largest of its queries': a scalar instead of a register per result row)
    float th1[MT][4], th2[MT][4], marw = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t qi = qblock + qw + mt * 16 + 4 * quad + r;
            th1[mt][r] = (float)ev.thr[qi]; th2[mt][r] = TWO ? (float)ev.thr2[qi] : 0.f; marw = fmaxf(marw, ev.qmar[qi]);
        }
    // a NaN margin (NaN in the tables) must stay NaN - every comparison false, as in the fp64 sweep - so: max, not fmax
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float x = __shfl_xor(marw, o, 64); marw = (x > marw || x != x) ? x : marw; }
    const float mar_w = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(marw)));

    float a[MT][KC];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int kc = 0; kc < KC; kc++) a[mt][kc] = (float)ev.Q[(qblock + qw + mt * 16 + col) * KQ + quad * KC + kc];

    int cnt[MT][4], cnt2[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) { cnt[mt][r] = 0; cnt2[mt][r] = 0; }

    auto fetch = [&](int item0, float4_t (&pre)[PER4], float& preb) {
        if constexpr (DENSE) {
            // K1 + K2 == KQ, both multiples of 4: a row is one run of 16-byte pieces (or one per table segment); the last
            // tile reads its missing rows from the tables' last row (never counted)
#pragma unroll
            for (int p = 0; p < PER4; p++) {
                const int e4 = tid + 256 * p, it = (4 * e4) / KQ, k = (4 * e4) % KQ;
                const int64_t j = (int64_t)item0 + it < I ? (int64_t)item0 + it : (int64_t)I - 1;
                const float* src = k < K1 ? seg0 + j * K1 + k : seg1 + j * K2 + (k - K1);
                pre[p] = *reinterpret_cast<const float4_t*>(src);
            }
            preb = (tid < NI && item0 + tid < I) ? (float)ev.col_b[item0 + tid] : INFINITY;   // a column past the end scores +inf: never counted, never within a margin
            return;
        }
        preb = (tid < NI && item0 + tid < I) ? (float)ev.col_b[item0 + tid] : INFINITY;   // a column past the end scores +inf: never counted, never within a margin
#pragma unroll
        for (int p = 0; p < PER4; p++) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int idx = 4 * (tid + 256 * p) + c, it = idx / KQ, k = idx % KQ;
                const int64_t j = (int64_t)item0 + it;
                const bool ok = j < I && k < K1 + K2;
                const int64_t jc = j < I ? j : (int64_t)I - 1;
                const int kk = k < K1 + K2 ? k : 0;
                float x = 0.f;
                if (K1 + K2 > 0) x = kk < K1 ? seg0[jc * K1 + kk] : seg1[jc * K2 + (kk - K1)];
                v[c] = ok ? x : 0.f;
            }
            pre[p] = (float4_t){v[0], v[1], v[2], v[3]};
        }
    };

    // this block's part of the catalog (a whole number of tiles; the last range ends with the catalog)
    const int i_begin = (int)blockIdx.y * ev.range_items;
    const int i_end = (int)min((int64_t)I, (int64_t)i_begin + ev.range_items);
    const bool shared_lists = ev.ranges > 1;
    float4_t pre[PER4]; float preb;
    fetch(i_begin, pre, preb);
    for (int item0 = i_begin; item0 < i_end; item0 += NI) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PER4; p++) {
            const int idx = 4 * (tid + 256 * p), it = idx / KQ, k = idx % KQ;
            *reinterpret_cast<float4_t*>(&tile[it * STR + k]) = pre[p];
        }
        if (tid < NI) tile_b[tid] = preb;
        __syncthreads();
        if (item0 + NI < i_end) fetch(item0 + NI, pre, preb);
#pragma unroll 1
        for (int sub = 0; sub < NI / 16; sub++) {
            const int jj = sub * 16 + col;
            const int j = item0 + jj;
            float4_t acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c4 = 0; c4 < KC / 4; c4++) {
                const float4_t b4 = *reinterpret_cast<const float4_t*>(&tile[jj * STR + quad * KC + 4 * c4]);
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][4 * c4 + c], b4[c], acc[mt], 0, 0, 0);
            }
            const float bj = tile_b[jj];
            // hot path: counts, and the smallest distance of any of this lane's scores to a threshold
            float mind = INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float sc = acc[mt][r] + bj;
                    const float d1 = th1[mt][r] - sc;
                    cnt[mt][r] += (int)(d1 > mar_w);
                    if (TWO) {
                        const float d2 = th2[mt][r] - sc;
                        cnt2[mt][r] += (int)(d2 > mar_w);
                        mind = __builtin_fminf(mind, __builtin_fminf(__builtin_fabsf(d1), __builtin_fabsf(d2)));
                    } else {
                        mind = __builtin_fminf(mind, __builtin_fabsf(d1));
                    }
                }
            if (!__ballot(mind <= mar_w)) continue;
            // some score of the wavefront lies within the margin of a threshold (one 16-item step in ten): which ones
            unsigned open1 = 0, open2 = 0;   // bit (4 mt + r): that query's (second) threshold is within the margin of this column's score
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float sc = acc[mt][r] + bj;
                    open1 |= (unsigned)(__builtin_fabsf(th1[mt][r] - sc) <= mar_w) << (4 * mt + r);
                    if (TWO) open2 |= (unsigned)(__builtin_fabsf(th2[mt][r] - sc) <= mar_w) << (4 * mt + r);
                }
            // the rare ones: a compact loop, not unrolled
            unsigned open = open1 | open2;
            while (open) {
                const int e = __ffs(open) - 1;
                open &= open - 1;
                const int qi = qw + (e >> 2) * 16 + 4 * quad + (e & 3);
                const int which = (int)((open1 >> e) & 1u) | ((int)((open2 >> e) & 1u) << 1);
                int at;
                if (shared_lists) {
                    at = atomicAdd(&ev.amb_cnt[qblock + qi], 1);
                    if (at >= ev.amb_cap) *ev.amb_overflow = 1;
                } else {
                    at = atomicAdd(&q_amb[qi], 1);
                }
                if (at < ev.amb_cap) ev.amb[(qblock + qi) * (int64_t)ev.amb_cap + at] = j | (which << 28);
            }
        }
    }
    __syncthreads();
    if (!shared_lists)
        for (int i = tid; i < QB; i += 256) {
            ev.amb_cnt[qblock + i] = q_amb[i];
            if (q_amb[i] > ev.amb_cap) *ev.amb_overflow = 1;
        }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int c = cnt[mt][r], c2 = cnt2[mt][r];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); if (TWO) c2 += __shfl_xor(c2, o, 64); }
            if (col == 0) {
                const int64_t qi = qblock + qw + mt * 16 + 4 * quad + r;
                if (shared_lists) {
                    atomicAdd(&ev.cnt[qi], c);
                    if (TWO) atomicAdd(&ev.cnt2[qi], c2);
                } else {
                    ev.cnt[qi] = c;
                    if (TWO) ev.cnt2[qi] = c2;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// The f16 screen: the same provable sweep on the 16-bit matrix pipe (v_mfma_f32_16x16x32_f16: 16 times the multiply-adds
// of the FP32 form per cycle).  Every operand is split into two halves of 11 significant bits each,
//     x * scale = h + l + e,   h = f16(x * scale),  l = f16(x * scale - h),   |e| <= 2^-22 |x * scale| + 2^-25  (2^-14 if f16 subnormals were flushed),
// with one power-of-two scale for the item tables and one for the queries that put their largest element in [2^13, 2^14)
// (f16's normal range then holds the low halves of everything within 2^-17 of the largest element; below that the error
// is an absolute one).  Products of two f16 numbers are exact in f32, so
//     <q, T> = sum (qh Th)  +  sum (qh Tl + ql Th)  +  [ ql Tl + eq T + q eT ]
//              main chain      cross chain (2^-11 of the main one)     dropped: <= 3.3 * 2^-22 ||q|| ||T|| + the 2^-25 floors
// is three MFMAs per 32 contraction indices.  One accumulator takes the cross chain FIRST and the main chain after it (b_j
// is added to the result): a main product then passes through at most KQ + 3 roundings (its own chain, b_j, the
// subtraction of the threshold) of relative size 2^-23 at most (twice the unit roundoff: the bound holds whether the
// matrix pipe rounds to nearest or truncates, and for any order of summation inside an instruction), and the cross
// products, 2^-10 of the main ones together, through at most 3 KQ + 3.  (The other order - cross products added to a full
// main sum - would round every one of them relative to the main sum.  Two accumulators and an addition, the first form of
// this kernel, have the same bound and cost a vector instruction per score: 88 -> 83 ms for 1M x 200k.)
// The margin of a query (screen16_queries_kernel) adds these up; a column
// whose score is further than the margin from the threshold is decided for certain, the others go to the fp64 recheck
// exactly as in the f32 screen.  Operand maps: lane l holds A[row l & 15][k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j]
// [col l & 15], j = 0..7; D[row 4 (l >> 4) + r][col l & 15].
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half_t;

template <int KQ, bool TWO> struct Screen16Shape {
    // 16-query tiles per wavefront.  Measured at 1M users x 200k items, K = 64: with two thresholds per query (BPRMF) two
    // tiles at three wavefronts per SIMD run 85 ms against 90 ms for four tiles at two; with one threshold it is 139 against 133
    static constexpr int MT = KQ <= 64 ? (TWO ? 2 : 4) : KQ <= 128 ? 2 : 1;
    static constexpr int WAVES = (KQ <= 64 && TWO) ? 3 : 2;       // wavefronts per SIMD the registers are held to
    static constexpr int NI = KQ <= 128 ? 64 : 32;                // items per tile
    static constexpr int QB = 4 * MT * 16;                        // queries per block
};
static int screen16_qb(int KQ16, bool two) { return KQ16 <= 64 ? (two ? 128 : 256) : KQ16 <= 128 ? 128 : 64; }

// power-of-two scale that puts a largest magnitude m into [2^13, 2^14); 1 when m is 0, infinite or NaN
__device__ __forceinline__ double screen16_scale(unsigned max_bits)
{
    const float m = __uint_as_float(max_bits);
    if (!(m > 0.f) || !(m < INFINITY)) return 1.0;
    int e; (void)frexpf(m, &e);                // m = f 2^e, 0.5 <= f < 1
    int s = 14 - e;
    s = s < -50 ? -50 : s > 50 ? 50 : s;
    return ldexp(1.0, s);
}

// bounds[0] = max ||T_j||, [1] = max |b_j|, [2] = max |T_jk|, [3] = max |q'_k| (bit patterns of non-negative floats)
__global__ void __launch_bounds__(256) screen16_bounds_kernel(EvalDev ev, const float* __restrict__ seg0, const float* __restrict__ seg1, int I,
                                                              unsigned* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
    unsigned nmax = 0, bmax = 0, emax = 0, qmax = 0;
    for (int j = wave; j < I; j += waves) {
        double s = 0; float em = 0.f;
        for (int k = lane; k < ev.K1; k += 64) { const float t = seg0[(int64_t)j * ev.K1 + k]; s += (double)t * t; em = (fabsf(t) > em || t != t) ? fabsf(t) : em; }
        for (int k = lane; k < ev.K2; k += 64) { const float t = seg1[(int64_t)j * ev.K2 + k]; s += (double)t * t; em = (fabsf(t) > em || t != t) ? fabsf(t) : em; }
        s = group_sum<64>(s);
        const float n = __double2float_ru(sqrt(s) * 1.000001), b = __double2float_ru(fabs(ev.bconst[j]) * 1.000001);
        nmax = max(nmax, __float_as_uint(n == n ? n : INFINITY));
        bmax = max(bmax, __float_as_uint(b == b ? b : INFINITY));
        emax = max(emax, __float_as_uint(em == em ? em : INFINITY));
    }
    for (int64_t w = wave; w < ev.nq; w += waves) {
        float qm = 0.f;
        for (int k = lane; k < ev.KQ; k += 64) { const float q = __double2float_ru(fabs(ev.Q[w * ev.KQ + k])); qm = (q > qm || q != q) ? q : qm; }
        qmax = max(qmax, __float_as_uint(qm == qm ? qm : INFINITY));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { emax = max(emax, (unsigned)__shfl_xor((int)emax, o, 64)); qmax = max(qmax, (unsigned)__shfl_xor((int)qmax, o, 64)); }
    if (lane == 0) { atomicMax(out, nmax); atomicMax(out + 1, bmax); atomicMax(out + 2, emax); atomicMax(out + 3, qmax); }
}

// the two f16 halves of x (already scaled)
__device__ __forceinline__ void split16(float x, half_t& h, half_t& l)
{
    h = (half_t)x;                       // round to nearest even
    l = (half_t)(x - (float)h);          // the difference is exact in f32
}

// item planes Th, Tl [I_pad][KQ16] (zero beyond the tables' width and beyond the catalog) and the scaled constants
// bs [I_pad] (past_end beyond the catalog: +inf for the counting sweep - such a column is never counted and never within a
// margin -, -inf for the top-k sweep, which lists what is NOT below its bar)
__global__ void __launch_bounds__(256) screen16_items_kernel(EvalDev ev, const float* __restrict__ seg0, const float* __restrict__ seg1, int I, int I_pad,
                                                             int KQ16, const unsigned* __restrict__ bounds, half_t* __restrict__ Th, half_t* __restrict__ Tl,
                                                             float* __restrict__ bs, int32_t* __restrict__ overflow, float past_end = INFINITY,
                                                             const int32_t* __restrict__ order = nullptr, bool own_margin = false)
{
    const double sT = screen16_scale(bounds[2]), sq = screen16_scale(bounds[3]);
    const float sTf = (float)sT;
    const int64_t n = (int64_t)I_pad * KQ16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / KQ16; const int k = (int)(i % KQ16);
        const int64_t j = (order && c < I) ? order[c] : c;      // the item behind column c
        float x = 0.f;
        if (c < I) { if (k < ev.K1) x = seg0[j * ev.K1 + k]; else if (k < ev.K1 + ev.K2) x = seg1[j * ev.K2 + (k - ev.K1)]; }
        half_t h, l; split16(x * sTf, h, l);   // a power of two: the product is exact
        Th[i] = h; Tl[i] = l;
        if (k == 0) {
            float b = c < I ? (float)(ev.bconst[j] * (sq * sT)) : past_end;
            if (own_margin && c < I) {
                // The constant's own share of a column's error - its rounding to f32 and the rounding of the sum it enters,
                // 3 u |b| at most - is not part of the queries' margins (which would have to carry max |b_j| over the whole
                // catalog: a handful of items with large biases then made every column of every query undecided).  It is
                // added to the constant itself, e = 6 u |b|: a column counts when score + e is below threshold - margin,
                // and is decided when it lies further than margin + 2 e above (the sweep widens its window by 2.03 e per column).
                b = b + 3.6e-7f * fabsf(b);
            }
            bs[c] = b;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the scaled constants must stay far from the end of f32's range
        const double bmax = (double)__uint_as_float(bounds[1]);
        const float tm = __uint_as_float(bounds[0]), em = __uint_as_float(bounds[2]), qm = __uint_as_float(bounds[3]);
        if (!(bmax * (sq * sT) <= 1e30) || !(tm < INFINITY) || !(em < INFINITY) || !(qm < INFINITY)) overflow[own_margin ? 1 : 0] = 1;
    }
}

// query planes Qh, Ql [nq_pad][KQ16], the scaled thresholds and the scaled margin of every query; one wavefront per query
__global__ void __launch_bounds__(256) screen16_queries_kernel(EvalDev ev, int KQ16, const unsigned* __restrict__ bounds, half_t* __restrict__ Qh,
                                                               half_t* __restrict__ Ql, float* __restrict__ thr_s, float* __restrict__ thr2_s,
                                                               int32_t* __restrict__ overflow)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= ev.nq_pad) return;
    const double sT = screen16_scale(bounds[2]), sq = screen16_scale(bounds[3]), S = sq * sT;
    double s = 0;
    for (int k = lane; k < KQ16; k += 64) {
        const double q = k < ev.KQ ? ev.Q[w * ev.KQ + k] : 0.0;
        s += q * q;
        half_t h, l; split16((float)(q * sq), h, l);
        Qh[w * KQ16 + k] = h; Ql[w * KQ16 + k] = l;
    }
    s = sqrt(group_sum<64>(s));
    if (lane == 0) {
        const double tmax = (double)__uint_as_float(bounds[0]), bmax = (double)__uint_as_float(bounds[1]);
        const double u = 5.9604644775390625e-08;   // 2^-24
        const double t1 = ev.thr[w], t2 = ev.nctx == 1 ? ev.thr2[w] : 0.0;
        const double ta = (isfinite(t1) ? fabs(t1) : 0.0) + (isfinite(t2) ? fabs(t2) : 0.0);
        const double rep = 3.3 * 4.0 * u * s * tmax;                                              // 3.3 * 2^-22: halves dropped or rounded away
        const double sub = 6.103515625e-05 * sqrt((double)KQ16) * (tmax / sq + s / sT);          // 2^-14 per element: holds even if the matrix pipe flushed f16 subnormals (rounding alone: 2^-25)
        const double acc = 2.0 * (KQ16 + 3) * u * 1.002 * s * tmax + KQ16 * (u / 64.0) * s * tmax;  // main products; cross products ((3 KQ + 3) 2 u 2^-10 <= KQ u / 64)
        (void)bmax;                                                                               // the constants carry their own margin (screen16_items_kernel)
        const double rnd = 4.0 * u * (ta + s * tmax);                                             // thresholds rounded to f32, the two final additions, the subtraction
        const double mq = 1.05 * (rep + sub + acc + rnd) + 1e-30;
        ev.qmar[w] = w < ev.nq ? __double2float_ru(mq * S * 1.000001) : 0.f;                       // a padding row decides nothing and lists nothing
        thr_s[w] = (float)(t1 * S);
        thr2_s[w] = (float)(t2 * S);
        // out of f32's comfortable range, or not finite (only a padding row's -inf threshold is): the fp64 sweep takes over
        const bool odd1 = !(isfinite(t1) ? fabs(t1 * S) <= 1e30 : (t1 < 0 && t1 == t1)), odd2 = !(isfinite(t2) ? fabs(t2 * S) <= 1e30 : (t2 < 0 && t2 == t2));
        if (odd1 || odd2 || (w < ev.nq && !(mq * S <= 1e30))) overflow[1] = 1;
    }
}

template <int KQ, bool TWO>
__global__ void __launch_bounds__(256, (Screen16Shape<KQ, TWO>::WAVES)) score_screen16_kernel(EvalDev ev, const half_t* __restrict__ Th, const half_t* __restrict__ Tl, const float* __restrict__ bs,
                                                                const half_t* __restrict__ Qh, const half_t* __restrict__ Ql, const float* __restrict__ thr_s,
                                                                const float* __restrict__ thr2_s, int I)
{
    constexpr int MT = Screen16Shape<KQ, TWO>::MT, NI = Screen16Shape<KQ, TWO>::NI, QB = Screen16Shape<KQ, TWO>::QB;
    constexpr int NCH = KQ / 32;              // MFMA steps of 32 contraction indices
    constexpr int STR = KQ + 8;               // halfs per tile row: KQ / 2 + 4 dwords, conflict-free 16-byte reads by 16 lanes
    constexpr int PER = NI * KQ / 8 / 256;    // 16-byte pieces of a plane's tile per thread
    static_assert(PER >= 1 && KQ % 32 == 0, "tile shape");
    __shared__ __align__(16) half_t tiles[2 * NI * STR];   // the high plane, then the low plane: one address register and immediate offsets reach both
    half_t* const tile_h = tiles;
    half_t* const tile_l = tiles + NI * STR;
    __shared__ float tile_b[NI];
    __shared__ int q_amb[QB];
    __shared__ __align__(16) float q_win[QB];          // every query's own window
    __shared__ int cnt_lds[(TWO ? 2 : 1) * QB * 16];   // counts so far per (threshold, query, column lane): registers are short
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    const int64_t qblock = (int64_t)blockIdx.x * QB;
    const int qw = wave * (MT * 16);

    for (int i = tid; i < QB; i += 256) q_amb[i] = 0;
    for (int i = tid; i < (TWO ? 2 : 1) * QB * 16; i += 256) cnt_lds[i] = 0;
    // One margin for the wavefront (the largest of its queries').  Per result the epilogue forms ONE number,
    //     y = score - (threshold - margin):   y < 0  <=>  the threshold is above the score by more than the margin: counted;
    //                                         0 <= y <= window  <=>  the score is within the margin of the threshold: undecided,
    // and reads both answers from y's bit pattern: the sign bit is shifted into a register of the last 32 signs (v_alignbit;
    // their set bits are added up every 32 sub-steps), and as unsigned integers the patterns of non-negative floats order like
    // the floats while every negative one is larger, so the smallest pattern seen (v_min3_u32) against the window's tells
    // whether any result of the sub-step is undecided.  Equality gives y = +0: not counted, listed - the strict comparison.
    // A padding row's threshold is -inf: y = +inf, neither.  window = 2 margin + what rounding (threshold - margin) and y can
    // add: 2 u (|threshold| + margin) + u window.  (A compare and an add-with-carry per result were a chain through VCC with
    // idle cycles in between, as long as the MFMAs themselves; NaN has no defined sign, so tables or queries that are not
    // finite never get here: the launcher sends them to the fp64 sweep.)
    // Margin and window are per QUERY (thresholds of one wavefront's queries can differ by orders of magnitude, and so do the
    // roundings they bring: one query with a threshold of 0.1 among 63 with thresholds of 1e-6 would otherwise hand its window
    // to all of them).  The hot loop tests the smallest y of a sub-step against the wavefront's WIDEST window (one scalar);
    // the rare sub-step that passes looks up each query's own window (q_win, LDS) before it lists anything.
    float thm1[MT][4], thm2[MT][4], winw = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t qi = qblock + qw + mt * 16 + 4 * quad + r;
            const float t1 = thr_s[qi], t2 = TWO ? thr2_s[qi] : 0.f, mq = ev.qmar[qi];
            float thq = 0.f;
            if (t1 > -INFINITY) thq = fabsf(t1);
            if (TWO && t2 > -INFINITY) thq = fmaxf(thq, fabsf(t2));
            const float wq = (2.f * mq + 1.2e-7f * (thq + mq)) * 1.000001f;
            if (col == 0) q_win[qw + mt * 16 + 4 * quad + r] = wq;
            winw = fmaxf(winw, wq);
            thm1[mt][r] = t1 - mq; thm2[mt][r] = TWO ? t2 - mq : 0.f;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) winw = fmaxf(winw, __shfl_xor(winw, o, 64));
    const float window_q = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(winw)));   // + twice the column's own constant margin: per lane

    half8_t ah[MT][NCH], al[MT][NCH];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            const int64_t at = (qblock + qw + mt * 16 + col) * KQ + ch * 32 + 8 * quad;
            ah[mt][ch] = *reinterpret_cast<const half8_t*>(Qh + at);
            al[mt][ch] = *reinterpret_cast<const half8_t*>(Ql + at);
        }

    unsigned bits1[MT][4], bits2[MT][4];
    int nbits = 0;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) { bits1[mt][r] = 0; bits2[mt][r] = 0; }
    // every (query, column lane) slot of cnt_lds belongs to one lane
    auto add_up = [&](unsigned mask) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = qw + mt * 16 + 4 * quad + r;
                cnt_lds[q * 16 + col] += __popc(bits1[mt][r] & mask);
                if (TWO) cnt_lds[(QB + q) * 16 + col] += __popc(bits2[mt][r] & mask);
            }
    };

    // a tile is one contiguous run of each plane (rows are dense, the planes are padded to whole tiles)
    auto fetch = [&](int item0, half8_t (&ph)[PER], half8_t (&pl)[PER], float& preb) {
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const int64_t at = (int64_t)item0 * KQ + 8 * (int64_t)(tid + 256 * p);
            ph[p] = *reinterpret_cast<const half8_t*>(Th + at);
            pl[p] = *reinterpret_cast<const half8_t*>(Tl + at);
        }
        preb = tid < NI ? bs[item0 + tid] : 0.f;
    };

    const int i_begin = (int)blockIdx.y * ev.range_items;
    const int i_end = (int)min((int64_t)I, (int64_t)i_begin + ev.range_items);
    const bool shared_lists = ev.ranges > 1;
    half8_t ph[PER], pl[PER]; float preb;
    const int frag0 = col * STR + 8 * quad;
    fetch(i_begin, ph, pl, preb);
    for (int item0 = i_begin; item0 < i_end; item0 += NI) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const int e = 8 * (tid + 256 * p), it = e / KQ, k = e % KQ;
            *reinterpret_cast<half8_t*>(&tile_h[it * STR + k]) = ph[p];
            *reinterpret_cast<half8_t*>(&tile_l[it * STR + k]) = pl[p];
        }
        if (tid < NI) tile_b[tid] = preb;
        __syncthreads();
        if (item0 + NI < i_end) fetch(item0 + NI, ph, pl, preb);
        // the item fragments of a sub-step are read from LDS while the epilogue of the sub-step before it runs (into the
        // registers its MFMAs have just finished with): a wait for them at the top of every sub-step was 40 % of the kernel
        half8_t bh[NCH], bl[NCH]; float bj;
        // (the lane's part of the address is formed once; per sub-step one addition of a scalar: the address arithmetic of
        // (16 sub + col) * STR was five vector instructions per sub-step, one of them a 32-bit multiply, in a loop bound by vector issue)
        auto read_frags = [&](int sub) {
            const int at = frag0 + sub * (16 * STR);
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                bh[ch] = *reinterpret_cast<const half8_t*>(&tiles[at + ch * 32]);
                bl[ch] = *reinterpret_cast<const half8_t*>(&tiles[NI * STR + at + ch * 32]);
            }
            bj = tile_b[sub * 16 + col];
        };
        // this lane's column: undecided when 0 <= y <= the queries' window + 2.03 e_j, e_j = 6 u |b_j| the constant's own margin
        // (one v_fma per sub-step; a column past the end of the catalog has b = 3e38: y is huge, the window 2e32)
        unsigned window; float ej;
        // (Writing the MFMAs of sub-step s + 1 next to the epilogue of sub-step s, so that the scheduler could interleave
        // them, was measured and changed nothing: 133.0 against 132.6 ms for 2M queries.)
        read_frags(0);
#pragma unroll 1
        for (int sub = 0; sub < NI / 16; sub++) {
            const int j = item0 + sub * 16 + col;
            unsigned mn = 0xffffffffu;
            float sc[MT][4];
            ej = 7.3e-7f * __builtin_fabsf(bj);
            window = __float_as_uint(window_q + ej);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                // b_j is added after the chains: as their starting value it would make every one of their roundings relative
                // to |b_j|, and where the constants dominate the scores (Fossil's tables start at 1e-6) the margin then
                // covered hundreds of columns per query
                float4_t am = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][ch], bl[ch], am, 0, 0, 0);
                    am = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt][ch], bh[ch], am, 0, 0, 0);
                }
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][ch], bh[ch], am, 0, 0, 0);
                // element by element, and the file is built without SLP vectorisation: next to MFMAs a v_pk_add_f32 costs
                // more issue time than the two v_add_f32 it replaces, and this loop is bound by vector issue
#pragma unroll
                for (int r = 0; r < 4; r++) sc[mt][r] = am[r] + bj;
            }
            if (sub + 1 < NI / 16) read_frags(sub + 1);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned y1 = __float_as_uint(sc[mt][r] - thm1[mt][r]);
                    bits1[mt][r] = __builtin_amdgcn_alignbit(bits1[mt][r], y1, 31);
                    if (TWO) {
                        const unsigned y2 = __float_as_uint(sc[mt][r] - thm2[mt][r]);
                        bits2[mt][r] = __builtin_amdgcn_alignbit(bits2[mt][r], y2, 31);
                        mn = min(min(mn, y1), y2);   // one v_min3_u32
                    } else {
                        mn = min(mn, y1);
                    }
                }
            }
            if (++nbits == 32) { nbits = 0; add_up(~0u); }
            if (__ballot(mn <= window)) {
            // some score of the wavefront lies within the widest window of a threshold: which ones lie within their own query's
            unsigned open1 = 0, open2 = 0;   // bit (4 mt + r)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const float4_t w4 = *reinterpret_cast<const float4_t*>(&q_win[qw + mt * 16 + 4 * quad]);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned wqe = __float_as_uint(w4[r] + ej);
                    open1 |= (unsigned)(__float_as_uint(sc[mt][r] - thm1[mt][r]) <= wqe) << (4 * mt + r);
                    if (TWO) open2 |= (unsigned)(__float_as_uint(sc[mt][r] - thm2[mt][r]) <= wqe) << (4 * mt + r);
                }
            }
            unsigned open = open1 | open2;
            while (open) {
                const int e = __ffs(open) - 1;
                open &= open - 1;
                const int qi = qw + (e >> 2) * 16 + 4 * quad + (e & 3);
                const int which = (int)((open1 >> e) & 1u) | ((int)((open2 >> e) & 1u) << 1);
                int at;
                if (shared_lists) {
                    at = atomicAdd(&ev.amb_cnt[qblock + qi], 1);
                    if (at >= ev.amb_cap) *ev.amb_overflow = 1;
                } else {
                    at = atomicAdd(&q_amb[qi], 1);
                }
                if (at < ev.amb_cap) ev.amb[(qblock + qi) * (int64_t)ev.amb_cap + at] = j | (which << 28);
            }
            }
        }
    }
    __syncthreads();
    if (!shared_lists)
        for (int i = tid; i < QB; i += 256) {
            ev.amb_cnt[qblock + i] = q_amb[i];
            if (q_amb[i] > ev.amb_cap) *ev.amb_overflow = 1;
        }
    if (nbits) add_up(~0u >> (32 - nbits));   // the sub-steps since the last addition
    __syncthreads();
    for (int i = tid; i < (TWO ? 2 : 1) * QB; i += 256) {
        int c = 0;
#pragma unroll
        for (int x = 0; x < 16; x++) c += cnt_lds[i * 16 + ((x + i) & 15)];
        int32_t* dst = (i < QB ? ev.cnt : ev.cnt2) + qblock + (i < QB ? i : i - QB);
        if (shared_lists) atomicAdd(dst, c); else *dst = c;
    }
}

// max_j ||T_j||_2 and max_j |b_j| (as the bit patterns of non-negative floats, which order like unsigned integers).  A
// wavefront walks items wave, wave + waves, ... and publishes its two maxima once (one atomic per item on the same two
// words serializes: 4.5 ms for 200k items)
__global__ void __launch_bounds__(256) screen_item_bounds_kernel(EvalDev ev, const float* __restrict__ seg0, const float* __restrict__ seg1, int I,
                                                                 unsigned* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
    unsigned nmax = 0, bmax = 0;
    for (int j = wave; j < I; j += waves) {
        double s = 0;
        for (int k = lane; k < ev.K1; k += 64) { const double t = (double)seg0[(int64_t)j * ev.K1 + k]; s += t * t; }
        for (int k = lane; k < ev.K2; k += 64) { const double t = (double)seg1[(int64_t)j * ev.K2 + k]; s += t * t; }
        s = group_sum<64>(s);
        const float n = __double2float_ru(sqrt(s) * 1.000001), b = __double2float_ru(fabs(ev.bconst[j]) * 1.000001);
        // NaN / infinity in the tables: the margins become NaN / infinite and every column goes to the fp64 path or is dropped as there
        nmax = max(nmax, __float_as_uint(n == n ? n : INFINITY));
        bmax = max(bmax, __float_as_uint(b == b ? b : INFINITY));
    }
    if (lane == 0) { atomicMax(out, nmax); atomicMax(out + 1, bmax); }
}

// the margin of every query; one wavefront per query
__global__ void __launch_bounds__(256) screen_margins_kernel(EvalDev ev, const unsigned* __restrict__ bounds)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= ev.nq_pad) return;
    double s = 0;
    for (int k = lane; k < ev.KQ; k += 64) { const double q = ev.Q[w * ev.KQ + k]; s += q * q; }
    s = sqrt(group_sum<64>(s));
    if (lane == 0) {
        const double tmax = (double)__uint_as_float(bounds[0]), bmax = (double)__uint_as_float(bounds[1]);
        const double u = 5.9604644775390625e-08;
        const double t1 = ev.thr[w], t2 = ev.nctx == 1 ? ev.thr2[w] : 0.0;
        const double ta = (isfinite(t1) ? fabs(t1) : 0.0) + (isfinite(t2) ? fabs(t2) : 0.0);
        const double mq = 1.05 * (ev.KQ + 2) * u * s * tmax + 4.0 * u * (bmax + ta + s * tmax) + 1e-30;
        // a padding row (no query) decides nothing and lists nothing
        ev.qmar[w] = w < ev.nq ? __double2float_ru(mq * 1.000001) : 0.f;
    }
}

// doubles as unsigned keys that order like the doubles (every real number and both infinities map to a key above 0)
__device__ __forceinline__ uint64_t order_key(double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double order_unkey(uint64_t k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

// Top-k lists out of the |.|_1 sweep.  The scores of that sweep are the exact fp64 scores already, so a query's list holds
// (score, item) and its bar is simply the k-th best score listed so far (clicked items left out): no margins, no rescoring.
// A wavefront of the sweep is 64 items against the same queries, so a list belongs to one wavefront - appended to with a
// ballot, no atomics - and when fewer than 64 places are left the wavefront selects the k-th best of it (a search over the
// 64 bits of the keys, counted with ballots), drops what is below and what the user has clicked, and raises the bar.
// More than L1_LIST - 64 entries tied at the k-th best: the query is left to the score-matrix path.
#define L1_LIST 256
struct TopkL1Dev {
    int k;
    double* cand_s;             // [nq_pad][L1_LIST]
    int32_t* cand_i;            // [nq_pad][L1_LIST]
    int32_t* cand_cnt;          // [nq_pad]
    int32_t* q_over;            // [nq_pad] 1: ties the list could not hold
    const int4* qmeta;          // [nq_pad] (user, positive, the other held-out item, -); user -1 on padding rows
    const int64_t* clk_off; const int32_t* clk_item;
    int32_t* items_out;         // [nq][k]
    double* scores_out;         // [nq][k]
};

// n: entries in the list of query qi (wave-uniform); returns the new bar.  slack: 0 when the listed scores are exact; with
// approximate scores (each within M of the exact one) 2 M - the k entries at or above the k-th best tau are exactly >=
// tau - M, so the exact k-th best is, and a column of the exact top k is listed at >= tau - 2 M: never refused, never dropped.
__device__ __forceinline__ double l1_list_select(const TopkL1Dev& tk, int64_t qi, int& n, int lane, double slack)
{
    double* cs = tk.cand_s + qi * L1_LIST; int32_t* ci = tk.cand_i + qi * L1_LIST;
    const int4 meta = tk.qmeta[qi];
    const int user = meta.x;
    const int32_t* clk = tk.clk_item + tk.clk_off[user]; const int nc = (int)(tk.clk_off[user + 1] - tk.clk_off[user]);
    __threadfence();                           // the appends of this wavefront have reached memory
    double sc[L1_LIST / 64]; int it[L1_LIST / 64]; uint64_t key[L1_LIST / 64];
#pragma unroll
    for (int i = 0; i < L1_LIST / 64; i++) {
        const int idx = lane + 64 * i;
        // loads that do not stop at this compute unit's vector cache (which may hold the lines as an earlier selection read them)
        sc[i] = idx < n ? __hip_atomic_load(cs + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        it[i] = idx < n ? __hip_atomic_load(ci + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
    }
#pragma unroll
    for (int i = 0; i < L1_LIST / 64; i++) {
        const bool ok = lane + 64 * i < n && it[i] != meta.y && it[i] != meta.z && !sorted_contains(clk, nc, it[i]);
        key[i] = ok ? order_key(sc[i]) : 0ull;
    }
    uint64_t prefix = 0;                       // the largest v with at least k keys >= v (0: fewer than k candidates)
#pragma unroll 1
    for (int bit = 63; bit >= 0; bit--) {
        const uint64_t v = prefix | (1ull << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < L1_LIST / 64; i++) c += __popcll(__ballot(key[i] >= v));
        if (c >= tk.k) prefix = v;
    }
    double nb = prefix ? order_unkey(prefix) - slack : -INFINITY;
    int off = 0;
#pragma unroll
    for (int i = 0; i < L1_LIST / 64; i++) {
        const bool keep = key[i] != 0ull && sc[i] >= nb;
        const unsigned long long bal = __ballot(keep);
        if (keep) { const int at = off + __popcll(bal & ((1ull << lane) - 1ull)); cs[at] = sc[i]; ci[at] = it[i]; }
        off += __popcll(bal);
    }
    if (off > L1_LIST - 64) {
        if (lane == 0) tk.q_over[qi] = 1;
        off = 0; nb = INFINITY;
    }
    n = off;
    return nb;
}

// |.|_1 scoring (TransRec_L1): not a contraction, so it runs on the vector ALU.  A block owns QT
// queries for the whole item sweep; items are staged through LDS in tiles of 64 (odd row stride:
// conflict-free when the 64 lanes of a wavefront read 64 different items), thread (item jj, query
// group qg) accumulates QT/4 distances per tile.
// MODE 0: counts against the queries' thresholds; 1: the scores themselves, to ev.S; 2: top-k lists (TI == 64 only)
template <class T, int QT, int TI, int MODE>
__global__ void __launch_bounds__(256) score_count_l1_kernel(EvalDev ev, const T* __restrict__ H, const int32_t* __restrict__ qtest,
                                                             const int32_t* __restrict__ qval, int I, int K, TopkL1Dev tk)
{
    constexpr bool STORE = MODE == 1;
    static_assert(MODE != 2 || TI == 64, "a list belongs to one wavefront");
    constexpr int NG = 256 / TI, QPT = QT / NG;   // query groups per block, queries per thread
    extern __shared__ double lds[];
    const int STR = K | 1;
    double* tile = lds;                 // [TI][STR]
    double* qs = lds + TI * STR;        // [QT][K]
    const int tid = threadIdx.x, jj = tid % TI, qg = tid / TI;
    const int64_t qbase = (int64_t)blockIdx.x * QT;
    for (int i = tid; i < QT * K; i += 256) qs[i] = ev.Q[(qbase + i / K) * ev.KQ + i % K];
    double thr[QPT]; int ex1[QPT], ex2[QPT], cnt[QPT];
#pragma unroll
    for (int a = 0; a < QPT; a++) {
        const int64_t qi = qbase + qg + NG * a;
        if constexpr (MODE == 2) {
            // the bar of a list; a padding row lists nothing
            const int4 meta = tk.qmeta[qi];
            thr[a] = meta.x >= 0 ? -INFINITY : INFINITY; ex1[a] = meta.y; ex2[a] = meta.z; cnt[a] = 0;
        } else {
            thr[a] = ev.thr[qi]; ex1[a] = qtest[qi]; ex2[a] = qval[qi]; cnt[a] = 0;
        }
    }
    // A tile is a contiguous stretch of the item table.  Up to PF elements per thread, the next tile is requested into
    // registers before the current one is scored and stored to LDS afterwards: staged where it is needed, every tile waited
    // for its loads with nothing else to do (two barriers per tile) - about 40 % of the sweep.
    constexpr int PF = 16;
    const bool ahead = TI * K <= 256 * PF;
    T pre[PF]; int lofs[PF];
#pragma unroll
    for (int e = 0; e < PF; e++) { const int i = tid + 256 * e; lofs[e] = i < TI * K ? (i / K) * STR + i % K : -1; pre[e] = 0; }
    auto request = [&](int item0) {
        const int64_t base = (int64_t)item0 * K, end = (int64_t)I * K;
#pragma unroll
        for (int e = 0; e < PF; e++) { const int64_t g = base + tid + 256 * e; pre[e] = (lofs[e] >= 0 && g < end) ? H[g] : (T)0; }
    };
    if (ahead) request(0);
    for (int item0 = 0; item0 < I; item0 += TI) {
        __syncthreads();
        if (ahead) {
#pragma unroll
            for (int e = 0; e < PF; e++) if (lofs[e] >= 0) tile[lofs[e]] = (double)pre[e];
        } else {
            for (int i = tid; i < TI * K; i += 256) {
                const int it = i / K, k = i % K; const int64_t j = (int64_t)item0 + it;
                tile[it * STR + k] = j < I ? (double)H[j * K + k] : 0.0;
            }
        }
        __syncthreads();
        if (ahead && item0 + TI < I) request(item0 + TI);
        const int j = item0 + jj;
        // a score is summed as the reference sums it (TransRec_L1.cpp:87-91): from b_j, k ascending
        const double bj0 = j < I ? ev.col_b[j] : 0.0;
        double acc[QPT];
#pragma unroll
        for (int a = 0; a < QPT; a++) acc[a] = bj0;
        if constexpr (TI == 64) {
            // a wavefront is one query group: its QPT queries are the same on every lane.  Read from the query array with a
            // wave-uniform address they arrive in scalar registers (scalar loads, four components at a time) instead of
            // through LDS - nine LDS reads per component and wavefront (one tile value, eight broadcast query values) against
            // sixteen fp64 vector operations left LDS the bound, at 2.3x the vector floor
            const int qgu = __builtin_amdgcn_readfirstlane(qg);
            // (the constant address space: the array is not written while this kernel runs, and only loads from that space
            // are issued as scalar loads when their address is wave-uniform)
            typedef const __attribute__((address_space(4))) double* const_double_ptr;
            const_double_ptr qrow = (const_double_ptr)(uintptr_t)(ev.Q + (qbase + qgu) * ev.KQ);
            const int64_t qstep = (int64_t)NG * ev.KQ;
            int k = 0;
            for (; k + 4 <= K; k += 4) {
                const double t0 = tile[jj * STR + k], t1 = tile[jj * STR + k + 1], t2 = tile[jj * STR + k + 2], t3 = tile[jj * STR + k + 3];
#pragma unroll
                for (int a = 0; a < QPT; a++) {
                    const_double_ptr qa = qrow + a * qstep + k;
                    acc[a] += fabs(qa[0] - t0); acc[a] += fabs(qa[1] - t1); acc[a] += fabs(qa[2] - t2); acc[a] += fabs(qa[3] - t3);
                }
            }
            for (; k < K; k++) {
                const double t = tile[jj * STR + k];
#pragma unroll
                for (int a = 0; a < QPT; a++) acc[a] += fabs(qrow[a * qstep + k] - t);
            }
        } else {
            for (int k = 0; k < K; k++) {
                const double t = tile[jj * STR + k];
#pragma unroll
                for (int a = 0; a < QPT; a++) acc[a] += fabs(qs[(qg + NG * a) * K + k] - t);
            }
        }
        if constexpr (MODE == 2) {
            const bool inr = j < I;
            const int jid = inr ? (ev.col_item ? ev.col_item[j] : j) : -1;
            bool crowded = false;
#pragma unroll
            for (int a = 0; a < QPT; a++) {
                const double sc = -acc[a] + 0.0;              // (+ 0.0: one zero, so that keys order exactly like the scores)
                const bool in = inr && jid != ex1[a] && jid != ex2[a] && sc >= thr[a];
                const unsigned long long bal = __ballot(in);
                if (bal) {
                    const int64_t qi = qbase + qg + NG * a;
                    if (in) {
                        const int at = cnt[a] + __popcll(bal & ((1ull << jj) - 1ull));
                        tk.cand_s[qi * L1_LIST + at] = sc; tk.cand_i[qi * L1_LIST + at] = jid;
                    }
                    cnt[a] += __popcll(bal);
                    crowded |= cnt[a] > L1_LIST - 64;
                }
            }
            // (selections after the tile's appends, where the accumulators are dead: called from inside the loop above
            // they doubled the kernel's registers)
            // ONE copy of the selection code, its query picked at run time: counts and bars go through LDS for it (the query
            // rows of LDS, which this tile shape does not read - its queries arrive through scalar loads); unrolled over
            // the queries it cost 90 more registers
            if (crowded) {
                double* sw = qs + qg * (2 * QPT);
                if (jj == 0) {
#pragma unroll
                    for (int a = 0; a < QPT; a++) { sw[2 * a] = (double)cnt[a]; sw[2 * a + 1] = thr[a]; }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll 1
                for (int a = 0; a < QPT; a++) {
                    int n = (int)sw[2 * a];
                    if (n > L1_LIST - 64) {
                        const double nb = l1_list_select(tk, qbase + qg + NG * a, n, jj, 0.0);
                        if (jj == 0) { sw[2 * a] = (double)n; sw[2 * a + 1] = nb; }
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int a = 0; a < QPT; a++) { cnt[a] = (int)sw[2 * a]; thr[a] = sw[2 * a + 1]; }
            }
        } else if (j < I) {
            const int jid = ev.col_item ? ev.col_item[j] : j;
#pragma unroll
            for (int a = 0; a < QPT; a++) {
                if (STORE) ev.S[(qbase + qg + NG * a) * (int64_t)I + j] = -acc[a];
                else cnt[a] += (jid != ex1[a] && jid != ex2[a] && thr[a] > -acc[a]) ? 1 : 0;
            }
        }
    }
    if constexpr (MODE == 2) {
#pragma unroll
        for (int a = 0; a < QPT; a++) if (jj == 0) tk.cand_cnt[qbase + qg + NG * a] = cnt[a];
        return;
    }
    if (STORE) return;
#pragma unroll
    for (int a = 0; a < QPT; a++) {
        int c = cnt[a];
#pragma unroll
        for (int o = TI / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (jj == 0) ev.cnt[qbase + qg + NG * a] = c;
    }
}

// launches the L1 scorer with the item tile that fits the 160 KB of LDS: 64 items up to K = 160, 32 above
template <class T, int MODE>
static int launch_score_l1(const EvalDev& ev, const T* H, const int32_t* qtest, const int32_t* qval, int n_cols, int K, hipStream_t st,
                           const TopkL1Dev& tk = TopkL1Dev{})
{
    constexpr int QT = 32;
    const unsigned blocks = (unsigned)(ev.nq_pad / QT);
    if (K <= 160) {
        const size_t lds_bytes = sizeof(double) * ((size_t)64 * (K | 1) + (size_t)QT * K);
        hipLaunchKernelGGL((score_count_l1_kernel<T, QT, 64, MODE>), dim3(blocks), dim3(256), lds_bytes, st, ev, H, qtest, qval, n_cols, K, tk);
    } else {
        const size_t lds_bytes = sizeof(double) * ((size_t)32 * (K | 1) + (size_t)QT * K);
        if constexpr (MODE == 2) { rec_set_error("top-k lists out of the L1 sweep need K <= 160"); return -1; }
        else hipLaunchKernelGGL((score_count_l1_kernel<T, QT, 32, MODE>), dim3(blocks), dim3(256), lds_bytes, st, ev, H, qtest, qval, n_cols, K, tk);
    }
    if (hipGetLastError() != hipSuccess) { rec_set_error("launch of the L1 scoring kernel failed"); return -2; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Integer screen of the |.|_1 sweep.  sum_k |q_k - t_k| is no contraction, but on fixed-point operands it is one vector
// instruction per term: v_sad_u32 (D = |a - b| + c, exact).  Tables and queries are mapped to unsigned integers,
//     x' = rint((x + A) / delta),   A = max |x| over both,   delta = 2 A / (R - 1),   R = floor((2^32 - 1) / K),
// so that every sum of K differences fits 32 bits.  Each operand is off by at most delta / 2 (+ the double rounding of the
// mapping), a difference by at most delta, the sum S' by at most K delta:  | sum_k |q_k - t_k| - delta S' | <= 1.0002 K delta.
// In units of delta, with B_j = b_j / delta and T_q = threshold / delta, y = S' + B_j + T_q decides
//     y >  M : the column counts (threshold > -(b_j + sum)),    y < -M : it does not,     M = 1.0002 K + 2,
// and what lies between (or is not a number) goes to the query's list of undecided columns, which correct_clicked_kernel decides in fp64 - the
// held-out pair included (its own column is undecided by construction and cancels there against the pair's entry).
// Half the vector instructions of the fp64 sweep and half its LDS and scalar-load bytes; the counts are identical.
__global__ void __launch_bounds__(256) l1_absmax_kernel(const void* __restrict__ H, int is_f64, int64_t nH, const double* __restrict__ Q, int64_t nQ,
                                                        unsigned* __restrict__ bound)
{
    float mx = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < nH; i += stride) {
        const double x = is_f64 ? ((const double*)H)[i] : (double)((const float*)H)[i];
        const float a = __double2float_ru(fabs(x));
        mx = (a > mx || a != a) ? a : mx;             // a NaN stays
    }
    for (int64_t i = t0; i < nQ; i += stride) {
        const float a = __double2float_ru(fabs(Q[i]));
        mx = (a > mx || a != a) ? a : mx;
    }
    // (bit patterns of non-negative floats order like the floats; NaN patterns are above infinity's)
    unsigned b = __float_as_uint(mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(bound, b);
}

__device__ __forceinline__ void l1_scale(const unsigned* bound, int K, double& A, double& inv_delta, bool& ok)
{
    const float a = __uint_as_float(bound[0]);
    ok = a == a && a < 3.0e38f;
    A = ok ? fmax((double)a, 1e-30) : 1.0;       // (all-zero operands: any scale does, and this one keeps 1 / delta finite)
    const double R = floor(4294967295.0 / (double)K);
    inv_delta = (R - 1.0) / (2.0 * A);
}

__global__ void __launch_bounds__(256) l1_quant_items_kernel(const void* __restrict__ H, int is_f64, int I, int K, const double* __restrict__ b,
                                                             const unsigned* __restrict__ bound, uint32_t* __restrict__ Hi, double* __restrict__ Bj,
                                                             int32_t* __restrict__ flags)
{
    double A, inv; bool ok; l1_scale(bound, K, A, inv, ok);
    const int64_t n = (int64_t)I * K, stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!ok) { if (t0 == 0) flags[1] = 1; return; }
    for (int64_t i = t0; i < n; i += stride) {
        const double x = is_f64 ? ((const double*)H)[i] : (double)((const float*)H)[i];
        Hi[i] = (uint32_t)rint((x + A) * inv);
    }
    for (int64_t j = t0; j < I; j += stride) Bj[j] = b[j] * inv;
}

__global__ void __launch_bounds__(256) l1_quant_queries_kernel(EvalDev ev, int K, const unsigned* __restrict__ bound, uint32_t* __restrict__ Qi,
                                                               double* __restrict__ Tq)
{
    double A, inv; bool ok; l1_scale(bound, K, A, inv, ok);
    if (!ok) return;
    const int64_t n = (int64_t)ev.nq_pad * K, stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n; i += stride) {
        const int64_t q = i / K; const int k = (int)(i % K);
        Qi[i] = (uint32_t)rint((ev.Q[q * ev.KQ + k] + A) * inv);
    }
    for (int64_t q = t0; q < ev.nq_pad; q += stride) Tq[q] = ev.thr[q] * inv;
}

// a block owns QT queries for the whole catalog (tiles of 64 items through LDS, the next tile requested while the current
// one is scored); a wavefront is 64 items against QT / 4 queries, which arrive through scalar loads
// LISTS: top-k lists instead of counts.  A list holds (v, item), v = -(S' + B_j) the score in units of delta, within M of
// the exact one; the bar follows the k-th best v at a distance of 2 M (l1_list_select), and topk_l1_finish_kernel scores
// what is left in the lists in fp64.
template <int QT, bool LISTS>
__global__ void __launch_bounds__(256) score_count_l1_sad_kernel(EvalDev ev, const uint32_t* __restrict__ Hi, const uint32_t* __restrict__ Qi,
                                                                 const double* __restrict__ Bj, const double* __restrict__ Tq, int I, int K, double M,
                                                                 typename std::conditional<LISTS, TopkL1Dev, int>::type tk)
{
    constexpr int TI = 64, NG = 4, QPT = QT / NG, PF = 16;
    extern __shared__ uint32_t tile_u[];          // [TI][STR], then (LISTS) two doubles per query
    const int STR = K | 1;
    const int tid = threadIdx.x, jj = tid & 63, qg = tid >> 6;
    const int64_t qbase = (int64_t)blockIdx.x * QT;
    double tq[QPT]; int cnt[QPT], open[QPT];
#pragma unroll
    for (int a = 0; a < QPT; a++) {
        // LISTS: tq is the bar of the list (a padding row lists nothing), cnt its length
        if constexpr (LISTS) tq[a] = tk.qmeta[qbase + qg + NG * a].x >= 0 ? -INFINITY : INFINITY;
        else tq[a] = Tq[qbase + qg + NG * a];
        cnt[a] = 0; open[a] = 0;
    }
    uint32_t pre[PF]; int lofs[PF];
#pragma unroll
    for (int e = 0; e < PF; e++) { const int i = tid + 256 * e; lofs[e] = i < TI * K ? (i / K) * STR + i % K : -1; pre[e] = 0; }
    const int64_t end = (int64_t)I * K;
    auto request = [&](int item0) {
        const int64_t base = (int64_t)item0 * K;
#pragma unroll
        for (int e = 0; e < PF; e++) { const int64_t g = base + tid + 256 * e; pre[e] = (lofs[e] >= 0 && g < end) ? Hi[g] : 0u; }
    };
    const bool ahead = TI * K <= 256 * PF;        // wider tiles are staged when they are needed
    if (ahead) request(0);
    const int qgu = __builtin_amdgcn_readfirstlane(qg);
    typedef const __attribute__((address_space(4))) uint32_t* const_u32_ptr;
    const_u32_ptr qrow = (const_u32_ptr)(uintptr_t)(Qi + (qbase + qgu) * K);
    const int64_t qstep = (int64_t)NG * K;
    for (int item0 = 0; item0 < I; item0 += TI) {
        __syncthreads();
        if (ahead) {
#pragma unroll
            for (int e = 0; e < PF; e++) if (lofs[e] >= 0) tile_u[lofs[e]] = pre[e];
        } else {
            const int64_t base = (int64_t)item0 * K;
            for (int i = tid; i < TI * K; i += 256) tile_u[(i / K) * STR + i % K] = base + i < end ? Hi[base + i] : 0u;
        }
        __syncthreads();
        if (ahead && item0 + TI < I) request(item0 + TI);
        const int j = item0 + jj;
        uint32_t acc[QPT];
#pragma unroll
        for (int a = 0; a < QPT; a++) acc[a] = 0;
        const uint32_t* trow = tile_u + jj * STR;
        int k = 0;
        // (four terms per round: with eight the queries of a round no longer fit the scalar registers, 1.41 -> 2.48 s per 2M queries)
        for (; k + 4 <= K; k += 4) {
            const uint32_t t0 = trow[k], t1 = trow[k + 1], t2 = trow[k + 2], t3 = trow[k + 3];
#pragma unroll
            for (int a = 0; a < QPT; a++) {
                const_u32_ptr qa = qrow + a * qstep + k;
                acc[a] = __usad(t0, qa[0], acc[a]); acc[a] = __usad(t1, qa[1], acc[a]);
                acc[a] = __usad(t2, qa[2], acc[a]); acc[a] = __usad(t3, qa[3], acc[a]);
            }
        }
        for (; k < K; k++) {
            const uint32_t t = trow[k];
#pragma unroll
            for (int a = 0; a < QPT; a++) acc[a] = __usad(t, qrow[a * qstep + k], acc[a]);
        }
        const bool inr = j < I;
        const double bj = inr ? Bj[j] : 0.0;
        if constexpr (LISTS) {
            bool crowded = false;
#pragma unroll
            for (int a = 0; a < QPT; a++) {
                const double v = -((double)acc[a] + bj) + 0.0;
                const bool in = inr && v >= tq[a];
                const unsigned long long bal = __ballot(in);
                if (bal) {
                    const int64_t qi = qbase + qg + NG * a;
                    if (in) {
                        const int at = cnt[a] + __popcll(bal & ((1ull << jj) - 1ull));
                        tk.cand_s[qi * L1_LIST + at] = v; tk.cand_i[qi * L1_LIST + at] = j;
                    }
                    cnt[a] += __popcll(bal);
                    crowded |= cnt[a] > L1_LIST - 64;
                }
            }
            if (crowded) {      // one copy of the selection code, counts and bars through LDS (as in score_count_l1_kernel)
                double* sw = reinterpret_cast<double*>(tile_u + ((TI * STR + 1) & ~1)) + qg * (2 * QPT);
                if (jj == 0) {
#pragma unroll
                    for (int a = 0; a < QPT; a++) { sw[2 * a] = (double)cnt[a]; sw[2 * a + 1] = tq[a]; }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll 1
                for (int a = 0; a < QPT; a++) {
                    int n = (int)sw[2 * a];
                    if (n > L1_LIST - 64) {
                        const double nb = l1_list_select(tk, qbase + qg + NG * a, n, jj, 2.0 * M);
                        if (jj == 0) { sw[2 * a] = (double)n; sw[2 * a + 1] = nb; }
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int a = 0; a < QPT; a++) { cnt[a] = (int)sw[2 * a]; tq[a] = sw[2 * a + 1]; }
            }
            continue;
        }
#pragma unroll
        for (int a = 0; a < QPT; a++) {
            const double y = (double)acc[a] + bj + tq[a];
            const bool yes = inr && y > M;
            const bool und = inr && !(y > M) && !(y < -M);      // (a y that is not a number is left to fp64 as well)
            cnt[a] += yes ? 1 : 0;
            const unsigned long long bal = __ballot(und);
            if (bal) {
                const int at = open[a] + __popcll(bal & ((1ull << jj) - 1ull));
                if (und && at < ev.amb_cap) ev.amb[(qbase + qg + NG * a) * (int64_t)ev.amb_cap + at] = j | (1 << 28);
                open[a] += __popcll(bal);
            }
        }
    }
    if constexpr (LISTS) {
#pragma unroll
        for (int a = 0; a < QPT; a++) if (jj == 0) tk.cand_cnt[qbase + qg + NG * a] = cnt[a];
        return;
    }
#pragma unroll
    for (int a = 0; a < QPT; a++) {
        int c = cnt[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (jj == 0) {
            ev.cnt[qbase + qg + NG * a] = c;
            ev.amb_cnt[qbase + qg + NG * a] = open[a];
            if (open[a] > ev.amb_cap) *ev.amb_overflow = 1;
        }
    }
}

// Removes the clicked items from the counts, decides the columns the f32 screen left open, and writes (count_val,
// count_test, max) per user.  One wavefront per query; the scores it needs are recomputed in fp64 on the vector ALU, four
// items at a time (16 lanes per item) - ONE summation order for every use, so that a column added by one rule and taken out
// by another is compared the same way both times.
template <class T>
__global__ void __launch_bounds__(256) correct_clicked_kernel(DevModel m, EvalDev ev)
{
    const int lane = threadIdx.x & 63, l16 = lane & 15, gq = lane >> 4;
    const int w = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= ev.nq) return;
    const int u = m.eval_users[w / ev.nctx];
    const int ctx = ev.nctx == 2 ? (w & 1) : 0;
    const int val = m.val_item[u], test = m.test_item[u];
    const int32_t* clk = m.clk_item + m.clk_off[u]; const int nc = (int)(m.clk_off[u + 1] - m.clk_off[u]);
    const double* q = ev.Q + (int64_t)w * ev.KQ;
    const T* seg0 = ev.seg0 >= 0 ? (const T*)m.tab[ev.seg0] : nullptr;
    const T* seg1 = ev.seg1 >= 0 ? (const T*)m.tab[ev.seg1] : nullptr;
    const int K1 = ev.K1, K2 = ev.K2;
    const bool l1 = m.model == REC_TRANSREC_L1;
    const double thr = ev.thr[w], thr2 = ev.nctx == 1 ? ev.thr2[w] : 0.0;
    // columns that are not candidates: every occurrence of a clicked item and of the held-out pair
    int excluded = 0; bool tc = false, vc = false;
    for (int i = lane; i < nc; i += 64) {
        const int e = clk[i];
        excluded += ev.item_mult ? ev.item_mult[e] : 1;
        tc |= e == test; vc |= e == val;
    }
    excluded = group_sum<64>(excluded);
    const bool test_clicked = __ballot(tc) != 0, val_clicked = __ballot(vc) != 0;
    if (!test_clicked) excluded += ev.item_mult ? ev.item_mult[test] : 1;
    if (!val_clicked && val != test) excluded += ev.item_mult ? ev.item_mult[val] : 1;

    // one list of (item, weight, thresholds it applies to): the clicked items (-multiplicity; the fp64 sweeps never count
    // the held-out pair, the screen counts it like any column), then the held-out pair where the screen counted it and
    // the click list does not take it out, then the columns the screen left open (+1 where the threshold is above the score)
    // (a query marked -1 overflowed its list and was swept again on its own in fp64, the held-out pair excluded there)
    const int acnt = ev.amb ? ev.amb_cnt[w] : 0;
    const bool heldout_counted = ev.heldout_counted && acnt >= 0;
    const int na = acnt < 0 ? 0 : acnt < ev.amb_cap ? acnt : ev.amb_cap;
    const int32_t* open = ev.amb ? ev.amb + (int64_t)w * ev.amb_cap : nullptr;
    const int total = nc + 2 + na;
    int dc = 0, dc2 = 0;
    // Sixteen entries of the list per round: lane l decodes entry base + (l & 15), then the four groups of 16 lanes score
    // entries 4s + group for s = 0 .. 3 - the rows of all four requested together (one entry at a time a round cost two
    // memory round trips: the id, then its row).  Each score is summed as before: k ascending within the 16 lanes, then
    // across them.
    for (int base = 0; base < total; base += 16) {
        const int idx = base + l16;
        int e = -1, wgt = 0, which = 3;
        if (idx < nc) {
            e = clk[idx];
            wgt = -(ev.item_mult ? ev.item_mult[e] : 1);
            if ((e == test || e == val) && !heldout_counted) wgt = 0;
        } else if (idx < nc + 2) {
            const int h = idx - nc;
            e = h == 0 ? test : val;
            const bool done = h == 0 ? test_clicked : (val_clicked || val == test);
            wgt = (heldout_counted && e >= 0 && !done) ? -1 : 0;
        } else if (idx < total) {
            const int ent = open[idx - nc - 2];
            e = ent & 0x0fffffff; which = ent >> 28; wgt = 1;
        }
        if (!__ballot(wgt != 0)) continue;
        int ws[4], whs[4]; int64_t ec[4]; double acc[4];
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
            const int src = 4 * sb + gq;
            const int es = __shfl(e, src, 64);
            ws[sb] = __shfl(wgt, src, 64); whs[sb] = __shfl(which, src, 64);
            ec[sb] = (ws[sb] != 0 && es >= 0) ? es : 0;   // an entry without work still runs the loads, on item 0
            acc[sb] = 0;
        }
        if (l1) {
            // from b_j, k ascending (TransRec_L1.cpp:87-91): the 16 lanes of a group load 16 terms, every lane adds them in order
#pragma unroll
            for (int sb = 0; sb < 4; sb++) acc[sb] = ev.bconst[ec[sb]];
            for (int k0 = 0; k0 < K1; k0 += 16) {
                const int k = k0 + l16;
                const double qk = k < K1 ? q[k] : 0.0;
                double t[4];
#pragma unroll
                for (int sb = 0; sb < 4; sb++) t[sb] = k < K1 ? fabs(qk - (double)seg0[ec[sb] * K1 + k]) : 0.0;
                const int n = min(16, K1 - k0);
                for (int l = 0; l < n; l++)
#pragma unroll
                    for (int sb = 0; sb < 4; sb++) acc[sb] += __shfl(t[sb], (lane & 48) + l, 64);
            }
        } else {
            for (int k = l16; k < K1; k += 16) {
                const double qk = q[k];
#pragma unroll
                for (int sb = 0; sb < 4; sb++) acc[sb] += qk * (double)seg0[ec[sb] * K1 + k];
            }
            for (int k = l16; k < K2; k += 16) {
                const double qk = q[K1 + k];
#pragma unroll
                for (int sb = 0; sb < 4; sb++) acc[sb] += qk * (double)seg1[ec[sb] * K2 + k];
            }
        }
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
            const double sc = l1 ? -acc[sb] : group_sum<16>(acc[sb]) + ev.bconst[ec[sb]];
            dc += ((whs[sb] & 1) && thr > sc) ? ws[sb] : 0;
            dc2 += ((whs[sb] & 2) && thr2 > sc) ? ws[sb] : 0;
        }
    }
    // every lane of a group holds the group's sum: add the four groups
    dc = dc + __shfl_xor(dc, 16, 64); dc = dc + __shfl_xor(dc, 32, 64);
    dc2 = dc2 + __shfl_xor(dc2, 16, 64); dc2 = dc2 + __shfl_xor(dc2, 32, 64);
    if (lane == 0) {
        const int c = ev.cnt[w] + dc, c2 = (ev.nctx == 1 ? ev.cnt2[w] : 0) + dc2;
        if (ev.nctx == 1) {
            ev.counts[3 * (int64_t)u] = c; ev.counts[3 * (int64_t)u + 1] = c2;
        } else {
            ev.counts[3 * (int64_t)u + ctx] = c;
        }
        if (ctx == 0) ev.counts[3 * (int64_t)u + 2] = ev.n_cols - excluded;
    }
}

// Candidate draw d of user u: what model::sampleAUC takes from rand() % nItems (model.cpp:122-123), from a counter-based
// hash instead (splitmix64 finaliser), so that a user's draws depend on (seed, user, draw) alone.
__host__ __device__ __forceinline__ uint64_t draw_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ int32_t user_draw(uint64_t seed, int u, int d, int n_items)
{
    const uint64_t z = draw_mix64(draw_mix64(seed ^ draw_mix64((uint64_t)(uint32_t)u)) + (uint64_t)(uint32_t)d);
    return (int32_t)(((z >> 32) * (uint64_t)(uint32_t)n_items) >> 32);
}

// model::sampleAUC (model.cpp:100-143) with the reference's sampling: n_draws candidate draws PER USER, a draw that
// hits a clicked or held-out item is skipped and not replaced, every remaining draw is compared under the validation
// and under the test context.  One wavefront per user, four draws per step (16 lanes each), scores in fp64 with the
// summation of correct_clicked_kernel.  Writes (count_val, count_test, max) of the user.
template <class T>
__global__ void __launch_bounds__(256) sample_count_kernel(DevModel m, EvalDev ev, int n_draws, uint64_t seed, int32_t* __restrict__ draws_out)
{
    const int lane = threadIdx.x & 63, l16 = lane & 15, gq = lane >> 4;
    const int w = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= m.n_eval) return;
    const int u = m.eval_users[w];
    const int val = m.val_item[u], test = m.test_item[u];
    const int32_t* clk = m.clk_item + m.clk_off[u]; const int nc = (int)(m.clk_off[u + 1] - m.clk_off[u]);
    const int64_t w0 = (int64_t)w * ev.nctx;
    const double* qv = ev.Q + w0 * ev.KQ;                                  // validation context
    const double* qt = ev.nctx == 2 ? qv + ev.KQ : qv;                     // test context (the same query when the model has one)
    const double thr_v = ev.thr[w0], thr_t = ev.nctx == 2 ? ev.thr[w0 + 1] : ev.thr2[w0];
    const T* seg0 = ev.seg0 >= 0 ? (const T*)m.tab[ev.seg0] : nullptr;
    const T* seg1 = ev.seg1 >= 0 ? (const T*)m.tab[ev.seg1] : nullptr;
    const int K1 = ev.K1, K2 = ev.K2;
    const bool l1 = m.model == REC_TRANSREC_L1, two = ev.nctx == 2;
    int cv = 0, ct = 0, mx = 0;
    for (int base = 0; base < n_draws; base += 4) {
        const int d = base + gq;
        const bool live = d < n_draws;
        const int j = user_draw(seed, u, live ? d : 0, m.I);
        if (draws_out && live && l16 == 0) draws_out[(int64_t)u * n_draws + d] = j;
        const bool cand = live && j != test && j != val && !sorted_contains(clk, nc, j);
        if (!__ballot(cand)) continue;
        const int64_t ec = j;
        double sv = 0, st = 0;
        if (l1) {
            // from b_j, k ascending, as in correct_clicked_kernel
            sv = st = ev.bconst[ec];
            for (int k0 = 0; k0 < K1; k0 += 16) {
                const int k = k0 + l16;
                const double t = k < K1 ? (double)seg0[ec * K1 + k] : 0.0;
                const double tv = k < K1 ? fabs(qv[k] - t) : 0.0, tt = (two && k < K1) ? fabs(qt[k] - t) : 0.0;
                const int n = min(16, K1 - k0);
                for (int l = 0; l < n; l++) { sv += __shfl(tv, (lane & 48) + l, 64); if (two) st += __shfl(tt, (lane & 48) + l, 64); }
            }
            sv = -sv; st = two ? -st : sv;
        } else {
            for (int k = l16; k < K1; k += 16) { const double t = (double)seg0[ec * K1 + k]; sv += qv[k] * t; if (two) st += qt[k] * t; }
            for (int k = l16; k < K2; k += 16) { const double t = (double)seg1[ec * K2 + k]; sv += qv[K1 + k] * t; if (two) st += qt[K1 + k] * t; }
            const double b = ev.bconst[ec];
            sv = group_sum<16>(sv) + b; st = two ? group_sum<16>(st) + b : sv;
        }
        mx += cand ? 1 : 0;
        cv += (cand && thr_v > sv) ? 1 : 0;
        ct += (cand && thr_t > st) ? 1 : 0;
    }
    // every lane of a group holds the group's counts: add the four groups
    cv += __shfl_xor(cv, 16, 64); cv += __shfl_xor(cv, 32, 64);
    ct += __shfl_xor(ct, 16, 64); ct += __shfl_xor(ct, 32, 64);
    mx += __shfl_xor(mx, 16, 64); mx += __shfl_xor(mx, 32, 64);
    if (lane == 0) { ev.counts[3 * (int64_t)u] = cv; ev.counts[3 * (int64_t)u + 1] = ct; ev.counts[3 * (int64_t)u + 2] = mx; }
}

template <class T>
__global__ void gather_rows_kernel(const T* __restrict__ src, const int32_t* __restrict__ ids, int64_t n, int K, T* __restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * K; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[(int64_t)ids[i / K] * K + i % K];
}

__global__ void fill_f64_kernel(double* p, int64_t n, double v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
// test / validation item of the user behind every query row (-1 on padding rows)
__global__ void query_items_kernel(DevModel m, EvalDev ev, int32_t* qtest, int32_t* qval)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ev.nq_pad) return;
    int t = -1, v = -1;
    if (i < ev.nq) { const int u = m.eval_users[i / ev.nctx]; t = m.test_item[u]; v = m.val_item[u]; }
    qtest[i] = t; qval[i] = v;
}

// the queries listed in idx as a compact batch of their own (one wavefront per query), and their counts back
__global__ void __launch_bounds__(256) gather_queries_kernel(EvalDev ev, EvalDev sub, const int32_t* __restrict__ idx, const int32_t* __restrict__ qtest,
                                                             const int32_t* __restrict__ qval, int32_t* __restrict__ sub_test, int32_t* __restrict__ sub_val)
{
    const int lane = threadIdx.x & 63;
    const int w = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= sub.nq) return;
    const int64_t q = idx[w];
    for (int k = lane; k < ev.KQ; k += 64) sub.Q[(int64_t)w * ev.KQ + k] = ev.Q[q * ev.KQ + k];
    if (lane == 0) { sub.thr[w] = ev.thr[q]; sub.thr2[w] = ev.thr2[q]; sub_test[w] = qtest[q]; sub_val[w] = qval[q]; }
}
__global__ void scatter_counts_kernel(EvalDev ev, EvalDev sub, const int32_t* __restrict__ idx)
{
    const int w = (int)((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= sub.nq) return;
    const int64_t q = idx[w];
    ev.cnt[q] = sub.cnt[w];
    if (ev.nctx == 1) ev.cnt2[q] = sub.cnt2[w];
    ev.amb_cnt[q] = -1;          // nothing left undecided for this query, and its held-out pair has not been counted
}

template <class T, bool TWO, bool STORE>
static int launch_score(const EvalDev& ev, const T* s0, const T* s1, int n_cols, const int32_t* qtest, const int32_t* qval, hipStream_t st)
{
#define SCORE_CASE(KQv)                                                                                         \
    case KQv: {                                                                                                 \
        const unsigned blocks = (unsigned)(ev.nq_pad / (4 * TileShape<KQv>::MT * 16));                          \
        hipLaunchKernelGGL((score_count_kernel<T, KQv, TWO, STORE>), dim3(blocks), dim3(256), 0, st, ev, s0, s1, qtest, qval, n_cols); \
        break;                                                                                                  \
    }
    switch (ev.KQ) {
        SCORE_CASE(16) SCORE_CASE(32) SCORE_CASE(64) SCORE_CASE(128) SCORE_CASE(256)
    default: rec_set_error("unsupported query width"); return -1;
    }
#undef SCORE_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}

static unsigned blocks_for_waves(int64_t waves) { return (unsigned)((waves * 64 + 255) / 256); }

static int g_screen_on = 1, g_screen_cap = 0;   // cap 0: by the size of the catalog (see eval_impl)
static int g_screen_ranges = 0;         // ranges the screen cuts the catalog into (0: by the number of query blocks, see eval_impl)
void eval_configure_ranges(int ranges) { g_screen_ranges = ranges; }
static int g_screen_mode = 0;           // how the last full-catalog sweep ran: 0 fp64, 1 f32 screen, 2 screen overflowed and fp64 took over
static int64_t g_screen_open = 0;       // columns the last screen left to fp64, over all queries
void eval_configure_screen(int on, int cap) { if (on >= 0) g_screen_on = on; if (cap >= 0) g_screen_cap = cap; }   // on: 0 fp64 sweep, 1 f16 screen, 2 f32 screen
void eval_screen_stats(int* mode, int64_t* open) { *mode = g_screen_mode; *open = g_screen_open; }

static int screen_qblock(int KQ) { return KQ <= 16 ? 256 : KQ <= 128 ? 128 : 64; }

template <bool TWO>
static int launch_screen(const EvalDev& ev, const float* s0, const float* s1, int n_cols, const int32_t* qtest, const int32_t* qval, hipStream_t st)
{
#define SCREEN_CASE(KQv)                                                                                        \
    case KQv: {                                                                                                 \
        const unsigned blocks = (unsigned)(ev.nq_pad / (4 * ScreenShape<KQv>::MT * 16));                        \
        if (ev.K1 + ev.K2 == KQv && ev.K1 % 4 == 0 && ev.K2 % 4 == 0) hipLaunchKernelGGL((score_screen_kernel<KQv, TWO, true>), dim3(blocks, ev.ranges), dim3(256), 0, st, ev, s0, s1, qtest, qval, n_cols); \
        else hipLaunchKernelGGL((score_screen_kernel<KQv, TWO, false>), dim3(blocks, ev.ranges), dim3(256), 0, st, ev, s0, s1, qtest, qval, n_cols); \
        break;                                                                                                  \
    }
    switch (ev.KQ) {
        SCREEN_CASE(16) SCREEN_CASE(32) SCREEN_CASE(64) SCREEN_CASE(128) SCREEN_CASE(256)
    default: rec_set_error("unsupported query width"); return -1;
    }
#undef SCREEN_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}

template <bool TWO>
static int launch_screen16(const EvalDev& ev, int KQ16, const half_t* Th, const half_t* Tl, const float* bs, const half_t* Qh, const half_t* Ql,
                           const float* thr_s, const float* thr2_s, int n_cols, hipStream_t st)
{
#define SCREEN16_CASE(KQv)                                                                                      \
    case KQv: {                                                                                                 \
        const unsigned blocks = (unsigned)(ev.nq_pad / Screen16Shape<KQv, TWO>::QB);                            \
        hipLaunchKernelGGL((score_screen16_kernel<KQv, TWO>), dim3(blocks, ev.ranges), dim3(256), 0, st, ev, Th, Tl, bs, Qh, Ql, thr_s, thr2_s, n_cols); \
        break;                                                                                                  \
    }
    switch (KQ16) {
        SCREEN16_CASE(32) SCREEN16_CASE(64) SCREEN16_CASE(128) SCREEN16_CASE(256)
    default: rec_set_error("unsupported query width"); return -1;
    }
#undef SCREEN16_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T>
static int eval_impl(const DevModel& m, int32_t* counts_host, double* xpos_host, hipStream_t st, double* score_ms,
                     int64_t* score_launches, bool profile, const int32_t* subset_host, int n_subset, int user_draws, uint64_t draw_seed,
                     int32_t* draws_host)
{
    EvalDev ev = {};
    const int U = m.U, I = m.I;
    const int NC = n_subset > 0 ? n_subset : I;   // scored columns
    ev.n_cols = NC;
    ev.nctx = (m.model == REC_BPRMF || m.model == REC_MP) ? 1 : 2;
    ev.seg0 = ev.seg1 = -1; ev.K1 = ev.K2 = 0;
    switch (m.model) {
    case REC_MP: break;
    case REC_BPRMF: ev.seg0 = BPRMF_GI; ev.K1 = m.K; break;
    case REC_MC: ev.seg0 = MC_ETA; ev.K1 = m.K; break;
    case REC_FPMC: ev.seg0 = FPMC_GI; ev.K1 = m.K; ev.seg1 = FPMC_ETA; ev.K2 = m.KK; break;
    case REC_PRME: ev.seg0 = PRME_GI; ev.K1 = m.K; ev.seg1 = PRME_ETA; ev.K2 = m.KK; break;
    case REC_HRM_AVG: case REC_HRM_MAX: ev.seg0 = HRM_V; ev.K1 = m.K; break;
    case REC_TRANSREC: case REC_TRANSREC_L1: ev.seg0 = TR_H; ev.K1 = m.K; break;
    case REC_FOSSIL_SIMPLE: ev.seg0 = FS_V; ev.K1 = m.K; break;
    case REC_FOSSIL: ev.seg0 = FO_V; ev.K1 = m.K; break;
    case REC_REBUS: ev.seg0 = RB_P; ev.K1 = m.K; break;
    default: rec_set_error("evaluation is not available for this model"); return -1;
    }
    const int kw = ev.K1 + ev.K2;
    ev.KQ = kw <= 16 ? 16 : kw <= 32 ? 32 : kw <= 64 ? 64 : kw <= 128 ? 128 : 256;
    if (kw > 256) { rec_set_error("query width above 256 is not supported"); return -1; }
    const int mt = ev.KQ <= 32 ? 4 : ev.KQ <= 64 ? 2 : 1;
    const bool l1 = m.model == REC_TRANSREC_L1;
    constexpr int L1_QT = 32;
    // f32 tables, the whole catalog, a contraction: the f32 screen decides what it can prove, fp64 the rest
    const bool screen = g_screen_on && sizeof(T) == 4 && !l1 && n_subset <= 0 && kw > 0 && I < (1 << 28);
    const bool screen16 = screen && g_screen_on == 1;    // 1: the f16 screen (default); 2: the f32 screen
    const int KQ16 = std::max(32, ev.KQ);
    const int qblock = l1 ? L1_QT : screen16 ? 256 : screen ? std::max(4 * mt * 16, screen_qblock(ev.KQ)) : 4 * mt * 16;
    ev.nq = m.n_eval * ev.nctx;
    ev.nq_pad = ((ev.nq + qblock - 1) / qblock) * qblock;

    // host result defaults
    for (int64_t i = 0; i < (int64_t)U * 3; i++) counts_host[i] = -1;
    if (xpos_host) for (int64_t i = 0; i < (int64_t)U * 2; i++) xpos_host[i] = 0;
    if (ev.nq == 0) return 0;

    int32_t *qtest = nullptr, *qval = nullptr;
    std::vector<void*> owned;
    auto alloc = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, bytes ? bytes : 8); if (e == hipSuccess) owned.push_back(*p); return e; };
    auto release = [&]() { for (void* p : owned) (void)hipFree(p); };
#define EV_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { rec_set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); release(); return -2; } } while (0)
    EV_TRY(alloc((void**)&ev.Q, sizeof(double) * (size_t)ev.nq_pad * ev.KQ));
    EV_TRY(alloc((void**)&ev.thr, sizeof(double) * ev.nq_pad));
    EV_TRY(alloc((void**)&ev.thr2, sizeof(double) * ev.nq_pad));
    EV_TRY(alloc((void**)&ev.cnt, sizeof(int32_t) * ev.nq_pad));
    EV_TRY(alloc((void**)&ev.cnt2, sizeof(int32_t) * ev.nq_pad));
    EV_TRY(alloc((void**)&ev.bconst, sizeof(double) * I));
    EV_TRY(alloc((void**)&ev.counts, sizeof(int32_t) * (size_t)U * 3));
    EV_TRY(alloc((void**)&ev.xpos, sizeof(double) * (size_t)U * 2));
    EV_TRY(alloc((void**)&qtest, sizeof(int32_t) * ev.nq_pad));
    EV_TRY(alloc((void**)&qval, sizeof(int32_t) * ev.nq_pad));
    EV_TRY(hipMemsetAsync(ev.Q, 0, sizeof(double) * (size_t)ev.nq_pad * ev.KQ, st));
    EV_TRY(hipMemsetAsync(ev.xpos, 0, sizeof(double) * (size_t)U * 2, st));
    // padding rows never win a comparison
    hipLaunchKernelGGL(fill_f64_kernel, dim3(256), dim3(256), 0, st, ev.thr, (int64_t)ev.nq_pad, -INFINITY);
    hipLaunchKernelGGL(fill_f64_kernel, dim3(256), dim3(256), 0, st, ev.thr2, (int64_t)ev.nq_pad, -INFINITY);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(256), dim3(256), 0, st, ev.counts, (int64_t)U * 3, -1);
    hipLaunchKernelGGL(query_items_kernel, dim3((ev.nq_pad + 255) / 256), dim3(256), 0, st, m, ev, qtest, qval);
    hipLaunchKernelGGL(item_consts_kernel<T>, dim3(blocks_for_waves(I)), dim3(256), 0, st, m, ev);
    hipLaunchKernelGGL(build_queries_kernel<T>, dim3(blocks_for_waves(ev.nq)), dim3(256), 0, st, m, ev);
    EV_TRY(hipGetLastError());
    ev.col_b = ev.bconst;
    if (user_draws > 0) {
        // per-user candidate draws: no sweep, one wavefront per user scores its own draws
        int32_t* d_draws = nullptr;
        if (draws_host) {
            EV_TRY(alloc((void**)&d_draws, sizeof(int32_t) * (size_t)U * user_draws));
            EV_TRY(hipMemsetAsync(d_draws, 0xff, sizeof(int32_t) * (size_t)U * user_draws, st));
        }
        hipEvent_t s0 = nullptr, s1 = nullptr;
        if (profile) { EV_TRY(hipEventCreate(&s0)); EV_TRY(hipEventCreate(&s1)); EV_TRY(hipEventRecord(s0, st)); }
        hipLaunchKernelGGL(sample_count_kernel<T>, dim3(blocks_for_waves(m.n_eval)), dim3(256), 0, st, m, ev, user_draws, draw_seed, d_draws);
        EV_TRY(hipGetLastError());
        if (profile) EV_TRY(hipEventRecord(s1, st));
        EV_TRY(hipMemcpyAsync(counts_host, ev.counts, sizeof(int32_t) * (size_t)U * 3, hipMemcpyDeviceToHost, st));
        if (xpos_host) EV_TRY(hipMemcpyAsync(xpos_host, ev.xpos, sizeof(double) * (size_t)U * 2, hipMemcpyDeviceToHost, st));
        if (draws_host) EV_TRY(hipMemcpyAsync(draws_host, d_draws, sizeof(int32_t) * (size_t)U * user_draws, hipMemcpyDeviceToHost, st));
        EV_TRY(hipStreamSynchronize(st));
        if (profile) {
            float ms = 0; EV_TRY(hipEventElapsedTime(&ms, s0, s1));
            *score_ms += ms; *score_launches += 1;
            (void)hipEventDestroy(s0); (void)hipEventDestroy(s1);
        }
        release();
        return 0;
    }
    const T* seg0_cols = ev.seg0 >= 0 ? (const T*)m.tab[ev.seg0] : nullptr;
    const T* seg1_cols = ev.seg1 >= 0 ? (const T*)m.tab[ev.seg1] : nullptr;
    if (n_subset > 0) {
        // the sampled columns as compact tables: rows of the item segments and b gathered by item id
        std::vector<int32_t> mult(I, 0);
        for (int i = 0; i < n_subset; i++) {
            if (subset_host[i] < 0 || subset_host[i] >= I) { rec_set_error("sampled item out of range"); release(); return -1; }
            mult[subset_host[i]]++;
        }
        int32_t *d_ids = nullptr, *d_mult = nullptr; double* d_b = nullptr;
        EV_TRY(alloc((void**)&d_ids, sizeof(int32_t) * n_subset));
        EV_TRY(alloc((void**)&d_mult, sizeof(int32_t) * I));
        EV_TRY(alloc((void**)&d_b, sizeof(double) * n_subset));
        EV_TRY(hipMemcpyAsync(d_ids, subset_host, sizeof(int32_t) * n_subset, hipMemcpyHostToDevice, st));
        EV_TRY(hipMemcpyAsync(d_mult, mult.data(), sizeof(int32_t) * I, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gather_rows_kernel<double>, dim3(64), dim3(256), 0, st, (const double*)ev.bconst, (const int32_t*)d_ids, (int64_t)n_subset, 1, d_b);
        if (seg0_cols) {
            T* g = nullptr; EV_TRY(alloc((void**)&g, sizeof(T) * (size_t)n_subset * ev.K1));
            hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(256), dim3(256), 0, st, seg0_cols, (const int32_t*)d_ids, (int64_t)n_subset, ev.K1, g);
            seg0_cols = g;
        }
        if (seg1_cols) {
            T* g = nullptr; EV_TRY(alloc((void**)&g, sizeof(T) * (size_t)n_subset * ev.K2));
            hipLaunchKernelGGL(gather_rows_kernel<T>, dim3(256), dim3(256), 0, st, seg1_cols, (const int32_t*)d_ids, (int64_t)n_subset, ev.K2, g);
            seg1_cols = g;
        }
        EV_TRY(hipStreamSynchronize(st));   // mult (host vector) has been copied
        ev.col_item = d_ids; ev.item_mult = d_mult; ev.col_b = d_b;
    }

    unsigned* bounds = nullptr;
    if (!l1 && n_subset <= 0) { g_screen_mode = 0; g_screen_open = 0; }
    if (screen) {
        // Undecided columns per query grow with the catalog (margin x columns per unit of score); a trained model leaves 10-25
        // per query at 200k items, a few users with histories of thousands of clicks some hundreds.  The lists are real
        // memory (nq x cap x 4 bytes: 3 GB for 2M queries at 384), so they are sized for the bulk - I / 512 columns within
        // 128 .. 1024 for the f16 screen, max(64, I / 2048) for the f32 screen - and the few queries that need more are swept
        // again on their own in fp64.  If the allocation fails the lists are halved (down to 32 columns) before the screen
        // is given up for the fp64 sweep.
        if (screen16) ev.amb_cap = g_screen_cap > 0 ? g_screen_cap : (int)std::min<int64_t>(1024, std::max<int64_t>(128, ((I / 512 + 31) / 32) * 32));
        else ev.amb_cap = g_screen_cap > 0 ? g_screen_cap : std::max(64, ((I / 2048 + 31) / 32) * 32);
        EV_TRY(alloc((void**)&ev.qmar, sizeof(float) * ev.nq_pad));
        while (alloc((void**)&ev.amb, sizeof(int32_t) * (size_t)ev.nq_pad * ev.amb_cap) != hipSuccess) {
            (void)hipGetLastError();
            if (ev.amb_cap <= 32) { rec_set_error("no device memory for the screen's lists of undecided columns"); release(); return -2; }
            ev.amb_cap = std::max(32, ev.amb_cap / 2);
        }
        EV_TRY(alloc((void**)&ev.amb_cnt, sizeof(int32_t) * ev.nq_pad));
        EV_TRY(alloc((void**)&ev.amb_overflow, sizeof(int32_t) * 2));   // [0] a list overflowed, [1] operands the screen cannot take
        EV_TRY(alloc((void**)&bounds, sizeof(unsigned) * 2));
        EV_TRY(hipMemsetAsync(ev.amb_overflow, 0, sizeof(int32_t) * 2, st));
        EV_TRY(hipMemsetAsync(bounds, 0, sizeof(unsigned) * 2, st));
        // A block keeps its queries for a sweep of the catalog, so a launch has nq_pad / qblock blocks: with few queries (a
        // rank's share of the users, a small corpus) they do not fill the chip evenly - 977 blocks on 768 resident places run
        // as one full round and a quarter-full one.  The catalog is then cut into ranges (grid.y) until there are about 8
        // rounds of blocks; a query's blocks add their counts and share its list through integer atomics.
        {
            int dev = 0, cus = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            const int64_t resident = (int64_t)cus * ((screen16 && !(KQ16 <= 64 && ev.nctx == 1)) ? 2 : 3), blocks = ev.nq_pad / (screen16 ? screen16_qb(KQ16, ev.nctx == 1) : qblock);
            const int tile = 128;                                       // a multiple of every tile shape's item count
            int ranges = g_screen_ranges > 0 ? std::min(g_screen_ranges, (I + tile - 1) / tile)
                                             : (int)std::min<int64_t>(std::min<int64_t>(16, (8 * resident + blocks - 1) / blocks), (I + 4095) / 4096);   // at least 4096 items per range
            ranges = std::max(1, ranges);
            ev.range_items = (int)((((int64_t)I + ranges - 1) / ranges + tile - 1) / tile * tile);
            ev.ranges = (I + ev.range_items - 1) / ev.range_items;
            if (ev.ranges > 1) {
                EV_TRY(hipMemsetAsync(ev.cnt, 0, sizeof(int32_t) * ev.nq_pad, st));
                EV_TRY(hipMemsetAsync(ev.cnt2, 0, sizeof(int32_t) * ev.nq_pad, st));
                EV_TRY(hipMemsetAsync(ev.amb_cnt, 0, sizeof(int32_t) * ev.nq_pad, st));
            }
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profile) { EV_TRY(hipEventCreate(&e0)); EV_TRY(hipEventCreate(&e1)); EV_TRY(hipEventRecord(e0, st)); }
    int rc = 0;
    // the sweep on the FP32 matrix pipe ("eval_screen" 2, and the second try when the f16 screen's wider margin has
    // overflowed a query's list)
    auto run_f32_screen = [&]() -> int {
        hipLaunchKernelGGL(screen_item_bounds_kernel, dim3(std::min(1024u, blocks_for_waves(I))), dim3(256), 0, st, ev, (const float*)seg0_cols, (const float*)seg1_cols, I, bounds);
        ev.heldout_counted = 1;
        hipLaunchKernelGGL(screen_margins_kernel, dim3(blocks_for_waves(ev.nq_pad)), dim3(256), 0, st, ev, (const unsigned*)bounds);
        return ev.nctx == 1 ? launch_screen<true>(ev, (const float*)seg0_cols, (const float*)seg1_cols, NC, qtest, qval, st)
                            : launch_screen<false>(ev, (const float*)seg0_cols, (const float*)seg1_cols, NC, qtest, qval, st);
    };
    if (l1) {
        // the integer screen (whole catalog, K <= 160: the 64-item tile, either table precision); "eval_screen" 0 or
        // operands it cannot take (not finite), or a query with more undecided columns than its list holds: the fp64 sweep
        bool held = false;
        if (g_screen_on && n_subset <= 0 && m.K <= 160 && I < (1 << 28)) {
            uint32_t *Hi = nullptr, *Qi = nullptr; double *Bj = nullptr, *Tq = nullptr; unsigned* bnd = nullptr;
            ev.amb_cap = g_screen_cap > 0 ? g_screen_cap : 64;
            EV_TRY(alloc((void**)&Hi, sizeof(uint32_t) * (size_t)I * m.K));
            EV_TRY(alloc((void**)&Qi, sizeof(uint32_t) * (size_t)ev.nq_pad * m.K));
            EV_TRY(alloc((void**)&Bj, sizeof(double) * (size_t)I));
            EV_TRY(alloc((void**)&Tq, sizeof(double) * (size_t)ev.nq_pad));
            EV_TRY(alloc((void**)&bnd, sizeof(unsigned)));
            EV_TRY(alloc((void**)&ev.amb, sizeof(int32_t) * (size_t)ev.nq_pad * ev.amb_cap));
            EV_TRY(alloc((void**)&ev.amb_cnt, sizeof(int32_t) * ev.nq_pad));
            EV_TRY(alloc((void**)&ev.amb_overflow, sizeof(int32_t) * 2));
            EV_TRY(hipMemsetAsync(ev.amb_overflow, 0, sizeof(int32_t) * 2, st));
            EV_TRY(hipMemsetAsync(bnd, 0, sizeof(unsigned), st));
            const int f64 = sizeof(T) == 8;
            hipLaunchKernelGGL(l1_absmax_kernel, dim3(1024), dim3(256), 0, st, (const void*)seg0_cols, f64, (int64_t)I * m.K, (const double*)ev.Q,
                               (int64_t)ev.nq_pad * ev.KQ, bnd);
            hipLaunchKernelGGL(l1_quant_items_kernel, dim3(2048), dim3(256), 0, st, (const void*)seg0_cols, f64, I, m.K, (const double*)ev.col_b,
                               (const unsigned*)bnd, Hi, Bj, ev.amb_overflow);
            hipLaunchKernelGGL(l1_quant_queries_kernel, dim3(2048), dim3(256), 0, st, ev, m.K, (const unsigned*)bnd, Qi, Tq);
            const size_t lds_bytes = sizeof(uint32_t) * (size_t)64 * (m.K | 1);
            hipLaunchKernelGGL((score_count_l1_sad_kernel<L1_QT, false>), dim3((unsigned)(ev.nq_pad / L1_QT)), dim3(256), lds_bytes, st, ev, (const uint32_t*)Hi,
                               (const uint32_t*)Qi, (const double*)Bj, (const double*)Tq, I, m.K, 1.0002 * m.K + 2.0, 0);
            EV_TRY(hipGetLastError());
            int32_t over2[2] = {0, 0};
            EV_TRY(hipMemcpyAsync(over2, ev.amb_overflow, sizeof over2, hipMemcpyDeviceToHost, st));
            EV_TRY(hipStreamSynchronize(st));
            held = !(over2[0] | over2[1]);
            g_screen_mode = held ? 1 : 2; g_screen_open = 0;
            if (held) {
                ev.heldout_counted = 1;
                if (profile) {
                    std::vector<int32_t> open(ev.nq);
                    EV_TRY(hipMemcpy(open.data(), ev.amb_cnt, sizeof(int32_t) * ev.nq, hipMemcpyDeviceToHost));
                    for (int32_t x : open) g_screen_open += x > 0 ? x : 0;
                }
            } else {
                ev.amb = nullptr; ev.heldout_counted = 0;
            }
        } else if (n_subset <= 0) { g_screen_mode = 0; g_screen_open = 0; }
        if (!held) rc = launch_score_l1<T, 0>(ev, seg0_cols, qtest, qval, NC, m.K, st);
    } else if (screen16) {
        // operands as pairs of f16 halves under two power-of-two scales, margins, sweep on the 16-bit matrix pipe
        const int NI16 = KQ16 <= 128 ? 64 : 32;
        const int I_pad = ((I + NI16 - 1) / NI16) * NI16;
        half_t *Th = nullptr, *Tl = nullptr, *Qh = nullptr, *Ql = nullptr; float *bs = nullptr, *thr_s = nullptr, *thr2_s = nullptr; unsigned* b4 = nullptr;
        EV_TRY(alloc((void**)&Th, sizeof(half_t) * (size_t)I_pad * KQ16));
        EV_TRY(alloc((void**)&Tl, sizeof(half_t) * (size_t)I_pad * KQ16));
        EV_TRY(alloc((void**)&bs, sizeof(float) * (size_t)I_pad));
        EV_TRY(alloc((void**)&Qh, sizeof(half_t) * (size_t)ev.nq_pad * KQ16));
        EV_TRY(alloc((void**)&Ql, sizeof(half_t) * (size_t)ev.nq_pad * KQ16));
        EV_TRY(alloc((void**)&thr_s, sizeof(float) * ev.nq_pad));
        EV_TRY(alloc((void**)&thr2_s, sizeof(float) * ev.nq_pad));
        EV_TRY(alloc((void**)&b4, sizeof(unsigned) * 4));
        EV_TRY(hipMemsetAsync(b4, 0, sizeof(unsigned) * 4, st));
        hipLaunchKernelGGL(screen16_bounds_kernel, dim3(1024), dim3(256), 0, st, ev, (const float*)seg0_cols, (const float*)seg1_cols, I, b4);
        hipLaunchKernelGGL(screen16_items_kernel, dim3(2048), dim3(256), 0, st, ev, (const float*)seg0_cols, (const float*)seg1_cols, I, I_pad, KQ16,
                           (const unsigned*)b4, Th, Tl, bs, ev.amb_overflow, 3.0e38f, (const int32_t*)nullptr, true);
        hipLaunchKernelGGL(screen16_queries_kernel, dim3(blocks_for_waves(ev.nq_pad)), dim3(256), 0, st, ev, KQ16, (const unsigned*)b4, Qh, Ql, thr_s, thr2_s,
                           ev.amb_overflow);
        ev.heldout_counted = 1;
        rc = ev.nctx == 1 ? launch_screen16<true>(ev, KQ16, Th, Tl, bs, Qh, Ql, thr_s, thr2_s, NC, st)
                          : launch_screen16<false>(ev, KQ16, Th, Tl, bs, Qh, Ql, thr_s, thr2_s, NC, st);
    } else if (screen) {
        rc = run_f32_screen();
    } else {
        rc = ev.nctx == 1 ? launch_score<T, true, false>(ev, seg0_cols, seg1_cols, NC, qtest, qval, st)
                          : launch_score<T, false, false>(ev, seg0_cols, seg1_cols, NC, qtest, qval, st);
    }
    if (rc) { release(); return rc; }
    if (screen) {
        // A query with more undecided columns than its list holds (a user with a history of thousands of clicks, tables
        // with huge norms, a model that has diverged).  A few such queries are swept again on their own in fp64; many send
        // the whole sweep to the next narrower margin (f16 -> f32 screen -> fp64); operands the screen cannot take (not
        // finite, out of range) send it to fp64 directly.
        int32_t over2[2] = {0, 0};
        EV_TRY(hipMemcpyAsync(over2, ev.amb_overflow, sizeof over2, hipMemcpyDeviceToHost, st));
        EV_TRY(hipStreamSynchronize(st));
        int32_t over = over2[0] | over2[1];
        bool second_try = false, partial = false;
        if (getenv("RECSYS_DEBUG_SCREEN")) {
            std::vector<int32_t> ac(ev.nq);
            (void)hipMemcpy(ac.data(), ev.amb_cnt, sizeof(int32_t) * ev.nq, hipMemcpyDeviceToHost);
            int64_t tot = 0, nover = 0; int mx = 0, arg = 0;
            for (int q = 0; q < ev.nq; q++) { tot += ac[q]; nover += ac[q] > ev.amb_cap; if (ac[q] > mx) { mx = ac[q]; arg = q; } }
            fprintf(stderr, "[screen] flags %d %d, cap %d, undecided total %lld, queries over %lld of %d, max %d at query %d\n", over2[0], over2[1], ev.amb_cap,
                    (long long)tot, (long long)nover, ev.nq, mx, arg);
        }
        auto redo_overflowing_in_fp64 = [&](bool* done) -> int {
            *done = false;
            std::vector<int32_t> ac(ev.nq);
            EV_TRY(hipMemcpy(ac.data(), ev.amb_cnt, sizeof(int32_t) * ev.nq, hipMemcpyDeviceToHost));
            std::vector<int32_t> idx;
            for (int q = 0; q < ev.nq; q++) if (ac[q] > ev.amb_cap) idx.push_back(q);
            if (idx.empty() || (int64_t)idx.size() * 8 > ev.nq) return 0;
            const int qb64 = 4 * mt * 16;
            EvalDev e2 = ev;
            e2.nq = (int)idx.size(); e2.nq_pad = ((e2.nq + qb64 - 1) / qb64) * qb64;
            e2.amb = nullptr; e2.ranges = 1;
            int32_t *d_idx = nullptr, *sub_test = nullptr, *sub_val = nullptr;
            EV_TRY(alloc((void**)&d_idx, sizeof(int32_t) * e2.nq));
            EV_TRY(alloc((void**)&sub_test, sizeof(int32_t) * e2.nq_pad));
            EV_TRY(alloc((void**)&sub_val, sizeof(int32_t) * e2.nq_pad));
            EV_TRY(alloc((void**)&e2.Q, sizeof(double) * (size_t)e2.nq_pad * ev.KQ));
            EV_TRY(alloc((void**)&e2.thr, sizeof(double) * e2.nq_pad));
            EV_TRY(alloc((void**)&e2.thr2, sizeof(double) * e2.nq_pad));
            EV_TRY(alloc((void**)&e2.cnt, sizeof(int32_t) * e2.nq_pad));
            EV_TRY(alloc((void**)&e2.cnt2, sizeof(int32_t) * e2.nq_pad));
            EV_TRY(hipMemcpyAsync(d_idx, idx.data(), sizeof(int32_t) * e2.nq, hipMemcpyHostToDevice, st));
            EV_TRY(hipMemsetAsync(sub_test, 0xff, sizeof(int32_t) * e2.nq_pad, st));
            EV_TRY(hipMemsetAsync(sub_val, 0xff, sizeof(int32_t) * e2.nq_pad, st));
            EV_TRY(hipMemsetAsync(e2.Q, 0, sizeof(double) * (size_t)e2.nq_pad * ev.KQ, st));
            hipLaunchKernelGGL(fill_f64_kernel, dim3(64), dim3(256), 0, st, e2.thr, (int64_t)e2.nq_pad, -INFINITY);
            hipLaunchKernelGGL(fill_f64_kernel, dim3(64), dim3(256), 0, st, e2.thr2, (int64_t)e2.nq_pad, -INFINITY);
            hipLaunchKernelGGL(gather_queries_kernel, dim3(blocks_for_waves(e2.nq)), dim3(256), 0, st, ev, e2, (const int32_t*)d_idx, (const int32_t*)qtest,
                               (const int32_t*)qval, sub_test, sub_val);
            // the fp64 sweep never counts the held-out pair (the positive's own column would be compared with a sum formed in
            // another order than the threshold's): these queries are marked, and correct_clicked_kernel leaves the pair alone
            int rc2 = ev.nctx == 1 ? launch_score<T, true, false>(e2, seg0_cols, seg1_cols, NC, sub_test, sub_val, st)
                                   : launch_score<T, false, false>(e2, seg0_cols, seg1_cols, NC, sub_test, sub_val, st);
            if (rc2) return rc2;
            hipLaunchKernelGGL(scatter_counts_kernel, dim3((e2.nq + 255) / 256), dim3(256), 0, st, ev, e2, (const int32_t*)d_idx);
            EV_TRY(hipGetLastError());
            EV_TRY(hipStreamSynchronize(st));      // idx (host vector) has been copied
            *done = true;
            return 0;
        };
        if (over2[0] && !over2[1]) {
            rc = redo_overflowing_in_fp64(&partial);
            if (rc) { release(); return rc; }
            if (partial) over = 0;
        }
        if (over && screen16 && !over2[1]) {
            // the f32 screen's margin is less than half as wide: it may hold where the f16 screen did not, at a quarter of
            // the fp64 sweep's time
            second_try = true;
            EV_TRY(hipMemsetAsync(ev.amb_overflow, 0, sizeof(int32_t) * 2, st));
            EV_TRY(hipMemsetAsync(bounds, 0, sizeof(unsigned) * 2, st));
            EV_TRY(hipMemsetAsync(ev.amb_cnt, 0, sizeof(int32_t) * ev.nq_pad, st));
            if (ev.ranges > 1) {
                EV_TRY(hipMemsetAsync(ev.cnt, 0, sizeof(int32_t) * ev.nq_pad, st));
                EV_TRY(hipMemsetAsync(ev.cnt2, 0, sizeof(int32_t) * ev.nq_pad, st));
            }
            rc = run_f32_screen();
            if (rc) { release(); return rc; }
            EV_TRY(hipMemcpyAsync(over2, ev.amb_overflow, sizeof over2, hipMemcpyDeviceToHost, st));
            EV_TRY(hipStreamSynchronize(st));
            over = over2[0] | over2[1];
            if (over2[0] && !over2[1]) {
                rc = redo_overflowing_in_fp64(&partial);
                if (rc) { release(); return rc; }
                if (partial) over = 0;
            }
        }
        g_screen_mode = over ? 2 : partial ? 4 : second_try ? 3 : 1; g_screen_open = 0;
        if (profile) {
            std::vector<int32_t> open(ev.nq);
            EV_TRY(hipMemcpy(open.data(), ev.amb_cnt, sizeof(int32_t) * ev.nq, hipMemcpyDeviceToHost));
            for (int32_t x : open) g_screen_open += x > 0 ? x : 0;
        }
        if (over) {
            ev.amb = nullptr; ev.heldout_counted = 0;
            rc = ev.nctx == 1 ? launch_score<T, true, false>(ev, seg0_cols, seg1_cols, NC, qtest, qval, st)
                              : launch_score<T, false, false>(ev, seg0_cols, seg1_cols, NC, qtest, qval, st);
            if (rc) { release(); return rc; }
        }
    }
    if (profile) EV_TRY(hipEventRecord(e1, st));
    hipLaunchKernelGGL(correct_clicked_kernel<T>, dim3(blocks_for_waves(ev.nq)), dim3(256), 0, st, m, ev);
    EV_TRY(hipGetLastError());
    EV_TRY(hipMemcpyAsync(counts_host, ev.counts, sizeof(int32_t) * (size_t)U * 3, hipMemcpyDeviceToHost, st));
    if (xpos_host) EV_TRY(hipMemcpyAsync(xpos_host, ev.xpos, sizeof(double) * (size_t)U * 2, hipMemcpyDeviceToHost, st));
    EV_TRY(hipStreamSynchronize(st));
    if (profile) {
        float ms = 0; EV_TRY(hipEventElapsedTime(&ms, e0, e1));
        *score_ms += ms; *score_launches += 1;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    release();
#undef EV_TRY
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Top-k lists.  The relative scores of a batch of queries are materialised ([rows][I] f64 in HBM),
// clicked / held-out items are masked, the context's positive enters with the score the reference
// formula gives it, and one block per row selects the k largest by an 8-pass radix select on
// order-preserving 64-bit keys, then orders them (score descending, item id ascending on ties).

__global__ void __launch_bounds__(256) topk_mask_kernel(DevModel m, EvalDev ev)
{
    const int lane = threadIdx.x & 63;
    const int w = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= ev.nq) return;
    const int u = m.eval_users[w >> 1]; const int ctx = w & 1;
    double* row = ev.S + (int64_t)w * m.I;
    const int32_t* clk = m.clk_item + m.clk_off[u]; const int nc = (int)(m.clk_off[u + 1] - m.clk_off[u]);
    // every address is written by exactly one lane: the positive is skipped here and set by lane 0
    const int val = m.val_item[u], test = m.test_item[u];
    const int pos = ctx ? test : val, other = ctx ? val : test;
    for (int i = lane; i < nc; i += 64) { const int j = clk[i]; if (j != pos && j != other) row[j] = -INFINITY; }
    if (lane == 0) { if (other != pos) row[other] = -INFINITY; row[pos] = ev.thr[w]; }
}

__global__ void __launch_bounds__(256) topk_select_kernel(EvalDev ev, int I, int k, int32_t* __restrict__ items_out, double* __restrict__ scores_out)
{
    __shared__ unsigned hist[256];
    __shared__ uint64_t s_prefix; __shared__ int s_need;
    __shared__ int s_gt, s_eq, s_wave[4];
    __shared__ double sel_s[256]; __shared__ int sel_i[256];
    const int tid = threadIdx.x, row = blockIdx.x;
    const double* S = ev.S + (int64_t)row * I;
    if (tid == 0) { s_prefix = 0; s_need = k; }
    __syncthreads();
    // radix select: after pass p the top 8(p+1) bits of the k-th largest key are known
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        const uint64_t prefix = s_prefix; const uint64_t mask = pass == 0 ? 0 : (~0ull << (shift + 8));
        for (int j = tid; j < I; j += 256) {
            const uint64_t key = order_key(S[j]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int need = s_need, d = 255;
            for (; d > 0; d--) { if ((int)hist[d] >= need) break; need -= (int)hist[d]; }
            s_prefix = prefix | ((uint64_t)d << shift); s_need = need;
        }
        __syncthreads();
    }
    const uint64_t T = s_prefix; const int quota_eq = s_need;   // keys > T are all taken, the first quota_eq keys == T by item id
    if (tid == 0) { s_gt = 0; s_eq = 0; }
    for (int i = tid; i < 256; i += 256) { sel_s[i] = -INFINITY; sel_i[i] = -1; }
    __syncthreads();
    const int n_gt = k - quota_eq;
    for (int j0 = 0; j0 < I; j0 += 256) {
        const int j = j0 + tid;
        const double sc = j < I ? S[j] : -INFINITY;
        const uint64_t key = order_key(sc);
        const bool gt = j < I && key > T, eq = j < I && key == T;
        if (gt) { const int p = atomicAdd(&s_gt, 1); sel_s[p] = sc; sel_i[p] = j; }
        // ordered compaction of the ties
        const unsigned long long bal = __ballot(eq);
        const int wv = tid >> 6, ln = tid & 63;
        if (ln == 0) s_wave[wv] = __popcll(bal);
        __syncthreads();
        int before = __popcll(bal & ((1ull << ln) - 1));
        for (int q = 0; q < wv; q++) before += s_wave[q];
        const int base = s_eq;
        if (eq && base + before < quota_eq) { sel_s[n_gt + base + before] = sc; sel_i[n_gt + base + before] = j; }
        __syncthreads();
        if (tid == 0) s_eq = base + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    // order the k selected entries
    if (tid < k) {
        const double si = sel_s[tid]; const int ii = sel_i[tid];
        int rank = 0;
        for (int q = 0; q < k; q++) rank += (sel_s[q] > si || (sel_s[q] == si && (sel_i[q] < ii || (sel_i[q] == ii && q < tid)))) ? 1 : 0;
        const bool real = si > -INFINITY;
        items_out[(int64_t)row * k + rank] = real ? ii : -1;
        scores_out[(int64_t)row * k + rank] = real ? si + ev.aq[row] : -INFINITY;
    }
}

template <class T>
static int topk_impl(const DevModel& m0, const int32_t* users_host, int n_users, int k, int32_t* items_host, double* scores_host,
                     hipStream_t st)
{
    if (k < 1 || k > 256) { rec_set_error("k must be in 1..256"); return -1; }
    DevModel m = m0;
    const int I = m.I;
    EvalDev ev = {};
    ev.nctx = 2; ev.force2 = 1;
    ev.seg0 = ev.seg1 = -1; ev.K1 = ev.K2 = 0;
    switch (m.model) {
    case REC_MP: break;
    case REC_BPRMF: ev.seg0 = BPRMF_GI; ev.K1 = m.K; break;
    case REC_MC: ev.seg0 = MC_ETA; ev.K1 = m.K; break;
    case REC_FPMC: ev.seg0 = FPMC_GI; ev.K1 = m.K; ev.seg1 = FPMC_ETA; ev.K2 = m.KK; break;
    case REC_PRME: ev.seg0 = PRME_GI; ev.K1 = m.K; ev.seg1 = PRME_ETA; ev.K2 = m.KK; break;
    case REC_HRM_AVG: case REC_HRM_MAX: ev.seg0 = HRM_V; ev.K1 = m.K; break;
    case REC_TRANSREC: case REC_TRANSREC_L1: ev.seg0 = TR_H; ev.K1 = m.K; break;
    case REC_FOSSIL_SIMPLE: ev.seg0 = FS_V; ev.K1 = m.K; break;
    case REC_FOSSIL: ev.seg0 = FO_V; ev.K1 = m.K; break;
    case REC_REBUS: ev.seg0 = RB_P; ev.K1 = m.K; break;
    default: rec_set_error("top-k is not available for this model"); return -1;
    }
    const int kw = ev.K1 + ev.K2;
    if (kw > 256) { rec_set_error("query width above 256 is not supported"); return -1; }
    ev.KQ = kw <= 16 ? 16 : kw <= 32 ? 32 : kw <= 64 ? 64 : kw <= 128 ? 128 : 256;
    const bool l1 = m.model == REC_TRANSREC_L1;
    constexpr int L1_QT = 32;
    const int mt = ev.KQ <= 32 ? 4 : ev.KQ <= 64 ? 2 : 1;
    const int qblock = l1 ? L1_QT : 4 * mt * 16;
    // rows of one batch: a quarter of the free device memory in scores, within 2 .. 16 GiB (a batch of a few hundred rows
    // leaves most of the chip idle: the scoring kernels launch one block per 32 .. 128 rows)
    size_t free_b = 0, total_b = 0; (void)hipMemGetInfo(&free_b, &total_b);
    const int64_t score_bytes = std::min<int64_t>(std::max<int64_t>((int64_t)(free_b / 4), (int64_t)1 << 31), (int64_t)1 << 34);
    int batch_users = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(4096, n_users), score_bytes / ((int64_t)I * 8) / 2));
    std::vector<void*> owned;
    auto alloc = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, bytes ? bytes : 8); if (e == hipSuccess) owned.push_back(*p); return e; };
    auto release = [&]() { for (void* p : owned) (void)hipFree(p); };
#define TK_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { rec_set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); release(); return -2; } } while (0)
    const int max_rows = ((2 * batch_users + qblock - 1) / qblock) * qblock;
    int32_t *d_users = nullptr, *qtest = nullptr, *qval = nullptr, *d_items = nullptr; double* d_scores = nullptr;
    TK_TRY(alloc((void**)&d_users, sizeof(int32_t) * batch_users));
    TK_TRY(alloc((void**)&ev.Q, sizeof(double) * (size_t)max_rows * ev.KQ));
    TK_TRY(alloc((void**)&ev.thr, sizeof(double) * max_rows));
    TK_TRY(alloc((void**)&ev.thr2, sizeof(double) * max_rows));
    TK_TRY(alloc((void**)&ev.aq, sizeof(double) * max_rows));
    TK_TRY(alloc((void**)&ev.cnt, sizeof(int32_t) * max_rows));
    TK_TRY(alloc((void**)&ev.bconst, sizeof(double) * I));
    TK_TRY(alloc((void**)&ev.xpos, sizeof(double) * (size_t)m.U * 2));
    TK_TRY(alloc((void**)&ev.S, sizeof(double) * (size_t)max_rows * I));
    TK_TRY(alloc((void**)&qtest, sizeof(int32_t) * max_rows));
    TK_TRY(alloc((void**)&qval, sizeof(int32_t) * max_rows));
    TK_TRY(alloc((void**)&d_items, sizeof(int32_t) * (size_t)max_rows * k));
    TK_TRY(alloc((void**)&d_scores, sizeof(double) * (size_t)max_rows * k));
    ev.cnt2 = ev.cnt; ev.counts = nullptr; ev.col_b = ev.bconst; ev.n_cols = I;
    hipLaunchKernelGGL(item_consts_kernel<T>, dim3(blocks_for_waves(I)), dim3(256), 0, st, m, ev);
    for (int u0 = 0; u0 < n_users; u0 += batch_users) {
        const int nu = std::min(batch_users, n_users - u0);
        TK_TRY(hipMemcpyAsync(d_users, users_host + u0, sizeof(int32_t) * nu, hipMemcpyHostToDevice, st));
        m.eval_users = d_users; m.n_eval = nu;
        ev.nq = 2 * nu; ev.nq_pad = ((ev.nq + qblock - 1) / qblock) * qblock;
        TK_TRY(hipMemsetAsync(ev.Q, 0, sizeof(double) * (size_t)ev.nq_pad * ev.KQ, st));
        hipLaunchKernelGGL(query_items_kernel, dim3((ev.nq_pad + 255) / 256), dim3(256), 0, st, m, ev, qtest, qval);
        hipLaunchKernelGGL(build_queries_kernel<T>, dim3(blocks_for_waves(ev.nq)), dim3(256), 0, st, m, ev);
        int rc = 0;
        if (l1) {
            rc = launch_score_l1<T, 1>(ev, (const T*)m.tab[TR_H], qtest, qval, I, m.K, st);
        } else {
            rc = launch_score<T, false, true>(ev, ev.seg0 >= 0 ? (const T*)m.tab[ev.seg0] : nullptr, ev.seg1 >= 0 ? (const T*)m.tab[ev.seg1] : nullptr, I, qtest, qval, st);
        }
        if (rc) { release(); return rc; }
        hipLaunchKernelGGL(topk_mask_kernel, dim3(blocks_for_waves(ev.nq)), dim3(256), 0, st, m, ev);
        hipLaunchKernelGGL(topk_select_kernel, dim3(ev.nq), dim3(256), 0, st, ev, I, k, d_items, d_scores);
        TK_TRY(hipGetLastError());
        TK_TRY(hipMemcpyAsync(items_host + (size_t)u0 * 2 * k, d_items, sizeof(int32_t) * (size_t)ev.nq * k, hipMemcpyDeviceToHost, st));
        TK_TRY(hipMemcpyAsync(scores_host + (size_t)u0 * 2 * k, d_scores, sizeof(double) * (size_t)ev.nq * k, hipMemcpyDeviceToHost, st));
        TK_TRY(hipStreamSynchronize(st));
    }
    release();
#undef TK_TRY
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Top-k lists fused into the sweep (f32 tables, k <= 64): no score matrix.  The f16 sweep above, with ONE moving threshold
// per query instead of the positive's score.  A query keeps a list of at most TOPK_CAP candidate columns (approximate
// score, item) per catalog range; a column enters when its approximate score is not below the query's bar
//     bar = tau - 2 M,     tau = the k-th best approximate score among the candidates seen so far (clicked and held-out
//                                items left out), M = the query's proven bound on |approximate - exact|,
// and whenever a list is nearly full the wavefront that owns the query selects the k-th best of it (a 32-step search
// over the bit patterns, counted with ballots), raises the bar and drops what fell below it.  Why 2 M: the k candidates
// at or above tau have exact scores >= tau - M, so the exact k-th best is >= tau - M, and a column of the exact top k
// has an approximate score >= tau - 2 M - it is never refused and never dropped.  The lists therefore always hold a
// superset of the exact top k; topk_finish_kernel scores the survivors in fp64 (the summation of correct_clicked_kernel)
// and orders them exactly (score descending, item id ascending).  Appends are rare after the first few thousand columns
// (the bar follows the k-th best of everything seen), so the hot epilogue is one subtraction and half a v_min3 per score.
// A list that stays nearly full after a selection (more than TOPK_CAP - 16 columns within 2 M of the k-th best) marks
// its query for the caller's fallback, the score-matrix path.
#define TOPK_CAP 256
#define TOPK_CLK 512            // click lists up to this length are searched in LDS
#define TOPK_NEG_BIG (-3.0e38f)

struct TopkDev {
    int k, ranges;
    unsigned long long* cand;   // [nq_pad * ranges][TOPK_CAP]: item << 32 | bit pattern of the approximate (scaled) score
    int32_t* cand_cnt;          // [nq_pad * ranges]
    int32_t* q_over;            // [nq_pad] 1: the query's lists could not hold its near-ties
    const int4* qmeta;          // [nq_pad] (user, positive, the other held-out item, -); user -1 on padding rows
    int32_t* items_out;         // [nq][k]
    double* scores_out;         // [nq][k]
};

// bit patterns of floats as unsigned keys that order like the floats (larger float, larger key); NaN patterns with the
// sign bit set map to keys near 0, and 0xffffffff - the mark of a dropped entry - to key 0
__device__ __forceinline__ unsigned f32_key(unsigned b) { return (b >> 31) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ unsigned f32_unkey(unsigned k) { return (k >> 31) ? (k & 0x7fffffffu) : ~k; }

// v[i] for a per-lane index i: a select tree over the bits of i (N - 1 v_bfi_b32; registers cannot be indexed per lane, and
// written as a conditional the compiler turns the pick into an indexed load from a copy of v in scratch memory)
template <int N>
__device__ __forceinline__ float pick_result(const float (&v)[N], int i)
{
    if constexpr (N == 1) return v[0];
    else {
        float h[N / 2];
        const unsigned odd = 0u - (unsigned)(i & 1);
#pragma unroll
        for (int x = 0; x < N / 2; x++) h[x] = __uint_as_float((__float_as_uint(v[2 * x + 1]) & odd) | (__float_as_uint(v[2 * x]) & ~odd));
        return pick_result<N / 2>(h, i >> 1);
    }
}

__global__ void topk_meta_kernel(DevModel m, EvalDev ev, int4* __restrict__ qmeta)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ev.nq_pad) return;
    int4 r = make_int4(-1, -1, -1, 0);
    if (i < ev.nq) {
        // one query per (user, context); or one per user when both contexts have the same query vector (BPRMF): then the
        // validation positive is y and the test positive z, and neither is a candidate of either list
        const int u = m.eval_users[ev.nctx == 2 ? i >> 1 : i]; const int ctx = ev.nctx == 2 ? (int)(i & 1) : 0;
        const int val = m.val_item[u], test = m.test_item[u];
        r = make_int4(u, ctx ? test : val, ctx ? val : test, 0);
    }
    qmeta[i] = r;
}

// query planes and margins for the top-k sweep: screen16_queries_kernel without thresholds (the moving bar is itself an
// approximate score, bounded by Bmax + ||q'|| Tmax like every score)
__global__ void __launch_bounds__(256) topk16_queries_kernel(EvalDev ev, int KQ16, const unsigned* __restrict__ bounds, half_t* __restrict__ Qh,
                                                             half_t* __restrict__ Ql, int32_t* __restrict__ overflow)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= ev.nq_pad) return;
    const double sT = screen16_scale(bounds[2]), sq = screen16_scale(bounds[3]), S = sq * sT;
    double s = 0;
    for (int k = lane; k < KQ16; k += 64) {
        const double q = k < ev.KQ ? ev.Q[w * ev.KQ + k] : 0.0;
        s += q * q;
        half_t h, l; split16((float)(q * sq), h, l);
        Qh[w * KQ16 + k] = h; Ql[w * KQ16 + k] = l;
    }
    s = sqrt(group_sum<64>(s));
    if (lane == 0) {
        const double tmax = (double)__uint_as_float(bounds[0]), bmax = (double)__uint_as_float(bounds[1]);
        const double u = 5.9604644775390625e-08;   // 2^-24
        const double rep = 3.3 * 4.0 * u * s * tmax;
        const double sub = 6.103515625e-05 * sqrt((double)KQ16) * (tmax / sq + s / sT);
        const double acc = 2.0 * (KQ16 + 3) * u * 1.002 * s * tmax + KQ16 * (u / 64.0) * s * tmax;
        const double rnd = 8.0 * u * (bmax + s * tmax);
        const double mq = 1.05 * (rep + sub + acc + rnd) + 1e-30;
        ev.qmar[w] = w < ev.nq ? __double2float_ru(mq * S * 1.000001) : 0.f;
        if (w < ev.nq && !((bmax + s * tmax + mq) * S <= 1e30)) *overflow = 1;   // scores must stay far from the end of f32's range (and finite)
    }
}

template <int KQ>
__global__ void __launch_bounds__(256, 2) topk_sweep16_kernel(EvalDev ev, TopkDev tk, const int64_t* __restrict__ clk_off, const int32_t* __restrict__ clk_item,
                                                              const int32_t* __restrict__ col_item, const half_t* __restrict__ Th, const half_t* __restrict__ Tl, const float* __restrict__ bs,
                                                              const half_t* __restrict__ Qh, const half_t* __restrict__ Ql, int I)
{
    constexpr int MT = Screen16Shape<KQ, false>::MT, NI = Screen16Shape<KQ, false>::NI, QB = Screen16Shape<KQ, false>::QB;
    constexpr int NCH = KQ / 32, STR = KQ + 8, PER = NI * KQ / 8 / 256;
    static_assert(PER >= 1 && KQ % 32 == 0 && MT * 16 <= 64, "tile shape");
    __shared__ __align__(16) half_t tiles[2 * NI * STR];
    half_t* const tile_h = tiles;
    half_t* const tile_l = tiles + NI * STR;
    __shared__ float tile_b[NI];
    __shared__ int q_cnt[QB];                 // entries in each query's list (this block's range)
    __shared__ int clk_lds[4][TOPK_CLK];      // per wavefront: the click list of the user whose list is being selected from
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    const int64_t qblock = (int64_t)blockIdx.x * QB;
    const int qw = wave * (MT * 16);
    const int k = tk.k;

    for (int i = tid; i < QB; i += 256) q_cnt[i] = 0;
    // bar[mt][r]: a column is listed when its score is not below it.  One margin per wavefront, the largest of its queries'.
    float bar[MT][4], marw = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t qi = qblock + qw + mt * 16 + 4 * quad + r;
            marw = fmaxf(marw, ev.qmar[qi]);
            bar[mt][r] = tk.qmeta[qi].x >= 0 ? TOPK_NEG_BIG : INFINITY;     // a padding row lists nothing
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) marw = fmaxf(marw, __shfl_xor(marw, o, 64));
    const float mar_w = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(marw)));

    half8_t ah[MT][NCH], al[MT][NCH];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            const int64_t at = (qblock + qw + mt * 16 + col) * KQ + ch * 32 + 8 * quad;
            ah[mt][ch] = *reinterpret_cast<const half8_t*>(Qh + at);
            al[mt][ch] = *reinterpret_cast<const half8_t*>(Ql + at);
        }

    auto fetch = [&](int item0, half8_t (&ph)[PER], half8_t (&pl)[PER], float& preb) {
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const int64_t at = (int64_t)item0 * KQ + 8 * (int64_t)(tid + 256 * p);
            ph[p] = *reinterpret_cast<const half8_t*>(Th + at);
            pl[p] = *reinterpret_cast<const half8_t*>(Tl + at);
        }
        preb = tid < NI ? bs[item0 + tid] : 0.f;
    };

    // The wavefront that owns query qi (a row of this block) raises its bar: the k-th best of the list, clicked and
    // held-out items dropped, then everything below the new bar.  Returns the new bar.
    auto select = [&](int qi) -> float {
        const int64_t qg = qblock + qi;
        unsigned long long* base = tk.cand + (qg * tk.ranges + blockIdx.y) * TOPK_CAP;
        const int n = min(q_cnt[qi], TOPK_CAP);
        const int4 meta = tk.qmeta[qg];
        const int32_t* clk = clk_item + clk_off[meta.x]; const int nc = (int)(clk_off[meta.x + 1] - clk_off[meta.x]);
        // the user's (sorted) click list goes through LDS when it fits: the membership searches below are chains of
        // dependent loads, and from memory they made a selection cost 20 us - during which the block's other wavefronts
        // wait at the next tile barrier
        const bool cached = nc <= TOPK_CLK;
        if (cached) for (int i = lane; i < nc; i += 64) clk_lds[wave][i] = clk[i];
        __threadfence();                       // the appends of this wavefront have reached memory
        unsigned long long ent[TOPK_CAP / 64]; unsigned key[TOPK_CAP / 64];
#pragma unroll
        for (int i = 0; i < TOPK_CAP / 64; i++) {
            const int idx = lane + 64 * i;
            // a load that does not stop at this compute unit's vector cache (which may hold the line as an earlier selection read it)
            ent[i] = idx < n ? __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffull;
        }
#pragma unroll
        for (int i = 0; i < TOPK_CAP / 64; i++) {
            const int idx = lane + 64 * i;
            const int item = (int)(ent[i] >> 32);
            const bool ok = idx < n && item != meta.y && item != meta.z && !(cached ? sorted_contains(clk_lds[wave], nc, item) : sorted_contains(clk, nc, item));
            key[i] = ok ? f32_key((unsigned)ent[i]) : 0u;
        }
        unsigned prefix = 0;                   // the largest v with at least k keys >= v (0: fewer than k valid entries)
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            const unsigned v = prefix | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < TOPK_CAP / 64; i++) c += __popcll(__ballot(key[i] >= v));
            if (c >= k) prefix = v;
        }
        float nb = TOPK_NEG_BIG;
        if (prefix) {
            const float tau = __uint_as_float(f32_unkey(prefix));
            nb = (tau - 2.f * mar_w) - 2.4e-7f * (fabsf(tau) + 2.f * mar_w);
        }
        int off = 0;
#pragma unroll
        for (int i = 0; i < TOPK_CAP / 64; i++) {
            const bool keep = key[i] != 0u && __uint_as_float((unsigned)ent[i]) >= nb;
            const unsigned long long bal = __ballot(keep);
            if (keep) base[off + __popcll(bal & ((1ull << lane) - 1ull))] = ent[i];
            off += __popcll(bal);
        }
        if (off > TOPK_CAP - 16) {             // too many columns within the margin of the k-th best: the caller's fallback decides this query
            if (lane == 0) tk.q_over[qg] = 1;
            off = 0; nb = INFINITY;
        }
        if (lane == 0) q_cnt[qi] = off;
        return nb;
    };

    const int i_begin = (int)blockIdx.y * ev.range_items;
    const int i_end = (int)min((int64_t)I, (int64_t)i_begin + ev.range_items);
    half8_t ph[PER], pl[PER]; float preb;
    const int frag0 = col * STR + 8 * quad;
    fetch(i_begin, ph, pl, preb);
    for (int item0 = i_begin; item0 < i_end; item0 += NI) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const int e = 8 * (tid + 256 * p), it = e / KQ, kk = e % KQ;
            *reinterpret_cast<half8_t*>(&tile_h[it * STR + kk]) = ph[p];
            *reinterpret_cast<half8_t*>(&tile_l[it * STR + kk]) = pl[p];
        }
        if (tid < NI) tile_b[tid] = preb;
        __syncthreads();
        if (item0 + NI < i_end) fetch(item0 + NI, ph, pl, preb);
        half8_t bh[NCH], bl[NCH]; float bj;
        auto read_frags = [&](int sub) {
            const int at = frag0 + sub * (16 * STR);
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                bh[ch] = *reinterpret_cast<const half8_t*>(&tiles[at + ch * 32]);
                bl[ch] = *reinterpret_cast<const half8_t*>(&tiles[NI * STR + at + ch * 32]);
            }
            bj = tile_b[sub * 16 + col];
        };
        read_frags(0);
#pragma unroll 1
        for (int sub = 0; sub < NI / 16; sub++) {
            const int j = item0 + sub * 16 + col;
            float sc[MT * 4];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                float4_t am = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][ch], bl[ch], am, 0, 0, 0);
                    am = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt][ch], bh[ch], am, 0, 0, 0);
                }
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][ch], bh[ch], am, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) sc[4 * mt + r] = am[r] + bj;
            }
            if (sub + 1 < NI / 16) read_frags(sub + 1);
            // y = score - bar: a column is listed when y is not negative.  The sign bits of a sub-step's MT * 4 results are
            // shifted into one register (v_alignbit: result i of the sub-step ends up at bit MT * 4 - 1 - i), so that the
            // rare sub-step that lists something has its mask at hand.  (A column past the end of the catalog scores -inf,
            // a padding row's bar is +inf: both give y = -inf.  NaN never gets here: the launcher checks the operands.)
            unsigned sg = 0;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) sg = __builtin_amdgcn_alignbit(sg, __float_as_uint(sc[4 * mt + r] - bar[mt][r]), 31);
            constexpr unsigned ALL = (1u << (MT * 4)) - 1u;
            const unsigned open = ~sg & ALL;
            if (!__ballot(open != 0u)) continue;
            // Every lane walks its own set bits; an iteration serves the first open result of ALL lanes with one LDS
            // atomic, one wait and one store (typically a handful of lanes hold one bit each: one iteration).  The result
            // a lane's bit names is picked out of its registers by a select tree over the bits of the index (a loop over
            // the MT * 4 results with a scalar skip paid an LDS round trip per result instead of per iteration).
            bool full = false;
            unsigned todo = open;
            while (__ballot(todo != 0u)) {
                if (todo) {
                    const int b = 31 - __clz((int)todo);
                    todo &= ~(1u << b);
                    const int i = MT * 4 - 1 - b;                 // result 4 mt + r
                    const float v = pick_result<MT * 4>(sc, i);
                    const int qi = qw + (i >> 2) * 16 + 4 * quad + (i & 3);
                    const int at = atomicAdd(&q_cnt[qi], 1);
                    if (at < TOPK_CAP)
                        tk.cand[((qblock + qi) * tk.ranges + blockIdx.y) * TOPK_CAP + at] = ((unsigned long long)(unsigned)col_item[j] << 32) | __float_as_uint(v);
                    full |= at + 1 > TOPK_CAP - 16;   // a sub-step adds at most 16 columns to a list: selected before the next one
                }
            }
            if (!__ballot(full)) continue;
            unsigned long long need = __ballot(lane < MT * 16 && q_cnt[qw + (lane < MT * 16 ? lane : 0)] > TOPK_CAP - 16);
            while (need) {
                const int ql = __ffsll((long long)need) - 1;
                need &= need - 1;
                const float nb = select(qw + ql);
                const int tmt = ql >> 4, tq = (ql & 15) >> 2, tr = ql & 3;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (mt == tmt && r == tr && quad == tq) bar[mt][r] = nb;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < QB; i += 256) tk.cand_cnt[(qblock + i) * tk.ranges + blockIdx.y] = min(q_cnt[i], TOPK_CAP);
}

// One wavefront per query: the k best of its lists, exactly.  Drops clicked / held-out items, finds the k-th best
// approximate score over all ranges, scores what is within 2 M of it in fp64 (four items per step, 16 lanes each: the
// summation of correct_clicked_kernel), adds the context's positive with the score the reference formula gives it, and
// orders by (score descending, item id ascending).
#define TOPK_SURV 192
template <class T>
__global__ void __launch_bounds__(256) topk_finish_kernel(DevModel m, EvalDev ev, TopkDev tk)
{
    __shared__ int s_item[4][TOPK_SURV];
    __shared__ double s_score[4][TOPK_SURV];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l16 = lane & 15, gq = lane >> 4;
    const int w = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= ev.nq) return;
    const int k = tk.k;
    // a query yields one list, or two (validation, test) when both contexts share it
    const int nlists = ev.nctx == 1 ? 2 : 1;
    const int64_t row0 = ev.nctx == 1 ? 2 * (int64_t)w : w;
    const int4 meta = tk.qmeta[w];
    const int32_t* clk = m.clk_item + m.clk_off[meta.x]; const int nc = (int)(m.clk_off[meta.x + 1] - m.clk_off[meta.x]);
    unsigned long long* lists = tk.cand + (int64_t)w * tk.ranges * TOPK_CAP;
    const int32_t* cnts = tk.cand_cnt + (int64_t)w * tk.ranges;
    bool over = tk.q_over[w] != 0;
    float nb = TOPK_NEG_BIG;
    int ns = 0;
    const float mq = ev.qmar[w];
    if (tk.ranges == 1 && !over) {
        // one list: its entries stay in registers (four per lane) from the membership test to the compaction
        __shared__ int clk_lds[4][TOPK_CLK];
        const int n = cnts[0];
        const bool cached = nc <= TOPK_CLK;
        if (cached) for (int i = lane; i < nc; i += 64) clk_lds[wv][i] = clk[i];
        unsigned long long ent[TOPK_CAP / 64]; unsigned key[TOPK_CAP / 64];
#pragma unroll
        for (int i = 0; i < TOPK_CAP / 64; i++) { const int idx = lane + 64 * i; ent[i] = idx < n ? lists[idx] : 0xffffffffull; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < TOPK_CAP / 64; i++) {
            const int idx = lane + 64 * i;
            const int item = (int)(ent[i] >> 32);
            const bool ok = idx < n && item != meta.y && item != meta.z && !(cached ? sorted_contains(clk_lds[wv], nc, item) : sorted_contains(clk, nc, item));
            key[i] = ok ? f32_key((unsigned)ent[i]) : 0u;
        }
        unsigned prefix = 0;
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            const unsigned v = prefix | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < TOPK_CAP / 64; i++) c += __popcll(__ballot(key[i] >= v));
            if (c >= k) prefix = v;
        }
        if (prefix) {
            const float tau = __uint_as_float(f32_unkey(prefix));
            nb = (tau - 2.f * mq) - 2.4e-7f * (fabsf(tau) + 2.f * mq);
        }
#pragma unroll
        for (int i = 0; i < TOPK_CAP / 64; i++) {
            const bool keep = key[i] != 0u && __uint_as_float((unsigned)ent[i]) >= nb;
            const unsigned long long bal = __ballot(keep);
            const int at = ns + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && at < TOPK_SURV - 1) s_item[wv][at] = (int)(ent[i] >> 32);
            ns += __popcll(bal);
        }
    } else if (!over) {
        // entries that are not candidates get the pattern of key 0
        for (int r = 0; r < tk.ranges; r++) {
            const int n = cnts[r];
            for (int idx = lane; idx < n; idx += 64) {
                const unsigned long long e = lists[r * TOPK_CAP + idx];
                const int item = (int)(e >> 32);
                if (item == meta.y || item == meta.z || sorted_contains(clk, nc, item)) lists[r * TOPK_CAP + idx] = e | 0xffffffffull;
            }
        }
        __threadfence();
        unsigned prefix = 0;
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            const unsigned v = prefix | (1u << bit);
            int c = 0;
            for (int r = 0; r < tk.ranges; r++) {
                const int n = cnts[r];
                for (int i0 = 0; i0 < n; i0 += 64) {
                    const int idx = i0 + lane;
                    const unsigned key = idx < n ? f32_key((unsigned)__hip_atomic_load(lists + r * TOPK_CAP + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0u;
                    c += __popcll(__ballot(key >= v));
                }
            }
            if (c >= k) prefix = v;
        }
        if (prefix) {
            const float tau = __uint_as_float(f32_unkey(prefix));
            nb = (tau - 2.f * mq) - 2.4e-7f * (fabsf(tau) + 2.f * mq);
        }
        for (int r = 0; r < tk.ranges; r++) {
            const int n = cnts[r];
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int idx = i0 + lane;
                const unsigned long long e = idx < n ? __hip_atomic_load(lists + r * TOPK_CAP + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffull;
                const bool keep = f32_key((unsigned)e) != 0u && __uint_as_float((unsigned)e) >= nb;
                const unsigned long long bal = __ballot(keep);
                const int at = ns + __popcll(bal & ((1ull << lane) - 1ull));
                if (keep && at < TOPK_SURV - 1) s_item[wv][at] = (int)(e >> 32);
                ns += __popcll(bal);
            }
        }
    }
    if (ns > TOPK_SURV - 1) over = true;
    if (over) {
        // left to the caller's fallback
        if (lane == 0) tk.q_over[w] = 1;
        for (int r = lane; r < nlists * k; r += 64) { tk.items_out[row0 * k + r] = -1; tk.scores_out[row0 * k + r] = -INFINITY; }
        return;
    }
    const double* q = ev.Q + (int64_t)w * ev.KQ;
    const T* seg0 = ev.seg0 >= 0 ? (const T*)m.tab[ev.seg0] : nullptr;
    const T* seg1 = ev.seg1 >= 0 ? (const T*)m.tab[ev.seg1] : nullptr;
    const int K1 = ev.K1, K2 = ev.K2;
    for (int base = 0; base < ns; base += 4) {
        const int idx = base + gq;
        const int64_t ec = idx < ns ? s_item[wv][idx] : 0;
        double s = 0;
        for (int kk = l16; kk < K1; kk += 16) s += q[kk] * (double)seg0[ec * K1 + kk];
        for (int kk = l16; kk < K2; kk += 16) s += q[K1 + kk] * (double)seg1[ec * K2 + kk];
        s = group_sum<16>(s) + ev.bconst[ec];
        if (l16 == 0 && idx < ns) s_score[wv][idx] = s;
    }
    const int nt = ns + 1;
    const double aq = ev.aq[w];
    for (int li = 0; li < nlists; li++) {
        int32_t* items = tk.items_out + (row0 + li) * k;
        double* scores = tk.scores_out + (row0 + li) * k;
        // the list's positive joins the survivors with the score the reference formula gives it
        if (lane == 0) { s_item[wv][ns] = li ? meta.z : meta.y; s_score[wv][ns] = li ? ev.thr2[w] : ev.thr[w]; }
        __builtin_amdgcn_wave_barrier();
        // LDS writes of this wavefront are visible to its later reads (in order); rank by counting
        for (int i = lane; i < nt; i += 64) {
            const double si = s_score[wv][i]; const int ii = s_item[wv][i];
            int rank = 0;
            for (int x = 0; x < nt; x++) { const double sx = s_score[wv][x]; rank += (sx > si || (sx == si && s_item[wv][x] < ii)) ? 1 : 0; }
            if (rank < k) { items[rank] = ii; scores[rank] = si + aq; }
        }
        for (int r = nt + lane; r < k; r += 64) { items[r] = -1; scores[r] = -INFINITY; }
        __builtin_amdgcn_wave_barrier();
    }
}

static int g_topk_fused = 1;
static int64_t g_topk_fused_lists = 0; static double g_topk_fused_ms = 0;
void eval_configure_topk(int fused) { g_topk_fused = fused; }
void eval_topk_stats(int64_t* lists, double* ms) { *lists = g_topk_fused_lists; *ms = g_topk_fused_ms; }

template <int KQ>
static void launch_topk_sweep(const EvalDev& ev, const TopkDev& tk, const DevModel& m, const half_t* Th, const half_t* Tl, const float* bs, const half_t* Qh,
                              const half_t* Ql, hipStream_t st)
{
    const unsigned blocks = (unsigned)(ev.nq_pad / Screen16Shape<KQ, false>::QB);
    hipLaunchKernelGGL((topk_sweep16_kernel<KQ>), dim3(blocks, ev.ranges), dim3(256), 0, st, ev, tk, m.clk_off, m.clk_item, m.pop_order, Th, Tl, bs, Qh, Ql, m.I);
}

// users_host == nullptr: the first n_users users of m.eval_users.  fallback: positions (in users order) whose lists the
// fused path could not produce; *all_fallback: the operands do not suit the f16 sweep at all.
static int topk_fused_impl(const DevModel& m0, const int32_t* users_host, int n_users, int k, int32_t* items_host, double* scores_host,
                           hipStream_t st, std::vector<int32_t>& fallback, bool* all_fallback)
{
    typedef float T;
    DevModel m = m0;
    const int I = m.I;
    *all_fallback = false;
    EvalDev ev = {};
    // BPRMF's prediction ignores the previous item (BPRMF.cpp:69-72): both contexts of a user share one query, and one
    // sweep serves both lists
    ev.nctx = m.model == REC_BPRMF ? 1 : 2; ev.force2 = 1;
    const int nctx = ev.nctx;
    ev.seg0 = ev.seg1 = -1; ev.K1 = ev.K2 = 0;
    switch (m.model) {
    case REC_BPRMF: ev.seg0 = BPRMF_GI; ev.K1 = m.K; break;
    case REC_MC: ev.seg0 = MC_ETA; ev.K1 = m.K; break;
    case REC_FPMC: ev.seg0 = FPMC_GI; ev.K1 = m.K; ev.seg1 = FPMC_ETA; ev.K2 = m.KK; break;
    case REC_PRME: ev.seg0 = PRME_GI; ev.K1 = m.K; ev.seg1 = PRME_ETA; ev.K2 = m.KK; break;
    case REC_HRM_AVG: case REC_HRM_MAX: ev.seg0 = HRM_V; ev.K1 = m.K; break;
    case REC_TRANSREC: ev.seg0 = TR_H; ev.K1 = m.K; break;
    case REC_FOSSIL_SIMPLE: ev.seg0 = FS_V; ev.K1 = m.K; break;
    case REC_FOSSIL: ev.seg0 = FO_V; ev.K1 = m.K; break;
    case REC_REBUS: ev.seg0 = RB_P; ev.K1 = m.K; break;
    default: *all_fallback = true; return 0;
    }
    const int kw = ev.K1 + ev.K2;
    ev.KQ = kw <= 16 ? 16 : kw <= 32 ? 32 : kw <= 64 ? 64 : kw <= 128 ? 128 : 256;
    const int KQ16 = std::max(32, ev.KQ);
    const int QBLK = screen16_qb(KQ16, false);
    const int NI16 = KQ16 <= 128 ? 64 : 32;
    const int I_pad = ((I + NI16 - 1) / NI16) * NI16;
    const int batch_users = std::min(n_users, (1 << 20) / nctx);
    const int max_rows = ((nctx * batch_users + QBLK - 1) / QBLK) * QBLK;
    // catalog ranges: enough blocks to fill the chip when the queries alone do not (a short list of users)
    int dev = 0, cus = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    {
        const int64_t blocks = max_rows / QBLK, resident = 2 * (int64_t)cus;
        const int tile = 128;
        int ranges = (int)std::min<int64_t>(std::min<int64_t>(64, (4 * resident + blocks - 1) / blocks), (I + 4095) / 4096);
        ranges = std::max(1, ranges);
        ev.range_items = (int)((((int64_t)I + ranges - 1) / ranges + tile - 1) / tile * tile);
        ev.ranges = (I + ev.range_items - 1) / ev.range_items;
    }
    std::vector<void*> owned;
    auto alloc = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, bytes ? bytes : 8); if (e == hipSuccess) owned.push_back(*p); return e; };
    auto release = [&]() { for (void* p : owned) (void)hipFree(p); };
#define TF_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { rec_set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); release(); return -2; } } while (0)
    TopkDev tk = {};
    tk.k = k; tk.ranges = ev.ranges;
    int32_t* d_users = nullptr; int4* qmeta = nullptr; int32_t* over = nullptr; unsigned* b4 = nullptr;
    half_t *Th = nullptr, *Tl = nullptr, *Qh = nullptr, *Ql = nullptr; float* bs = nullptr;
    if (users_host) TF_TRY(alloc((void**)&d_users, sizeof(int32_t) * batch_users));
    TF_TRY(alloc((void**)&ev.Q, sizeof(double) * (size_t)max_rows * ev.KQ));
    TF_TRY(alloc((void**)&ev.thr, sizeof(double) * max_rows));
    TF_TRY(alloc((void**)&ev.thr2, sizeof(double) * max_rows));
    TF_TRY(alloc((void**)&ev.aq, sizeof(double) * max_rows));
    TF_TRY(alloc((void**)&ev.qmar, sizeof(float) * max_rows));
    TF_TRY(alloc((void**)&ev.bconst, sizeof(double) * I));
    TF_TRY(alloc((void**)&ev.xpos, sizeof(double) * (size_t)m.U * 2));
    TF_TRY(alloc((void**)&qmeta, sizeof(int4) * max_rows));
    TF_TRY(alloc((void**)&tk.cand, sizeof(unsigned long long) * (size_t)max_rows * ev.ranges * TOPK_CAP));
    TF_TRY(alloc((void**)&tk.cand_cnt, sizeof(int32_t) * (size_t)max_rows * ev.ranges));
    TF_TRY(alloc((void**)&tk.q_over, sizeof(int32_t) * max_rows));
    TF_TRY(alloc((void**)&tk.items_out, sizeof(int32_t) * (size_t)batch_users * 2 * k));
    TF_TRY(alloc((void**)&tk.scores_out, sizeof(double) * (size_t)batch_users * 2 * k));
    TF_TRY(alloc((void**)&over, sizeof(int32_t)));
    TF_TRY(alloc((void**)&b4, sizeof(unsigned) * 4));
    TF_TRY(alloc((void**)&Th, sizeof(half_t) * (size_t)I_pad * KQ16));
    TF_TRY(alloc((void**)&Tl, sizeof(half_t) * (size_t)I_pad * KQ16));
    TF_TRY(alloc((void**)&bs, sizeof(float) * (size_t)I_pad));
    TF_TRY(alloc((void**)&Qh, sizeof(half_t) * (size_t)max_rows * KQ16));
    TF_TRY(alloc((void**)&Ql, sizeof(half_t) * (size_t)max_rows * KQ16));
    tk.qmeta = qmeta;
    ev.col_b = ev.bconst; ev.n_cols = I;
    const float* seg0 = ev.seg0 >= 0 ? (const float*)m.tab[ev.seg0] : nullptr;
    const float* seg1 = ev.seg1 >= 0 ? (const float*)m.tab[ev.seg1] : nullptr;
    hipLaunchKernelGGL(item_consts_kernel<T>, dim3(blocks_for_waves(I)), dim3(256), 0, st, m, ev);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    TF_TRY(hipEventCreate(&e0)); TF_TRY(hipEventCreate(&e1));
    std::vector<int32_t> over_host;
    g_topk_fused_lists = 0; g_topk_fused_ms = 0;
    const int32_t* all_users = m0.eval_users;
    for (int u0 = 0; u0 < n_users; u0 += batch_users) {
        const int nu = std::min(batch_users, n_users - u0);
        if (users_host) { TF_TRY(hipMemcpyAsync(d_users, users_host + u0, sizeof(int32_t) * nu, hipMemcpyHostToDevice, st)); m.eval_users = d_users; }
        else m.eval_users = all_users + u0;
        m.n_eval = nu;
        ev.nq = nctx * nu; ev.nq_pad = ((ev.nq + QBLK - 1) / QBLK) * QBLK;
        TF_TRY(hipEventRecord(e0, st));
        TF_TRY(hipMemsetAsync(ev.Q, 0, sizeof(double) * (size_t)ev.nq_pad * ev.KQ, st));
        TF_TRY(hipMemsetAsync(tk.q_over, 0, sizeof(int32_t) * ev.nq_pad, st));
        TF_TRY(hipMemsetAsync(over, 0, sizeof(int32_t), st));
        TF_TRY(hipMemsetAsync(b4, 0, sizeof(unsigned) * 4, st));
        hipLaunchKernelGGL(topk_meta_kernel, dim3((ev.nq_pad + 255) / 256), dim3(256), 0, st, m, ev, qmeta);
        hipLaunchKernelGGL(build_queries_kernel<T>, dim3(blocks_for_waves(ev.nq)), dim3(256), 0, st, m, ev);
        hipLaunchKernelGGL(screen16_bounds_kernel, dim3(1024), dim3(256), 0, st, ev, seg0, seg1, I, b4);
        hipLaunchKernelGGL(screen16_items_kernel, dim3(2048), dim3(256), 0, st, ev, seg0, seg1, I, I_pad, KQ16, (const unsigned*)b4, Th, Tl, bs, over, -INFINITY, m.pop_order);
        hipLaunchKernelGGL(topk16_queries_kernel, dim3(blocks_for_waves(ev.nq_pad)), dim3(256), 0, st, ev, KQ16, (const unsigned*)b4, Qh, Ql, over);
        switch (KQ16) {
        case 32: launch_topk_sweep<32>(ev, tk, m, Th, Tl, bs, Qh, Ql, st); break;
        case 64: launch_topk_sweep<64>(ev, tk, m, Th, Tl, bs, Qh, Ql, st); break;
        case 128: launch_topk_sweep<128>(ev, tk, m, Th, Tl, bs, Qh, Ql, st); break;
        default: launch_topk_sweep<256>(ev, tk, m, Th, Tl, bs, Qh, Ql, st); break;
        }
        hipLaunchKernelGGL(topk_finish_kernel<T>, dim3(blocks_for_waves(ev.nq)), dim3(256), 0, st, m, ev, tk);
        TF_TRY(hipGetLastError());
        TF_TRY(hipEventRecord(e1, st));
        int32_t ov = 0;
        over_host.resize(ev.nq);
        TF_TRY(hipMemcpyAsync(&ov, over, sizeof ov, hipMemcpyDeviceToHost, st));
        TF_TRY(hipMemcpyAsync(over_host.data(), tk.q_over, sizeof(int32_t) * ev.nq, hipMemcpyDeviceToHost, st));
        TF_TRY(hipMemcpyAsync(items_host + (size_t)u0 * 2 * k, tk.items_out, sizeof(int32_t) * (size_t)nu * 2 * k, hipMemcpyDeviceToHost, st));
        if (scores_host) TF_TRY(hipMemcpyAsync(scores_host + (size_t)u0 * 2 * k, tk.scores_out, sizeof(double) * (size_t)nu * 2 * k, hipMemcpyDeviceToHost, st));
        TF_TRY(hipStreamSynchronize(st));
        float ms = 0; TF_TRY(hipEventElapsedTime(&ms, e0, e1)); g_topk_fused_ms += ms;
        if (ov) {   // tables or queries that are not finite, or out of the f16 sweep's range
            *all_fallback = true;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); release();
            g_topk_fused_lists = 0;
            return 0;
        }
        for (int a = 0; a < nu; a++) {
            if (over_host[nctx * a] || over_host[nctx * a + nctx - 1]) fallback.push_back(u0 + a);
            else g_topk_fused_lists += 2;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    release();
#undef TF_TRY
    return 0;
}

// One wavefront per query of the |.|_1 sweep: clicked items out of what the list still holds, the context's positive in
// with the score the reference formula gives it, ordered by (score descending, item id ascending).
// rescore: the lists hold approximate scores (the fixed-point sweep): what they still hold is scored in fp64, summed as the
// reference sums it (from b_j, k ascending; four items per step, 16 lanes each, as in correct_clicked_kernel).
template <class T>
__global__ void __launch_bounds__(256) topk_l1_finish_kernel(DevModel m, EvalDev ev, TopkL1Dev tk, int rescore)
{
    __shared__ int s_item[4][L1_LIST + 1];
    __shared__ double s_score[4][L1_LIST + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= ev.nq) return;
    const int k = tk.k;
    int32_t* items = tk.items_out + (int64_t)w * k;
    double* scores = tk.scores_out + (int64_t)w * k;
    if (tk.q_over[w]) {                        // left to the caller's fallback
        for (int r = lane; r < k; r += 64) { items[r] = -1; scores[r] = -INFINITY; }
        return;
    }
    const int4 meta = tk.qmeta[w];
    const int32_t* clk = m.clk_item + m.clk_off[meta.x]; const int nc = (int)(m.clk_off[meta.x + 1] - m.clk_off[meta.x]);
    const double* cs = tk.cand_s + (int64_t)w * L1_LIST; const int32_t* ci = tk.cand_i + (int64_t)w * L1_LIST;
    const int n = tk.cand_cnt[w];
    int ns = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int idx = i0 + lane;
        const int item = idx < n ? ci[idx] : -1;
        const double sc = idx < n ? cs[idx] : 0.0;
        const bool keep = idx < n && item != meta.y && item != meta.z && !sorted_contains(clk, nc, item);
        const unsigned long long bal = __ballot(keep);
        if (keep) { const int at = ns + __popcll(bal & ((1ull << lane) - 1ull)); s_item[wv][at] = item; s_score[wv][at] = sc; }
        ns += __popcll(bal);
    }
    if (rescore) {
        __builtin_amdgcn_wave_barrier();
        const int l16 = lane & 15, gq = lane >> 4, K = ev.K1;
        const double* q = ev.Q + (int64_t)w * ev.KQ;
        const T* H = (const T*)m.tab[ev.seg0];
        for (int base = 0; base < ns; base += 4) {
            const int idx = base + gq;
            const int64_t ec = idx < ns ? s_item[wv][idx] : 0;
            double pr = ev.bconst[ec];
            for (int k0 = 0; k0 < K; k0 += 16) {
                const int k = k0 + l16;
                const double t = k < K ? fabs(q[k] - (double)H[ec * K + k]) : 0.0;
                const int nn = min(16, K - k0);
                for (int l = 0; l < nn; l++) pr += __shfl(t, (lane & 48) + l, 64);
            }
            if (l16 == 0 && idx < ns) s_score[wv][idx] = -pr + 0.0;
        }
    }
    if (lane == 0) { s_item[wv][ns] = meta.y; s_score[wv][ns] = ev.thr[w]; }
    const int nt = ns + 1;
    const double aq = ev.aq[w];
    __builtin_amdgcn_wave_barrier();
    // LDS writes of this wavefront are visible to its later reads (in order); rank by counting
    for (int i = lane; i < nt; i += 64) {
        const double si = s_score[wv][i]; const int ii = s_item[wv][i];
        int rank = 0;
        for (int x = 0; x < nt; x++) { const double sx = s_score[wv][x]; rank += (sx > si || (sx == si && s_item[wv][x] < ii)) ? 1 : 0; }
        if (rank < k) { items[rank] = ii; scores[rank] = si + aq; }
    }
    for (int r = nt + lane; r < k; r += 64) { items[r] = -1; scores[r] = -INFINITY; }
}

// users_host == nullptr: the first n_users users of m.eval_users.  fallback: positions (in users order) whose lists held
// too many ties; *all_fallback: this shape is not served (K above 160).
template <class T>
static int topk_l1_impl(const DevModel& m0, const int32_t* users_host, int n_users, int k, int32_t* items_host, double* scores_host,
                        hipStream_t st, std::vector<int32_t>& fallback, bool* all_fallback)
{
    DevModel m = m0;
    const int I = m.I;
    *all_fallback = false;
    if (m.K > 160) { *all_fallback = true; return 0; }
    EvalDev ev = {};
    ev.nctx = 2; ev.force2 = 1;
    ev.seg0 = TR_H; ev.seg1 = -1; ev.K1 = m.K; ev.K2 = 0;
    ev.KQ = m.K <= 16 ? 16 : m.K <= 32 ? 32 : m.K <= 64 ? 64 : m.K <= 128 ? 128 : 256;
    constexpr int QT = 32;
    const int batch_users = std::min(n_users, 1 << 17);
    const int max_rows = ((2 * batch_users + QT - 1) / QT) * QT;
    std::vector<void*> owned;
    auto alloc = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, bytes ? bytes : 8); if (e == hipSuccess) owned.push_back(*p); return e; };
    auto release = [&]() { for (void* p : owned) (void)hipFree(p); };
#define TL_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { rec_set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); release(); return -2; } } while (0)
    TopkL1Dev tk = {};
    tk.k = k; tk.clk_off = m.clk_off; tk.clk_item = m.clk_item;
    int32_t* d_users = nullptr; int4* qmeta = nullptr;
    if (users_host) TL_TRY(alloc((void**)&d_users, sizeof(int32_t) * batch_users));
    TL_TRY(alloc((void**)&ev.Q, sizeof(double) * (size_t)max_rows * ev.KQ));
    TL_TRY(alloc((void**)&ev.thr, sizeof(double) * max_rows));
    TL_TRY(alloc((void**)&ev.thr2, sizeof(double) * max_rows));
    TL_TRY(alloc((void**)&ev.aq, sizeof(double) * max_rows));
    TL_TRY(alloc((void**)&ev.bconst, sizeof(double) * I));
    TL_TRY(alloc((void**)&ev.xpos, sizeof(double) * (size_t)m.U * 2));
    TL_TRY(alloc((void**)&qmeta, sizeof(int4) * max_rows));
    TL_TRY(alloc((void**)&tk.cand_s, sizeof(double) * (size_t)max_rows * L1_LIST));
    TL_TRY(alloc((void**)&tk.cand_i, sizeof(int32_t) * (size_t)max_rows * L1_LIST));
    TL_TRY(alloc((void**)&tk.cand_cnt, sizeof(int32_t) * max_rows));
    TL_TRY(alloc((void**)&tk.q_over, sizeof(int32_t) * max_rows));
    TL_TRY(alloc((void**)&tk.items_out, sizeof(int32_t) * (size_t)batch_users * 2 * k));
    TL_TRY(alloc((void**)&tk.scores_out, sizeof(double) * (size_t)batch_users * 2 * k));
    uint32_t *Hi = nullptr, *Qi = nullptr; double *Bj = nullptr, *Tq = nullptr; unsigned* bnd = nullptr; int32_t* flags = nullptr;
    if (g_screen_on) {
        TL_TRY(alloc((void**)&Hi, sizeof(uint32_t) * (size_t)I * m.K));
        TL_TRY(alloc((void**)&Qi, sizeof(uint32_t) * (size_t)max_rows * m.K));
        TL_TRY(alloc((void**)&Bj, sizeof(double) * (size_t)I));
        TL_TRY(alloc((void**)&Tq, sizeof(double) * (size_t)max_rows));
        TL_TRY(alloc((void**)&bnd, sizeof(unsigned)));
        TL_TRY(alloc((void**)&flags, sizeof(int32_t) * 2));
    }
    tk.qmeta = qmeta;
    ev.col_b = ev.bconst; ev.n_cols = I;
    hipLaunchKernelGGL(item_consts_kernel<T>, dim3(blocks_for_waves(I)), dim3(256), 0, st, m, ev);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    TL_TRY(hipEventCreate(&e0)); TL_TRY(hipEventCreate(&e1));
    std::vector<int32_t> over_host;
    g_topk_fused_lists = 0; g_topk_fused_ms = 0;
    const int32_t* all_users = m0.eval_users;
    for (int u0 = 0; u0 < n_users; u0 += batch_users) {
        const int nu = std::min(batch_users, n_users - u0);
        if (users_host) { TL_TRY(hipMemcpyAsync(d_users, users_host + u0, sizeof(int32_t) * nu, hipMemcpyHostToDevice, st)); m.eval_users = d_users; }
        else m.eval_users = all_users + u0;
        m.n_eval = nu;
        ev.nq = 2 * nu; ev.nq_pad = ((ev.nq + QT - 1) / QT) * QT;
        TL_TRY(hipEventRecord(e0, st));
        TL_TRY(hipMemsetAsync(ev.Q, 0, sizeof(double) * (size_t)ev.nq_pad * ev.KQ, st));
        TL_TRY(hipMemsetAsync(tk.q_over, 0, sizeof(int32_t) * ev.nq_pad, st));
        hipLaunchKernelGGL(topk_meta_kernel, dim3((ev.nq_pad + 255) / 256), dim3(256), 0, st, m, ev, qmeta);
        hipLaunchKernelGGL(build_queries_kernel<T>, dim3(blocks_for_waves(ev.nq)), dim3(256), 0, st, m, ev);
        // the sweep on fixed-point operands (approximate scores in the lists, rescored in fp64 by the finish); operands that
        // are not finite, or "eval_screen" 0: the fp64 sweep, whose lists hold the exact scores
        int32_t declined = 1;
        if (g_screen_on) {
            TL_TRY(hipMemsetAsync(flags, 0, sizeof(int32_t) * 2, st));
            TL_TRY(hipMemsetAsync(bnd, 0, sizeof(unsigned), st));
            const int f64 = sizeof(T) == 8;
            hipLaunchKernelGGL(l1_absmax_kernel, dim3(1024), dim3(256), 0, st, (const void*)m.tab[TR_H], f64, (int64_t)I * m.K, (const double*)ev.Q,
                               (int64_t)ev.nq_pad * ev.KQ, bnd);
            hipLaunchKernelGGL(l1_quant_items_kernel, dim3(2048), dim3(256), 0, st, (const void*)m.tab[TR_H], f64, I, m.K, (const double*)ev.col_b,
                               (const unsigned*)bnd, Hi, Bj, flags);
            hipLaunchKernelGGL(l1_quant_queries_kernel, dim3(2048), dim3(256), 0, st, ev, m.K, (const unsigned*)bnd, Qi, Tq);
            TL_TRY(hipMemcpyAsync(&declined, flags + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            TL_TRY(hipStreamSynchronize(st));
        }
        if (!declined) {
            const size_t lds_bytes = sizeof(uint32_t) * (((size_t)64 * (m.K | 1) + 1) & ~(size_t)1) + sizeof(double) * 2 * QT;
            hipLaunchKernelGGL((score_count_l1_sad_kernel<QT, true>), dim3((unsigned)(ev.nq_pad / QT)), dim3(256), lds_bytes, st, ev, (const uint32_t*)Hi,
                               (const uint32_t*)Qi, (const double*)Bj, (const double*)Tq, I, m.K, 1.0002 * m.K + 2.0, tk);
        } else if (int rc = launch_score_l1<T, 2>(ev, (const T*)m.tab[TR_H], nullptr, nullptr, I, m.K, st, tk)) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); release(); return rc; }
        hipLaunchKernelGGL(topk_l1_finish_kernel<T>, dim3(blocks_for_waves(ev.nq)), dim3(256), 0, st, m, ev, tk, declined ? 0 : 1);
        TL_TRY(hipGetLastError());
        over_host.resize(ev.nq);
        if (!declined) {
            // A list that could not hold the columns within two margins of its k-th best (operands with an outlier far above the
            // rest leave the fixed-point scale too coarse to tell the others apart): the batch again through the fp64 sweep,
            // whose lists only overflow on exact ties.
            TL_TRY(hipMemcpyAsync(over_host.data(), tk.q_over, sizeof(int32_t) * ev.nq, hipMemcpyDeviceToHost, st));
            TL_TRY(hipStreamSynchronize(st));
            bool any = false;
            for (int32_t x : over_host) any |= x != 0;
            if (any) {
                TL_TRY(hipMemsetAsync(tk.q_over, 0, sizeof(int32_t) * ev.nq_pad, st));
                if (int rc = launch_score_l1<T, 2>(ev, (const T*)m.tab[TR_H], nullptr, nullptr, I, m.K, st, tk)) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); release(); return rc; }
                hipLaunchKernelGGL(topk_l1_finish_kernel<T>, dim3(blocks_for_waves(ev.nq)), dim3(256), 0, st, m, ev, tk, 0);
                TL_TRY(hipGetLastError());
            }
        }
        TL_TRY(hipEventRecord(e1, st));
        TL_TRY(hipMemcpyAsync(over_host.data(), tk.q_over, sizeof(int32_t) * ev.nq, hipMemcpyDeviceToHost, st));
        TL_TRY(hipMemcpyAsync(items_host + (size_t)u0 * 2 * k, tk.items_out, sizeof(int32_t) * (size_t)nu * 2 * k, hipMemcpyDeviceToHost, st));
        if (scores_host) TL_TRY(hipMemcpyAsync(scores_host + (size_t)u0 * 2 * k, tk.scores_out, sizeof(double) * (size_t)nu * 2 * k, hipMemcpyDeviceToHost, st));
        TL_TRY(hipStreamSynchronize(st));
        float ms = 0; TL_TRY(hipEventElapsedTime(&ms, e0, e1)); g_topk_fused_ms += ms;
        for (int a = 0; a < nu; a++) {
            if (over_host[2 * a] || over_host[2 * a + 1]) fallback.push_back(u0 + a);
            else g_topk_fused_lists += 2;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    release();
#undef TL_TRY
    return 0;
}

int eval_topk(const DevModel& m, int precision, const int32_t* users_host, int n_users, int k, int32_t* items_host,
              double* scores_host, hipStream_t st)
{
    if (k < 1 || k > 256) { rec_set_error("k must be in 1..256"); return -1; }
    g_topk_fused_lists = 0; g_topk_fused_ms = 0;
    std::vector<int32_t> all;
    if (!users_host && (n_users < 0 || n_users > m.n_eval)) { rec_set_error("more users than have a test item"); return -1; }
    const bool fused = g_topk_fused && precision == REC_F32 && k <= 64 && m.K + m.KK > 0 && m.I < (1 << 28) &&
                       m.model != REC_MP && m.model != REC_TRANSREC_L1;
    // TransRec_L1: the lists come out of the |.|_1 sweep (fp64 scores, either table precision)
    const bool l1_lists = g_topk_fused && k <= 64 && m.model == REC_TRANSREC_L1;
    std::vector<int32_t> fallback; bool all_fallback = !fused && !l1_lists;
    if (fused) { if (int rc = topk_fused_impl(m, users_host, n_users, k, items_host, scores_host, st, fallback, &all_fallback)) return rc; }
    if (l1_lists) {
        if (int rc = precision == REC_F64 ? topk_l1_impl<double>(m, users_host, n_users, k, items_host, scores_host, st, fallback, &all_fallback)
                                          : topk_l1_impl<float>(m, users_host, n_users, k, items_host, scores_host, st, fallback, &all_fallback)) return rc;
    }
    if (!all_fallback && fallback.empty()) return 0;
    // the score-matrix path: everything (f64 tables, k > 64, MostPopular, operands out of the f16 sweep's range) or the
    // users whose lists the fused path gave up on
    if (!users_host) {
        all.resize(n_users);
        HIP_TRY(hipMemcpy(all.data(), m.eval_users, sizeof(int32_t) * n_users, hipMemcpyDeviceToHost));
        users_host = all.data();
    }
    std::vector<double> tmp_scores;
    if (all_fallback) {
        if (!scores_host) { tmp_scores.resize((size_t)n_users * 2 * k); scores_host = tmp_scores.data(); }
        return precision == REC_F64 ? topk_impl<double>(m, users_host, n_users, k, items_host, scores_host, st)
                                    : topk_impl<float>(m, users_host, n_users, k, items_host, scores_host, st);
    }
    std::vector<int32_t> fu(fallback.size()), fi(fallback.size() * 2 * k); std::vector<double> fs(fallback.size() * 2 * k);
    for (size_t a = 0; a < fallback.size(); a++) fu[a] = users_host[fallback[a]];
    if (int rc = precision == REC_F64 ? topk_impl<double>(m, fu.data(), (int)fu.size(), k, fi.data(), fs.data(), st)
                                      : topk_impl<float>(m, fu.data(), (int)fu.size(), k, fi.data(), fs.data(), st)) return rc;
    for (size_t a = 0; a < fallback.size(); a++) {
        memcpy(items_host + (size_t)fallback[a] * 2 * k, fi.data() + a * 2 * k, sizeof(int32_t) * 2 * k);
        if (scores_host) memcpy(scores_host + (size_t)fallback[a] * 2 * k, fs.data() + a * 2 * k, sizeof(double) * 2 * k);
    }
    return 0;
}

int eval_run(const DevModel& m, int precision, int32_t* counts_host, double* xpos_host, hipStream_t st, double* score_ms,
             int64_t* score_launches, bool profile, const int32_t* subset_host, int n_subset, int user_draws, uint64_t draw_seed,
             int32_t* draws_host)
{
    return precision == REC_F64 ? eval_impl<double>(m, counts_host, xpos_host, st, score_ms, score_launches, profile, subset_host, n_subset, user_draws, draw_seed, draws_host)
                                : eval_impl<float>(m, counts_host, xpos_host, st, score_ms, score_launches, profile, subset_host, n_subset, user_draws, draw_seed, draws_host);
}
