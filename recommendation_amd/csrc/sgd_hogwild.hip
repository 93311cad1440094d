// Hogwild BPR-SGD step: one lane group (16, 32 or 64 lanes of a wavefront) per tuple, B tuples in
// flight per group.
//
// Lane g of a group owns components k = g, g+LPT, ... of every row the tuple names, so each row
// is read with one coalesced wave-instruction per EPL step (a 64-component f32 row = one 256-B
// request) and written back the same way: as hardware float atomics (one contiguous dword run per
// row, the shape global_atomic_add_f32 sustains its full rate on), or as plain racy stores.
// The loop is software-pipelined against HBM latency: while the rows of batch i are turned into
// deltas, the tuple indices of batch i+2 and the rows of batch i+1 are already in flight, and the
// atomics of batch i are issued behind them (vector-memory operations retire in issue order, so a
// wait for the next rows never waits for these atomics).
// All gradients of a tuple are computed from the values loaded at its start (the reference reads
// pre-update copies in the same places; where it does not, the difference is O(lr^2)).
// Rows that every tuple touches (TransRec's r) are not scattered per tuple: their gradients are
// summed per thread, per block through LDS, and land in DevModel::gacc with one atomic per block;
// launch_apply_global folds them into the row after the launch.
#include "common.h"

template <class T>
__device__ __forceinline__ void scatter(T* p, T delta, T oldv, int atomic)
{
    if (atomic) unsafeAtomicAdd(p, delta);
    else *p = oldv + delta;
}

template <class T> __device__ __forceinline__ T sigmoid_neg(T x) { return (T)1 / ((T)1 + rexp(x)); }

// ---- which rows a tuple names: R rows (table, row index, width) and up to two bias scalars
template <int MODEL> struct Spec;
template <> struct Spec<REC_BPRMF> { static constexpr int R = 3, NB = 2; };
template <> struct Spec<REC_MC> { static constexpr int R = 3, NB = 0; };
template <> struct Spec<REC_FPMC> { static constexpr int R = 6, NB = 0; };
template <> struct Spec<REC_PRME> { static constexpr int R = 6, NB = 0; };
template <> struct Spec<REC_HRM_AVG> { static constexpr int R = 4, NB = 0; };
template <> struct Spec<REC_HRM_MAX> { static constexpr int R = 4, NB = 0; };
template <> struct Spec<REC_TRANSREC> { static constexpr int R = 4, NB = 2; };
template <> struct Spec<REC_TRANSREC_L1> { static constexpr int R = 4, NB = 2; };

template <int MODEL, class T>
__device__ __forceinline__ T* row_ptr(const DevModel& m, const int4& t, int r, int& width)
{
    const int K = m.K, KK = m.KK;
    width = K;
    if constexpr (MODEL == REC_BPRMF) {           // (u, pos, neg): gamma_u, gamma_pos, gamma_neg
        return r == 0 ? (T*)m.tab[BPRMF_GU] + (int64_t)t.x * K : (T*)m.tab[BPRMF_GI] + (int64_t)(r == 1 ? t.y : t.z) * K;
    } else if constexpr (MODEL == REC_MC) {       // (i, pos, neg): gamma_i, eta_pos, eta_neg
        return r == 0 ? (T*)m.tab[MC_GI] + (int64_t)t.x * K : (T*)m.tab[MC_ETA] + (int64_t)(r == 1 ? t.y : t.z) * K;
    } else if constexpr (MODEL == REC_FPMC) {     // (u, prev, pos, neg): gamma_u, gamma_pos, gamma_neg, kappa_prev, eta_pos, eta_neg
        if (r == 0) return (T*)m.tab[FPMC_GU] + (int64_t)t.x * K;
        if (r < 3) return (T*)m.tab[FPMC_GI] + (int64_t)(r == 1 ? t.z : t.w) * K;
        width = KK;
        if (r == 3) return (T*)m.tab[FPMC_KAPPA] + (int64_t)t.y * KK;
        return (T*)m.tab[FPMC_ETA] + (int64_t)(r == 4 ? t.z : t.w) * KK;
    } else if constexpr (MODEL == REC_PRME) {     // (u, prev, pos, neg): gamma_u, gamma_pos, gamma_neg, eta_prev, eta_pos, eta_neg
        if (r == 0) return (T*)m.tab[PRME_GU] + (int64_t)t.x * K;
        if (r < 3) return (T*)m.tab[PRME_GI] + (int64_t)(r == 1 ? t.z : t.w) * K;
        width = KK;
        return (T*)m.tab[PRME_ETA] + (int64_t)(r == 3 ? t.y : r == 4 ? t.z : t.w) * KK;
    } else if constexpr (MODEL == REC_HRM_AVG || MODEL == REC_HRM_MAX) {  // (u, prev, pos, neg): U_u, V_prev, V_pos, V_neg
        return r == 0 ? (T*)m.tab[HRM_U] + (int64_t)t.x * K : (T*)m.tab[HRM_V] + (int64_t)(r == 1 ? t.y : r == 2 ? t.z : t.w) * K;
    } else {                                      // TransRec (u, x, y, yn): R_u, H_x, H_y, H_yn
        return r == 0 ? (T*)m.tab[TR_R] + (int64_t)t.x * K : (T*)m.tab[TR_H] + (int64_t)(r == 1 ? t.y : r == 2 ? t.z : t.w) * K;
    }
}

// item behind row r of the tuple (-1: the row is indexed by user)
template <int MODEL>
__device__ __forceinline__ int row_item(const int4& t, int r)
{
    if constexpr (MODEL == REC_BPRMF) return r == 0 ? -1 : r == 1 ? t.y : t.z;
    else if constexpr (MODEL == REC_MC) return r == 0 ? t.x : r == 1 ? t.y : t.z;
    else if constexpr (MODEL == REC_FPMC || MODEL == REC_PRME) return r == 0 ? -1 : (r == 1 || r == 4) ? t.z : (r == 2 || r == 5) ? t.w : t.y;
    else return r == 0 ? -1 : r == 1 ? t.y : r == 2 ? t.z : t.w;   // HRM, TransRec
}

template <int MODEL, class T>
__device__ __forceinline__ T* bias_ptr(const DevModel& m, const int4& t, int s)
{
    if constexpr (MODEL == REC_BPRMF) return (T*)m.tab[BPRMF_BETA] + (s == 0 ? t.y : t.z);
    else return (T*)m.tab[TR_BETA] + (s == 0 ? t.z : t.w);
}

// ---- the step itself: rows v[R][EPL] and biases b[2] in, deltas out (in place)
template <int MODEL, class T, int LPT, int EPL>
__device__ __forceinline__ T step(const DevModel& m, T lr, int g, T (&v)[Spec<MODEL>::R][EPL], T b0, T b1,
                                  const T (&rg)[EPL], T (&racc)[EPL], const bool (&at)[Spec<MODEL>::R])
{
    const T lam = (T)m.lambda;
    auto gsum = [](T x) { return group_sum<LPT>(x); };
    if constexpr (MODEL == REC_BPRMF) {
        // BPRMF.cpp:85-105
        T s = 0;
#pragma unroll
        for (int j = 0; j < EPL; j++) s += v[0][j] * (v[1][j] - v[2][j]);
        const T deri = sigmoid_neg((b0 - b1) + gsum(s));
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            const T u = v[0][j], p = v[1][j], q = v[2][j];
            v[0][j] = lr * (deri * (p - q) - lam * u);
            v[1][j] = lr * (deri * u - lam * p);
            v[2][j] = lr * (-deri * u - lam / (T)10.0 * q);
        }
        return deri;
    } else if constexpr (MODEL == REC_MC) {
        // MC.cpp:91-110
        T s = 0;
#pragma unroll
        for (int j = 0; j < EPL; j++) s += v[0][j] * (v[1][j] - v[2][j]);
        const T deri = sigmoid_neg(gsum(s));
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            const T u = v[0][j], p = v[1][j], q = v[2][j];
            v[0][j] = lr * (deri * (p - q) - lam * u);
            v[1][j] = lr * (deri * u - lam * p);
            v[2][j] = lr * (-deri * u - lam * q);
        }
        return deri;
    } else if constexpr (MODEL == REC_FPMC) {
        // FPMC.cpp:88-122
        T s = 0;
#pragma unroll
        for (int j = 0; j < EPL; j++) s += v[0][j] * (v[1][j] - v[2][j]) + v[3][j] * (v[4][j] - v[5][j]);
        const T deri = sigmoid_neg(gsum(s));
#pragma unroll
        for (int j = 0; j < EPL; j++) {
#pragma unroll
            for (int h = 0; h < 6; h += 3) {
                const T u = v[h][j], p = v[h + 1][j], q = v[h + 2][j];
                v[h][j] = lr * (deri * (p - q) - lam * u);
                v[h + 1][j] = lr * (deri * u - lam * p);
                v[h + 2][j] = lr * (-deri * u - lam * q);
            }
        }
        return deri;
    } else if constexpr (MODEL == REC_PRME) {
        // PRME.cpp:82-103; x_uij = -(distance to pos) + (distance to neg)
        const T alpha = (T)m.alpha;
        T s = 0;
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            s += alpha * ((v[0][j] - v[2][j]) * (v[0][j] - v[2][j]) - (v[0][j] - v[1][j]) * (v[0][j] - v[1][j]));
            s += (1 - alpha) * ((v[3][j] - v[5][j]) * (v[3][j] - v[5][j]) - (v[3][j] - v[4][j]) * (v[3][j] - v[4][j]));
        }
        const T deri = sigmoid_neg(gsum(s));
#pragma unroll
        for (int j = 0; j < EPL; j++) {
#pragma unroll
            for (int h = 0; h < 6; h += 3) {
                const T w = h == 0 ? alpha : 1 - alpha;
                const T u = v[h][j], p = v[h + 1][j], q = v[h + 2][j];
                v[h][j] = lr * (deri * w * (p - q) - lam * u);
                v[h + 1][j] = lr * (deri * w * (u - p) - lam * p);
                v[h + 2][j] = lr * (deri * (-w) * (u - q) - lam * q);
            }
        }
        return deri;
    } else if constexpr (MODEL == REC_HRM_AVG || MODEL == REC_HRM_MAX) {
        // HRM_avg.cpp:168-195, HRM_max.cpp:178-208
        T h[EPL]; T s = 0;
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            h[j] = MODEL == REC_HRM_MAX ? (v[0][j] > v[1][j] ? v[0][j] : v[1][j]) : (v[0][j] + v[1][j]);
            s += h[j] * (v[2][j] - v[3][j]);
        }
        const T deri = sigmoid_neg(gsum(s));
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            const T u = v[0][j], a = v[1][j], p = v[2][j], q = v[3][j];
            const T deri_k = deri * (p - q);
            const bool upd_u = MODEL == REC_HRM_AVG || u > a;
            v[0][j] = upd_u ? lr * (deri_k - lam * u) : (T)0;
            v[1][j] = (MODEL == REC_HRM_AVG || !upd_u) ? lr * (deri_k - lam * a) : (T)0;
            v[2][j] = lr * (deri * h[j] - lam * p);
            v[3][j] = lr * (-deri * h[j] - lam * q);
        }
        return deri;
    } else {
        // TransRec.cpp:208-258, TransRec_L1.cpp:207-259
        constexpr bool L1 = MODEL == REC_TRANSREC_L1;
        const T rreg = (T)m.relation_reg;
        // An item row may sit outside the unit ball while a launch is running (see below): it is read
        // as its projection, which is what the reference would have stored (TransRec.cpp:245-258).
        T h[3][EPL];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            T s = 0;
#pragma unroll
            for (int j = 0; j < EPL; j++) s += v[r + 1][j] * v[r + 1][j];
            s = rsqrt_(gsum(s));
#pragma unroll
            for (int j = 0; j < EPL; j++) h[r][j] = s > 1 ? v[r + 1][j] / s : v[r + 1][j];
        }
        T d[EPL], dn[EPL]; T zs = 0;
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            d[j] = rg[j] + v[0][j] + h[0][j] - h[1][j];
            dn[j] = rg[j] + v[0][j] + h[0][j] - h[2][j];
            zs += L1 ? (rabs(d[j]) - rabs(dn[j])) : (d[j] * d[j] - dn[j] * dn[j]);
        }
        const T deri = sigmoid_neg((-b0 + b1) - gsum(zs));
        // padded lanes (k >= K) hold zeros and must stay zero
        T nw[3][EPL]; T sq[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            const bool live = g + LPT * j < m.K;
            T gx, gy, gn, gr;
            if (!L1) {
                gx = deri * 2 * (-d[j] + dn[j]) - lam * h[0][j];
                gn = deri * 2 * (-dn[j]) - lam * h[2][j];
                gy = deri * 2 * (d[j]) - lam * h[1][j];
                gr = deri * 2 * (-d[j] + dn[j]);
            } else {
                const T tmp = (T)((d[j] > 0 ? -1 : 1) + (dn[j] > 0 ? 1 : -1));
                gx = deri * tmp;
                gy = deri * (T)(d[j] > 0 ? 1 : -1);
                gn = deri * (T)(dn[j] > 0 ? -1 : 1);
                gr = deri * tmp;
            }
            if (!live) { gx = gy = gn = gr = 0; }
            nw[0][j] = h[0][j] + lr * gx; nw[1][j] = h[1][j] + lr * gy; nw[2][j] = h[2][j] + lr * gn;
            racc[j] += gr;
            sq[0] += nw[0][j] * nw[0][j]; sq[1] += nw[1][j] * nw[1][j]; sq[2] += nw[2][j] * nw[2][j];
            v[0][j] = lr * (gr - rreg * v[0][j]);
        }
        // A row written with a plain store gets the projected new value (TransRec.cpp:237-239).  A row
        // scattered with atomics only receives its gradient step: a projection expressed as a delta would be
        // applied once per tuple in flight and overshoot; launch_apply_global projects every row after the launch.
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const T s = rsqrt_(gsum(sq[r]));
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                if (at[r + 1]) v[r + 1][j] = nw[r][j] - h[r][j];
                else v[r + 1][j] = (s > 1 ? nw[r][j] / s : nw[r][j]) - v[r + 1][j];
            }
        }
        return deri;
    }
}

// ------------------------------------------------------------------------------------------------
template <int MODEL, class T, int LPT, int EPL, int B>
__global__ void __launch_bounds__(256) sgd_hogwild_kernel(DevModel m, const int4* __restrict__ tuples, int64_t n, T lr,
                                                         int atomic)
{
    constexpr int R = Spec<MODEL>::R, NB = Spec<MODEL>::NB;
    constexpr bool TR = MODEL == REC_TRANSREC || MODEL == REC_TRANSREC_L1;
    const int g = threadIdx.x & (LPT - 1);
    const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / LPT);
    const int64_t gid = (int64_t)blockIdx.x * (blockDim.x / LPT) + threadIdx.x / LPT;
    const int64_t nbatch = (n + B - 1) / B;

    T racc[EPL], rg[EPL];
#pragma unroll
    for (int j = 0; j < EPL; j++) { racc[j] = 0; rg[j] = 0; }
    if constexpr (TR) {
        // the global translation r is read once: it does not change during a launch
#pragma unroll
        for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < m.K) rg[j] = ((const T*)m.tab[TR_r])[k]; }
    }

    auto load_idx = [&](int64_t batch, int4 (&tp)[B]) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int64_t i = batch * B + b;
            tp[b] = (batch < nbatch && i < n) ? tuples[i] : make_int4(-1, 0, 0, 0);
        }
    };
    // bias s of tuple b is held by lane 2b+s of the group: one load and one scatter instruction per batch
    auto my_bias_ptr = [&](const int4 (&tp)[B]) -> T* {
        T* p = nullptr;
        if constexpr (NB == 2) {
#pragma unroll
            for (int b = 0; b < B; b++)
                if ((g >> 1) == b && g < 2 * B && tp[b].x >= 0) p = bias_ptr<MODEL, T>(m, tp[b], g & 1);
        }
        return p;
    };
    auto load_rows = [&](const int4 (&tp)[B], T (&v)[B][R][EPL], T& bs) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            const bool ok = tp[b].x >= 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                int width; const T* row = row_ptr<MODEL, T>(m, tp[b], r, width);
#pragma unroll
                for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; v[b][r][j] = (ok && k < width) ? row[k] : (T)0; }
            }
        }
        bs = 0;
        if constexpr (NB == 2) { const T* p = my_bias_ptr(tp); if (p) bs = *p; }
    };

    int4 tp[B], tn[B];
    T v[B][R][EPL], bs;
    load_idx(gid, tp);
    load_idx(gid + ngroups, tn);
    load_rows(tp, v, bs);
    for (int64_t batch = gid; batch < nbatch; batch += ngroups) {
        // indices two batches ahead, rows one batch ahead
        int4 tnn[B];
        load_idx(batch + 2 * ngroups, tnn);
        T vn[B][R][EPL], bn;
        load_rows(tn, vn, bn);
        // keep the values the plain-store mode adds its deltas to
        T old[B][R][EPL]; T mine = 0;  // mine: deri of the tuple whose bias this lane holds
        bool at[B][R];
#pragma unroll
        for (int b = 0; b < B; b++) {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int j = 0; j < EPL; j++) old[b][r][j] = v[b][r][j];
            const T b0 = NB == 2 ? __shfl(bs, 2 * b, LPT) : (T)0, b1 = NB == 2 ? __shfl(bs, 2 * b + 1, LPT) : (T)0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                // rows of hot items fall back to plain stores: their atomics would serialize on one address
                const int it = row_item<MODEL>(tp[b], r);
                at[b][r] = ((atomic >> r) & 1) && !(tp[b].x >= 0 && it >= 0 && m.hot[it]);
            }
            const T deri = step<MODEL, T, LPT, EPL>(m, lr, g, v[b], b0, b1, rg, racc, at[b]);
            if ((g >> 1) == b) mine = deri;
        }
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (tp[b].x < 0) continue;
#pragma unroll
            for (int r = 0; r < R; r++) {
                int width; T* row = row_ptr<MODEL, T>(m, tp[b], r, width);
#pragma unroll
                for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < width) scatter(row + k, v[b][r][j], old[b][r][j], at[b][r]); }
            }
        }
        if constexpr (NB == 2) {
            // BPRMF.cpp:92-93: +deri on the positive, -deri on the negative; TransRec.cpp:225-226: the opposite signs
            T* p = my_bias_ptr(tp);
            if (p) {
                const T sgn = ((g & 1) == 0) == (MODEL == REC_BPRMF) ? (T)1 : (T)-1;
                const T* base = (const T*)m.tab[MODEL == REC_BPRMF ? BPRMF_BETA : TR_BETA];
                scatter(p, lr * (sgn * mine - (T)m.bias_reg * bs), bs, ((atomic >> 8) & 1) && !m.hot[p - base]);
            }
        }
        bs = bn;
#pragma unroll
        for (int b = 0; b < B; b++) {
            tp[b] = tn[b]; tn[b] = tnn[b];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int j = 0; j < EPL; j++) v[b][r][j] = vn[b][r][j];
        }
    }

    if constexpr (TR) {
        // block-level sum of the r gradients: component k lives in lane (k % LPT) of every group
        __shared__ T red[REC_MAX_K];
        for (int k = threadIdx.x; k < REC_MAX_K; k += blockDim.x) red[k] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < m.K) unsafeAtomicAdd(&red[k], racc[j]); }
        __syncthreads();
        T* gacc = (T*)m.gacc;
        for (int k = threadIdx.x; k < m.K; k += blockDim.x) unsafeAtomicAdd(gacc + k, red[k]);
    }
}

// Folds the summed gradients of the always-touched rows into the parameters after a launch of
// n tuples.  n sequential steps of the reference move such a row by about lr * (sum of gradients
// - n * lambda * row) as long as it barely changes; the sum is applied with at most `cap` steps'
// worth of learning rate so that one launch cannot overshoot the point where the sequential
// steps would have stopped moving it (the engine keeps launches of these models short instead).
template <class T>
__global__ void apply_global_kernel(DevModel m, T lr, int64_t n, T cap)
{
    T* gacc = (T*)m.gacc;
    const T scale = n > 0 ? ((T)n < cap ? (T)1 : cap / (T)n) : (T)0;
    T* r = (T*)m.tab[TR_r];
    for (int k = threadIdx.x; k < m.K; k += blockDim.x) {
        r[k] += lr * scale * (gacc[k] - (T)n * (T)m.lambda * r[k]);
        gacc[k] = 0;
    }
}

// every item row back onto the unit ball (TransRec.cpp:245-258), one wavefront per row
template <class T>
__global__ void __launch_bounds__(256) project_rows_kernel(T* __restrict__ tab, int64_t rows, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = w; r < rows; r += nw) {
        T* row = tab + r * K; T x[REC_MAX_EPL]; T s = 0;
#pragma unroll
        for (int j = 0; j < REC_MAX_EPL; j++) { const int k = lane + 64 * j; x[j] = k < K ? row[k] : (T)0; s += x[j] * x[j]; }
        s = rsqrt_(group_sum<64>(s));
        if (s > 1) {
#pragma unroll
            for (int j = 0; j < REC_MAX_EPL; j++) { const int k = lane + 64 * j; if (k < K) row[k] = x[j] / s; }
        }
    }
}

template <int MODEL, class T>
static int dispatch_shape(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, hipStream_t st)
{
    const int Kmax = m.K > m.KK ? m.K : m.KK;
    // 2048 blocks of 256 threads keep 8 blocks per CU resident; groups stride over the batches
    auto go = [&](auto kern, int lpt, int b) {
        const int64_t groups_per_block = 256 / lpt;
        int64_t blocks = ((n + b - 1) / b + groups_per_block - 1) / groups_per_block;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, m, tuples, n, (T)lr, atomic);
    };
    constexpr int R = Spec<MODEL>::R;
    constexpr int B1 = R <= 4 ? 4 : 2;  // tuples in flight per group, by register budget
    if (Kmax <= 16) go(sgd_hogwild_kernel<MODEL, T, 16, 1, B1>, 16, B1);
    else if (Kmax <= 32) go(sgd_hogwild_kernel<MODEL, T, 32, 1, B1>, 32, B1);
    else if (Kmax <= 64) go(sgd_hogwild_kernel<MODEL, T, 64, 1, B1>, 64, B1);
    else if (Kmax <= 128) go(sgd_hogwild_kernel<MODEL, T, 64, 2, 2>, 64, 2);
    else if (Kmax <= 256) go(sgd_hogwild_kernel<MODEL, T, 64, 4, 1>, 64, 1);
    else { rec_set_error("K > 256 is not supported"); return -1; }
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T>
static int dispatch_model(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, hipStream_t st)
{
    switch (m.model) {
    case REC_BPRMF: return dispatch_shape<REC_BPRMF, T>(m, tuples, n, lr, atomic, st);
    case REC_MC: return dispatch_shape<REC_MC, T>(m, tuples, n, lr, atomic, st);
    case REC_FPMC: return dispatch_shape<REC_FPMC, T>(m, tuples, n, lr, atomic, st);
    case REC_PRME: return dispatch_shape<REC_PRME, T>(m, tuples, n, lr, atomic, st);
    case REC_HRM_AVG: return dispatch_shape<REC_HRM_AVG, T>(m, tuples, n, lr, atomic, st);
    case REC_HRM_MAX: return dispatch_shape<REC_HRM_MAX, T>(m, tuples, n, lr, atomic, st);
    case REC_TRANSREC: return dispatch_shape<REC_TRANSREC, T>(m, tuples, n, lr, atomic, st);
    case REC_TRANSREC_L1: return dispatch_shape<REC_TRANSREC_L1, T>(m, tuples, n, lr, atomic, st);
    default:
        rec_set_error("Hogwild update kernel not available for this model");
        return -1;
    }
}

int launch_sgd_hogwild(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, int atomic,
                       hipStream_t st)
{
    if (n <= 0) return 0;
    return precision == REC_F64 ? dispatch_model<double>(m, tuples, n, lr, atomic, st)
                                : dispatch_model<float>(m, tuples, n, lr, atomic, st);
}

int launch_apply_global(const DevModel& m, int precision, double lr, int64_t n_tuples, double cap, hipStream_t st)
{
    if (!(m.model == REC_TRANSREC || m.model == REC_TRANSREC_L1)) return 0;
    int64_t blocks = ((int64_t)m.I + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (precision == REC_F64) {
        hipLaunchKernelGGL(apply_global_kernel<double>, dim3(1), dim3(256), 0, st, m, lr, n_tuples, cap);
        hipLaunchKernelGGL(project_rows_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (double*)m.tab[TR_H], (int64_t)m.I, m.K);
    } else {
        hipLaunchKernelGGL(apply_global_kernel<float>, dim3(1), dim3(256), 0, st, m, (float)lr, n_tuples, (float)cap);
        hipLaunchKernelGGL(project_rows_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (float*)m.tab[TR_H], (int64_t)m.I, m.K);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
