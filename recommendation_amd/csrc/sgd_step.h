// The BPR-SGD step of the dot-product / metric models on registers: which rows a tuple names
// (row_ptr, row_item, bias_ptr) and the arithmetic that turns loaded rows into deltas (step).
// Shared by the Hogwild kernels; device code only.
#pragma once
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

// table of row r of a tuple, and the row's number in it (row_ptr = row_table + row_number * width)
template <int MODEL, class T>
__device__ __forceinline__ T* row_table(const DevModel& m, int r)
{
    if constexpr (MODEL == REC_BPRMF) return (T*)m.tab[r == 0 ? BPRMF_GU : BPRMF_GI];
    else if constexpr (MODEL == REC_MC) return (T*)m.tab[r == 0 ? MC_GI : MC_ETA];
    else if constexpr (MODEL == REC_FPMC) return (T*)m.tab[r == 0 ? FPMC_GU : r < 3 ? FPMC_GI : r == 3 ? FPMC_KAPPA : FPMC_ETA];
    else if constexpr (MODEL == REC_PRME) return (T*)m.tab[r == 0 ? PRME_GU : r < 3 ? PRME_GI : PRME_ETA];
    else if constexpr (MODEL == REC_HRM_AVG || MODEL == REC_HRM_MAX) return (T*)m.tab[r == 0 ? HRM_U : HRM_V];
    else return (T*)m.tab[r == 0 ? TR_R : TR_H];
}
template <int MODEL>
__device__ __forceinline__ int row_number(const int4& t, int r)
{
    const int item = row_item<MODEL>(t, r);
    return item >= 0 ? item : t.x;   // rows without an item are indexed by the tuple's user
}

template <int MODEL, class T>
__device__ __forceinline__ T* bias_ptr(const DevModel& m, const int4& t, int s)
{
    if constexpr (MODEL == REC_BPRMF) return (T*)m.tab[BPRMF_BETA] + (s == 0 ? t.y : t.z);
    else return (T*)m.tab[TR_BETA] + (s == 0 ? t.z : t.w);
}

// ---- owner mode (see sgd_hogwild.hip)
// rows of a tuple indexed by its negative item, and where each sits inside the item's accumulator row
template <int MODEL> __device__ __forceinline__ constexpr bool owned_row(int r)
{
    if constexpr (MODEL == REC_BPRMF || MODEL == REC_MC) return r == 2;
    else if constexpr (MODEL == REC_FPMC || MODEL == REC_PRME) return r == 2 || r == 5;
    else return r == 3;   // HRM, TransRec
}
template <int MODEL> __device__ __forceinline__ int owned_off(int r, int K) { return ((MODEL == REC_FPMC || MODEL == REC_PRME) && r == 5) ? K : 0; }
template <int MODEL> __device__ __forceinline__ int neg_item(const int4& t) { return (MODEL == REC_BPRMF || MODEL == REC_MC) ? t.z : t.w; }

// reads an LDS accumulator element and leaves zero behind, atomically per element (an addition by another group lands
// either before the exchange, and is taken, or after it, and stays for the next flush)
__device__ __forceinline__ float lds_take(float* p) { return __uint_as_float(atomicExch((unsigned*)p, 0u)); }
__device__ __forceinline__ double lds_take(double* p) { return __longlong_as_double((long long)atomicExch((unsigned long long*)p, 0ull)); }

struct OwnerArgs {
    const int64_t* qoff;   // [slices + 1] where each slice's tuples start in the partitioned buffer
    int slice_rows;        // items per slice
    int rows_res;          // items of a slice with an accumulator row in LDS
    int acc_width;         // elements of an accumulator row (K, or K + KK)
    int flush_at;          // additions after which a row's accumulator is moved to memory
    int claim;             // plain mode: batches a group takes from the launch's counter at a time
};

// ------------------------------------------------------------------------------------------------
// ---- the step itself: rows v[R][EPL] and biases b[2] in, deltas out (in place).  LAYOUT: which components lane g
// holds - 0: g, g + LPT, ... (one dword per lane and access), 1: EPL * g .. EPL * g + EPL - 1 (one wide access per row)
template <int MODEL, class T, int LPT, int EPL, int LAYOUT = 0>
__device__ __forceinline__ T step(const DevModel& m, T lr, int g, T (&v)[Spec<MODEL>::R][EPL], T b0, T b1,
                                  const T (&rg)[EPL], T (&racc)[EPL], const bool (&at)[Spec<MODEL>::R])
{
    const T lam = (T)m.lambda;
    auto gsum = [](T x) { return group_sum_dpp<LPT>(x); };
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
            const bool live = (LAYOUT == 0 ? g + LPT * j : EPL * g + j) < m.K;
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
