// Hogwild BPR-SGD step: one lane group (16, 32 or 64 lanes of a wavefront) per tuple.
//
// Lane g of a group owns components k = g, g+LPT, ... of every row the tuple names, so each row
// is read with one coalesced wave-instruction per EPL step (a 64-component f32 row = one 256-B
// request) and written back the same way: as hardware float atomics (one contiguous dword run per
// row, the shape global_atomic_add_f32 sustains its full rate on), or as plain racy stores.
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

template <class T, int LPT, int EPL>
struct Grp {
    int g;  // lane within the group
    __device__ Grp() : g(threadIdx.x & (LPT - 1)) {}
    __device__ __forceinline__ void load(const T* row, int K, T (&v)[EPL]) const
    {
#pragma unroll
        for (int j = 0; j < EPL; j++) { int k = g + LPT * j; v[j] = k < K ? row[k] : (T)0; }
    }
    __device__ __forceinline__ void add(T* row, int K, const T (&d)[EPL], const T (&old)[EPL], int atomic) const
    {
#pragma unroll
        for (int j = 0; j < EPL; j++) { int k = g + LPT * j; if (k < K) scatter(row + k, d[j], old[j], atomic); }
    }
    __device__ __forceinline__ static T sum(T v) { return group_sum<LPT>(v); }
    // value held by the group's lane 0
    __device__ __forceinline__ static T first(T v) { return __shfl(v, 0, LPT); }
};

template <class T> __device__ __forceinline__ T sigmoid_neg(T x) { return (T)1 / ((T)1 + rexp(x)); }

// ------------------------------------------------------------------------------------------------
template <int MODEL, class T, int LPT, int EPL>
__global__ void __launch_bounds__(256) sgd_hogwild_kernel(DevModel m, const int4* __restrict__ tuples, int64_t n, T lr,
                                                         int atomic)
{
    const Grp<T, LPT, EPL> G;
    const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / LPT);
    const int64_t gid = (int64_t)blockIdx.x * (blockDim.x / LPT) + threadIdx.x / LPT;
    const int K = m.K, KK = m.KK;
    const T lam = (T)m.lambda, breg = (T)m.bias_reg;

    // TransRec: gradient of the global translation r, summed over this thread's tuples
    T racc[EPL];
#pragma unroll
    for (int j = 0; j < EPL; j++) racc[j] = 0;

    for (int64_t i = gid; i < n; i += ngroups) {
        const int4 t = tuples[i];
        if constexpr (MODEL == REC_BPRMF) {
            // BPRMF.cpp:85-105
            T* beta = (T*)m.tab[BPRMF_BETA];
            T* gu = (T*)m.tab[BPRMF_GU] + (int64_t)t.x * K;
            T* gp = (T*)m.tab[BPRMF_GI] + (int64_t)t.y * K;
            T* gn = (T*)m.tab[BPRMF_GI] + (int64_t)t.z * K;
            T u[EPL], p[EPL], q[EPL];
            G.load(gu, K, u); G.load(gp, K, p); G.load(gn, K, q);
            T bp = 0, bn = 0;
            if (G.g == 0) { bp = beta[t.y]; bn = beta[t.z]; }
            T s = 0;
#pragma unroll
            for (int j = 0; j < EPL; j++) s += u[j] * (p[j] - q[j]);
            s = G.sum(s);
            const T deri = sigmoid_neg(G.first(bp - bn) + s);
            T du[EPL], dp[EPL], dq[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                du[j] = lr * (deri * (p[j] - q[j]) - lam * u[j]);
                dp[j] = lr * (deri * u[j] - lam * p[j]);
                dq[j] = lr * (-deri * u[j] - lam / (T)10.0 * q[j]);
            }
            G.add(gu, K, du, u, atomic); G.add(gp, K, dp, p, atomic); G.add(gn, K, dq, q, atomic);
            if (G.g == 0) {
                scatter(beta + t.y, lr * (deri - breg * bp), bp, atomic);
                scatter(beta + t.z, lr * (-deri - breg * bn), bn, atomic);
            }
        } else if constexpr (MODEL == REC_MC) {
            // MC.cpp:91-110
            T* gi = (T*)m.tab[MC_GI] + (int64_t)t.x * K;
            T* ep = (T*)m.tab[MC_ETA] + (int64_t)t.y * K;
            T* en = (T*)m.tab[MC_ETA] + (int64_t)t.z * K;
            T u[EPL], p[EPL], q[EPL];
            G.load(gi, K, u); G.load(ep, K, p); G.load(en, K, q);
            T s = 0;
#pragma unroll
            for (int j = 0; j < EPL; j++) s += u[j] * (p[j] - q[j]);
            const T deri = sigmoid_neg(G.sum(s));
            T du[EPL], dp[EPL], dq[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                du[j] = lr * (deri * (p[j] - q[j]) - lam * u[j]);
                dp[j] = lr * (deri * u[j] - lam * p[j]);
                dq[j] = lr * (-deri * u[j] - lam * q[j]);
            }
            G.add(gi, K, du, u, atomic); G.add(ep, K, dp, p, atomic); G.add(en, K, dq, q, atomic);
        } else if constexpr (MODEL == REC_FPMC) {
            // FPMC.cpp:88-122
            T* gu = (T*)m.tab[FPMC_GU] + (int64_t)t.x * K;
            T* gp = (T*)m.tab[FPMC_GI] + (int64_t)t.z * K;
            T* gn = (T*)m.tab[FPMC_GI] + (int64_t)t.w * K;
            T* ka = (T*)m.tab[FPMC_KAPPA] + (int64_t)t.y * KK;
            T* ep = (T*)m.tab[FPMC_ETA] + (int64_t)t.z * KK;
            T* en = (T*)m.tab[FPMC_ETA] + (int64_t)t.w * KK;
            T u[EPL], p[EPL], q[EPL], a[EPL], b[EPL], c[EPL];
            G.load(gu, K, u); G.load(gp, K, p); G.load(gn, K, q);
            G.load(ka, KK, a); G.load(ep, KK, b); G.load(en, KK, c);
            T s = 0;
#pragma unroll
            for (int j = 0; j < EPL; j++) s += u[j] * (p[j] - q[j]) + a[j] * (b[j] - c[j]);
            const T deri = sigmoid_neg(G.sum(s));
            T d0[EPL], d1[EPL], d2[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                d0[j] = lr * (deri * (p[j] - q[j]) - lam * u[j]);
                d1[j] = lr * (deri * u[j] - lam * p[j]);
                d2[j] = lr * (-deri * u[j] - lam * q[j]);
            }
            G.add(gu, K, d0, u, atomic); G.add(gp, K, d1, p, atomic); G.add(gn, K, d2, q, atomic);
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                d0[j] = lr * (deri * (b[j] - c[j]) - lam * a[j]);
                d1[j] = lr * (deri * a[j] - lam * b[j]);
                d2[j] = lr * (-deri * a[j] - lam * c[j]);
            }
            G.add(ka, KK, d0, a, atomic); G.add(ep, KK, d1, b, atomic); G.add(en, KK, d2, c, atomic);
        } else if constexpr (MODEL == REC_PRME) {
            // PRME.cpp:82-103
            const T alpha = (T)m.alpha;
            T* gu = (T*)m.tab[PRME_GU] + (int64_t)t.x * K;
            T* gp = (T*)m.tab[PRME_GI] + (int64_t)t.z * K;
            T* gn = (T*)m.tab[PRME_GI] + (int64_t)t.w * K;
            T* ei = (T*)m.tab[PRME_ETA] + (int64_t)t.y * KK;
            T* ep = (T*)m.tab[PRME_ETA] + (int64_t)t.z * KK;
            T* en = (T*)m.tab[PRME_ETA] + (int64_t)t.w * KK;
            T u[EPL], p[EPL], q[EPL], a[EPL], b[EPL], c[EPL];
            G.load(gu, K, u); G.load(gp, K, p); G.load(gn, K, q);
            G.load(ei, KK, a); G.load(ep, KK, b); G.load(en, KK, c);
            T s = 0;  // x_uij = -(dist to pos) + (dist to neg)
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                s += alpha * ((u[j] - q[j]) * (u[j] - q[j]) - (u[j] - p[j]) * (u[j] - p[j]));
                s += (1 - alpha) * ((a[j] - c[j]) * (a[j] - c[j]) - (a[j] - b[j]) * (a[j] - b[j]));
            }
            const T deri = sigmoid_neg(G.sum(s));
            T d0[EPL], d1[EPL], d2[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                d0[j] = lr * (deri * alpha * (p[j] - q[j]) - lam * u[j]);
                d1[j] = lr * (deri * alpha * (u[j] - p[j]) - lam * p[j]);
                d2[j] = lr * (deri * (-alpha) * (u[j] - q[j]) - lam * q[j]);
            }
            G.add(gu, K, d0, u, atomic); G.add(gp, K, d1, p, atomic); G.add(gn, K, d2, q, atomic);
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                d0[j] = lr * (deri * (1 - alpha) * (b[j] - c[j]) - lam * a[j]);
                d1[j] = lr * (deri * (1 - alpha) * (a[j] - b[j]) - lam * b[j]);
                d2[j] = lr * (deri * (alpha - 1) * (a[j] - c[j]) - lam * c[j]);
            }
            G.add(ei, KK, d0, a, atomic); G.add(ep, KK, d1, b, atomic); G.add(en, KK, d2, c, atomic);
        } else if constexpr (MODEL == REC_HRM_AVG || MODEL == REC_HRM_MAX) {
            // HRM_avg.cpp:168-195, HRM_max.cpp:178-208
            T* Uu = (T*)m.tab[HRM_U] + (int64_t)t.x * K;
            T* Vp = (T*)m.tab[HRM_V] + (int64_t)t.y * K;
            T* Vi = (T*)m.tab[HRM_V] + (int64_t)t.z * K;
            T* Vn = (T*)m.tab[HRM_V] + (int64_t)t.w * K;
            T u[EPL], a[EPL], p[EPL], q[EPL], h[EPL];
            G.load(Uu, K, u); G.load(Vp, K, a); G.load(Vi, K, p); G.load(Vn, K, q);
            T s = 0;
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                h[j] = MODEL == REC_HRM_MAX ? (u[j] > a[j] ? u[j] : a[j]) : (u[j] + a[j]);
                s += h[j] * (p[j] - q[j]);
            }
            const T deri = sigmoid_neg(G.sum(s));
            T d0[EPL], d1[EPL], d2[EPL], d3[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                const T deri_k = deri * (p[j] - q[j]);
                d2[j] = lr * (deri * h[j] - lam * p[j]);
                d3[j] = lr * (-deri * h[j] - lam * q[j]);
                const bool upd_u = MODEL == REC_HRM_AVG || u[j] > a[j];
                const bool upd_p = MODEL == REC_HRM_AVG || !(u[j] > a[j]);
                d0[j] = upd_u ? lr * (deri_k - lam * u[j]) : (T)0;
                d1[j] = upd_p ? lr * (deri_k - lam * a[j]) : (T)0;
            }
            G.add(Vi, K, d2, p, atomic); G.add(Vn, K, d3, q, atomic);
            G.add(Uu, K, d0, u, atomic); G.add(Vp, K, d1, a, atomic);
        } else if constexpr (MODEL == REC_TRANSREC || MODEL == REC_TRANSREC_L1) {
            // TransRec.cpp:208-258, TransRec_L1.cpp:207-259
            constexpr bool L1 = MODEL == REC_TRANSREC_L1;
            const T rreg = (T)m.relation_reg;
            T* beta = (T*)m.tab[TR_BETA];
            T* R = (T*)m.tab[TR_R] + (int64_t)t.x * K;
            T* Hx = (T*)m.tab[TR_H] + (int64_t)t.y * K;
            T* Hy = (T*)m.tab[TR_H] + (int64_t)t.z * K;
            T* Hn = (T*)m.tab[TR_H] + (int64_t)t.w * K;
            T r[EPL], u[EPL], x[EPL], y[EPL], yn[EPL], v[EPL], vn[EPL];
            G.load((T*)m.tab[TR_r], K, r); G.load(R, K, u); G.load(Hx, K, x); G.load(Hy, K, y); G.load(Hn, K, yn);
            T by = 0, bn = 0;
            if (G.g == 0) { by = beta[t.z]; bn = beta[t.w]; }
            T zs = 0;
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                v[j] = r[j] + u[j] + x[j] - y[j];
                vn[j] = r[j] + u[j] + x[j] - yn[j];
                zs += L1 ? (rabs(v[j]) - rabs(vn[j])) : (v[j] * v[j] - vn[j] * vn[j]);
            }
            const T z = G.first(-by + bn) - G.sum(zs);
            const T deri = sigmoid_neg(z);
            // new rows, then the unit-ball projection of the three item rows (TransRec.cpp:237-239);
            // padded lanes (k >= K) hold zeros and stay zero
            T nx[EPL], ny[EPL], nn[EPL], du[EPL];
            T sx = 0, sy = 0, sn = 0;
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                const bool live = G.g + LPT * j < K;
                T gx, gy, gn_, gr;
                if (!L1) {
                    gx = deri * 2 * (-v[j] + vn[j]) - lam * x[j];
                    gn_ = deri * 2 * (-vn[j]) - lam * yn[j];
                    gy = deri * 2 * (v[j]) - lam * y[j];
                    gr = deri * 2 * (-v[j] + vn[j]);
                } else {
                    const T tmp = (T)((v[j] > 0 ? -1 : 1) + (vn[j] > 0 ? 1 : -1));
                    gx = deri * tmp;
                    gy = deri * (T)(v[j] > 0 ? 1 : -1);
                    gn_ = deri * (T)(vn[j] > 0 ? -1 : 1);
                    gr = deri * tmp;
                }
                if (!live) { gx = gy = gn_ = gr = 0; }
                nx[j] = x[j] + lr * gx; ny[j] = y[j] + lr * gy; nn[j] = yn[j] + lr * gn_;
                du[j] = lr * (gr - rreg * u[j]);
                racc[j] += gr;
                sx += nx[j] * nx[j]; sy += ny[j] * ny[j]; sn += nn[j] * nn[j];
            }
            sx = rsqrt_(G.sum(sx)); sy = rsqrt_(G.sum(sy)); sn = rsqrt_(G.sum(sn));
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                if (sx > 1) nx[j] /= sx;
                if (sy > 1) ny[j] /= sy;
                if (sn > 1) nn[j] /= sn;
                nx[j] -= x[j]; ny[j] -= y[j]; nn[j] -= yn[j];
            }
            G.add(Hx, K, nx, x, atomic); G.add(Hy, K, ny, y, atomic); G.add(Hn, K, nn, yn, atomic);
            G.add(R, K, du, u, atomic);
            if (G.g == 0) {
                scatter(beta + t.z, lr * (-deri - breg * by), by, atomic);
                scatter(beta + t.w, lr * (deri - breg * bn), bn, atomic);
            }
        }
    }

    if constexpr (MODEL == REC_TRANSREC || MODEL == REC_TRANSREC_L1) {
        // block-level sum of the r gradients: component k lives in lane (k % LPT) of every group
        __shared__ T red[REC_MAX_K];
        for (int k = threadIdx.x; k < REC_MAX_K; k += blockDim.x) red[k] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPL; j++) { int k = G.g + LPT * j; if (k < K) unsafeAtomicAdd(&red[k], racc[j]); }
        __syncthreads();
        T* gacc = (T*)m.gacc;
        for (int k = threadIdx.x; k < K; k += blockDim.x) unsafeAtomicAdd(gacc + k, red[k]);
    }
}

// Folds the summed gradients of the always-touched rows into the parameters after a launch of
// n tuples.  n sequential steps of the reference move such a row by about lr * (sum of gradients
// - n * lambda * row) as long as it barely changes; the sum is applied with at most `cap` steps'
// worth of learning rate so that a large launch cannot overshoot.
template <class T>
__global__ void apply_global_kernel(DevModel m, T lr, int64_t n, T cap)
{
    T* gacc = (T*)m.gacc;
    const T scale = n > 0 ? ((T)n < cap ? (T)1 : cap / (T)n) : (T)0;
    if (m.model == REC_TRANSREC || m.model == REC_TRANSREC_L1) {
        T* r = (T*)m.tab[TR_r];
        for (int k = threadIdx.x; k < m.K; k += blockDim.x) {
            r[k] += lr * scale * (gacc[k] - (T)n * (T)m.lambda * r[k]);
            gacc[k] = 0;
        }
    }
}

template <int MODEL, class T>
static int dispatch_shape(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, hipStream_t st)
{
    const int Kmax = m.K > m.KK ? m.K : m.KK;
    // 2048 blocks of 256 threads keep 8 blocks per CU resident; groups stride over the tuples
    auto go = [&](auto kern, int lpt) {
        int64_t groups_per_block = 256 / lpt;
        int64_t blocks = (n + groups_per_block - 1) / groups_per_block;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, m, tuples, n, (T)lr, atomic);
    };
    if (Kmax <= 16) go(sgd_hogwild_kernel<MODEL, T, 16, 1>, 16);
    else if (Kmax <= 32) go(sgd_hogwild_kernel<MODEL, T, 32, 1>, 32);
    else if (Kmax <= 64) go(sgd_hogwild_kernel<MODEL, T, 64, 1>, 64);
    else if (Kmax <= 128) go(sgd_hogwild_kernel<MODEL, T, 64, 2>, 64);
    else if (Kmax <= 256) go(sgd_hogwild_kernel<MODEL, T, 64, 4>, 64);
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

int launch_apply_global(const DevModel& m, int precision, double lr, int64_t n_tuples, hipStream_t st)
{
    if (!(m.model == REC_TRANSREC || m.model == REC_TRANSREC_L1)) return 0;
    const double cap = 256.0;
    if (precision == REC_F64)
        hipLaunchKernelGGL(apply_global_kernel<double>, dim3(1), dim3(256), 0, st, m, lr, n_tuples, cap);
    else
        hipLaunchKernelGGL(apply_global_kernel<float>, dim3(1), dim3(256), 0, st, m, (float)lr, n_tuples, (float)cap);
    HIP_TRY(hipGetLastError());
    return 0;
}
