// Hogwild BPR-SGD step of the models whose tuple touches a variable number of rows: Fossil,
// FossilSimple and REBUS.  One wavefront per tuple; lane l owns components k = l, l+64, ...
//
// A tuple reads every item row of the user's history (Fossil: the whole training sequence with
// repeats, Fossil.cpp:140-153; REBUS: the distinct items of the prefix, REBUS.cpp:305-318) twice:
// pass 1 streams the rows into the two history sums (eight rows in flight per wave at K <= 64), pass 2 re-reads
// each row (an L2 / Infinity Cache hit) for its own regulariser term and scatters its delta with
// float atomics.  The rows of the CSR are consecutive in seq_item, so the item ids themselves are
// read as coalesced runs.
// Parameters that every tuple touches are not scattered per tuple (their atomics would all hit one
// address): Fossil's global weights WT[0..L) and FossilSimple's alpha collect their gradient sums
// per block and are folded in by launch_apply_global_history after the launch; they stay constant
// during a launch.
#include "common.h"

#define E REC_MAX_EPL

template <class T>
__device__ __forceinline__ void scat(T* p, T delta, T oldv, bool atomic)
{
    if (atomic) unsafeAtomicAdd(p, delta);
    else *p = oldv + delta;
}

template <class T> __device__ __forceinline__ T sigm(T x) { return (T)1 / ((T)1 + rexp(x)); }
template <class T> __device__ __forceinline__ T wsum(T v) { return group_sum<64>(v); }

// gacc layout: [0, L) gradient sums, [L, 2L) number of tuples that contributed (Fossil); [0], [1] for FossilSimple
template <int MODEL, class T, int EPL>
__global__ void __launch_bounds__(256) sgd_history_kernel(DevModel m, const int4* __restrict__ tuples, int64_t n, T lr, int atomic)
{
    constexpr int RIF = EPL == 1 ? 8 : 4;   // history rows in flight per wavefront
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int K = m.K, L = m.L, I = m.I;
    const T lam = (T)m.lambda, breg = (T)m.bias_reg;
    const bool at_item = (atomic >> 1) & 1, at_user = atomic & 1, at_bias = (atomic >> 8) & 1;

    // per-lane accumulators of the global gradients: lane `ind` holds WT[ind]'s (Fossil), lane 0 alpha's (FossilSimple)
    T gsum = 0, gcnt = 0;

    for (int64_t i = wid; i < n; i += nwaves) {
        const int4 t = tuples[i];
        const int u = t.x;
        const int64_t so = m.seq_off[u];
        const int32_t* seq = m.seq_item + so;
        const int nu = (int)(m.seq_off[u + 1] - so);

        if constexpr (MODEL == REC_FOSSIL || MODEL == REC_FOSSIL_SIMPLE) {
            constexpr bool SIMPLE = MODEL == REC_FOSSIL_SIMPLE;
            T* beta = (T*)m.tab[SIMPLE ? FS_BETA : FO_BETA];
            T* Ui = (T*)m.tab[SIMPLE ? FS_UI : FO_UI];
            T* V = (T*)m.tab[SIMPLE ? FS_V : FO_V];
            // tuple: Fossil (u, t, neg, -), FossilSimple (u, prev, pos, neg)
            const int tpos = SIMPLE ? 0 : t.y;
            const int pos = SIMPLE ? t.z : seq[tpos];
            const int neg = SIMPLE ? t.w : t.z;
            const int np = SIMPLE ? 1 : (tpos < L ? tpos : L);
            T* Vp = V + (int64_t)pos * K; T* Vn = V + (int64_t)neg * K;
            T vp[EPL], vn[EPL], sp[EPL], sn[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                const int k = lane + 64 * j; sp[j] = sn[j] = 0;
                vp[j] = k < K ? Vp[k] : (T)0; vn[j] = k < K ? Vn[k] : (T)0;
            }
            // pass 1: history sums, Fossil.cpp:140-153
            int cp = 0, cn = 0;
            for (int h0 = 0; h0 < nu; h0 += RIF) {
                int it[RIF]; T x[RIF][EPL];
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    it[q] = h0 + q < nu ? seq[h0 + q] : -1;
#pragma unroll
                    for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; x[q][j] = (it[q] >= 0 && k < K) ? Ui[(int64_t)it[q] * K + k] : (T)0; }
                }
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    if (it[q] < 0) continue;
                    if (it[q] != pos) cp++;
                    if (it[q] != neg) cn++;
#pragma unroll
                    for (int j = 0; j < EPL; j++) { if (it[q] != pos) sp[j] += x[q][j]; if (it[q] != neg) sn[j] += x[q][j]; }
                }
            }
            const T wp = cp > 0 ? rpow((T)cp, (T)-0.2) : (T)0, wn = cn > 0 ? rpow((T)cn, (T)-0.2) : (T)0;
            // short-term weights: lane ind holds WT[ind] + WTu[u][ind] (Fossil) / lane 0 alpha + alpha_u[u] (FossilSimple)
            T wt_g = 0, wt_u = 0;
            if (lane < np) {
                if (SIMPLE) { wt_g = ((T*)m.tab[FS_ALPHA])[0]; wt_u = ((T*)m.tab[FS_ALPHA_U])[u]; }
                else { wt_g = ((T*)m.tab[FO_WT])[lane]; wt_u = ((T*)m.tab[FO_WTU])[(int64_t)u * L + lane]; }
            }
            const T wt = wt_g + wt_u;
#pragma unroll
            for (int j = 0; j < EPL; j++) { sp[j] *= wp; sn[j] *= wn; }
            T grad_mine = 0;   // d loss / d weight of this lane's ind, filled below
            for (int ind = 0; ind < np; ind++) {
                const int c = SIMPLE ? t.y : seq[tpos - 1 - ind];
                const T w = __shfl(wt, ind, 64);
                T d1 = 0, d2 = 0;
#pragma unroll
                for (int j = 0; j < EPL; j++) {
                    const int k = lane + 64 * j;
                    const T x = k < K ? Ui[(int64_t)c * K + k] : (T)0;
                    // FossilSimple adds the short-term term on both sides unconditionally (FossilSimple.cpp:180-181)
                    if (SIMPLE || c != pos) sp[j] += w * x;
                    if (SIMPLE || c != neg) sn[j] += w * x;
                    d1 += x * vp[j]; d2 += x * vn[j];
                }
                d1 = wsum(d1); d2 = wsum(d2);
                // Fossil.cpp:194-206 (three cases); FossilSimple.cpp:195-196 (always the difference)
                const T gI = (SIMPLE || (c != pos && c != neg)) ? d1 - d2 : (c != pos ? d1 : -d2);
                if (lane == ind) grad_mine = gI;
            }
            T s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 0; j < EPL; j++) { s1 += sp[j] * vp[j]; s2 += sn[j] * vn[j]; }
            T bpos = 0, bneg = 0;
            if (lane == 0) { bpos = beta[pos]; bneg = beta[neg]; }
            const T x_uij = __shfl(bpos - bneg, 0, 64) + (wsum(s1) - wsum(s2));
            const T deri = sigm(x_uij);
            if (lane == 0) {
                scat(beta + pos, lr * (deri - breg * bpos), bpos, at_bias && !m.hot[pos]);
                scat(beta + neg, lr * (-deri - breg * bneg), bneg, at_bias && !m.hot[neg]);
            }
            if (lane < np) {
                // per-user weight now; the global weight only collects its gradient
                gsum += deri * grad_mine; gcnt += 1;
                if (SIMPLE) {
                    T* au = (T*)m.tab[FS_ALPHA_U] + u;
                    scat(au, lr * (deri * grad_mine - lam * wt_u), wt_u, at_user);
                } else {
                    const int c = seq[tpos - 1 - lane];
                    // "- lambda * WT[ind]" in the two aliasing branches, Fossil.cpp:201,205
                    const T regv = (c != pos && c != neg) ? wt_u : wt_g;
                    T* wu = (T*)m.tab[FO_WTU] + (int64_t)u * L + lane;
                    scat(wu, lr * (deri * grad_mine - lam * regv), wt_u, at_user);
                }
            }
            // pass 2: every history row, Fossil.cpp:209-247 / FossilSimple.cpp:198-226
            for (int h0 = 0; h0 < nu; h0 += RIF) {
                int it[RIF]; T x[RIF][EPL];
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    it[q] = h0 + q < nu ? seq[h0 + q] : -1;
#pragma unroll
                    for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; x[q][j] = (it[q] >= 0 && k < K) ? Ui[(int64_t)it[q] * K + k] : (T)0; }
                }
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    if (it[q] < 0) continue;
                    const int c = it[q];
                    T a = 0; bool inprev = false;
                    if (SIMPLE) { inprev = c == t.y; a = __shfl(wt, 0, 64); }
                    else {
                        int pt = -1;
                        for (int ind = 0; ind < np; ind++) if (seq[tpos - 1 - ind] == c) { pt = ind; break; }
                        inprev = pt >= 0;
                        a = __shfl(wt, pt >= 0 ? pt : 0, 64);
                    }
                    if (!inprev) a = 0;
                    T cpz, cnz;   // delta = lr * (deri * (cpz * Vp - cnz * Vn) - lambda * U_c)
                    if (SIMPLE && inprev) { cpz = wp + a; cnz = wn + a; }                       // FossilSimple.cpp:199-203
                    else if (c != pos) { cpz = wp + a; cnz = c != neg ? wn + a : (T)0; }
                    else { cpz = 0; cnz = wn + a; }
                    const bool atc = at_item && !m.hot[c];
                    T* row = Ui + (int64_t)c * K;
#pragma unroll
                    for (int j = 0; j < EPL; j++) {
                        const int k = lane + 64 * j;
                        if (k < K) scat(row + k, lr * (deri * (cpz * vp[j] - cnz * vn[j]) - lam * x[q][j]), x[q][j], atc);
                    }
                }
            }
            // Fossil.cpp:249-252
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                const int k = lane + 64 * j;
                if (k < K) {
                    scat(Vp + k, lr * (deri * sp[j] - lam * vp[j]), vp[j], at_item && !m.hot[pos]);
                    scat(Vn + k, lr * (-deri * sn[j] - lam * vn[j]), vn[j], at_item && !m.hot[neg]);
                }
            }
        } else {
            // ---------------- REBUS, REBUS.cpp:282-444; tuple (u, t-1, neg, -)
            T* beta = (T*)m.tab[RB_BETA]; T* P = (T*)m.tab[RB_P];
            const int tpos = t.y + 1, neg = t.z;
            const int pos = seq[tpos];
            const uint8_t* fo = m.first_occ + so;
            const int64_t ctx = so + 2 * (int64_t)u + (tpos - 1);
            const int32_t* path = m.path_item + m.path_off[ctx];
            const int np = (int)(m.path_off[ctx + 1] - m.path_off[ctx]);
            const double* etas = m.eta + (int64_t)np * L;
            T pp[EPL], pn[EPL], sp[EPL], sn[EPL], sl[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                const int k = lane + 64 * j; sp[j] = sn[j] = sl[j] = 0;
                pp[j] = k < K ? P[(int64_t)pos * K + k] : (T)0; pn[j] = k < K ? P[(int64_t)neg * K + k] : (T)0;
            }
            int cp = 0, cn = 0; bool pos_in = false, neg_in = false;
            for (int h0 = 0; h0 < tpos; h0 += RIF) {
                int it[RIF]; T x[RIF][EPL];
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    it[q] = (h0 + q < tpos && fo[h0 + q]) ? seq[h0 + q] : -1;
#pragma unroll
                    for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; x[q][j] = (it[q] >= 0 && k < K) ? P[(int64_t)it[q] * K + k] : (T)0; }
                }
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    if (it[q] < 0) continue;
                    if (it[q] != pos) cp++; else pos_in = true;
                    if (it[q] != neg) cn++; else neg_in = true;
#pragma unroll
                    for (int j = 0; j < EPL; j++) { if (it[q] != pos) sp[j] += x[q][j]; if (it[q] != neg) sn[j] += x[q][j]; }
                }
            }
            const T wp = cp > 0 ? rpow((T)cp, (T)m.alpha) : (T)0, wn = cn > 0 ? rpow((T)cn, (T)m.alpha) : (T)0;
            T proot[EPL];   // the Root row, when the path is empty (REBUS.cpp:414-418)
            for (int r = 0; r < np; r++) {
                const int c = path[r]; const T e = (T)etas[r];
#pragma unroll
                for (int j = 0; j < EPL; j++) {
                    const int k = lane + 64 * j;
                    const T x = k < K ? P[(int64_t)c * K + k] : (T)0;
                    sl[j] += e * x; proot[j] = x;
                }
            }
            T dp[EPL], dn[EPL]; T qp = 0, qn = 0;
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                dp[j] = wp * sp[j] + sl[j] - pp[j]; dn[j] = wn * sn[j] + sl[j] - pn[j];
                qp += dp[j] * dp[j]; qn += dn[j] * dn[j];
            }
            T bpos = 0, bneg = 0;
            if (lane == 0) { bpos = beta[pos]; bneg = beta[neg]; }
            const T x_uij = __shfl(-bpos + bneg, 0, 64) - wsum(qp) + wsum(qn);
            const T deri = sigm(x_uij);
            if (lane == 0) {
                scat(beta + pos, lr * (-deri - breg * bpos), bpos, at_bias && !m.hot[pos]);   // [9]
                scat(beta + neg, lr * (deri - breg * bneg), bneg, at_bias && !m.hot[neg]);    // [10]
            }
            for (int h0 = 0; h0 < tpos; h0 += RIF) {
                int it[RIF]; T x[RIF][EPL];
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    it[q] = (h0 + q < tpos && fo[h0 + q]) ? seq[h0 + q] : -1;
#pragma unroll
                    for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; x[q][j] = (it[q] >= 0 && k < K) ? P[(int64_t)it[q] * K + k] : (T)0; }
                }
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    if (it[q] < 0) continue;
                    const int c = it[q];
                    bool inpath = false; T se = 0;
                    for (int r = 0; r < np; r++) if (path[r] == c) { inpath = true; se += (T)etas[r]; }
                    T a, b;   // delta = lr * (deri * (a * dist_pos + b * dist_neg) - lambda * P_c)
                    if (inpath) {
                        if (c != pos) { if (c == neg) continue; a = -(wp + se); b = wn + se; }   // [1]
                        else { a = 1 - se; b = wn + se; }                                        // [4]
                    } else {
                        if (c != pos) { a = -wp; b = c != neg ? wn : (T)-1; }                    // [2] / [7]
                        else { a = 1; b = wn; }                                                  // [5]
                    }
                    const bool atc = at_item && !m.hot[c];
                    T* row = P + (int64_t)c * K;
#pragma unroll
                    for (int j = 0; j < EPL; j++) {
                        const int k = lane + 64 * j;
                        if (k < K) scat(row + k, lr * (deri * (a * dp[j] + b * dn[j]) - lam * x[q][j]), x[q][j], atc);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                const int k = lane + 64 * j;
                if (k >= K) continue;
                if (path[0] == I) scat(P + (int64_t)I * K + k, lr * (deri * (-dp[j] + dn[j]) - lam * proot[j]), proot[j], at_item);     // [3]
                if (!neg_in) scat(P + (int64_t)neg * K + k, lr * (deri * (-dn[j]) - lam * pn[j]), pn[j], at_item && !m.hot[neg]);       // [8]
                if (!pos_in) scat(P + (int64_t)pos * K + k, lr * (deri * (dp[j]) - lam * pp[j]), pp[j], at_item && !m.hot[pos]);        // [6]
            }
        }
    }

    if constexpr (MODEL == REC_FOSSIL || MODEL == REC_FOSSIL_SIMPLE) {
        __shared__ T red[2 * REC_MAX_L];
        for (int k = threadIdx.x; k < 2 * REC_MAX_L; k += blockDim.x) red[k] = 0;
        __syncthreads();
        const int nl = MODEL == REC_FOSSIL ? L : 1;
        if (lane < nl && gcnt > 0) { unsafeAtomicAdd(&red[lane], gsum); unsafeAtomicAdd(&red[REC_MAX_L + lane], gcnt); }
        __syncthreads();
        T* gacc = (T*)m.gacc;
        if ((int)threadIdx.x < nl && red[REC_MAX_L + threadIdx.x] > 0) {
            unsafeAtomicAdd(gacc + threadIdx.x, red[threadIdx.x]);
            unsafeAtomicAdd(gacc + REC_MAX_L + threadIdx.x, red[REC_MAX_L + threadIdx.x]);
        }
    }
}

// WT[ind] += lr/df * (deri*g - lambda/df * WT[ind]) per tuple (Fossil.cpp:197,200,204) and
// alpha += lr/10 * (deri*g - lambda/10 * alpha) (FossilSimple.cpp:195), summed over the launch and applied
// with at most `cap` steps' worth of learning rate.
template <class T>
__global__ void apply_global_history_kernel(DevModel m, T lr, T cap)
{
    T* gacc = (T*)m.gacc;
    const int ind = threadIdx.x;
    const bool fossil = m.model == REC_FOSSIL;
    const int nl = fossil ? m.L : 1;
    if (ind >= nl) return;
    const T n = gacc[REC_MAX_L + ind], g = gacc[ind];
    gacc[ind] = 0; gacc[REC_MAX_L + ind] = 0;
    if (n <= 0) return;
    const T df = fossil ? (T)m.df : (T)10;
    const T scale = n < cap ? (T)1 : cap / n;
    T* w = fossil ? (T*)m.tab[FO_WT] + ind : (T*)m.tab[FS_ALPHA];
    *w += lr / df * scale * (g - n * (T)m.lambda / df * (*w));
}

template <int MODEL, class T>
static int launch_history(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, int64_t max_in_flight, hipStream_t st)
{
    // one tuple per wavefront at a time: the number of wavefronts launched is the number of tuples in flight
    int64_t waves = n;
    if (max_in_flight > 0 && waves > max_in_flight) waves = max_in_flight;
    int64_t blocks = (waves + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks), block(256);
    if (m.K <= 64) hipLaunchKernelGGL((sgd_history_kernel<MODEL, T, 1>), grid, block, 0, st, m, tuples, n, (T)lr, atomic);
    else if (m.K <= 128) hipLaunchKernelGGL((sgd_history_kernel<MODEL, T, 2>), grid, block, 0, st, m, tuples, n, (T)lr, atomic);
    else hipLaunchKernelGGL((sgd_history_kernel<MODEL, T, 4>), grid, block, 0, st, m, tuples, n, (T)lr, atomic);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T>
static int dispatch_history(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, int64_t max_in_flight, hipStream_t st)
{
    switch (m.model) {
    case REC_FOSSIL: return launch_history<REC_FOSSIL, T>(m, tuples, n, lr, atomic, max_in_flight, st);
    case REC_FOSSIL_SIMPLE: return launch_history<REC_FOSSIL_SIMPLE, T>(m, tuples, n, lr, atomic, max_in_flight, st);
    case REC_REBUS: return launch_history<REC_REBUS, T>(m, tuples, n, lr, atomic, max_in_flight, st);
    default: rec_set_error("not a history model"); return -1;
    }
}

int launch_sgd_history(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, int atomic, int64_t max_in_flight,
                       hipStream_t st)
{
    if (n <= 0) return 0;
    return precision == REC_F64 ? dispatch_history<double>(m, tuples, n, lr, atomic, max_in_flight, st)
                                : dispatch_history<float>(m, tuples, n, lr, atomic, max_in_flight, st);
}

int launch_apply_global_history(const DevModel& m, int precision, double lr, double cap, hipStream_t st)
{
    if (!(m.model == REC_FOSSIL || m.model == REC_FOSSIL_SIMPLE)) return 0;
    if (precision == REC_F64) hipLaunchKernelGGL(apply_global_history_kernel<double>, dim3(1), dim3(64), 0, st, m, lr, cap);
    else hipLaunchKernelGGL(apply_global_history_kernel<float>, dim3(1), dim3(64), 0, st, m, (float)lr, (float)cap);
    HIP_TRY(hipGetLastError());
    return 0;
}
