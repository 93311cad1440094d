// Device-side tuple sampling, parameter initialisation and precision conversion.
//
// One thread draws one tuple.  All randomness comes from Philox4x32-10 keyed by the seed and
// counted by (step, epoch, round), so a tuple depends only on (seed, epoch, step), never on the
// launch geometry.  The supports are those of the reference's samplers (cited per model); the
// differences are that eligible users are drawn from a precomputed list instead of by rejection
// (same distribution) and that BPRMF / MC positives are drawn with replacement (the reference
// swap-pops them without replacement inside an epoch, BPRMF.cpp:128-138, MC.cpp:132-142).
#include "common.h"

struct Rng {
    uint2 key; uint32_t s0, s1, ep, round; uint4 cur; int have;
    __device__ Rng(uint64_t seed, uint64_t epoch, uint64_t step)
        : key(make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(epoch >> 32))), s0((uint32_t)step),
          s1((uint32_t)(step >> 32)), ep((uint32_t)epoch), round(0), have(0) {}
    __device__ uint32_t next()
    {
        if (have == 0) { cur = philox4x32(make_uint4(s0, s1, ep, round++), key); have = 4; }
        uint32_t r = have == 4 ? cur.x : have == 3 ? cur.y : have == 2 ? cur.z : cur.w;
        have--;
        return r;
    }
};

#define MAX_TRIES 4096  // every rejection loop ends after this many draws

template <int MODEL>
__global__ void __launch_bounds__(256) sampler_kernel(DevModel m, int4* __restrict__ out, int64_t n, int64_t first_step,
                                                      uint64_t seed, uint64_t epoch)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rng(seed, epoch, (uint64_t)(first_step + i));
    const int e = m.elig[bounded(rng.next(), (uint32_t)m.n_elig)];
    int4 t = make_int4(e, 0, 0, 0);
    if constexpr (MODEL == REC_MC) {
        // MC.cpp:80-89, 139-147: source item with successors, a successor, a non-successor
        const int64_t a = m.succ_off[e]; const int cnt = (int)(m.succ_off[e + 1] - a);
        const int pos = m.succ_item[a + bounded(rng.next(), (uint32_t)cnt)];
        const int32_t* ss = m.sset_item + m.sset_off[e]; const int ns = (int)(m.sset_off[e + 1] - m.sset_off[e]);
        int neg = 0;
        for (int tr = 0; tr < MAX_TRIES; tr++) { neg = (int)bounded(rng.next(), (uint32_t)m.I); if (!sorted_contains(ss, ns, neg)) break; }
        t = make_int4(e, pos, neg, 0);
    } else {
        const int64_t so = m.seq_off[e]; const int nu = (int)(m.seq_off[e + 1] - so);
        const int32_t* seq = m.seq_item + so;
        const int32_t* clk = m.clk_item + m.clk_off[e]; const int nc = (int)(m.clk_off[e + 1] - m.clk_off[e]);
        if constexpr (MODEL == REC_BPRMF) {
            // BPRMF.cpp:74-83, 135-143
            const int pos = seq[bounded(rng.next(), (uint32_t)nu)];
            int neg = 0;
            for (int tr = 0; tr < MAX_TRIES; tr++) { neg = (int)bounded(rng.next(), (uint32_t)m.I); if (!sorted_contains(clk, nc, neg)) break; }
            t = make_int4(e, pos, neg, 0);
        } else if constexpr (MODEL == REC_FPMC || MODEL == REC_PRME || MODEL == REC_FOSSIL_SIMPLE || MODEL == REC_HRM_AVG ||
                             MODEL == REC_HRM_MAX) {
            // FPMC.cpp:140-151, PRME.cpp:120-132, FossilSimple.cpp:253-264, HRM_avg.cpp:153-159 + 171-174
            const int p = (int)bounded(rng.next(), (uint32_t)(nu - 1));
            const int prev = seq[p], pos = seq[p + 1];
            int neg = 0;
            for (int tr = 0; tr < MAX_TRIES; tr++) {
                neg = (int)bounded(rng.next(), (uint32_t)m.I);
                if (neg != pos && !(MODEL == REC_FOSSIL_SIMPLE && neg == prev)) break;
            }
            t = make_int4(e, prev, pos, neg);
        } else if constexpr (MODEL == REC_TRANSREC || MODEL == REC_TRANSREC_L1) {
            // TransRec.cpp:188-199
            const int p = (int)bounded(rng.next(), (uint32_t)(nu - 1));
            int neg = 0;
            for (int tr = 0; tr < MAX_TRIES; tr++) { neg = (int)bounded(rng.next(), (uint32_t)m.I); if (!sorted_contains(clk, nc, neg)) break; }
            t = make_int4(e, seq[p], seq[p + 1], neg);
        } else if constexpr (MODEL == REC_FOSSIL) {
            // Fossil.cpp:282-293: target position 1..nu-1, negative outside {pos} + the <= L previous items
            const int tp = 1 + (int)bounded(rng.next(), (uint32_t)(nu - 1));
            const int pos = seq[tp]; const int lo = tp - m.L > 0 ? tp - m.L : 0;
            int neg = 0;
            for (int tr = 0; tr < MAX_TRIES; tr++) {
                neg = (int)bounded(rng.next(), (uint32_t)m.I);
                bool bad = neg == pos;
                for (int j = lo; j < tp && !bad; j++) bad = seq[j] == neg;
                if (!bad) break;
            }
            t = make_int4(e, tp, neg, 0);
        } else if constexpr (MODEL == REC_REBUS) {
            // REBUS.cpp:536-550: negative outside {pos} + the matched path
            const int tp = 1 + (int)bounded(rng.next(), (uint32_t)(nu - 1));
            const int pos = seq[tp];
            const int64_t ctx = so + 2 * (int64_t)e + (tp - 1);
            const int32_t* path = m.path_item + m.path_off[ctx]; const int np = (int)(m.path_off[ctx + 1] - m.path_off[ctx]);
            int neg = 0;
            for (int tr = 0; tr < MAX_TRIES; tr++) {
                neg = (int)bounded(rng.next(), (uint32_t)m.I);
                bool bad = neg == pos;
                for (int j = 0; j < np && !bad; j++) bad = path[j] == neg;
                if (!bad) break;
            }
            t = make_int4(e, tp - 1, neg, 0);
        }
    }
    out[i] = t;
}

int launch_sampler(const DevModel& m, int4* tuples, int64_t n, int64_t first_step, uint64_t seed, uint64_t epoch,
                   hipStream_t st)
{
    if (n <= 0) return 0;
    if (m.n_elig <= 0) { rec_set_error("no user (or source item) is eligible for sampling"); return -1; }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define SAMPLER_CASE(M) case M: hipLaunchKernelGGL(sampler_kernel<M>, grid, block, 0, st, m, tuples, n, first_step, seed, epoch); break;
    switch (m.model) {
        SAMPLER_CASE(REC_BPRMF) SAMPLER_CASE(REC_MC) SAMPLER_CASE(REC_FPMC) SAMPLER_CASE(REC_PRME)
        SAMPLER_CASE(REC_HRM_AVG) SAMPLER_CASE(REC_HRM_MAX) SAMPLER_CASE(REC_TRANSREC) SAMPLER_CASE(REC_TRANSREC_L1)
        SAMPLER_CASE(REC_FOSSIL_SIMPLE) SAMPLER_CASE(REC_FOSSIL) SAMPLER_CASE(REC_REBUS)
    default: rec_set_error("model has no sampler"); return -1;
    }
#undef SAMPLER_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ precision conversion
template <class T>
__global__ void convert_in_kernel(const double* __restrict__ src, T* __restrict__ dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (T)src[i];
}
template <class T>
__global__ void convert_out_kernel(const T* __restrict__ src, double* __restrict__ dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}

static unsigned grid_for(int64_t n)
{
    int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : b > 8192 ? 8192 : b);
}

int launch_convert_in(const double* src, void* dst, int64_t n, int precision, hipStream_t st)
{
    if (n <= 0) return 0;
    if (precision == REC_F64) hipLaunchKernelGGL(convert_in_kernel<double>, dim3(grid_for(n)), dim3(256), 0, st, src, (double*)dst, n);
    else hipLaunchKernelGGL(convert_in_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, src, (float*)dst, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_convert_out(const void* src, double* dst, int64_t n, int precision, hipStream_t st)
{
    if (n <= 0) return 0;
    if (precision == REC_F64) hipLaunchKernelGGL(convert_out_kernel<double>, dim3(grid_for(n)), dim3(256), 0, st, (const double*)src, dst, n);
    else hipLaunchKernelGGL(convert_out_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)src, dst, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ parameter initialisation
// v = lo + (hi - lo) * uniform[0,1), one Philox call per 4 values
template <class T>
__global__ void fill_uniform_kernel(T* __restrict__ dst, int64_t n, double lo, double hi, uint64_t seed, uint32_t stream)
{
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += (int64_t)gridDim.x * blockDim.x) {
        const uint4 r = philox4x32(make_uint4((uint32_t)q, (uint32_t)(q >> 32), stream, 0u), key);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
        for (int j = 0; j < 4; j++)
            if (q * 4 + j < n) dst[q * 4 + j] = (T)(lo + (hi - lo) * ((double)rr[j] * (1.0 / 4294967296.0)));
    }
}

// rows onto the unit ball, TransRec.cpp:245-258 (one wavefront per row)
template <class T>
__global__ void unit_ball_rows_kernel(T* __restrict__ tab, int64_t rows, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = w; r < rows; r += nw) {
        T* row = tab + r * K; T s = 0;
        for (int k = lane; k < K; k += 64) s += row[k] * row[k];
        s = rsqrt_(group_sum<64>(s));
        if (s > 1) for (int k = lane; k < K; k += 64) row[k] /= s;
    }
}

template <class T>
static int param_init(const DevModel& m, uint64_t seed, hipStream_t st)
{
    const int64_t U = m.U, I = m.I, K = m.K, KK = m.KK, L = m.L;
    auto fill = [&](int t, int64_t n, double lo, double hi) {
        if (n > 0) hipLaunchKernelGGL(fill_uniform_kernel<T>, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, (T*)m.tab[t], n, lo, hi, seed, (uint32_t)t);
    };
    auto ball = [&](int t, int64_t rows) {
        hipLaunchKernelGGL(unit_ball_rows_kernel<T>, dim3(grid_for(rows * 64)), dim3(256), 0, st, (T*)m.tab[t], rows, (int)K);
    };
    switch (m.model) {
    case REC_BPRMF:  // BPRMF.cpp:11-21
        fill(BPRMF_BETA, I, 0, 0); fill(BPRMF_GU, U * K, 0, 1); fill(BPRMF_GI, I * K, 0, 1); break;
    case REC_MC:     // MC.cpp:30-32
        fill(MC_GI, I * K, 0, 1); fill(MC_ETA, I * K, 0, 1); break;
    case REC_FPMC:   // FPMC.cpp:11-13
        fill(FPMC_GU, U * K, 0, 1); fill(FPMC_GI, I * K, 0, 1); fill(FPMC_KAPPA, I * KK, 0, 1); fill(FPMC_ETA, I * KK, 0, 1); break;
    case REC_PRME:   // PRME.cpp:11-13
        fill(PRME_GU, U * K, 0, 1); fill(PRME_GI, I * K, 0, 1); fill(PRME_ETA, I * KK, 0, 1); break;
    case REC_HRM_AVG: case REC_HRM_MAX:  // HRM_avg.cpp:15-17
        fill(HRM_U, U * K, -1.0 / K, 1.0 / K); fill(HRM_V, I * K, -1.0 / K, 1.0 / K); break;
    case REC_TRANSREC: case REC_TRANSREC_L1: {  // TransRec.cpp:14-30
        const double range = 6.0 / sqrt((double)K);
        fill(TR_BETA, I, 0, 0); fill(TR_H, I * K, range, -range); fill(TR_R, U * K, 0, 0); fill(TR_r, K, range, -range);
        ball(TR_H, I); ball(TR_r, 1);
        break;
    }
    case REC_FOSSIL_SIMPLE:  // FossilSimple.cpp:11-17
        fill(FS_ALPHA, 1, -1e-6, 1e-6); fill(FS_ALPHA_U, U, -1e-6, 1e-6); fill(FS_BETA, I, 0, 0);
        fill(FS_UI, I * K, -1e-6, 1e-6); fill(FS_V, I * K, -1e-6, 1e-6); break;
    case REC_FOSSIL:         // Fossil.cpp:11-13
        fill(FO_BETA, I, -1e-6, 1e-6); fill(FO_WT, L, -1e-6, 1e-6); fill(FO_WTU, U * L, -1e-6, 1e-6);
        fill(FO_UI, I * K, -1e-6, 1e-6); fill(FO_V, I * K, -1e-6, 1e-6); break;
    case REC_REBUS:          // REBUS.cpp:102-104
        fill(RB_BETA, I + 1, 0, 1); fill(RB_P, (I + 1) * K, 0, 1); break;
    default: break;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_param_init(const DevModel& m, int precision, uint64_t seed, hipStream_t st)
{
    return precision == REC_F64 ? param_init<double>(m, seed, st) : param_init<float>(m, seed, st);
}

// ------------------------------------------------------------------ hot-item flags
__global__ void mark_hot_kernel(const int32_t* __restrict__ count, uint8_t* __restrict__ hot, int I, int64_t min_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= I) hot[i] = (i < I && min_count > 0 && count[i] >= min_count) ? 1 : 0;
}

int launch_mark_hot(const DevModel& m, uint8_t* hot, int64_t min_count, hipStream_t st)
{
    hipLaunchKernelGGL(mark_hot_kernel, dim3((m.I + 256) / 256), dim3(256), 0, st, m.train_count, hot, m.I, min_count);
    HIP_TRY(hipGetLastError());
    return 0;
}
