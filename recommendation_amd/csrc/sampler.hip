// Device-side tuple sampling, parameter initialisation and precision conversion.
//
// One thread draws one tuple.  All randomness comes from Philox4x32-10 keyed by the seed and
// counted by (step, epoch, round), so a tuple depends only on (seed, epoch, step), never on the
// launch geometry.  The supports are those of the reference's samplers (cited per model); the
// differences are that eligible users are drawn from a precomputed list instead of by rejection
// (same distribution) and that BPRMF / MC positives are drawn with replacement (the reference
// swap-pops them without replacement inside an epoch, BPRMF.cpp:128-138, MC.cpp:132-142).
#include "common.h"
#include "sampler_dev.h"

template <int MODEL>
__global__ void __launch_bounds__(256) sampler_kernel(DevModel m, int4* __restrict__ out, int64_t n, int64_t first_step,
                                                      uint64_t seed, uint64_t epoch, int user_batch, int neg_share)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = sample_tuple<MODEL>(m, (uint64_t)(first_step + i), seed, epoch, user_batch, neg_share);
}


int launch_sampler(const DevModel& m, int4* tuples, int64_t n, int64_t first_step, uint64_t seed, uint64_t epoch,
                   hipStream_t st, int user_batch, int neg_share)
{
    if (n <= 0) return 0;
    if (m.n_elig <= 0) { rec_set_error("no user (or source item) is eligible for sampling"); return -1; }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define SAMPLER_CASE(M) case M: hipLaunchKernelGGL(sampler_kernel<M>, grid, block, 0, st, m, tuples, n, first_step, seed, epoch, user_batch, neg_share); break;
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

// ------------------------------------------------------------------ pairs of steps that share their positive item
// One thread draws steps 2p and 2p+1.  The first is drawn as sampler_kernel draws it: owner uniform, then one of the
// owner's events uniform.  The second is an event of the first one's positive item, taken with the events' own
// weights 1 / (events of the owner) through the item's alias table: summed over the first step's item this gives
// every event the probability the first step gives it, so both steps have the reference's distribution and only
// their joint distribution differs (same positive).  Both take their negative from one stream of candidates, each
// accepting the first one its own tuple allows.
template <int MODEL>
__global__ void __launch_bounds__(256) pair_sampler_kernel(DevModel m, int4* __restrict__ out, int64_t n, int64_t first_step,
                                                           uint64_t seed, uint64_t epoch)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * p >= n) return;
    const uint64_t step = (uint64_t)(first_step + 2 * p);
    Rng rng(seed, epoch, step);
    const int pick = (int)bounded(rng.next(), (uint32_t)m.n_elig);
    const int oa = m.elig_identity ? pick : m.elig[pick];
    const uint4 rec = *reinterpret_cast<const uint4*>(m.owner_rec + (int64_t)oa * 16);
    const int64_t off = (int64_t)(((uint64_t)rec.y << 32) | rec.x); const int cnt = (int)rec.z;
    int aux_a = 0, pos;
    if constexpr (MODEL == REC_MC) pos = m.succ_item[off + bounded(rng.next(), (uint32_t)cnt)];
    else if constexpr (MODEL == REC_BPRMF) pos = m.seq_item[off + bounded(rng.next(), (uint32_t)cnt)];
    else { const int q = (int)bounded(rng.next(), (uint32_t)(cnt - 1)); aux_a = m.seq_item[off + q]; pos = m.seq_item[off + q + 1]; }
    // the partner: an event of `pos`
    Rng rb(seed ^ 0x5851F42D4C957F2Dull, epoch, step + 1);
    const int64_t e0 = m.ev_off[pos]; const int ecnt = (int)(m.ev_off[pos + 1] - e0);
    int ob = oa, aux_b = aux_a;
    if (ecnt > 0) {
        int slot = (int)bounded(rb.next(), (uint32_t)ecnt);
        const float u01 = (float)(rb.next() >> 8) * (1.0f / 16777216.0f);
        if (!(u01 < m.ev_alias_p[e0 + slot])) slot = m.ev_alias_i[e0 + slot];
        const int4 ev = m.ev[e0 + slot];
        ob = ev.x; aux_b = ev.y;
    }
    auto rejected = [&](int owner, int item) {
        if constexpr (MODEL == REC_BPRMF) {
            if (!filter_may_contain(m.owner_rec, owner, item)) return false;
            const int64_t c0 = m.clk_off[owner];
            return sorted_contains(m.clk_item + c0, (int)(m.clk_off[owner + 1] - c0), item);
        } else if constexpr (MODEL == REC_MC) {
            if (!filter_may_contain(m.owner_rec, owner, item)) return false;
            const int64_t s0 = m.sset_off[owner];
            return sorted_contains(m.sset_item + s0, (int)(m.sset_off[owner + 1] - s0), item);
        } else {
            return item == pos;
        }
    };
    Rng rn(seed ^ 0x8CB92BA72F3D8DD7ull, epoch, step >> 1);
    int neg_a = -1, neg_b = -1;
    for (int tr = 0; tr < MAX_TRIES && (neg_a < 0 || neg_b < 0); tr++) {
        const int c = (int)bounded(rn.next(), (uint32_t)m.I);
        if (neg_a < 0 && !rejected(oa, c)) neg_a = c;
        if (neg_b < 0 && !rejected(ob, c)) neg_b = c;
    }
    auto fallback = [&](int owner) {
        if constexpr (MODEL == REC_BPRMF) {
            const int64_t c0 = m.clk_off[owner]; const int nc = (int)(m.clk_off[owner + 1] - c0);
            return kth_nonmember(m.clk_item + c0, nc, (int)bounded(rn.next(), (uint32_t)(m.I - nc)));
        } else if constexpr (MODEL == REC_MC) {
            const int64_t s0 = m.sset_off[owner]; const int ns = (int)(m.sset_off[owner + 1] - s0);
            return kth_nonmember(m.sset_item + s0, ns, (int)bounded(rn.next(), (uint32_t)(m.I - ns)));
        } else {
            const int c = (int)bounded(rn.next(), (uint32_t)(m.I - 1));   // uniform over the catalog without pos
            return c >= pos ? c + 1 : c;
        }
    };
    if (neg_a < 0) neg_a = fallback(oa);
    if (neg_b < 0) neg_b = fallback(ob);
    constexpr bool THREE = MODEL == REC_BPRMF || MODEL == REC_MC;
    out[2 * p] = THREE ? make_int4(oa, pos, neg_a, 0) : make_int4(oa, aux_a, pos, neg_a);
    if (2 * p + 1 < n) out[2 * p + 1] = THREE ? make_int4(ob, pos, neg_b, 0) : make_int4(ob, aux_b, pos, neg_b);
}

int launch_pair_sampler(const DevModel& m, int4* tuples, int64_t n, int64_t first_step, uint64_t seed, uint64_t epoch, hipStream_t st)
{
    if (n <= 0) return 0;
    if (m.n_elig <= 0) { rec_set_error("no user (or source item) is eligible for sampling"); return -1; }
    if (m.n_ev <= 0 || !m.ev_off) { rec_set_error("the event index has not been built"); return -1; }
    if (first_step & 1) { rec_set_error("a launch of the pair sampler starts on an even step"); return -1; }
    const int64_t pairs = (n + 1) / 2;
    const dim3 grid((unsigned)((pairs + 255) / 256)), block(256);
#define PAIR_CASE(M) case M: hipLaunchKernelGGL(pair_sampler_kernel<M>, grid, block, 0, st, m, tuples, n, first_step, seed, epoch); break;
    switch (m.model) {
        PAIR_CASE(REC_BPRMF) PAIR_CASE(REC_MC) PAIR_CASE(REC_FPMC) PAIR_CASE(REC_PRME) PAIR_CASE(REC_HRM_AVG) PAIR_CASE(REC_HRM_MAX)
    default: rec_set_error("the pair sampler is not available for this model"); return -1;
    }
#undef PAIR_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ event-major sampling
// The tuple-major sampler draws an owner uniformly and then one of the owner's events (a training
// position; for MC a transition) uniformly: event e is hit with probability w_e / n_elig per tuple,
// w_e = 1 / (events of its owner).  n independent tuples therefore contain event e Binomial(n, p_e)
// times; here every event draws its own count Poisson(n * p_e) instead - the same process with the
// total Poisson(n) rather than exactly n (relative spread n^-1/2) - which needs no communication
// between events.  The events are stored ordered by positive item, so the chunk comes out grouped by
// positive item without a sort.  Negatives are drawn per tuple exactly as in sampler_kernel.
constexpr int EV_PER_THREAD = 4, EV_TILE = 256 * EV_PER_THREAD;

__device__ __forceinline__ int poisson_draw(float lam, uint32_t r0, uint32_t r1)
{
    if (!(lam > 0.f)) return 0;
    const float u = ((float)(r0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    if (lam < 64.f) {
        // inversion by sequential search
        float p = __expf(-lam), s = p; int k = 0;
        while (u > s && k < 512) { k++; p *= lam / (float)k; s += p; }
        return k;
    }
    // normal approximation with continuity correction
    const float u2 = (float)(r1 >> 8) * (1.0f / 16777216.0f);
    const float z = sqrtf(-2.f * __logf(u)) * __cosf(6.28318530718f * u2);
    const float k = floorf(lam + sqrtf(lam) * z + 0.5f);
    return k < 0.f ? 0 : (int)k;
}

__device__ __forceinline__ int event_count(const DevModel& m, int64_t e, float scale, uint64_t seed, uint64_t epoch, uint64_t chunk_no)
{
    if (e >= m.n_ev) return 0;
    Rng rng(seed ^ 0x9E3779B97F4A7C15ull, epoch, (chunk_no << 40) | (uint64_t)e);
    const uint32_t r0 = rng.next(), r1 = rng.next();
    return poisson_draw(scale * m.ev_w[e], r0, r1);
}

__device__ __forceinline__ int block_exclusive_scan_256(int x, int* total)
{
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < wv; q++) base += wsum[q];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - x;
}

__global__ void __launch_bounds__(256) event_count_kernel(DevModel m, float scale, uint64_t seed, uint64_t epoch, uint64_t chunk_no,
                                                         int64_t* __restrict__ block_sum)
{
    const int64_t base = (int64_t)blockIdx.x * EV_TILE + threadIdx.x * EV_PER_THREAD;
    int s = 0;
#pragma unroll
    for (int q = 0; q < EV_PER_THREAD; q++) s += event_count(m, base + q, scale, seed, epoch, chunk_no);
    int total;
    (void)block_exclusive_scan_256(s, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

// exclusive scan of block_sum[0, nblocks) in place by one block; *n_out = the total
__global__ void __launch_bounds__(256) event_scan_kernel(int64_t* __restrict__ block_sum, int64_t nblocks, int64_t* __restrict__ n_out)
{
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    // a thread owns 16 consecutive sums per round: 4096 per round
    for (int64_t t0 = 0; t0 < nblocks; t0 += 4096) {
        const int64_t b = t0 + threadIdx.x * 16;
        int64_t x[16]; int s = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) { x[q] = b + q < nblocks ? block_sum[b + q] : 0; s += (int)x[q]; }
        int total;
        const int ex = block_exclusive_scan_256(s, &total);
        int64_t run = carry_s + ex;
#pragma unroll
        for (int q = 0; q < 16; q++) { if (b + q < nblocks) block_sum[b + q] = run; run += x[q]; }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = carry_s;
}

template <int MODEL>
__global__ void __launch_bounds__(256) event_emit_kernel(DevModel m, int4* __restrict__ out, int64_t cap, float scale, int64_t first_step,
                                                        uint64_t chunk_no, uint64_t seed, uint64_t epoch,
                                                        const int64_t* __restrict__ block_off)
{
    __shared__ int prefix[EV_TILE + 1];
    const int64_t ebase = (int64_t)blockIdx.x * EV_TILE;
    int c[EV_PER_THREAD], s = 0;
#pragma unroll
    for (int q = 0; q < EV_PER_THREAD; q++) { c[q] = event_count(m, ebase + threadIdx.x * EV_PER_THREAD + q, scale, seed, epoch, chunk_no); s += c[q]; }
    int total;
    int ex = block_exclusive_scan_256(s, &total);
#pragma unroll
    for (int q = 0; q < EV_PER_THREAD; q++) { prefix[threadIdx.x * EV_PER_THREAD + q] = ex; ex += c[q]; }
    if (threadIdx.x == 255) prefix[EV_TILE] = ex;
    __syncthreads();
    const int64_t obase = block_off[blockIdx.x];
    for (int j = threadIdx.x; j < total; j += 256) {
        // the event this tuple belongs to: last le with prefix[le] <= j
        int lo = 0, hi = EV_TILE;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (prefix[mid] <= j) lo = mid; else hi = mid; }
        const int4 ev = m.ev[ebase + lo];
        const int64_t oi = obase + j;
        if (oi >= cap) continue;
        Rng rng(seed, epoch, (uint64_t)(first_step + oi));
        int neg = 0;
        int4 t;
        if constexpr (MODEL == REC_BPRMF) {
            const int32_t* clk = m.clk_item + m.clk_off[ev.x]; const int nc = (int)(m.clk_off[ev.x + 1] - m.clk_off[ev.x]);
            neg = -1;
            for (int tr = 0; tr < MAX_TRIES && neg < 0; tr++) { const int c = (int)bounded(rng.next(), (uint32_t)m.I); if (!filter_may_contain(m.owner_rec, ev.x, c) || !sorted_contains(clk, nc, c)) neg = c; }
            if (neg < 0) neg = kth_nonmember(clk, nc, (int)bounded(rng.next(), (uint32_t)(m.I - nc)));
            t = make_int4(ev.x, ev.z, neg, 0);
        } else if constexpr (MODEL == REC_MC) {
            const int32_t* ss = m.sset_item + m.sset_off[ev.x]; const int ns = (int)(m.sset_off[ev.x + 1] - m.sset_off[ev.x]);
            neg = -1;
            for (int tr = 0; tr < MAX_TRIES && neg < 0; tr++) { const int c = (int)bounded(rng.next(), (uint32_t)m.I); if (!filter_may_contain(m.owner_rec, ev.x, c) || !sorted_contains(ss, ns, c)) neg = c; }
            if (neg < 0) neg = kth_nonmember(ss, ns, (int)bounded(rng.next(), (uint32_t)(m.I - ns)));
            t = make_int4(ev.x, ev.z, neg, 0);
        } else {
            for (int tr = 0; tr < MAX_TRIES; tr++) { neg = (int)bounded(rng.next(), (uint32_t)m.I); if (neg != ev.z) break; }
            t = make_int4(ev.x, ev.y, ev.z, neg);
        }
        out[oi] = t;
    }
}

size_t event_scratch_bytes(int64_t n_ev)
{
    return sizeof(int64_t) * (size_t)((n_ev + EV_TILE - 1) / EV_TILE + 8);
}

int launch_event_sampler(const DevModel& m, int4* tuples, int64_t cap, int64_t n_expected, int64_t first_step, uint64_t chunk_no,
                         uint64_t seed, uint64_t epoch, void* scratch, int64_t* n_out_dev, hipStream_t st)
{
    if (m.n_ev <= 0 || m.n_elig <= 0) { rec_set_error("no sampling events for this model"); return -1; }
    const int64_t nblocks = (m.n_ev + EV_TILE - 1) / EV_TILE;
    int64_t* block_sum = (int64_t*)scratch;
    const float scale = (float)((double)n_expected / (double)m.n_elig);
    hipLaunchKernelGGL(event_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, m, scale, seed, epoch, chunk_no, block_sum);
    hipLaunchKernelGGL(event_scan_kernel, dim3(1), dim3(256), 0, st, block_sum, nblocks, n_out_dev);
#define EMIT_CASE(M) case M: hipLaunchKernelGGL(event_emit_kernel<M>, dim3((unsigned)nblocks), dim3(256), 0, st, m, tuples, cap, scale, first_step, chunk_no, seed, epoch, (const int64_t*)block_sum); break;
    switch (m.model) {
        EMIT_CASE(REC_BPRMF) EMIT_CASE(REC_MC) EMIT_CASE(REC_FPMC) EMIT_CASE(REC_PRME) EMIT_CASE(REC_HRM_AVG) EMIT_CASE(REC_HRM_MAX)
    default: rec_set_error("the event-major sampler is not available for this model"); return -1;
    }
#undef EMIT_CASE
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
// elements [first, last) of the table whose element 0 is (or would be) at dst: a value depends only on its position in
// the whole table, so an engine that holds a range of the rows gets the values the whole table would have there
__global__ void fill_uniform_kernel(T* __restrict__ dst, int64_t first, int64_t last, double lo, double hi, uint64_t seed, uint32_t stream)
{
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    for (int64_t q = first / 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < last; q += (int64_t)gridDim.x * blockDim.x) {
        const uint4 r = philox4x32(make_uint4((uint32_t)q, (uint32_t)(q >> 32), stream, 0u), key);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
        for (int j = 0; j < 4; j++)
            if (q * 4 + j >= first && q * 4 + j < last) dst[q * 4 + j] = (T)(lo + (hi - lo) * ((double)rr[j] * (1.0 / 4294967296.0)));
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
    const int64_t I = m.I, K = m.K, KK = m.KK, L = m.L;
    auto fill = [&](int t, int64_t n, double lo, double hi) {
        if (n > 0) hipLaunchKernelGGL(fill_uniform_kernel<T>, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, (T*)m.tab[t], (int64_t)0, n, lo, hi, seed, (uint32_t)t);
    };
    // a table indexed by user: the rows of the engine's own users
    auto fill_users = [&](int t, int64_t cols, double lo, double hi) {
        const int64_t a = (int64_t)m.user_begin * cols, b = (int64_t)m.user_end * cols;
        if (b > a) hipLaunchKernelGGL(fill_uniform_kernel<T>, dim3(grid_for((b - a + 3) / 4 + 1)), dim3(256), 0, st, (T*)m.tab[t], a, b, lo, hi, seed, (uint32_t)t);
    };
    auto ball = [&](int t, int64_t rows) {
        hipLaunchKernelGGL(unit_ball_rows_kernel<T>, dim3(grid_for(rows * 64)), dim3(256), 0, st, (T*)m.tab[t], rows, (int)K);
    };
    switch (m.model) {
    case REC_BPRMF:  // BPRMF.cpp:11-21
        fill(BPRMF_BETA, I, 0, 0); fill_users(BPRMF_GU, K, 0, 1); fill(BPRMF_GI, I * K, 0, 1); break;
    case REC_MC:     // MC.cpp:30-32
        fill(MC_GI, I * K, 0, 1); fill(MC_ETA, I * K, 0, 1); break;
    case REC_FPMC:   // FPMC.cpp:11-13
        fill_users(FPMC_GU, K, 0, 1); fill(FPMC_GI, I * K, 0, 1); fill(FPMC_KAPPA, I * KK, 0, 1); fill(FPMC_ETA, I * KK, 0, 1); break;
    case REC_PRME:   // PRME.cpp:11-13
        fill_users(PRME_GU, K, 0, 1); fill(PRME_GI, I * K, 0, 1); fill(PRME_ETA, I * KK, 0, 1); break;
    case REC_HRM_AVG: case REC_HRM_MAX:  // HRM_avg.cpp:15-17
        fill_users(HRM_U, K, -1.0 / K, 1.0 / K); fill(HRM_V, I * K, -1.0 / K, 1.0 / K); break;
    case REC_TRANSREC: case REC_TRANSREC_L1: {  // TransRec.cpp:14-30
        const double range = 6.0 / sqrt((double)K);
        fill(TR_BETA, I, 0, 0); fill(TR_H, I * K, range, -range); fill_users(TR_R, K, 0, 0); fill(TR_r, K, range, -range);
        ball(TR_H, I); ball(TR_r, 1);
        break;
    }
    case REC_FOSSIL_SIMPLE:  // FossilSimple.cpp:11-17
        fill(FS_ALPHA, 1, -1e-6, 1e-6); fill_users(FS_ALPHA_U, 1, -1e-6, 1e-6); fill(FS_BETA, I, 0, 0);
        fill(FS_UI, I * K, -1e-6, 1e-6); fill(FS_V, I * K, -1e-6, 1e-6); break;
    case REC_FOSSIL:         // Fossil.cpp:11-13
        fill(FO_BETA, I, -1e-6, 1e-6); fill(FO_WT, L, -1e-6, 1e-6); fill_users(FO_WTU, L, -1e-6, 1e-6);
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
