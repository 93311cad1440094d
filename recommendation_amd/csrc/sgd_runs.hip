// Hogwild BPR-SGD over tuples grouped by their positive item.
//
// Float atomics are the scarce resource of the scatter (they execute at the memory side at a fixed
// chip-wide byte rate), and in every model here one item of the tuple - the positive - is drawn from
// the data, so a chunk of millions of tuples names each such item many times.  The event-major
// sampler (sampler.hip) emits a chunk with those tuples adjacent; sgd_runs_kernel walks a contiguous
// segment per lane group, keeps the deltas of the positive's rows in registers while the
// positive stays the same, feeds them back into the values it loads (so the run sees its own
// updates, as the sequential reference would), and issues one atomic row per run instead of
// one per tuple.  The other rows are scattered per tuple exactly as in sgd_hogwild_kernel.
// On tuples in any other order the kernel is still correct: every run then has length one.
//
// How long a run may stay private depends on who else works on the item at the same time:
//   * a run inside one segment has the item to itself: one scatter at its end (sequentially exact);
//   * a run that continues in a neighbouring segment is shared by two or three groups: it is
//     scattered every `shared` tuples, so the others see it with bounded delay;
//   * an item that fills whole segments is worked on by many groups at once.  Private sums of such
//     an item would be applied once per group and overshoot, and its atomics would queue on one
//     address, so its rows are written back with plain stores every `hot` tuples: concurrent
//     writers overwrite each other (the classic lock-free Hogwild store), which damps instead of
//     amplifying.
// After a scatter in the middle of a run the rows already loaded for the next batch do not hold it
// yet; those tuples read values that are at most one scatter interval old.
#include "common.h"
#include "sgd_step.h"

// the positive item of a tuple and the rows (and bias slot 0) it names
template <int MODEL> struct KeyRows;
template <> struct KeyRows<REC_BPRMF> { static constexpr int NK = 1; static constexpr int rows[2] = {1, -1}; };
template <> struct KeyRows<REC_MC> { static constexpr int NK = 1; static constexpr int rows[2] = {1, -1}; };
template <> struct KeyRows<REC_FPMC> { static constexpr int NK = 2; static constexpr int rows[2] = {1, 4}; };
template <> struct KeyRows<REC_PRME> { static constexpr int NK = 2; static constexpr int rows[2] = {1, 4}; };
template <> struct KeyRows<REC_HRM_AVG> { static constexpr int NK = 1; static constexpr int rows[2] = {2, -1}; };
template <> struct KeyRows<REC_HRM_MAX> { static constexpr int NK = 1; static constexpr int rows[2] = {2, -1}; };

__host__ __device__ __forceinline__ int key_item(int model, const int4& t)
{
    return (model == REC_BPRMF || model == REC_MC) ? t.y : t.z;
}

template <int MODEL> __device__ __forceinline__ constexpr bool is_key_row(int r)
{
    return r == KeyRows<MODEL>::rows[0] || r == KeyRows<MODEL>::rows[1];
}


template <int MODEL, class T, int LPT, int EPL, int B>
__global__ void __launch_bounds__(256) sgd_runs_kernel(DevModel m, const int4* __restrict__ tuples, int64_t n, T lr, int atomic,
                                                      int64_t batches_per_group, RunPolicy rp)
{
    constexpr int R = Spec<MODEL>::R, NB = Spec<MODEL>::NB, NK = KeyRows<MODEL>::NK;
    const int g = threadIdx.x & (LPT - 1);
    const int64_t gid = (int64_t)blockIdx.x * (blockDim.x / LPT) + threadIdx.x / LPT;
    const int64_t nbatch = (n + B - 1) / B;
    const int64_t first = gid * batches_per_group;
    const int64_t last = first + batches_per_group < nbatch ? first + batches_per_group : nbatch;
    if (first >= last) return;
    // who shares this segment's items: the tuples just outside it and its two ends
    const int64_t t_first = first * B, t_last = (last * B < n ? last * B : n) - 1;
    const int k_first = key_item(MODEL, tuples[t_first]), k_last = key_item(MODEL, tuples[t_last]);
    const int k_before = t_first > 0 ? key_item(MODEL, tuples[t_first - 1]) : -1;
    const int k_after = t_last + 1 < n ? key_item(MODEL, tuples[t_last + 1]) : -1;
    const bool hot_segment = k_first == k_last && (k_before == k_first || k_after == k_first);
    int pend_n = 0;

    T racc[EPL], rg[EPL];   // unused by these models; step() takes them
#pragma unroll
    for (int j = 0; j < EPL; j++) { racc[j] = 0; rg[j] = 0; }

    auto load_idx = [&](int64_t batch, int4 (&tp)[B]) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int64_t i = batch * B + b;
            tp[b] = (batch < last && i < n) ? tuples[i] : make_int4(-1, 0, 0, 0);
        }
    };
    // bias s of tuple b is held by lane 2b+s of the group
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
    load_idx(first, tp);
    load_idx(first + 1, tn);
    load_rows(tp, v, bs);
    // deltas of the current run that memory has not seen yet
    T pend[NK][EPL], pend_bias = 0;
#pragma unroll
    for (int i = 0; i < NK; i++)
#pragma unroll
        for (int j = 0; j < EPL; j++) pend[i][j] = 0;

    for (int64_t batch = first; batch < last; batch++) {
        int4 tnn[B];
        load_idx(batch + 2, tnn);
        T vn[B][R][EPL], bn;
        load_rows(tn, vn, bn);
        T mine = 0;  // deri of the tuple whose bias this lane holds
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (tp[b].x < 0) continue;
            const int key = key_item(MODEL, tp[b]);
            int next;
            if (b + 1 < B) next = tp[b + 1 < B ? b + 1 : b].x >= 0 ? key_item(MODEL, tp[b + 1 < B ? b + 1 : b]) : -1;
            else next = tn[0].x >= 0 ? key_item(MODEL, tn[0]) : -1;
            T old[R][EPL]; bool at[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
#pragma unroll
                for (int j = 0; j < EPL; j++) old[r][j] = v[b][r][j];
                const int it = row_item<MODEL>(tp[b], r);
                at[r] = ((atomic >> r) & 1) && (is_key_row<MODEL>(r) || !(it >= 0 && m.hot[it]));
            }
            // the run's own earlier deltas
#pragma unroll
            for (int i = 0; i < NK; i++)
#pragma unroll
                for (int j = 0; j < EPL; j++) v[b][KeyRows<MODEL>::rows[i]][j] += pend[i][j];
            const T b0 = NB == 2 ? __shfl(bs, 2 * b, LPT) : (T)0, b1 = NB == 2 ? __shfl(bs, 2 * b + 1, LPT) : (T)0;
            const T b0_eff = b0 + pend_bias;
            const T deri = step<MODEL, T, LPT, EPL>(m, lr, g, v[b], b0_eff, b1, rg, racc, at);
            if ((g >> 1) == b) mine = deri;
#pragma unroll
            for (int i = 0; i < NK; i++)
#pragma unroll
                for (int j = 0; j < EPL; j++) pend[i][j] += v[b][KeyRows<MODEL>::rows[i]][j];
            if constexpr (NB == 2) pend_bias += lr * (deri - (T)m.bias_reg * b0_eff);   // BPRMF.cpp:92
            pend_n++;
            const bool shared = key == k_before || key == k_after;
            const int limit = hot_segment ? rp.hot : shared ? rp.shared : 0x7fffffff;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (is_key_row<MODEL>(r)) continue;
                int width; T* row = row_ptr<MODEL, T>(m, tp[b], r, width);
#pragma unroll
                for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < width) scatter(row + k, v[b][r][j], old[r][j], at[r]); }
            }
            if (next != key || pend_n >= limit) {
                // end of the run (or of its private stretch): one scatter for everything it accumulated
                pend_n = 0;
#pragma unroll
                for (int i = 0; i < NK; i++) {
                    const int r = KeyRows<MODEL>::rows[i];
                    int width; T* row = row_ptr<MODEL, T>(m, tp[b], r, width);
#pragma unroll
                    for (int j = 0; j < EPL; j++) {
                        const int k = g + LPT * j;
                        if (k < width) scatter(row + k, pend[i][j], old[r][j], at[r] && (!hot_segment || rp.hot_atomic));
                        pend[i][j] = 0;
                    }
                }
                if constexpr (NB == 2) {
                    if (g == 0) scatter(bias_ptr<MODEL, T>(m, tp[b], 0), pend_bias, b0, ((atomic >> 8) & 1) && (!hot_segment || rp.hot_atomic));
                    pend_bias = 0;
                }
            }
        }
        if constexpr (NB == 2) {
            // the negatives' biases (BPRMF.cpp:93), one scatter instruction per batch
            T* p = my_bias_ptr(tp);
            if (p && (g & 1)) {
                const T* base = (const T*)m.tab[BPRMF_BETA];
                scatter(p, lr * (-mine - (T)m.bias_reg * bs), bs, ((atomic >> 8) & 1) && !m.hot[p - base]);
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
}

template <int MODEL, class T>
static int dispatch_shape(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, int64_t segment, RunPolicy rp, hipStream_t st)
{
    const int Kmax = m.K > m.KK ? m.K : m.KK;
    // every lane group walks one contiguous segment; about 8 blocks per CU when there is enough work,
    // and segments of at least 8 batches
    auto go = [&](auto kern, int lpt, int b) {
        const int64_t groups_per_block = 256 / lpt;
        const int64_t nbatch = (n + b - 1) / b;
        int64_t bpg = (nbatch + 2048 * groups_per_block - 1) / (2048 * groups_per_block);
        if (bpg < 8) bpg = 8;
        if (segment > 0) bpg = segment;
        const int64_t groups = (nbatch + bpg - 1) / bpg;
        const int64_t blocks = (groups + groups_per_block - 1) / groups_per_block;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, m, tuples, n, (T)lr, atomic, bpg, rp);
    };
    constexpr int R = Spec<MODEL>::R;
    constexpr int B1 = R <= 4 ? 4 : 2;
    if (Kmax <= 16) go(sgd_runs_kernel<MODEL, T, 16, 1, B1>, 16, B1);
    else if (Kmax <= 32) go(sgd_runs_kernel<MODEL, T, 32, 1, B1>, 32, B1);
    else if (Kmax <= 64) go(sgd_runs_kernel<MODEL, T, 64, 1, B1>, 64, B1);
    else if (Kmax <= 128) go(sgd_runs_kernel<MODEL, T, 64, 2, 2>, 64, 2);
    else if (Kmax <= 256) go(sgd_runs_kernel<MODEL, T, 64, 4, 1>, 64, 1);
    else { rec_set_error("K > 256 is not supported"); return -1; }
    HIP_TRY(hipGetLastError());
    return 0;
}

bool sgd_runs_supported(int model)
{
    return model == REC_BPRMF || model == REC_MC || model == REC_FPMC || model == REC_PRME || model == REC_HRM_AVG || model == REC_HRM_MAX;
}

template <class T>
static int dispatch_model(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, int64_t segment, RunPolicy rp, hipStream_t st)
{
    switch (m.model) {
    case REC_BPRMF: return dispatch_shape<REC_BPRMF, T>(m, tuples, n, lr, atomic, segment, rp, st);
    case REC_MC: return dispatch_shape<REC_MC, T>(m, tuples, n, lr, atomic, segment, rp, st);
    case REC_FPMC: return dispatch_shape<REC_FPMC, T>(m, tuples, n, lr, atomic, segment, rp, st);
    case REC_PRME: return dispatch_shape<REC_PRME, T>(m, tuples, n, lr, atomic, segment, rp, st);
    case REC_HRM_AVG: return dispatch_shape<REC_HRM_AVG, T>(m, tuples, n, lr, atomic, segment, rp, st);
    case REC_HRM_MAX: return dispatch_shape<REC_HRM_MAX, T>(m, tuples, n, lr, atomic, segment, rp, st);
    default: rec_set_error("the run-combining update is not available for this model"); return -1;
    }
}

int launch_sgd_runs(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, int atomic, int64_t segment, RunPolicy rp, hipStream_t st)
{
    if (n <= 0) return 0;
    return precision == REC_F64 ? dispatch_model<double>(m, tuples, n, lr, atomic, segment, rp, st) : dispatch_model<float>(m, tuples, n, lr, atomic, segment, rp, st);
}
