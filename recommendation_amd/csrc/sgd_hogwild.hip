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
//
// Owner mode (template parameter OWNER).  Float atomics execute at the memory side at a fixed chip-wide byte rate,
// far below the rate of loads and plain stores, so a scatter that adds two item rows per tuple is bound by them.  The
// negative item of a tuple is uniform over the catalog, so the tuples of a launch are first partitioned by the
// negative's slice of the catalog (partition.hip: 256 slices of consecutive items, equally loaded whatever the
// popularity of the items), and workgroup b applies the tuples of slice b.  It keeps an f32 accumulator row in LDS for
// the first `rows_res` items of its slice (as many as 160 KB hold): the negative's delta is added there
// (ds_add_f32) instead of in memory, and a row's accumulated delta is moved to memory with ONE atomic row per
// `flush_at` additions (and at the end of the launch).  Only this workgroup touches the item in the negative's role,
// and it reads the row as memory + pending accumulator, so its own steps on an item see each other exactly; other
// readers (the item as somebody's positive) see its negative updates at most flush_at - 1 steps late.  Items of the
// slice beyond rows_res, and every other row, are scattered as in the plain mode.
#include "common.h"
#include "sgd_step.h"
#include <algorithm>

constexpr int OWNER_THREADS = 768;   // 12 waves: what the registers of this kernel allow a CU to hold

template <int MODEL, class T, int LPT, int EPL, int B, bool OWNER>
__global__ void __launch_bounds__(OWNER ? OWNER_THREADS : 256) sgd_hogwild_kernel(DevModel m, const int4* __restrict__ tuples, int64_t n, T lr,
                                                                                 int atomic, OwnerArgs ow)
{
    constexpr int R = Spec<MODEL>::R, NB = Spec<MODEL>::NB;
    constexpr bool TR = MODEL == REC_TRANSREC || MODEL == REC_TRANSREC_L1;
    const int g = threadIdx.x & (LPT - 1);
    // plain mode: the groups of all blocks stride over the batches of the launch; owner mode: the groups of block b
    // stride over the batches of slice b
    extern __shared__ __align__(16) unsigned char owner_lds[];
    T* acc = (T*)owner_lds;                                                        // [rows_res][acc_width]
    unsigned* pend = (unsigned*)(owner_lds + (size_t)ow.rows_res * ow.acc_width * sizeof(T));   // [rows_res] additions since the last flush
    const int slice_base = OWNER ? (int)blockIdx.x * ow.slice_rows : 0;
    if constexpr (OWNER) {
        const int64_t q0 = ow.qoff[blockIdx.x];
        tuples += q0; n = ow.qoff[blockIdx.x + 1] - q0;
        for (int i = threadIdx.x; i < ow.rows_res * ow.acc_width; i += blockDim.x) acc[i] = 0;
        for (int i = threadIdx.x; i < ow.rows_res; i += blockDim.x) pend[i] = 0;
        __syncthreads();
    }
    const int64_t ngroups = OWNER ? (int64_t)(blockDim.x / LPT) : (int64_t)gridDim.x * (blockDim.x / LPT);
    const int64_t gid = OWNER ? (int64_t)(threadIdx.x / LPT) : (int64_t)blockIdx.x * (blockDim.x / LPT) + threadIdx.x / LPT;
    const int64_t nbatch = (n + B - 1) / B;
    // accumulator slot of a tuple's negative item (-1: not resident here)
    auto slot_of = [&](const int4& t) -> int {
        if constexpr (!OWNER) return -1;
        const int s = neg_item<MODEL>(t) - slice_base;
        return (t.x >= 0 && s >= 0 && s < ow.rows_res) ? s : -1;
    };

    T racc[EPL], rg[EPL];
#pragma unroll
    for (int j = 0; j < EPL; j++) { racc[j] = 0; rg[j] = 0; }
    if constexpr (TR) {
        // the global translation r is read once: it does not change during a launch
#pragma unroll
        for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < m.K) rg[j] = ((const T*)m.tab[TR_r])[k]; }
    }

    // folding the rows two neighbouring tuples share: not for TransRec, whose step is bound by its arithmetic (the
    // projections), not by its atomics - there the extra compares cost more than the saved scatter
    constexpr bool FOLD = B % 2 == 0 && !TR;
    const int fold_reach = (atomic >> 16) & 3;   // 0: no folding, 1: aligned pairs of tuples, 2: every tuple of a batch
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
    // hf: lane b*R+r of the group holds the hot flag of row r of tuple b when that row would otherwise be scattered
    // with atomics; bias_hot: the same for the bias this lane holds.  The flags travel with the rows: asked for only when the scatter needs them, every one of them would cost the
    // wavefront a memory round trip of its own, behind the prefetched rows.
    auto load_rows = [&](const int4 (&tp)[B], T (&v)[B][R][EPL], T& bs, int& hf, uint8_t& bias_hot) {
        static_assert(B * R <= LPT, "one lane of the group per row flag");
        int flag_item = -1;   // lane b*R+r of the group fetches the flag of row r of tuple b
#pragma unroll
        for (int b = 0; b < B; b++) {
            const bool ok = tp[b].x >= 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                int width; const T* row = row_ptr<MODEL, T>(m, tp[b], r, width);
#pragma unroll
                for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; v[b][r][j] = (ok && k < width) ? row[k] : (T)0; }
                if (g == b * R + r && ok && ((atomic >> r) & 1)) flag_item = row_item<MODEL>(tp[b], r);
            }
        }
        hf = flag_item >= 0 ? (int)m.hot[flag_item] : 0;
        bs = 0; bias_hot = 0;
        if constexpr (NB == 2) {
            const T* p = my_bias_ptr(tp);
            if (p) {
                bs = *p;
                if ((atomic >> 8) & 1) bias_hot = m.hot[p - (const T*)m.tab[MODEL == REC_BPRMF ? BPRMF_BETA : TR_BETA]];
            }
        }
    };

    int4 tp[B], tn[B];
    T v[B][R][EPL], bs;
    int hf; uint8_t bias_hot;
    // Batches are claimed, not assigned by stride: lane groups advance at different speeds (their rows come from
    // different places, their CU shares its memory path with others), and with an equal share each the launch ends
    // with its slowest group.  Owner mode: one batch per claim from a counter in LDS; plain mode: CLAIM batches per
    // claim from a counter in memory (one returning atomic per 256 tuples).
    // batches per claim: 1 in owner mode (a counter in LDS), 64 per trip to the device counter in plain mode.  A launch too
    // short to give every group a few claims of that size (ow.claim == 0) is split by stride instead: one claim per
    // batch would put 100k returning atomics on one address (0.8 ms for 200k tuples).
    const int CLAIM = OWNER ? 1 : ow.claim;
    const int64_t stride_groups = (int64_t)gridDim.x * (blockDim.x / LPT);
    int64_t stride_next = (int64_t)blockIdx.x * (blockDim.x / LPT) + threadIdx.x / LPT;
    __shared__ int lds_next;
    if constexpr (OWNER) {
        if (threadIdx.x == 0) lds_next = 0;
        __syncthreads();
    }
    int64_t claim_cur = 0; int claim_left = 0;
    auto next_batch = [&]() -> int64_t {
        if (!OWNER && CLAIM == 0) { const int64_t b = stride_next; stride_next += stride_groups; return b; }
        if (claim_left == 0) {
            int c = 0;
            if (g == 0) c = OWNER ? atomicAdd(&lds_next, CLAIM) : atomicAdd(m.work_counter, CLAIM);
            claim_cur = __shfl(c, 0, LPT); claim_left = CLAIM;
        }
        claim_left--;
        return claim_cur++;
    };
    // the batch being applied, and the two behind it in the pipeline (claims only grow, so once one of them lies past
    // the end all later ones do)
    int64_t b_p = next_batch(), b_n = next_batch();
    load_idx(b_p, tp);
    load_idx(b_n, tn);
    load_rows(tp, v, bs, hf, bias_hot);
    while (b_p < nbatch) {
        // indices two batches ahead, rows one batch ahead
        int4 tnn[B];
        const int64_t b_nn = next_batch();
        load_idx(b_nn, tnn);
        T vn[B][R][EPL], bn;
        int hfn; uint8_t bias_hot_n;
        load_rows(tn, vn, bn, hfn, bias_hot_n);
        if constexpr (OWNER) {
            // an item's value as this workgroup knows it: memory + what it has added since the last flush.  Read here,
            // when the batch is about to be used (its rows have been in flight for a whole iteration), not where the rows
            // are requested: there the addition would make the wavefront wait for the rows it has just asked for.
#pragma unroll
            for (int b = 0; b < B; b++) {
                const int sl = slot_of(tp[b]);
                if (sl < 0) continue;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (!owned_row<MODEL>(r)) continue;
                    const int width = (MODEL == REC_FPMC || MODEL == REC_PRME) && r == 5 ? m.KK : m.K;
                    const T* a = acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K);
#pragma unroll
                    for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < width) v[b][r][j] += a[k]; }
                }
            }
        }
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
                // rows of hot items fall back to plain stores in either mode: their atomics would serialize on one address
                at[b][r] = ((atomic >> r) & 1) && !__shfl(hf, b * R + r, LPT);
            }
            const T deri = step<MODEL, T, LPT, EPL>(m, lr, g, v[b], b0, b1, rg, racc, at[b]);
            if ((g >> 1) == b) mine = deri;
        }
        // Tuples of a batch that name the same item in the same role (the sampler's shared negatives) scatter that
        // row once, with the sum of their deltas: a fraction of the atomics for it, the same sum.
        bool folded[B][R];
#pragma unroll
        for (int b = 0; b < B; b++)
#pragma unroll
            for (int r = 0; r < R; r++) folded[b][r] = false;
        if (FOLD && fold_reach == 1) {
            // aligned pairs only (what the sampler's shared negatives produce by default): one compare per row
#pragma unroll
            for (int b = 0; b + 1 < B; b += 2)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int it0 = row_item<MODEL>(tp[b], r), it1 = row_item<MODEL>(tp[b + 1], r);
                    const bool same = tp[b].x >= 0 && tp[b + 1].x >= 0 && it0 >= 0 && it0 == it1 && at[b][r] && at[b + 1][r];
                    folded[b + 1][r] = same;
#pragma unroll
                    for (int j = 0; j < EPL; j++) v[b][r][j] += same ? v[b + 1][r][j] : (T)0;
                }
        }
        if (FOLD && fold_reach == 2) {
            // every tuple of the batch folds into the first one that names the same item in the same role
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int b = B - 1; b > 0; b--) {
                    const int it1 = row_item<MODEL>(tp[b], r);
                    const bool can = tp[b].x >= 0 && it1 >= 0 && at[b][r];
                    int target = -1;
#pragma unroll
                    for (int c = b - 1; c >= 0; c--)
                        if (can && tp[c].x >= 0 && at[c][r] && row_item<MODEL>(tp[c], r) == it1) target = c;
                    folded[b][r] = target >= 0;
#pragma unroll
                    for (int c = 0; c < b; c++)
#pragma unroll
                        for (int j = 0; j < EPL; j++) v[c][r][j] += target == c ? v[b][r][j] : (T)0;
                }
        }
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (tp[b].x < 0) continue;
            const int sl = slot_of(tp[b]);
            bool accumulated = false;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (folded[b][r]) continue;
                int width; T* row = row_ptr<MODEL, T>(m, tp[b], r, width);
                if (OWNER && owned_row<MODEL>(r) && sl >= 0 && at[b][r]) {
                    T* a = acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K);
#pragma unroll
                    for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < width) unsafeAtomicAdd(a + k, v[b][r][j]); }
                    accumulated = true;
                    continue;
                }
#pragma unroll
                for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < width) scatter(row + k, v[b][r][j], old[b][r][j], at[b][r]); }
            }
            if constexpr (OWNER) {
                // every flush_at-th addition to an item moves its accumulated delta to memory: one atomic row for all of them
                unsigned before = 0;
                if (accumulated && g == 0) before = atomicInc(pend + sl, (unsigned)(ow.flush_at - 1));
                before = __shfl(before, 0, LPT);
                if (accumulated && before == (unsigned)(ow.flush_at - 1)) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (!owned_row<MODEL>(r)) continue;
                        int width; T* row = row_ptr<MODEL, T>(m, tp[b], r, width);
                        T* a = acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K);
#pragma unroll
                        for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < width) unsafeAtomicAdd(row + k, lds_take(a + k)); }
                    }
                }
            }
        }
        if constexpr (NB == 2) {
            // BPRMF.cpp:92-93: +deri on the positive, -deri on the negative; TransRec.cpp:225-226: the opposite signs
            T* p = my_bias_ptr(tp);
            const T sgn = ((g & 1) == 0) == (MODEL == REC_BPRMF) ? (T)1 : (T)-1;
            T delta = p ? lr * (sgn * mine - (T)m.bias_reg * bs) : (T)0;
            // a bias named by several tuples of the batch in the same role is written once, by the first of them, with
            // all their deltas: as plain stores they would otherwise lose all but one
            const T* base = (const T*)m.tab[MODEL == REC_BPRMF ? BPRMF_BETA : TR_BETA];
            const int mine_idx = p ? (int)(p - base) : -1;
            const T own = delta;
            bool folded_bias = false;
            if constexpr (FOLD) {
#pragma unroll
                for (int d = 1; d < B; d++) {   // the same role of the tuple d places further on / back: 2d lanes away
                    if (fold_reach == 0 || (fold_reach == 1 && d > 1)) break;
                    const int next_idx = __shfl_down(mine_idx, 2 * d, LPT), prev_idx = __shfl_up(mine_idx, 2 * d, LPT);
                    const T next_delta = __shfl_down(own, 2 * d, LPT);
                    const bool pair_ok = fold_reach == 2 || (((g >> 1) & 1) == 0);   // pairs: the even tuple takes the odd one's delta
                    if (p && pair_ok && g + 2 * d < 2 * B && next_idx == mine_idx) delta += next_delta;
                    if (p && (fold_reach == 2 || ((g >> 1) & 1) == 1) && g - 2 * d >= 0 && prev_idx == mine_idx) folded_bias = true;
                }
            }
            // owner mode: the tuples of a slice run side by side in one workgroup and name the same few negatives - as plain
            // stores ("value loaded a batch ago + step") their bias steps would overwrite each other, so there the biases
            // are always float atomics (the wide kernel sums them in LDS next to their rows)
            if (p && !folded_bias) scatter(p, delta, bs, OWNER || (((atomic >> 8) & 1) && !bias_hot));
        }
        bs = bn; bias_hot = bias_hot_n; hf = hfn;
        b_p = b_n; b_n = b_nn;
#pragma unroll
        for (int b = 0; b < B; b++) {
            tp[b] = tn[b]; tn[b] = tnn[b];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int j = 0; j < EPL; j++) v[b][r][j] = vn[b][r][j];
        }
    }

    if constexpr (OWNER) {
        // what is still pending goes to memory: one group per accumulator row
        __syncthreads();
        for (int sl = (int)gid; sl < ow.rows_res; sl += (int)ngroups) {
            if (pend[sl] == 0) continue;
            const int4 t = (MODEL == REC_BPRMF || MODEL == REC_MC) ? make_int4(0, 0, slice_base + sl, 0) : make_int4(0, 0, 0, slice_base + sl);
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (!owned_row<MODEL>(r)) continue;
                int width; T* row = row_ptr<MODEL, T>(m, t, r, width);
                const T* a = acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K);
#pragma unroll
                for (int j = 0; j < EPL; j++) { const int k = g + LPT * j; if (k < width) unsafeAtomicAdd(row + k, a[k]); }
            }
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

// slices of the catalog in owner mode: one workgroup each, all resident at once on 256 CUs
constexpr int OWNER_SLICES = 256;
constexpr size_t OWNER_LDS_BYTES = 160 * 1024 - 2048;   // per workgroup, less what the kernel declares statically

bool sgd_owner_supported(int model)
{
    return model == REC_BPRMF || model == REC_MC || model == REC_FPMC || model == REC_PRME || model == REC_HRM_AVG ||
           model == REC_HRM_MAX || model == REC_TRANSREC || model == REC_TRANSREC_L1;
}
int sgd_owner_slices() { return OWNER_SLICES; }
int sgd_owner_slice_rows(const DevModel& m) { return (m.I + OWNER_SLICES - 1) / OWNER_SLICES; }
int sgd_owner_neg_field(int model) { return (model == REC_BPRMF || model == REC_MC) ? 2 : 3; }

template <int MODEL, class T>
static int dispatch_shape(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, const int64_t* qoff, int flush_at, hipStream_t st)
{
    const int Kmax = m.K > m.KK ? m.K : m.KK;
    OwnerArgs ow = {nullptr, 0, 0, 0, 1, 64};
    size_t lds = 0;
    if (qoff) {
        ow.qoff = qoff; ow.slice_rows = sgd_owner_slice_rows(m); ow.flush_at = flush_at < 1 ? 1 : flush_at;
        ow.acc_width = (MODEL == REC_FPMC || MODEL == REC_PRME) ? m.K + m.KK : m.K;
        const size_t per_row = (size_t)ow.acc_width * sizeof(T) + sizeof(unsigned);
        ow.rows_res = (int)std::min<size_t>((size_t)ow.slice_rows, OWNER_LDS_BYTES / per_row);
        lds = (size_t)ow.rows_res * per_row;
    }
    // plain mode: 2048 blocks of 256 threads keep 8 blocks per CU resident; groups stride over the batches
    auto go = [&](auto kern, auto kern_owner, int lpt, int b) -> int {
        if (qoff) {
            HIP_TRY(hipFuncSetAttribute((const void*)kern_owner, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern_owner, dim3(OWNER_SLICES), dim3(OWNER_THREADS), lds, st, m, tuples, n, (T)lr, atomic, ow);
            return 0;
        }
        const int64_t groups_per_block = 256 / lpt;
        // a short launch gets as many groups as give each about four batches (fewer tuples in flight at the same time, and
        // so fewer steps computed from the same stale rows, than one batch per group), a long one the 2048 blocks the chip holds
        const int64_t nbatch_h = (n + b - 1) / b;
        int64_t blocks = (nbatch_h + 4 * groups_per_block - 1) / (4 * groups_per_block);
        if (blocks > 2048) blocks = 2048;
        // claims of 64 batches where every group gets at least four of them, a split by stride below that
        ow.claim = nbatch_h >= 4 * 64 * blocks * groups_per_block ? 64 : 0;
        HIP_TRY(hipMemsetAsync(m.work_counter, 0, sizeof(int), st));   // the batch counter the groups claim from
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, m, tuples, n, (T)lr, atomic, ow);
        return 0;
    };
    constexpr int R = Spec<MODEL>::R;
    constexpr int B1 = R <= 4 ? 4 : 2;  // tuples in flight per group, by register budget
    int rc = 0;
#define SHAPE(LPT, EPL, BB) rc = go(sgd_hogwild_kernel<MODEL, T, LPT, EPL, BB, false>, sgd_hogwild_kernel<MODEL, T, LPT, EPL, BB, true>, LPT, BB)
    if (Kmax <= 16) SHAPE(16, 1, B1);
    else if (Kmax <= 32) SHAPE(32, 1, B1);
    // 32 lanes per tuple from K = 33 on: a wave-instruction then serves two tuples with two 128-byte row segments
    // (the shape float atomics still run at full rate on) and the per-tuple arithmetic costs half the issue slots
    else if (Kmax <= 64) SHAPE(32, 2, B1);
    else if (Kmax <= 128) SHAPE(32, 4, 2);
    else if (Kmax <= 256) SHAPE(64, 4, 1);
    else { rec_set_error("K > 256 is not supported"); return -1; }
#undef SHAPE
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T>
static int dispatch_model(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, const int64_t* qoff, int flush_at, hipStream_t st)
{
    switch (m.model) {
    case REC_BPRMF: return dispatch_shape<REC_BPRMF, T>(m, tuples, n, lr, atomic, qoff, flush_at, st);
    case REC_MC: return dispatch_shape<REC_MC, T>(m, tuples, n, lr, atomic, qoff, flush_at, st);
    case REC_FPMC: return dispatch_shape<REC_FPMC, T>(m, tuples, n, lr, atomic, qoff, flush_at, st);
    case REC_PRME: return dispatch_shape<REC_PRME, T>(m, tuples, n, lr, atomic, qoff, flush_at, st);
    case REC_HRM_AVG: return dispatch_shape<REC_HRM_AVG, T>(m, tuples, n, lr, atomic, qoff, flush_at, st);
    case REC_HRM_MAX: return dispatch_shape<REC_HRM_MAX, T>(m, tuples, n, lr, atomic, qoff, flush_at, st);
    case REC_TRANSREC: return dispatch_shape<REC_TRANSREC, T>(m, tuples, n, lr, atomic, qoff, flush_at, st);
    case REC_TRANSREC_L1: return dispatch_shape<REC_TRANSREC_L1, T>(m, tuples, n, lr, atomic, qoff, flush_at, st);
    default:
        rec_set_error("Hogwild update kernel not available for this model");
        return -1;
    }
}

int launch_sgd_hogwild(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, int atomic,
                       hipStream_t st)
{
    if (n <= 0) return 0;
    return precision == REC_F64 ? dispatch_model<double>(m, tuples, n, lr, atomic, nullptr, 1, st)
                                : dispatch_model<float>(m, tuples, n, lr, atomic, nullptr, 1, st);
}

int launch_sgd_owner(const DevModel& m, int precision, const int4* partitioned, const int64_t* qoff, int64_t n, double lr, int atomic,
                     int flush_at, hipStream_t st)
{
    if (n <= 0) return 0;
    if (!qoff) { rec_set_error("owner mode needs the slice offsets"); return -1; }
    return precision == REC_F64 ? dispatch_model<double>(m, partitioned, n, lr, atomic, qoff, flush_at, st)
                                : dispatch_model<float>(m, partitioned, n, lr, atomic, qoff, flush_at, st);
}

int launch_apply_global(const DevModel& m, int precision, double lr, int64_t n_tuples, double cap, hipStream_t st, bool project)
{
    if (!(m.model == REC_TRANSREC || m.model == REC_TRANSREC_L1)) return 0;
    if (precision == REC_F64) hipLaunchKernelGGL(apply_global_kernel<double>, dim3(1), dim3(256), 0, st, m, lr, n_tuples, cap);
    else hipLaunchKernelGGL(apply_global_kernel<float>, dim3(1), dim3(256), 0, st, m, (float)lr, n_tuples, (float)cap);
    HIP_TRY(hipGetLastError());
    return project ? launch_project_rows(m, precision, st) : 0;
}

int launch_project_rows(const DevModel& m, int precision, hipStream_t st)
{
    if (!(m.model == REC_TRANSREC || m.model == REC_TRANSREC_L1)) return 0;
    int64_t blocks = ((int64_t)m.I + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (precision == REC_F64) hipLaunchKernelGGL(project_rows_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (double*)m.tab[TR_H], (int64_t)m.I, m.K);
    else hipLaunchKernelGGL(project_rows_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (float*)m.tab[TR_H], (int64_t)m.I, m.K);
    HIP_TRY(hipGetLastError());
    return 0;
}
