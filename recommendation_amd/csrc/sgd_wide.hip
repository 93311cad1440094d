// Hogwild BPR-SGD step, wide form (f32 tables, K a multiple of 4): a lane holds FOUR consecutive components of every
// row its tuple names, so a row of K = 64 is 16 lanes, a wavefront works on four tuples per instruction, and every
// row is fetched (and a plainly stored row written) by one 16-byte access per lane.
//
// Why a second form of sgd_hogwild.hip's kernel: that one (one dword per lane and access, 32 lanes per tuple) spends
// its time issuing instructions, not waiting for memory - counters on the 1M x 200k corpus: the SIMDs issue during
// 80 % of the kernel, 44 % of the wave cycles are issue stalls, 560 vector + 290 scalar instructions per 8 tuples.  Most
// of them are per-instruction overhead (addresses, masks, reductions, flags) that does not depend on how many tuples an
// instruction serves; twice the tuples per instruction and a quarter of the memory instructions per row halve it.
//
// Float atomics want the opposite shape - consecutive lanes on consecutive dwords (a wave-instruction that adds 64
// contiguous floats of one row, or 2 x 32 in two rows, runs at the full atomic rate; 16 lanes x 4 strided dwords in 4
// rows at a fraction of it).  Rows that are scattered with atomics therefore take a detour through a 1 KB staging area
// of the wavefront in LDS: every lane writes its four deltas with one ds_write_b128 (the image is row after row), and
// instruction j of the four atomic instructions adds floats 64j .. 64j+63 of that image: whole rows again.
//
// Owner mode keeps its LDS accumulators in 32-bit fixed point (quantum 2^-25): ds_add_f32 retires about one lane every
// three cycles per CU on this chip (3.2 G rows of 64 floats per second chip-wide, measured: scripts/ubench/lds_atomic.hip -
// 31 ms for the negatives of one epoch of the 100M-interaction corpus, more than the rest of the kernel), ds_add_u32 runs
// at the LDS rate (76-120 G rows/s).  A delta is rounded to the nearest quantum (at most 1.5e-8 off, below the f32
// spacing of a weight of magnitude >= 0.25) before it is added; the sums stay exact, so nothing is lost or doubled, and
// |sum| must stay below 64 between two flushes.
//
// Pipeline, batch claiming and owner mode are those of sgd_hogwild.hip (which documents them); steps that share rows
// by construction ("share_negative", "share_positive") are not folded here and stay with that kernel.
#include "common.h"
#include "sgd_step.h"
#include "sampler_dev.h"
#include <algorithm>

// SAMPLE: the kernel draws its own tuples - step first_step + i of (seed, epoch), the tuple sampler_kernel would have written
// to tuples[i] - instead of reading them: no sampler launch in front of it and no 16 bytes written and read back per tuple
struct SampleArgs { uint64_t seed, epoch; int64_t first_step; };

constexpr int WIDE_EPL = 4;
constexpr int WIDE_OWNER_THREADS = 1024;   // one workgroup per CU in owner mode
constexpr float FIX_SCALE = 33554432.f, FIX_INV = 1.f / 33554432.f;   // 2^25

__device__ __forceinline__ void fix_add(int* p, float x) { atomicAdd(p, __float2int_rn(x * FIX_SCALE)); }
__device__ __forceinline__ void fix_add4(int* p, const float (&x)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) fix_add(p + j, x[j]);
}
__device__ __forceinline__ float4 fix_read4(const int* p)
{
    const int4 a = *reinterpret_cast<const int4*>(p);
    return make_float4((float)a.x * FIX_INV, (float)a.y * FIX_INV, (float)a.z * FIX_INV, (float)a.w * FIX_INV);
}
// reads an accumulator element and leaves zero behind, atomically per element
__device__ __forceinline__ float fix_take(int* p) { return (float)atomicExch(p, 0) * FIX_INV; }
__device__ __forceinline__ float4 fix_take4(int* p) { return make_float4(fix_take(p), fix_take(p + 1), fix_take(p + 2), fix_take(p + 3)); }

// FULL: K (and KK) == 4 * LPT, every lane of a group holds four live components (no width tests)
template <int MODEL, int LPT, int B, bool OWNER, bool FULL, bool SAMPLE = false>
__global__ void __launch_bounds__(OWNER ? WIDE_OWNER_THREADS : 256) sgd_wide_kernel(DevModel m, const int4* __restrict__ tuples, int64_t n,
                                                                                   float lr, int atomic, OwnerArgs ow, SampleArgs sa)
{
    static_assert(!(SAMPLE && OWNER), "the owner mode applies tuples that a partition pass has ordered");
    using T = float;
    constexpr int R = Spec<MODEL>::R, NB = Spec<MODEL>::NB, EPL = WIDE_EPL;
    constexpr bool TR = MODEL == REC_TRANSREC || MODEL == REC_TRANSREC_L1;
    constexpr int ROWF = EPL * LPT;                     // floats of a group's row image in the staging area
    static_assert(B * R <= LPT && 2 * B <= LPT, "one lane of the group per row flag / bias");
    const int g = threadIdx.x & (LPT - 1);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // dynamic LDS: [staging: 64 x 4 floats per wavefront][owner mode: accumulators [rows_res][acc_width], counters, bias sums]
    extern __shared__ __align__(16) unsigned char wide_lds[];
    T* stage = (T*)wide_lds + (size_t)wave * 256;
    int* acc = (int*)wide_lds + (size_t)(blockDim.x >> 6) * 256;
    // owner mode, behind the accumulators: pend [rows_res] additions to an accumulator row since its last flush; models with
    // item biases: acc_bias [rows_res] this workgroup's pending additions to the bias of each resident item in the
    // negative's role (plain stores of the tuples it applies at the same moment would overwrite each other)
    unsigned* pend = (unsigned*)(acc + (size_t)ow.rows_res * ow.acc_width);
    int* acc_bias = (int*)(pend + ow.rows_res);
    __shared__ int lds_next;
    const int slice_base = OWNER ? (int)blockIdx.x * ow.slice_rows : 0;
    if constexpr (OWNER) {
        const int64_t q0 = ow.qoff[blockIdx.x];
        tuples += q0; n = ow.qoff[blockIdx.x + 1] - q0;
        for (int i = threadIdx.x; i < ow.rows_res * ow.acc_width; i += blockDim.x) acc[i] = 0;
        for (int i = threadIdx.x; i < ow.rows_res; i += blockDim.x) pend[i] = 0;
        if constexpr (NB == 2) for (int i = threadIdx.x; i < ow.rows_res; i += blockDim.x) acc_bias[i] = 0;
        if (threadIdx.x == 0) lds_next = 0;
        __syncthreads();
    }
    const int64_t nbatch = (n + B - 1) / B;
    auto slot_of = [&](const int4& t) -> int {
        if constexpr (!OWNER) return -1;
        const int s = neg_item<MODEL>(t) - slice_base;
        return (t.x >= 0 && s >= 0 && s < ow.rows_res) ? s : -1;
    };

    T racc[EPL], rg[EPL];
#pragma unroll
    for (int j = 0; j < EPL; j++) { racc[j] = 0; rg[j] = 0; }
    if constexpr (TR) {
#pragma unroll
        for (int j = 0; j < EPL; j++) { const int k = EPL * g + j; if (k < m.K) rg[j] = ((const T*)m.tab[TR_r])[k]; }
    }

    // claimed batches (see sgd_hogwild.hip)
    // batches per claim: 1 in owner mode (a counter in LDS), 64 per trip to the device counter in plain mode.  A launch too
    // short to give every group a few claims of that size (ow.claim == 0) is split by stride instead: one claim per
    // batch would put 100k returning atomics on one address (0.8 ms for 200k tuples).
    const int CLAIM = OWNER ? 1 : ow.claim;
    const int64_t stride_groups = (int64_t)gridDim.x * (blockDim.x / LPT);
    int64_t stride_next = (int64_t)blockIdx.x * (blockDim.x / LPT) + threadIdx.x / LPT;
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
    auto load_idx = [&](int64_t batch, int4 (&tp)[B]) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int64_t i = batch * B + b;
            tp[b] = (batch < nbatch && i < n) ? tuples[i] : make_int4(-1, 0, 0, 0);
        }
    };
    auto my_bias_ptr = [&](const int4 (&tp)[B]) -> T* {
        T* p = nullptr;
        if constexpr (NB == 2) {
#pragma unroll
            for (int b = 0; b < B; b++)
                if ((g >> 1) == b && g < 2 * B && tp[b].x >= 0) p = bias_ptr<MODEL, T>(m, tp[b], g & 1);
        }
        return p;
    };
    // rows as one 16-byte load per lane; hf: lane b*R+r of the group holds the hot flag of row r of tuple b
    auto load_rows = [&](const int4 (&tp)[B], float4 (&v)[B][R], T& bs, int& hf, uint8_t& bias_hot) {
        int flag_item = -1;
#pragma unroll
        for (int b = 0; b < B; b++) {
            const bool ok = tp[b].x >= 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                int width; const T* row = row_ptr<MODEL, T>(m, tp[b], r, width);
                v[b][r] = (ok && (FULL || EPL * g < width)) ? *reinterpret_cast<const float4*>(row + EPL * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (g == b * R + r && ok && ((atomic >> r) & 1)) flag_item = row_item<MODEL>(tp[b], r);
            }
        }
        hf = flag_item >= 0 ? (int)m.hot[flag_item] : 0;
        bs = 0; bias_hot = 0;
        if constexpr (NB == 2) {
            const T* p = my_bias_ptr(tp);
            if (p) {
                bs = *p;
                if (((atomic >> 8) & 1) || OWNER) bias_hot = m.hot[p - (const T*)m.tab[MODEL == REC_BPRMF ? BPRMF_BETA : TR_BETA]];
            }
        }
    };
    // Adds the wavefront's staged row images to memory: `code` = row index in table `tab` for the groups whose row goes
    // out this way, -1 for the others; `width` = floats of a row of that table.  Instruction j covers floats 64j .. 64j+63 of
    // the image: with 16 lanes per tuple one whole row.
    auto flush_stage = [&](T* tab, int width, int code) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int f = 64 * j + lane;
            const int e = f % ROWF;
            // the group whose image these 64 floats belong to is the same for the whole wavefront (or for each half of it)
            int c;
            if constexpr (ROWF >= 64) c = __builtin_amdgcn_readlane(code, ((64 * j) / ROWF) * LPT);
            else {
                static_assert(ROWF == 32, "8 lanes per tuple: two row images per instruction");
                const int c0 = __builtin_amdgcn_readlane(code, (2 * j) * LPT), c1 = __builtin_amdgcn_readlane(code, (2 * j + 1) * LPT);
                c = lane < 32 ? c0 : c1;
            }
            const T x = stage[f];
            if (c >= 0 && (FULL || e < width)) unsafeAtomicAdd(tab + (int64_t)c * width + e, x);
        }
    };

    // SAMPLE: a group draws the tuples of its next LPT / B batches in one go, one tuple per lane (every lane of the wavefront
    // busy with a tuple of its own, as in sampler_kernel), and hands them out batch by batch.  The two dependent loads of
    // a draw (the owner's record, then the item at the drawn position) are waited for once per LPT / B batches.
    constexpr int QN = LPT / B;
    int4 qt = make_int4(-1, 0, 0, 0); int64_t qb = 0; int qpos = QN;
    auto fetch = [&](int64_t& bidx, int4 (&t)[B]) {
        if constexpr (!SAMPLE) {
            bidx = next_batch();
            load_idx(bidx, t);
        } else {
            if (qpos == QN) {
                int64_t mine = 0;
#pragma unroll
                for (int j = 0; j < QN; j++) { const int64_t bj = next_batch(); if (g / B == j) mine = bj; }
                const int64_t i = mine * B + (g % B);
                qb = mine;
                qt = (mine < nbatch && i < n) ? sample_tuple<MODEL>(m, (uint64_t)(sa.first_step + i), sa.seed, sa.epoch, 1, 1) : make_int4(-1, 0, 0, 0);
                qpos = 0;
            }
            bidx = __shfl(qb, qpos * B, LPT);
#pragma unroll
            for (int b = 0; b < B; b++) {
                const int src = qpos * B + b;
                t[b] = make_int4(__shfl(qt.x, src, LPT), __shfl(qt.y, src, LPT), __shfl(qt.z, src, LPT), __shfl(qt.w, src, LPT));
            }
            qpos++;
        }
    };

    int4 tp[B], tn[B];
    float4 v[B][R]; T bs;
    int hf; uint8_t bias_hot;
    int64_t b_p, b_n;
    fetch(b_p, tp);
    fetch(b_n, tn);
    load_rows(tp, v, bs, hf, bias_hot);
    // The loop is left by the wavefront as a whole (the staged scatter reads across its lane groups): a group whose
    // claims have run past the end keeps going on invalid tuples, which load and write nothing.
    while (__ballot(b_p < nbatch)) {
        int4 tnn[B];
        int64_t b_nn;
        fetch(b_nn, tnn);
        float4 vn[B][R]; T bn;
        int hfn; uint8_t bias_hot_n;
        load_rows(tn, vn, bn, hfn, bias_hot_n);
        T mine = 0;
        int bias_slot = -1;   // owner mode: accumulator slot of the (negative's) bias this lane holds, -1: the bias lives in memory
#pragma unroll
        for (int b = 0; b < B; b++) {
            const bool ok = tp[b].x >= 0;
            const int sl = slot_of(tp[b]);
            if constexpr (OWNER && NB == 2) {
                // the negative's bias goes the way of its row: summed in LDS unless the row gets plain stores (whatever the
                // scatter policy says about biases: the tuples a workgroup applies at the same moment share their negatives,
                // and plain stores of their bias steps would overwrite each other)
                constexpr int NEG_ROW = MODEL == REC_BPRMF ? 2 : 3;
                if (sl >= 0 && g == 2 * b + 1 && ((atomic >> NEG_ROW) & 1) && !bias_hot) { bias_slot = sl; bs += (float)acc_bias[sl] * FIX_INV; }
            }
            T w[R][EPL], old[R][EPL];
#pragma unroll
            for (int r = 0; r < R; r++) { w[r][0] = v[b][r].x; w[r][1] = v[b][r].y; w[r][2] = v[b][r].z; w[r][3] = v[b][r].w; }
            if constexpr (OWNER) {
                // the negative's rows as this workgroup knows them: memory + what it has added since the last flush
                if (sl >= 0) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (!owned_row<MODEL>(r)) continue;
                        const int width = ((MODEL == REC_FPMC || MODEL == REC_PRME) && r == 5) ? m.KK : m.K;
                        if (FULL || EPL * g < width) {
                            const float4 a = fix_read4(acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K) + EPL * g);
                            w[r][0] += a.x; w[r][1] += a.y; w[r][2] += a.z; w[r][3] += a.w;
                        }
                    }
                }
            }
            // Rows of hot items are not scattered with atomics in either mode: from all over the chip they would serialize on
            // one address, and summing them exactly (here, or in an accumulator that is flushed now and then) applies
            // hundreds of steps computed from the same stale row at once - with the regularisation term in every one of them
            // the row overshoots.  They get plain racy stores, which lose most of the concurrent steps instead.
            bool at[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int flag = __shfl(hf, b * R + r, LPT);
                at[r] = ((atomic >> r) & 1) && !flag;
#pragma unroll
                for (int j = 0; j < EPL; j++) old[r][j] = w[r][j];
            }
            const T b0 = NB == 2 ? __shfl(bs, 2 * b, LPT) : (T)0, b1 = NB == 2 ? __shfl(bs, 2 * b + 1, LPT) : (T)0;
            const T deri = step<MODEL, T, LPT, EPL, 1>(m, lr, g, w, b0, b1, rg, racc, at);
            if ((g >> 1) == b) mine = deri;
            // moves what a group has taken out of an LDS accumulator (x, zeros for groups that take nothing) to memory
            auto stage_and_flush = [&](const float4& x, T* tab, int width, int rownum) {
                *reinterpret_cast<float4*>(stage + EPL * lane) = x;
                flush_stage(tab, width, rownum);
            };
            bool accumulated = false;
#pragma unroll
            for (int r = 0; r < R; r++) {
                int width; T* row = row_ptr<MODEL, T>(m, tp[b], r, width);
                const bool act = ok && (FULL || EPL * g < width);
                const bool to_acc = OWNER && owned_row<MODEL>(r) && sl >= 0 && at[r];
                if (to_acc) {
                    if (act) {
                        fix_add4(acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K) + EPL * g, w[r]);
                    }
                    accumulated = true;
                } else if (!at[r]) {
                    if (act) *reinterpret_cast<float4*>(row + EPL * g) = make_float4(old[r][0] + w[r][0], old[r][1] + w[r][1], old[r][2] + w[r][2], old[r][3] + w[r][3]);
                }
                // atomics: through the staging image, if any group of the wavefront needs them for this row
                const bool wants = ok && at[r] && !to_acc;
                if (__ballot(wants))
                    stage_and_flush(make_float4(w[r][0], w[r][1], w[r][2], w[r][3]), row_table<MODEL, T>(m, r), width, wants ? row_number<MODEL>(tp[b], r) : -1);
            }
            if constexpr (OWNER) {
                // every flush_at-th addition to an item moves its accumulated delta to memory: one atomic row for all of them
                bool fire0 = false;   // known to the first lane of the group, which counts the additions
                if (accumulated && g == 0) fire0 = atomicInc(pend + sl, (unsigned)(ow.flush_at - 1)) == (unsigned)(ow.flush_at - 1);
                const unsigned long long fired = __ballot(fire0);
                if (fired) {
                    const bool fire = (fired >> (lane & ~(LPT - 1))) & 1ull;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (!owned_row<MODEL>(r)) continue;
                        const int width = ((MODEL == REC_FPMC || MODEL == REC_PRME) && r == 5) ? m.KK : m.K;
                        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (fire && (FULL || EPL * g < width)) {
                            x = fix_take4(acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K) + EPL * g);
                        }
                        stage_and_flush(x, row_table<MODEL, T>(m, r), width, fire ? row_number<MODEL>(tp[b], r) : -1);
                    }
                    if constexpr (NB == 2) {
                        if (fire && g == 0) unsafeAtomicAdd(bias_ptr<MODEL, T>(m, tp[b], 1), fix_take(acc_bias + sl));
                    }
                }
            }
        }
        if constexpr (NB == 2) {
            // BPRMF.cpp:92-93: +deri on the positive, -deri on the negative; TransRec.cpp:225-226: the opposite signs
            T* p = my_bias_ptr(tp);
            const T sgn = ((g & 1) == 0) == (MODEL == REC_BPRMF) ? (T)1 : (T)-1;
            const T delta = p ? lr * (sgn * mine - (T)m.bias_reg * bs) : (T)0;
            if (p && bias_slot >= 0) fix_add(acc_bias + bias_slot, delta);
            // (owner mode: atomics on the positive's bias as well when its item's rows get them - the negative's steps arrive
            // as atomic sums, and a plain store of "value loaded a batch ago + step" would wipe out those that landed since)
            else if (p) scatter(p, delta, bs, (((atomic >> 8) & 1) || (OWNER && ((atomic >> 1) & 1))) && !bias_hot);
        }
        bs = bn; bias_hot = bias_hot_n; hf = hfn;
        b_p = b_n; b_n = b_nn;
#pragma unroll
        for (int b = 0; b < B; b++) {
            tp[b] = tn[b]; tn[b] = tnn[b];
#pragma unroll
            for (int r = 0; r < R; r++) v[b][r] = vn[b][r];
        }
    }

    if constexpr (OWNER) {
        // what is still pending goes to memory: one group per accumulator row
        __syncthreads();
        const int groups = blockDim.x / LPT, gid = threadIdx.x / LPT;
        for (int s0 = 0; s0 < ow.rows_res; s0 += groups) {
            const int sl = s0 + gid;
            bool dirty = sl < ow.rows_res && pend[sl] != 0;
            if constexpr (NB == 2) dirty = dirty || (sl < ow.rows_res && acc_bias[sl] != 0);   // a bias delta is added after its tuple's count
            if (!__ballot(dirty)) continue;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (!owned_row<MODEL>(r)) continue;
                const int width = ((MODEL == REC_FPMC || MODEL == REC_PRME) && r == 5) ? m.KK : m.K;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dirty && (FULL || EPL * g < width)) x = fix_read4(acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K) + EPL * g);
                *reinterpret_cast<float4*>(stage + EPL * lane) = x;
                flush_stage(row_table<MODEL, T>(m, r), width, dirty ? slice_base + sl : -1);
            }
            if constexpr (NB == 2) {
                if (dirty && g == 0) unsafeAtomicAdd((T*)m.tab[MODEL == REC_BPRMF ? BPRMF_BETA : TR_BETA] + slice_base + sl, (float)acc_bias[sl] * FIX_INV);
            }
        }
    }

    if constexpr (TR) {
        // block-level sum of the r gradients
        __shared__ T red[REC_MAX_K];
        for (int k = threadIdx.x; k < REC_MAX_K; k += blockDim.x) red[k] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPL; j++) { const int k = EPL * g + j; if (k < m.K) unsafeAtomicAdd(&red[k], racc[j]); }
        __syncthreads();
        T* gacc = (T*)m.gacc;
        for (int k = threadIdx.x; k < m.K; k += blockDim.x) unsafeAtomicAdd(gacc + k, red[k]);
    }
}

// ------------------------------------------------------------------------------------------------ launch
static size_t wide_owner_lds_budget() { return 160 * 1024 - 1024 - 256; }   // less the kernel's static LDS (TransRec's reduction row)

// LDS layout of the owner mode for `slices` slices of the catalog: fills ow (all but qoff and the flush periods) and
// returns the dynamic LDS bytes.  rows_res < slice_rows when a slice's accumulators do not fit.
static size_t wide_owner_layout(const DevModel& m, int slices, OwnerArgs& ow)
{
    const bool two = m.model == REC_FPMC || m.model == REC_PRME;
    const bool biased = m.model == REC_BPRMF || m.model == REC_TRANSREC || m.model == REC_TRANSREC_L1;
    const size_t stage_bytes = (size_t)(WIDE_OWNER_THREADS / 64) * 1024;
    ow.slice_rows = (m.I + slices - 1) / slices;
    ow.acc_width = two ? m.K + m.KK : m.K;
    const size_t per_row = (size_t)ow.acc_width * sizeof(float) + sizeof(unsigned) + (biased ? sizeof(float) : 0);
    ow.rows_res = (int)std::min<size_t>((size_t)ow.slice_rows, (wide_owner_lds_budget() - stage_bytes) / per_row);
    return stage_bytes + (size_t)ow.rows_res * per_row;
}

// Slices the owner mode cuts the catalog into: the fewest of 256 / 512 / 1024 (one, two or four workgroups per CU, one
// after the other) with which every item of a slice has its accumulator row in LDS; 1024 if none does.
int sgd_wide_owner_slices(const DevModel& m)
{
    for (int slices = 256; slices < 1024; slices *= 2) {
        OwnerArgs ow = {};
        wide_owner_layout(m, slices, ow);
        if (ow.rows_res >= ow.slice_rows) return slices;
    }
    return 1024;
}

template <int MODEL>
static int wide_dispatch(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, const int64_t* qoff, int slices, int flush_at, hipStream_t st,
                         const SampleArgs* sample = nullptr)
{
    const int Kmax = m.K > m.KK ? m.K : m.KK;
    OwnerArgs ow = {nullptr, 0, 0, 0, 1, 64};
    const SampleArgs sa = sample ? *sample : SampleArgs{0, 0, 0};
    auto go = [&](auto kern, auto kern_owner, auto kern_sample, int lpt, int b) -> int {
        if (qoff) {
            const size_t lds = wide_owner_layout(m, slices, ow);
            ow.qoff = qoff; ow.flush_at = flush_at < 1 ? 1 : flush_at;
            HIP_TRY(hipFuncSetAttribute((const void*)kern_owner, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern_owner, dim3(slices), dim3(WIDE_OWNER_THREADS), lds, st, m, tuples, n, (float)lr, atomic, ow, sa);
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
        if (sample) hipLaunchKernelGGL(kern_sample, dim3((unsigned)blocks), dim3(256), 4 * 1024, st, m, tuples, n, (float)lr, atomic, ow, sa);
        else hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 4 * 1024, st, m, tuples, n, (float)lr, atomic, ow, sa);
        return 0;
    };
    constexpr int R = Spec<MODEL>::R;
    int rc = 0;
    const bool full_k = m.K == m.KK || !(MODEL == REC_FPMC || MODEL == REC_PRME);   // every row of the tuple has K components
#define WIDE_SHAPE(LPT, BB)                                                                                                               \
    rc = (full_k && m.K == 4 * LPT) ? go(sgd_wide_kernel<MODEL, LPT, BB, false, true>, sgd_wide_kernel<MODEL, LPT, BB, true, true>,               \
                                         sgd_wide_kernel<MODEL, LPT, BB, false, true, true>, LPT, BB)                                       \
                                    : go(sgd_wide_kernel<MODEL, LPT, BB, false, false>, sgd_wide_kernel<MODEL, LPT, BB, true, false>,             \
                                         sgd_wide_kernel<MODEL, LPT, BB, false, false, true>, LPT, BB)
    if (Kmax <= 32) WIDE_SHAPE(8, (R <= 4 ? 2 : 1));
    else if (Kmax <= 64) WIDE_SHAPE(16, 2);
    else if (Kmax <= 128) WIDE_SHAPE(32, 2);
    else WIDE_SHAPE(64, 1);
#undef WIDE_SHAPE
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

bool sgd_wide_supported(const DevModel& m, int precision)
{
    if (precision != REC_F32) return false;
    if (m.K % 4 != 0 || m.KK % 4 != 0 || m.K < 8) return false;
    switch (m.model) {
    case REC_BPRMF: case REC_MC: case REC_FPMC: case REC_PRME: case REC_HRM_AVG: case REC_HRM_MAX: case REC_TRANSREC: case REC_TRANSREC_L1: return true;
    default: return false;
    }
}

// qoff == nullptr: plain mode on the tuples as they come; else owner mode on tuples partitioned by the negative's slice
// (`slices` of them, qoff[slices + 1]).  sample != nullptr (plain mode only): no tuples are read, the kernel draws step
// first_step + i of (seed, epoch) for its i-th tuple itself.
static int wide_launch(const DevModel& m, const int4* tuples, const int64_t* qoff, int slices, int64_t n, double lr, int atomic, int flush_at, hipStream_t st,
                       const SampleArgs* sample)
{
    if (n <= 0) return 0;
    switch (m.model) {
    case REC_BPRMF: return wide_dispatch<REC_BPRMF>(m, tuples, n, lr, atomic, qoff, slices, flush_at, st, sample);
    case REC_MC: return wide_dispatch<REC_MC>(m, tuples, n, lr, atomic, qoff, slices, flush_at, st, sample);
    case REC_FPMC: return wide_dispatch<REC_FPMC>(m, tuples, n, lr, atomic, qoff, slices, flush_at, st, sample);
    case REC_PRME: return wide_dispatch<REC_PRME>(m, tuples, n, lr, atomic, qoff, slices, flush_at, st, sample);
    case REC_HRM_AVG: return wide_dispatch<REC_HRM_AVG>(m, tuples, n, lr, atomic, qoff, slices, flush_at, st, sample);
    case REC_HRM_MAX: return wide_dispatch<REC_HRM_MAX>(m, tuples, n, lr, atomic, qoff, slices, flush_at, st, sample);
    case REC_TRANSREC: return wide_dispatch<REC_TRANSREC>(m, tuples, n, lr, atomic, qoff, slices, flush_at, st, sample);
    case REC_TRANSREC_L1: return wide_dispatch<REC_TRANSREC_L1>(m, tuples, n, lr, atomic, qoff, slices, flush_at, st, sample);
    default: rec_set_error("the wide update kernel is not available for this model"); return -1;
    }
}

int launch_sgd_wide(const DevModel& m, const int4* tuples, const int64_t* qoff, int slices, int64_t n, double lr, int atomic, int flush_at, hipStream_t st)
{
    return wide_launch(m, tuples, qoff, slices, n, lr, atomic, flush_at, st, nullptr);
}

int launch_sgd_wide_sampled(const DevModel& m, int64_t n, int64_t first_step, uint64_t seed, uint64_t epoch, double lr, int atomic, hipStream_t st)
{
    if (m.n_elig <= 0) { rec_set_error("no user (or source item) is eligible for sampling"); return -1; }
    const SampleArgs sa = {seed, epoch, first_step};
    return wide_launch(m, nullptr, nullptr, 0, n, lr, atomic, 1, st, &sa);
}
