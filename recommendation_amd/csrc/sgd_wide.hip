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
// Pipeline, batch claiming and owner mode are those of sgd_hogwild.hip (which documents them); steps that share rows
// by construction ("share_negative", "share_positive") are not folded here and stay with that kernel.
#include "common.h"
#include "sgd_step.h"
#include <algorithm>

constexpr int WIDE_EPL = 4;
constexpr int WIDE_OWNER_THREADS = 1024;   // one workgroup per CU in owner mode
constexpr int WIDE_SLICES = 256;

// FULL: K (and KK) == 4 * LPT, every lane of a group holds four live components (no width tests)
template <int MODEL, int LPT, int B, bool OWNER, bool FULL>
__global__ void __launch_bounds__(OWNER ? WIDE_OWNER_THREADS : 256) sgd_wide_kernel(DevModel m, const int4* __restrict__ tuples, int64_t n,
                                                                                   float lr, int atomic, OwnerArgs ow)
{
    using T = float;
    constexpr int R = Spec<MODEL>::R, NB = Spec<MODEL>::NB, EPL = WIDE_EPL;
    constexpr bool TR = MODEL == REC_TRANSREC || MODEL == REC_TRANSREC_L1;
    constexpr int ROWF = EPL * LPT;                     // floats of a group's row image in the staging area
    static_assert(B * R <= LPT && 2 * B <= LPT, "one lane of the group per row flag / bias");
    const int g = threadIdx.x & (LPT - 1);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // dynamic LDS: [staging: 64 x 4 floats per wavefront][owner mode: accumulators [rows_res][acc_width], ...]
    extern __shared__ __align__(16) unsigned char wide_lds[];
    T* stage = (T*)wide_lds + (size_t)wave * 256;
    T* acc = (T*)wide_lds + (size_t)(blockDim.x >> 6) * 256;
    // owner mode: [hot_slots][hot_width] sums of this workgroup's deltas to the hot items' other rows (behind acc, so that
    // both stay 16-byte aligned for the four-float accesses); then the counters: pend [rows_res] additions to an
    // accumulator row since its last flush, hot_pend [hot_slots][tables] the same per hot item and table
    constexpr int NSEG = HotSegs<MODEL>::N;
    T* hot_acc = acc + (size_t)ow.rows_res * ow.acc_width;
    unsigned* pend = (unsigned*)(hot_acc + (size_t)ow.hot_slots * ow.hot_width);
    unsigned* hot_pend = pend + ow.rows_res;
    // owner mode, models with item biases: [rows_res] this workgroup's pending additions to the bias of each resident item
    // in the negative's role (plain stores of the tuples it applies at the same moment would overwrite each other)
    T* acc_bias = (T*)(hot_pend + ow.hot_slots * NSEG);
    __shared__ int lds_next;
    const int slice_base = OWNER ? (int)blockIdx.x * ow.slice_rows : 0;
    if constexpr (OWNER) {
        const int64_t q0 = ow.qoff[blockIdx.x];
        tuples += q0; n = ow.qoff[blockIdx.x + 1] - q0;
        for (int i = threadIdx.x; i < ow.rows_res * ow.acc_width; i += blockDim.x) acc[i] = 0;
        for (int i = threadIdx.x; i < ow.rows_res; i += blockDim.x) pend[i] = 0;
        for (int i = threadIdx.x; i < ow.hot_slots * ow.hot_width; i += blockDim.x) hot_acc[i] = 0;
        for (int i = threadIdx.x; i < ow.hot_slots * NSEG; i += blockDim.x) hot_pend[i] = 0;
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
    constexpr int CLAIM = OWNER ? 1 : 64;
    int64_t claim_cur = 0; int claim_left = 0;
    auto next_batch = [&]() -> int64_t {
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
                if ((atomic >> 8) & 1) bias_hot = m.hot[p - (const T*)m.tab[MODEL == REC_BPRMF ? BPRMF_BETA : TR_BETA]];
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

    int4 tp[B], tn[B];
    float4 v[B][R]; T bs;
    int hf; uint8_t bias_hot;
    int64_t b_p = next_batch(), b_n = next_batch();
    load_idx(b_p, tp);
    load_idx(b_n, tn);
    load_rows(tp, v, bs, hf, bias_hot);
    // The loop is left by the wavefront as a whole (the staged scatter reads across its lane groups): a group whose
    // claims have run past the end keeps going on invalid tuples, which load and write nothing.
    while (__ballot(b_p < nbatch)) {
        int4 tnn[B];
        const int64_t b_nn = next_batch();
        load_idx(b_nn, tnn);
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
                if (sl >= 0 && g == 2 * b + 1) { bias_slot = sl; bs += acc_bias[sl]; }
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
                            const float4 a = *reinterpret_cast<const float4*>(acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K) + EPL * g);
                            w[r][0] += a.x; w[r][1] += a.y; w[r][2] += a.z; w[r][3] += a.w;
                        }
                    }
                }
            }
            // Rows of hot items are not scattered with per-tuple atomics: from all over the chip they would serialize on
            // one address.  Plain mode: plain racy stores instead.  Owner mode: the negative's rows are summed in LDS whether
            // hot or not (plain stores of the tuples one workgroup applies at the same moment would overwrite each other),
            // and a hot item's other rows go to the workgroup's accumulator of that item: no update is lost, and memory
            // sees one atomic row per hot_flush additions of a workgroup.
            bool at[R]; int hs[R];   // hs: accumulator slot of the row's hot item (-1: none)
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int flag = __shfl(hf, b * R + r, LPT);
                hs[r] = (OWNER && hot_seg<MODEL>(r) >= 0 && flag >= 1 && flag <= ow.hot_slots) ? flag - 1 : -1;
                // (owner mode: a hot item beyond the accumulator slots takes per-tuple atomics - its negative's sums arrive as
                // atomics, and a plain store of old + delta would wipe them out)
                at[r] = ((atomic >> r) & 1) && (OWNER || !flag);
                if constexpr (OWNER) {
                    if (hs[r] >= 0) {
                        const int width = (hot_seg<MODEL>(r) > 0 && (MODEL == REC_FPMC || MODEL == REC_PRME)) ? m.KK : m.K;
                        if (FULL || EPL * g < width) {
                            const float4 a = *reinterpret_cast<const float4*>(hot_acc + (size_t)hs[r] * ow.hot_width + hot_seg_off<MODEL>(hot_seg<MODEL>(r), m.K, m.KK) + EPL * g);
                            w[r][0] += a.x; w[r][1] += a.y; w[r][2] += a.z; w[r][3] += a.w;
                        }
                    }
                }
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
                const bool to_hot = OWNER && hs[r] >= 0 && at[r] && ok;
                if (to_acc) {
                    if (act) {
                        T* a = acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K) + EPL * g;
#pragma unroll
                        for (int j = 0; j < EPL; j++) unsafeAtomicAdd(a + j, w[r][j]);
                    }
                    accumulated = true;
                } else if (to_hot) {
                    if (act) {
                        T* a = hot_acc + (size_t)hs[r] * ow.hot_width + hot_seg_off<MODEL>(hot_seg<MODEL>(r), m.K, m.KK) + EPL * g;
#pragma unroll
                        for (int j = 0; j < EPL; j++) unsafeAtomicAdd(a + j, w[r][j]);
                    }
                } else if (!at[r]) {
                    if (act) *reinterpret_cast<float4*>(row + EPL * g) = make_float4(old[r][0] + w[r][0], old[r][1] + w[r][1], old[r][2] + w[r][2], old[r][3] + w[r][3]);
                }
                if constexpr (OWNER) {
                    if (hot_seg<MODEL>(r) >= 0) {
                        // every hot_flush-th addition of this workgroup to the hot item's table moves the sum to memory
                        bool fire0 = false;
                        if (to_hot && g == 0) fire0 = atomicInc(hot_pend + hs[r] * NSEG + hot_seg<MODEL>(r), (unsigned)(ow.hot_flush - 1)) == (unsigned)(ow.hot_flush - 1);
                        const unsigned long long fired = __ballot(fire0);
                        if (fired) {
                            const bool fire = (fired >> (lane & ~(LPT - 1))) & 1ull;
                            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (fire && (FULL || EPL * g < width)) {
                                T* a = hot_acc + (size_t)hs[r] * ow.hot_width + hot_seg_off<MODEL>(hot_seg<MODEL>(r), m.K, m.KK) + EPL * g;
                                x = make_float4(lds_take(a), lds_take(a + 1), lds_take(a + 2), lds_take(a + 3));
                            }
                            stage_and_flush(x, row_table<MODEL, T>(m, r), width, fire ? row_number<MODEL>(tp[b], r) : -1);
                        }
                    }
                }
                // atomics: through the staging image, if any group of the wavefront needs them for this row
                const bool wants = ok && at[r] && !to_acc && !to_hot;
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
                            T* a = acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K) + EPL * g;
                            x = make_float4(lds_take(a), lds_take(a + 1), lds_take(a + 2), lds_take(a + 3));
                        }
                        stage_and_flush(x, row_table<MODEL, T>(m, r), width, fire ? row_number<MODEL>(tp[b], r) : -1);
                    }
                    if constexpr (NB == 2) {
                        if (fire && g == 0) unsafeAtomicAdd(bias_ptr<MODEL, T>(m, tp[b], 1), lds_take(acc_bias + sl));
                    }
                }
            }
        }
        if constexpr (NB == 2) {
            // BPRMF.cpp:92-93: +deri on the positive, -deri on the negative; TransRec.cpp:225-226: the opposite signs
            T* p = my_bias_ptr(tp);
            const T sgn = ((g & 1) == 0) == (MODEL == REC_BPRMF) ? (T)1 : (T)-1;
            const T delta = p ? lr * (sgn * mine - (T)m.bias_reg * bs) : (T)0;
            if (p && bias_slot >= 0) unsafeAtomicAdd(acc_bias + bias_slot, delta);
            else if (p) scatter(p, delta, bs, ((atomic >> 8) & 1) && !bias_hot);
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
            const bool dirty = sl < ow.rows_res && pend[sl] != 0;
            if (!__ballot(dirty)) continue;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (!owned_row<MODEL>(r)) continue;
                const int width = ((MODEL == REC_FPMC || MODEL == REC_PRME) && r == 5) ? m.KK : m.K;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dirty && (FULL || EPL * g < width)) x = *reinterpret_cast<const float4*>(acc + (size_t)sl * ow.acc_width + owned_off<MODEL>(r, m.K) + EPL * g);
                *reinterpret_cast<float4*>(stage + EPL * lane) = x;
                flush_stage(row_table<MODEL, T>(m, r), width, dirty ? slice_base + sl : -1);
            }
            if constexpr (NB == 2) {
                if (dirty && g == 0) unsafeAtomicAdd((T*)m.tab[MODEL == REC_BPRMF ? BPRMF_BETA : TR_BETA] + slice_base + sl, acc_bias[sl]);
            }
        }
        // and the hot items' sums, one group per (slot, table)
        for (int s0 = 0; s0 < ow.hot_slots * NSEG; s0 += groups) {
            const int idx = s0 + gid;
            const bool dirty = idx < ow.hot_slots * NSEG && hot_pend[idx] != 0;
            if (!__ballot(dirty)) continue;
            const int slot = idx / NSEG, seg = idx % NSEG;
            const int item = dirty ? m.hot_items[slot] : -1;
#pragma unroll
            for (int sg = 0; sg < NSEG; sg++) {
                // a wave-instruction of the staged scatter serves one table: the groups take turns by table
                const bool mine_now = dirty && seg == sg;
                if (!__ballot(mine_now)) continue;
                const int rr = sg == 0 ? hot_seg_row<MODEL>(0) : sg == 1 ? hot_seg_row<MODEL>(1) : hot_seg_row<MODEL>(2);
                const int width = (sg > 0 && (MODEL == REC_FPMC || MODEL == REC_PRME)) ? m.KK : m.K;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mine_now && (FULL || EPL * g < width)) x = *reinterpret_cast<const float4*>(hot_acc + (size_t)slot * ow.hot_width + hot_seg_off<MODEL>(sg, m.K, m.KK) + EPL * g);
                *reinterpret_cast<float4*>(stage + EPL * lane) = x;
                flush_stage(row_table<MODEL, T>(m, rr), width, mine_now ? item : -1);
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

template <int MODEL>
static int wide_dispatch(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, const int64_t* qoff, int flush_at, int hot_flush, hipStream_t st)
{
    const int Kmax = m.K > m.KK ? m.K : m.KK;
    OwnerArgs ow = {nullptr, 0, 0, 0, 1, 0, 0, 1};
    auto go = [&](auto kern, auto kern_owner, int lpt, int b) -> int {
        if (qoff) {
            const size_t stage_bytes = (size_t)(WIDE_OWNER_THREADS / 64) * 1024;
            ow.qoff = qoff; ow.slice_rows = (m.I + WIDE_SLICES - 1) / WIDE_SLICES; ow.flush_at = flush_at < 1 ? 1 : flush_at;
            ow.acc_width = (MODEL == REC_FPMC || MODEL == REC_PRME) ? m.K + m.KK : m.K;
            ow.hot_slots = m.n_hot_slots; ow.hot_width = hot_width<MODEL>(m.K, m.KK); ow.hot_flush = hot_flush < 1 ? 1 : hot_flush;
            const size_t hot_bytes = (size_t)ow.hot_slots * ((size_t)ow.hot_width * sizeof(float) + sizeof(unsigned) * HotSegs<MODEL>::N);
            const size_t per_row = (size_t)ow.acc_width * sizeof(float) + sizeof(unsigned) + (Spec<MODEL>::NB == 2 ? sizeof(float) : 0);
            ow.rows_res = (int)std::min<size_t>((size_t)ow.slice_rows, (wide_owner_lds_budget() - stage_bytes - hot_bytes) / per_row);
            const size_t lds = stage_bytes + (size_t)ow.rows_res * per_row + hot_bytes;
            HIP_TRY(hipFuncSetAttribute((const void*)kern_owner, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern_owner, dim3(WIDE_SLICES), dim3(WIDE_OWNER_THREADS), lds, st, m, tuples, n, (float)lr, atomic, ow);
            return 0;
        }
        const int64_t groups_per_block = 256 / lpt;
        int64_t blocks = ((n + b - 1) / b + groups_per_block - 1) / groups_per_block;
        if (blocks > 2048) blocks = 2048;
        HIP_TRY(hipMemsetAsync(m.work_counter, 0, sizeof(int), st));   // the batch counter the groups claim from
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 4 * 1024, st, m, tuples, n, (float)lr, atomic, ow);
        return 0;
    };
    constexpr int R = Spec<MODEL>::R;
    int rc = 0;
    const bool full_k = m.K == m.KK || !(MODEL == REC_FPMC || MODEL == REC_PRME);   // every row of the tuple has K components
#define WIDE_SHAPE(LPT, BB)                                                                                                               \
    rc = (full_k && m.K == 4 * LPT) ? go(sgd_wide_kernel<MODEL, LPT, BB, false, true>, sgd_wide_kernel<MODEL, LPT, BB, true, true>, LPT, BB) \
                                    : go(sgd_wide_kernel<MODEL, LPT, BB, false, false>, sgd_wide_kernel<MODEL, LPT, BB, true, false>, LPT, BB)
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
int launch_sgd_wide(const DevModel& m, const int4* tuples, const int64_t* qoff, int64_t n, double lr, int atomic, int flush_at, int hot_flush, hipStream_t st)
{
    if (n <= 0) return 0;
    switch (m.model) {
    case REC_BPRMF: return wide_dispatch<REC_BPRMF>(m, tuples, n, lr, atomic, qoff, flush_at, hot_flush, st);
    case REC_MC: return wide_dispatch<REC_MC>(m, tuples, n, lr, atomic, qoff, flush_at, hot_flush, st);
    case REC_FPMC: return wide_dispatch<REC_FPMC>(m, tuples, n, lr, atomic, qoff, flush_at, hot_flush, st);
    case REC_PRME: return wide_dispatch<REC_PRME>(m, tuples, n, lr, atomic, qoff, flush_at, hot_flush, st);
    case REC_HRM_AVG: return wide_dispatch<REC_HRM_AVG>(m, tuples, n, lr, atomic, qoff, flush_at, hot_flush, st);
    case REC_HRM_MAX: return wide_dispatch<REC_HRM_MAX>(m, tuples, n, lr, atomic, qoff, flush_at, hot_flush, st);
    case REC_TRANSREC: return wide_dispatch<REC_TRANSREC>(m, tuples, n, lr, atomic, qoff, flush_at, hot_flush, st);
    case REC_TRANSREC_L1: return wide_dispatch<REC_TRANSREC_L1>(m, tuples, n, lr, atomic, qoff, flush_at, hot_flush, st);
    default: rec_set_error("the wide update kernel is not available for this model"); return -1;
    }
}
