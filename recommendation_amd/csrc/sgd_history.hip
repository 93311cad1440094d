// Hogwild BPR-SGD step of the models whose tuple touches a variable number of rows: Fossil,
// FossilSimple and REBUS (Fossil.cpp:140-252, FossilSimple.cpp:151-235, REBUS.cpp:282-444).
// One wavefront per group of G consecutive tuples of one user (G = 1 unless the sampler option
// user_batch groups them); lane l owns components k = l, l+64, ...
//
// A tuple reads every item row of the user's history (Fossil: the whole training sequence with
// repeats, Fossil.cpp:140-153; REBUS: the distinct items of the prefix, REBUS.cpp:305-318) twice:
// pass 1 streams the rows into the history sums, pass 2 re-reads each row (an L2 / Infinity Cache
// hit) for its own regulariser term and scatters its delta with float atomics.  The rows of the CSR
// are consecutive in seq_item, so the item ids themselves are read as coalesced runs.
// The G tuples of a group read the same rows, so the rows are streamed once per group.  Pass 1 keeps
// ONE running sum for the group: a tuple's sum is that sum (REBUS: as it stood where the tuple's prefix
// ended) less the rows of its own positive / negative, which are found by comparing ids, a round of
// them at a time.  Pass 2 scatters the sum of the G deltas with one atomic per component and row: for
// a row that is none of the group's positives, negatives or short-term / path items that sum is the
// same vector for every row (formed once per group; REBUS: once per stretch between two prefix ends)
// plus the row's own regulariser term - one multiply-add per row; the few other rows go through the
// per-tuple case analysis of the reference.  All G tuples see the parameters as they were when the
// group started, which is what tuples in flight on different wavefronts see of each other anyway.
// What bounds the kernels at 1M users x 200k items x 100M clicks, K = 64, f32 (PMC): REBUS, four
// tuples per walk, the float atomics (0.91 of the 1.3 TB/s the memory side executes them at); Fossil /
// FossilSimple, eight per walk, the per-tuple work outside the walks (atomics at 0.7 / 0.8 of that
// rate).  Float atomics on the few most clicked rows run far below that rate (thousands of adds queue
// on one address: all-atomics costs REBUS 4x the time per epoch): those rows are plain stores.
// Parameters that every tuple touches are not scattered per tuple (their atomics would all hit one
// address): Fossil's global weights WT[0..L) and FossilSimple's alpha collect their gradient sums
// per block and are folded in by launch_apply_global_history after the launch; they stay constant
// during a launch.
#include <algorithm>
#include "common.h"

namespace {
template <class T>
__device__ __forceinline__ void scat(T* p, T delta, T oldv, bool atomic)
{
    if (atomic) unsafeAtomicAdd(p, delta);
    else *p = oldv + delta;
}
// A value that is the same on every lane of the wavefront, moved to a scalar register: comparisons on it become
// scalar branches and row addresses built from it a scalar base plus the lane's offset.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni(int64_t v)
{
    return (int64_t)(((uint64_t)(uint32_t)uni((int)((uint64_t)v >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)v));
}
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
template <class T> __device__ __forceinline__ T sigm(T x) { return (T)1 / ((T)1 + rexp(x)); }
template <class T> __device__ __forceinline__ T wsum(T v) { return group_sum<64>(v); }

// Walks item rows of one user's history TWICE (the pass that sums them, then the pass that updates them), RIF rows per
// round: tab[seq[h]] for the positions h in [0, hend), or, when DENSE, the rows of the first `hend` entries {item,
// position} of `seq` (pairs, positions ascending) - a list that holds only the positions a walk keeps, so that no
// round is spent on the others.  Both passes are one stream of rounds: a pass is `R3` rounds (padded to a multiple of
// three), round t of the stream is round t - R3 of the second pass when t >= R3, and the rows and ids a round needs are
// requested two and five rounds ahead IN THE STREAM - the last rounds of the first pass request the first rows of the
// second, so the work between the passes waits for no load of the walk.
//
// round(t, ids, hpos, it, x) is called once per round that holds a row: lane q < RIF of `ids` holds the item of row q
// of the round (-1: past the end) and of `hpos` its position, it[q] is the same id as a wave-uniform value, x[q] this
// lane's components of its row (UNDEFINED where it[q] < 0, and on lanes whose component is past K: the caller clears
// what it sums on those lanes).  In the pass that updates (SCATTER) it returns two bit masks over the rows of the
// round: the rows of the low 16 bits are updated by the common step cD - creg * x (float atomics, or plain stores for
// hot items / !at_item), the rows of the high 16 bits are handed one by one to special(position, item, row components,
// delta out), which returns whether to update the row; after() runs when the round's rows have been dealt with (it
// may change cD / creg for the next round).
//
// Memory operations retire in order on a wavefront, a wait is "until at most N are outstanding", and the compiler can
// only count the operations that are certain to have been issued.  So every load of the loops is unconditional
// (entries past the end re-read the last one, rows past the end read row 0, lanes past K the last component) and a
// round is laid out as: request the item ids of the fifth round from here (one load: lane q fetches entry h0+q, the
// ids are read back lane by lane); request the rows (and hot flags: one load, lane q fetches the flag of row q) of
// the round after the next; compute this round's deltas from registers; issue this round's scatters back to back.
// The rows a round computes with were requested two rounds earlier, and nothing a scatter needs comes from a load
// that is younger than the previous round's scatters.  Round t keeps its rows in buffer t % 3 and its raw ids in slot
// t % 3; the loops are unrolled three times so that no register that a load is still going to write has to be moved
// (a move would wait for the load).
// A row that round r updates and rounds r+1 / r+2 read again (a repeated item) is read before its update.
template <int RIF, int EPL, bool DENSE, class T>
struct HistoryWalk {
    static_assert(RIF <= 16, "one lane per row of a round, two masks of one bit per row");
    T* tab; const int32_t* seq; const uint8_t* hot;
    int K, lane, hend, last, R3;
    uint32_t kc[EPL], row_bytes;
    int id[3], hp[3], hotf[3], rid[3], rpos[3];
    T x[3][RIF][EPL];

    __device__ __forceinline__ void load_ids(int t, int slot)
    {
        const int tl = t < R3 ? t : t - R3;          // (rounds past the second pass read its padding again: never used)
        int h = tl * RIF + lane; h = h < last ? h : last;
        if (DENSE) { const int2 e = ((const int2*)seq)[h]; rid[slot] = e.x; rpos[slot] = e.y; }
        else { rid[slot] = seq[h]; rpos[slot] = 0; }
    }
    __device__ __forceinline__ void resolve(int t, int slot)
    {
        const int tl = t < R3 ? t : t - R3;
        const int h = tl * RIF + lane;
        hp[slot] = DENSE ? rpos[slot] : h;
        id[slot] = (lane < RIF && h < hend && t < 2 * R3) ? rid[slot] : -1;
    }
    __device__ __forceinline__ void load_rows(int slot)
    {
        const int idc = id[slot] > 0 ? id[slot] : 0;
        const uint64_t ob = (uint64_t)(uint32_t)idc * row_bytes;   // lane q: byte offset of row q
        const int olo = (int)(uint32_t)ob, ohi = (int)(uint32_t)(ob >> 32);
#pragma unroll
        for (int q = 0; q < RIF; q++) {
            const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(ohi, q) << 32) | (uint32_t)__builtin_amdgcn_readlane(olo, q);
            const T* row = (const T*)((const char*)tab + o);
#pragma unroll
            for (int j = 0; j < EPL; j++) x[slot][q][j] = row[kc[j]];
        }
        hotf[slot] = (int)hot[idc];
    }
    // rows_ = the rows of the walk (positions, or entries of the dense list): requests the ids of the first two rounds
    __device__ __forceinline__ void begin(T* tab_, const int32_t* seq_, const uint8_t* hot_, int K_, int lane_, int rows_)
    {
        tab = tab_; seq = seq_; hot = hot_; K = K_; lane = lane_; hend = rows_; last = rows_ > 0 ? rows_ - 1 : 0;
        R3 = 3 * ((rows_ + 3 * RIF - 1) / (3 * RIF));
#pragma unroll
        for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; kc[j] = (uint32_t)(k < K ? k : K - 1); }
        row_bytes = (uint32_t)K * (uint32_t)sizeof(T);
        load_ids(0, 0); load_ids(1, 1);
    }
    // requests the rows of the first two rounds (and the ids of the next three)
    __device__ __forceinline__ void first_rows()
    {
        resolve(0, 0); resolve(1, 1);
        load_ids(2, 2); load_ids(3, 0); load_ids(4, 1);
        load_rows(0); load_rows(1);
    }
    template <int S, bool SCATTER, class Round, class Special, class After>
    __device__ __forceinline__ void stage(int t, bool at_item, const T (&cD)[EPL], const T& creg, Round& round, Special& special, After& after)
    {
        constexpr int LD = (S + 2) % 3;
        resolve(t + 2, LD);
        load_ids(t + 5, LD);
        load_rows(LD);
        if (__builtin_amdgcn_readlane(id[S], 0) < 0) return;   // a padding round (rows past the end are all at the end)
        int it[RIF];
#pragma unroll
        for (int q = 0; q < RIF; q++) it[q] = __builtin_amdgcn_readlane(id[S], q);
        unsigned masks = round(t, id[S], hp[S], it, x[S]);
        if constexpr (SCATTER) {
            // (this round's rows are needed from here on whatever the masks say: asking for them once, outside the
            // conditional code below, keeps the compiler from waiting for EVERY outstanding load in front of the loop
            // over the special rows)
#pragma unroll
            for (int q = 0; q < RIF; q++)
#pragma unroll
                for (int j = 0; j < EPL; j++) asm volatile("" :: "v"(x[S][q][j]));
#pragma unroll
            for (int q = 0; q < RIF; q++) {
                if (!((masks >> q) & 1)) continue;
                T* row = tab + (int64_t)it[q] * K;
                const bool atc = at_item && !__builtin_amdgcn_readlane(hotf[S], q);
#pragma unroll
                for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; if (k < K) scat(row + k, cD[j] - creg * x[S][q][j], x[S][q][j], atc); }
            }
            unsigned sm = (unsigned)uni((int)(masks >> 16));
            while (sm) {
                const int q = __builtin_ctz(sm); sm &= sm - 1;
                const int c = __builtin_amdgcn_readlane(id[S], q);
                T xs[EPL], ds[EPL];
#pragma unroll
                for (int j = 0; j < EPL; j++) {
                    // row q of the round, q known at run time only: masked together word by word (a chain of selects is
                    // turned into an indexed load from a copy of the rows in scratch memory)
                    uint32_t w[sizeof(T) / 4];
#pragma unroll
                    for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = 0;
#pragma unroll
                    for (int qq = 0; qq < RIF; qq++) {
                        uint32_t v[sizeof(T) / 4];
                        __builtin_memcpy(v, &x[S][qq][j], sizeof(T));
                        const uint32_t msk = q == qq ? ~0u : 0u;
#pragma unroll
                        for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] |= v[i] & msk;
                    }
                    __builtin_memcpy(&xs[j], w, sizeof(T));
                    ds[j] = 0;
                }
                if (!special(__builtin_amdgcn_readlane(hp[S], q), c, xs, ds)) continue;
                T* row = tab + (int64_t)c * K;
                const bool atc = at_item && !__builtin_amdgcn_readlane(hotf[S], q);
#pragma unroll
                for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; if (k < K) scat(row + k, ds[j], xs[j], atc); }
            }
            after();
        }
    }
    // one pass: SCATTER = false the first (rounds 0 .. R3 of the stream), true the second (R3 .. 2 R3)
    template <bool SCATTER, class Round, class Special, class After>
    __device__ __forceinline__ void pass(bool at_item, const T (&cD)[EPL], const T& creg, Round round, Special special, After after)
    {
        const int t0 = SCATTER ? R3 : 0;
        for (int t = t0; t < t0 + R3; t += 3) {
            stage<0, SCATTER>(t, at_item, cD, creg, round, special, after);
            stage<1, SCATTER>(t + 1, at_item, cD, creg, round, special, after);
            stage<2, SCATTER>(t + 2, at_item, cD, creg, round, special, after);
        }
    }
};

// lane l of `v` as a wave-uniform double
__device__ __forceinline__ double lane_double(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// The items of a group of G tuples whose rows need more than the common step of pass 2 (the positive, the negative
// and the up to Ls short-term / path items of every tuple), one per lane: slot g * (Ls + 2) + r.
template <int G>
__device__ __forceinline__ int special_items(const int (&shortl)[G], const int (&pos)[G], const int (&neg)[G], int Ls, int lane)
{
    int spec = -2;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int r = lane - g * (Ls + 2);
        const int v = __shfl(shortl[g], r & 63, 64);
        if (r >= 0 && r < Ls + 2) spec = r < Ls ? v : (r == Ls ? pos[g] : neg[g]);
    }
    return spec;
}

// (f32, K <= 64: one or two tuples per walk held to 128 registers, four wavefronts per SIMD; Fossil / FossilSimple with eight
// per walk to 168, three per SIMD, for ten spilled registers: Fossil 387 -> 327 ms per epoch at 1M users x 200k items,
// FossilSimple unchanged; at 128 registers both lose 30 .. 90 ms to the spills)
template <int MODEL, class T, int EPL, int G>
__global__ void __launch_bounds__(256, (sizeof(T) == 4 && EPL == 1 && G <= 2) ? 4 : (sizeof(T) == 4 && EPL == 1 && G == 8 && MODEL != REC_REBUS) ? 3 : 1) sgd_history_batch_kernel(DevModel m, const int4* __restrict__ tuples, int64_t n, T lr, int atomic, int CLAIM)
{
    constexpr int RIF = EPL == 1 ? 8 : 4;   // history rows per round
    const int lane = threadIdx.x & 63;
    const int K = m.K, L = m.L, I = m.I;
    const T lam = (T)m.lambda, breg = (T)m.bias_reg;
    const bool at_item = (atomic >> 1) & 1, at_user = atomic & 1, at_bias = (atomic >> 8) & 1;
    const int64_t ngroups = (n + G - 1) / G;

    T gsum = 0, gcnt = 0;   // Fossil: lane ind collects WT[ind]'s gradient; FossilSimple: lane 0 alpha's

    // Groups are claimed from a counter, CLAIM at a time: a group's work is proportional to the length of its
    // user's history, which varies too much for a fixed assignment of groups to wavefronts to balance.
    // The claim that follows is issued before the claimed groups' work and read after it.
    int next = 0;
    if (lane == 0) next = atomicAdd(m.work_counter, 1);
    for (;;) {
        const int64_t first = (int64_t)uni(next) * CLAIM;
        if (first >= ngroups) break;
        if (lane == 0) next = atomicAdd(m.work_counter, 1);
      const int64_t t_end = (first + CLAIM) * G < n ? (first + CLAIM) * G : n;
      int step = G;
      for (int64_t base = first * G; base < t_end; base += step) {
        const int u = uni(tuples[base].x);
        const int64_t so = uni(m.seq_off[u]);
        const int32_t* seq = m.seq_item + so;
        const int nu = uni((int)(m.seq_off[u + 1] - so));
        // The tuples of a group all step from the parameters the group found.  On a user with a handful of clicks they
        // mostly name the same few rows, so the summed step is close to a single step of G times the learning rate (REBUS
        // on a corpus of 3-click users diverged that way).  Such a user's tuples are applied one after the other
        // instead - its history is short, so the shared walk saves little - each waiting for the scatters of the one
        // before it and dropping what the vector cache holds of the rows.
        const bool alone = G > 1 && nu < m.pair_min_hist;
        const int in_group = (int)(base % G);                        // non-zero only inside a group that is applied tuple by tuple
        const int left = (int)(n - base < G - in_group ? n - base : G - in_group);
        const int ng = alone ? 1 : left;
        step = ng;
        if (alone && in_group != 0) { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
        int4 tg[G];
#pragma unroll
        for (int g = 0; g < G; g++) tg[g] = tuples[base + (g < ng ? g : 0)];

        if constexpr (MODEL == REC_FOSSIL || MODEL == REC_FOSSIL_SIMPLE) {
            constexpr bool SIMPLE = MODEL == REC_FOSSIL_SIMPLE;
            T* beta = (T*)m.tab[SIMPLE ? FS_BETA : FO_BETA];
            T* Ui = (T*)m.tab[SIMPLE ? FS_UI : FO_UI];
            T* V = (T*)m.tab[SIMPLE ? FS_V : FO_V];
            // tuple: Fossil (u, t, neg, -), FossilSimple (u, prev, pos, neg)
            HistoryWalk<RIF, EPL, false, T> walk;
            walk.begin(Ui, seq, m.hot, K, lane, nu);   // (its first loads travel with the tuples' own rows)
            // Every load of the set-up is unconditional (the slots past the group's end repeat tuple 0, lanes past K read
            // the last component) and the loads of one dependency level are issued together for all G tuples: a branch,
            // or a wave-uniform value taken out of a loaded register, between two loads makes the second wait for the
            // first - set up tuple by tuple, a group of eight waited for memory some seventy times.
            uint32_t kc[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; kc[j] = (uint32_t)(k < K ? k : K - 1); }
            // (a zero the compiler cannot see through: added to a wave-uniform index it keeps the loaded value in a vector
            // register until it is asked for, instead of a copy to a scalar register - and the wait that needs - at the load)
            int vz = 0; asm volatile("" : "+v"(vz));
            int tpos[G], pos[G], neg[G], np[G], prevl[G], pos_r[G];   // prevl: lane ind holds the ind-th previous item of tuple g
#pragma unroll
            for (int g = 0; g < G; g++) {   // level 2: what the tuples and the sequence say
                tpos[g] = SIMPLE ? 0 : uni(tg[g].y);
                neg[g] = uni(SIMPLE ? tg[g].w : tg[g].z);
                np[g] = SIMPLE ? 1 : (tpos[g] < L ? tpos[g] : L);
                if (SIMPLE) { pos_r[g] = tg[g].z; prevl[g] = tg[g].y; }
                else { const int back = tpos[g] - 1 - lane; pos_r[g] = seq[tpos[g] + vz]; prevl[g] = seq[back > 0 ? back : 0]; }
            }
            // short-term weights: lane ind holds WT[ind] + WTu[u][ind] (Fossil) / lane 0 alpha + alpha_u[u] (FossilSimple)
            const int wl = lane < (SIMPLE ? 1 : L) ? lane : 0;
            T wt_g = SIMPLE ? ((T*)m.tab[FS_ALPHA])[0] : ((T*)m.tab[FO_WT])[wl];
            T wt_u = SIMPLE ? ((T*)m.tab[FS_ALPHA_U])[u] : ((T*)m.tab[FO_WTU])[(int64_t)u * L + wl];
#pragma unroll
            for (int g = 0; g < G; g++) { pos[g] = uni(pos_r[g]); prevl[g] = lane < np[g] ? prevl[g] : -2; }
            // level 3: the rows of the tuples' own items (the positive's and the negative's rows in the history table too:
            // pass 1 sums every position and takes these out afterwards) and of their short-term items
            constexpr int LR = 3;   // short-term rows per tuple held in registers; longer contexts fetch the rest in place
            int hp[G], hn[G];       // hot flags of pos / neg
            T bp[G], bn[G];         // the biases as the group found them
            T vp[G][EPL], vn[G][EPL], sp[G][EPL], sn[G][EPL], uip[G][EPL], uin[G][EPL], px[G][LR][EPL];
#pragma unroll
            for (int g = 0; g < G; g++) {
                hp[g] = m.hot[pos[g] + vz]; hn[g] = m.hot[neg[g] + vz];
                bp[g] = beta[pos[g]]; bn[g] = beta[neg[g]];
#pragma unroll
                for (int j = 0; j < EPL; j++) {
                    vp[g][j] = V[(int64_t)pos[g] * K + kc[j]]; vn[g][j] = V[(int64_t)neg[g] * K + kc[j]];
                    uip[g][j] = Ui[(int64_t)pos[g] * K + kc[j]]; uin[g][j] = Ui[(int64_t)neg[g] * K + kc[j]];
                }
#pragma unroll
                for (int ind = 0; ind < LR; ind++) {
                    const int cl = uni(__shfl(prevl[g], ind, 64));   // -2 past the tuple's short-term items: row 0 is read then, and not used
                    const int c = cl > 0 ? cl : 0;
#pragma unroll
                    for (int j = 0; j < EPL; j++) px[g][ind][j] = Ui[(int64_t)c * K + kc[j]];
                }
            }
            // (the weight of a sum of c rows, c^-0.2, from the table of all lengths: lane l fetches the entry of nu - l - the sums
            // leave out the occurrences of the positive / the negative, a few positions at most)
            const double lenw = m.len_pow[nu - lane > 0 ? nu - lane : 0];
            walk.first_rows();
            __builtin_amdgcn_sched_barrier(0);   // (the instruction scheduler would move the masking below up between the loads, tuple by tuple)
            // what the slots past the group's end and the lanes past K must not contribute
            if (lane >= (SIMPLE ? 1 : L)) wt_g = wt_u = 0;
#pragma unroll
            for (int g = 0; g < G; g++) {
                hp[g] = uni(hp[g]); hn[g] = uni(hn[g]);
                if (g >= ng) { pos[g] = neg[g] = -1; np[g] = 0; prevl[g] = -2; hp[g] = hn[g] = 0; }
                if (!(at_item || at_bias)) hp[g] = hn[g] = 0;
#pragma unroll
                for (int j = 0; j < EPL; j++) { sp[g][j] = sn[g][j] = 0; if (lane + 64 * j >= K) vp[g][j] = vn[g][j] = 0; }
            }
            // pass 1: history sums, Fossil.cpp:140-153.  Every tuple of the group sums the whole sequence except the
            // positions that hold its positive / its negative: ONE sum over all positions (in fp64 whatever the tables'
            // type, see REBUS below), the occurrences of each tuple's two items counted on the way (lane q of `ids`
            // against a wave-uniform id: one compare per round and item) and their rows taken out afterwards.
            double S[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) S[j] = 0;
            int cp[G], cn[G];   // occurrences of pos / neg first, then the lengths of the two sums
#pragma unroll
            for (int g = 0; g < G; g++) cp[g] = cn[g] = 0;
            auto no_special = [](int, int, const T (&)[EPL], T (&)[EPL]) { return false; };
            auto no_after = []() {};
            const T none[EPL] = {}, zero = 0;
            walk.template pass<false>(false, none, zero,
                [&](int, int ids, int, const int (&it)[RIF], const T (&x)[RIF][EPL]) {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    cp[g] += __popcll(ballot64(ids == pos[g] && ids >= 0));
                    cn[g] += __popcll(ballot64(ids == neg[g] && ids >= 0));
                }
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    if (it[q] < 0) continue;
#pragma unroll
                    for (int j = 0; j < EPL; j++) S[j] += (double)x[q][j];
                }
                return 0u;
            }, no_special, no_after);
#pragma unroll
            for (int j = 0; j < EPL; j++) if (lane + 64 * j >= K) S[j] = 0;
            double spd[G][EPL], snd[G][EPL];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int op = cp[g], on = cn[g];
                cp[g] = nu - op; cn[g] = nu - on;
#pragma unroll
                for (int j = 0; j < EPL; j++) {
                    spd[g][j] = S[j] - (double)op * (double)uip[g][j];
                    snd[g][j] = S[j] - (double)on * (double)uin[g][j];
                }
            }
            const T wt = wt_g + wt_u;
            T wp[G], wn[G], deri[G];
            T user_delta = 0;   // this lane's WTu[u][lane] / alpha_u[u] delta summed over the group
#pragma unroll
            for (int g = 0; g < G; g++) {
                wp[g] = (T)(nu - cp[g] < 64 ? lane_double(lenw, nu - cp[g]) : m.len_pow[cp[g]]);
                wn[g] = (T)(nu - cn[g] < 64 ? lane_double(lenw, nu - cn[g]) : m.len_pow[cn[g]]);
                if (g >= ng) { wp[g] = wn[g] = 0; deri[g] = 0; continue; }
#pragma unroll
                for (int j = 0; j < EPL; j++) { sp[g][j] = (T)((double)wp[g] * spd[g][j]); sn[g][j] = (T)((double)wn[g] * snd[g][j]); }
                T grad_mine = 0;
                auto short_term = [&](int ind, const T (&x)[EPL]) {
                    const int c = uni(__shfl(prevl[g], ind, 64));
                    const T w = __shfl(wt, ind, 64);
                    T d1 = 0, d2 = 0;
#pragma unroll
                    for (int j = 0; j < EPL; j++) {
                        // FossilSimple adds the short-term term on both sides unconditionally (FossilSimple.cpp:180-181)
                        if (SIMPLE || c != pos[g]) sp[g][j] += w * x[j];
                        if (SIMPLE || c != neg[g]) sn[g][j] += w * x[j];
                        d1 += x[j] * vp[g][j]; d2 += x[j] * vn[g][j];
                    }
                    d1 = wsum(d1); d2 = wsum(d2);
                    // Fossil.cpp:194-206 (three cases); FossilSimple.cpp:195-196 (always the difference)
                    const T gI = (SIMPLE || (c != pos[g] && c != neg[g])) ? d1 - d2 : (c != pos[g] ? d1 : -d2);
                    if (lane == ind) grad_mine = gI;
                };
#pragma unroll
                for (int ind = 0; ind < LR; ind++)
                    if (ind < np[g]) short_term(ind, px[g][ind]);
                for (int ind = LR; ind < np[g]; ind++) {
                    const int c = uni(__shfl(prevl[g], ind, 64));
                    T x[EPL];
#pragma unroll
                    for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; x[j] = k < K ? Ui[(int64_t)c * K + k] : (T)0; }
                    short_term(ind, x);
                }
                T s1 = 0, s2 = 0;
#pragma unroll
                for (int j = 0; j < EPL; j++) { s1 += sp[g][j] * vp[g][j]; s2 += sn[g][j] * vn[g][j]; }
                const T x_uij = bp[g] - bn[g] + (wsum(s1) - wsum(s2));
                deri[g] = sigm(x_uij);
                if (lane < np[g]) {
                    gsum += deri[g] * grad_mine; gcnt += 1;
                    // "- lambda * WT[ind]" in the two aliasing branches, Fossil.cpp:201,205
                    const T regv = (SIMPLE || (prevl[g] != pos[g] && prevl[g] != neg[g])) ? wt_u : wt_g;
                    user_delta += lr * (deri[g] * grad_mine - lam * regv);
                }
            }
            // (no store before every tuple of the group is computed: the wait for a later tuple's loaded values would wait for it)
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (g >= ng || lane != 0) continue;
                scat(beta + pos[g], lr * (deri[g] - breg * bp[g]), bp[g], at_bias && !hp[g]);
                scat(beta + neg[g], lr * (-deri[g] - breg * bn[g]), bn[g], at_bias && !hn[g]);
            }
            if (lane < (SIMPLE ? 1 : L) && user_delta != (T)0) {
                T* wu = SIMPLE ? (T*)m.tab[FS_ALPHA_U] + u : (T*)m.tab[FO_WTU] + (int64_t)u * L + lane;
                scat(wu, user_delta, wt_u, at_user);
            }
            // pass 2: every history row once for the whole group, Fossil.cpp:209-247 / FossilSimple.cpp:198-226.
            // A row that is none of the group's positives, negatives or short-term items receives the same step from
            // every tuple - deri * (wp * Vp - wn * Vn), summed once per group - and its own regulariser term: one
            // multiply-add per row.  The others (found by comparing the row's id with the group's list of such items,
            // one per lane) go through the per-tuple case analysis.
            const T wt0 = __shfl(wt, 0, 64);
            const T reg = lam * (T)ng;
            const int Ls = SIMPLE ? 1 : L;
            const bool listed = G * (Ls + 2) <= 64;
            const int spec = special_items<G>(prevl, pos, neg, Ls, lane);
            T lrD[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                T D = 0;
#pragma unroll
                for (int g = 0; g < G; g++) D += deri[g] * (wp[g] * vp[g][j] - wn[g] * vn[g][j]);   // deri = wp = wn = 0 past the group's end
                lrD[j] = lr * D;
            }
            const T lrreg = lr * reg;
            walk.template pass<true>(at_item, lrD, lrreg,
                [&](int, int, int, const int (&it)[RIF], const T (&)[RIF][EPL]) {
                unsigned common = 0, sm = 0;
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    const int c = it[q];
                    if (c < 0) continue;
                    if (!listed || ballot64(spec == c) != 0) sm |= 1u << q; else common |= 1u << q;
                }
                return common | (sm << 16);
            },
                [&](int, int c, const T (&x)[EPL], T (&d)[EPL]) {
                // the common step, corrected for the tuples that list this item (the others' case is the common one)
                const uint64_t slots = ballot64(spec == c);
#pragma unroll
                for (int g = 0; g < G; g++) {
                    if (listed && (((slots >> (g * (Ls + 2))) & ((1ull << (Ls + 2)) - 1)) == 0)) continue;
                    const uint64_t hit = ballot64(prevl[g] == c);
                    const bool inprev = hit != 0;
                    T a = 0;
                    if (inprev) a = SIMPLE ? wt0 : __shfl(wt, (int)__ffsll((unsigned long long)hit) - 1, 64);
                    T cpz, cnz;   // delta_g = deri * (cpz * Vp - cnz * Vn)
                    if (SIMPLE && inprev) { cpz = wp[g] + a; cnz = wn[g] + a; }                       // FossilSimple.cpp:199-203
                    else if (c != pos[g]) { cpz = wp[g] + a; cnz = c != neg[g] ? wn[g] + a : (T)0; }
                    else { cpz = 0; cnz = wn[g] + a; }
                    cpz = (cpz - wp[g]) * deri[g]; cnz = (cnz - wn[g]) * deri[g];
#pragma unroll
                    for (int j = 0; j < EPL; j++) d[j] += cpz * vp[g][j] - cnz * vn[g][j];
                }
#pragma unroll
                for (int j = 0; j < EPL; j++) d[j] = lrD[j] + lr * d[j] - lrreg * x[j];
                return true;
            }, no_after);
            // Fossil.cpp:249-252
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (g >= ng) continue;
                T* Vp = V + (int64_t)pos[g] * K; T* Vn = V + (int64_t)neg[g] * K;
                const bool ap = at_item && !hp[g], an = at_item && !hn[g];
#pragma unroll
                for (int j = 0; j < EPL; j++) {
                    const int k = lane + 64 * j;
                    if (k < K) {
                        scat(Vp + k, lr * (deri[g] * sp[g][j] - lam * vp[g][j]), vp[g][j], ap);
                        scat(Vn + k, lr * (-deri[g] * sn[g][j] - lam * vn[g][j]), vn[g][j], an);
                    }
                }
            }
        } else {
            // ---------------- REBUS, REBUS.cpp:282-444; tuple (u, t-1, neg, -): the history of tuple g is seq[0, tpos[g])
            T* beta = (T*)m.tab[RB_BETA]; T* P = (T*)m.tab[RB_P];
            const int32_t* kept = m.kept_pair + 2 * so;   // the positions a REBUS walk keeps (first occurrences), as {item, position}
            int hp[G], hn[G];                               // hot flags of pos / neg, fetched with their rows
            int tpos[G], pos[G], neg[G], np[G], pathl[G];   // pathl / etal: lane r holds the r-th item of tuple g's matched path and its weight
            T etal[G];
            T pp[G][EPL], pn[G][EPL], sp[G][EPL], sn[G][EPL];
            int tmax = 0;
            // Every load of the set-up is unconditional (the slots past the group's end repeat tuple 0, lanes past K read
            // the last component) and the loads of one dependency level are issued together for all G tuples: a branch,
            // or a wave-uniform value taken out of a loaded register, between two loads makes the second wait for the first.
            uint32_t kc[EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; kc[j] = (uint32_t)(k < K ? k : K - 1); }
            // (a zero the compiler cannot see through: added to a wave-uniform index it keeps the loaded value in a vector
            // register until it is asked for, instead of a copy to a scalar register - and the wait that needs - at the load)
            int vz = 0; asm volatile("" : "+v"(vz));
            const int32_t* rank = m.kept_rank + so;
            int pos_r[G], kept_r[G]; int64_t po_r[G], pe_r[G];
#pragma unroll
            for (int g = 0; g < G; g++) {   // level 2: what the tuples and the sequence say
                tpos[g] = uni(tg[g].y) + 1;
                neg[g] = uni(tg[g].z);
                tmax = tpos[g] > tmax ? tpos[g] : tmax;
                const int64_t ctx = so + 2 * (int64_t)u + (tpos[g] - 1);
                pos_r[g] = seq[tpos[g] + vz]; kept_r[g] = rank[tpos[g] + vz]; po_r[g] = m.path_off[ctx + vz]; pe_r[g] = m.path_off[ctx + 1 + vz];
            }
            const int nkept = uni(rank[tmax + vz]);   // the rows of the walk: first occurrences among the positions [0, tmax)
            HistoryWalk<RIF, EPL, true, T> walk;
            walk.begin(P, kept, m.hot, K, lane, nkept);   // (its first loads travel with the tuples' own rows)
            // level 3: the tuples' own rows, their matched paths, the Root row (REBUS.cpp:414-418: it stands for an empty path)
            int cp[G], cn[G];
            T bp[G], bn[G], proot[EPL];     // bp / bn: the biases as the group found them
#pragma unroll
            for (int j = 0; j < EPL; j++) proot[j] = P[(int64_t)I * K + kc[j]];
#pragma unroll
            for (int g = 0; g < G; g++) {
                pos[g] = uni(pos_r[g]);
                cp[g] = cn[g] = uni(kept_r[g]);
                const int64_t po = uni(po_r[g]);
                np[g] = uni((int)(pe_r[g] - po));
                pathl[g] = m.path_item[po + (lane < np[g] ? lane : 0)];
                etal[g] = (T)m.eta[(int64_t)np[g] * L + (lane < L ? lane : 0)];
                hp[g] = m.hot[pos[g] + vz]; hn[g] = m.hot[neg[g] + vz];
                bp[g] = beta[pos[g]]; bn[g] = beta[neg[g]];
#pragma unroll
                for (int j = 0; j < EPL; j++) { pp[g][j] = P[(int64_t)pos[g] * K + kc[j]]; pn[g][j] = P[(int64_t)neg[g] * K + kc[j]]; }
            }
            walk.first_rows();
            __builtin_amdgcn_sched_barrier(0);   // (the instruction scheduler would move what follows up between the loads, tuple by tuple)
            // level 4: the rows of the matched paths (used after pass 1), and the weight of a sum of c rows, c^alpha, from the
            // table of all lengths: lane 2g + i fetches the entry of (rows of tuple g's prefix) - i, i = 1 in case the positive /
            // the negative turns out to be one of them
            int kl = 0;
#pragma unroll
            for (int g = 0; g < G; g++) kl = (lane >> 1) == g ? cp[g] : kl;
            const double lenw = m.len_pow[kl - (lane & 1) > 0 ? kl - (lane & 1) : 0];
            constexpr int LR = 3;   // path rows per tuple held in registers; longer paths fetch the rest in place
            T px[G][LR][EPL];
#pragma unroll
            for (int g = 0; g < G; g++) {
                pathl[g] = lane < np[g] ? pathl[g] : -2; etal[g] = lane < np[g] ? etal[g] : (T)0;
#pragma unroll
                for (int r = 0; r < LR; r++) {
                    const int cl = uni(__shfl(pathl[g], r, 64));   // -2 past the path's end: row 0 is read then, and not used
                    const int c = cl > 0 ? cl : 0;
#pragma unroll
                    for (int j = 0; j < EPL; j++) px[g][r][j] = P[(int64_t)c * K + kc[j]];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // what the slots past the group's end and the lanes past K must not contribute
#pragma unroll
            for (int g = 0; g < G; g++) {
                hp[g] = uni(hp[g]); hn[g] = uni(hn[g]);
                if (g >= ng) { tpos[g] = 0; pos[g] = neg[g] = -1; np[g] = 0; cp[g] = cn[g] = 0; pathl[g] = -2; etal[g] = 0; hp[g] = hn[g] = 0; }
                if (!(at_item || at_bias)) hp[g] = hn[g] = 0;
#pragma unroll
                for (int j = 0; j < EPL; j++)
                    if (lane + 64 * j >= K) { pp[g][j] = pn[g][j] = 0; for (int r = 0; r < LR; r++) px[g][r][j] = 0; }
            }
            // pass 1.  Tuple g sums the kept rows of positions [0, tpos[g]) except its positive's and its negative's: ONE
            // running sum over the walk, copied for tuple g where the walk passes tpos[g]; the two rows are taken out
            // afterwards (they are in registers already), what each position's id says about each tuple (inside its
            // prefix? its positive? its negative?) is asked once per round, lane q of `ids` against wave-uniform values.
            // The sums are kept in fp64 whatever the tables' type: a history of thousands of items summed in f32 is off
            // by ~1e-4 of the sum, which the two squared distances then turn into an error of that size in x_uij.
            double S[EPL], snap[G][EPL];
#pragma unroll
            for (int j = 0; j < EPL; j++) {
                S[j] = 0;
#pragma unroll
                for (int g = 0; g < G; g++) snap[g][j] = 0;
            }
            unsigned pos_in = 0, neg_in = 0;
            auto no_special = [](int, int, const T (&)[EPL], T (&)[EPL]) { return false; };
            auto no_after = []() {};
            const T none[EPL] = {}, zero = 0;
            unsigned ended = 0;   // tuples whose prefix has ended: their copy of the sum is taken
            walk.template pass<false>(false, none, zero,
                [&](int, int ids, int hpos, const int (&it)[RIF], const T (&x)[RIF][EPL]) {
                const uint64_t all = ballot64(ids >= 0);
                unsigned ending = 0, inq[G];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const bool in = ids >= 0 && hpos < tpos[g];
                    const uint64_t ing = ballot64(in);
                    if (ballot64(in && ids == pos[g]) != 0) pos_in |= 1u << g;
                    if (ballot64(in && ids == neg[g]) != 0) neg_in |= 1u << g;
                    inq[g] = (unsigned)ing;
                    if (ing != all && !((ended >> g) & 1)) ending |= 1u << g;
                }
                if (ending) {   // a prefix ends inside this round (or ended with the round before)
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        if (!((ending >> g) & 1)) continue;
#pragma unroll
                        for (int j = 0; j < EPL; j++) snap[g][j] = S[j];
#pragma unroll
                        for (int q = 0; q < RIF; q++) {
                            if (!((inq[g] >> q) & 1)) continue;
#pragma unroll
                            for (int j = 0; j < EPL; j++) snap[g][j] += (double)x[q][j];
                        }
                    }
                    ended |= ending;
                }
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    if (it[q] < 0) continue;
#pragma unroll
                    for (int j = 0; j < EPL; j++) S[j] += (double)x[q][j];
                }
                return 0u;
            }, no_special, no_after);
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (!((ended >> g) & 1)) {   // (a prefix that ends where the walk does)
#pragma unroll
                    for (int j = 0; j < EPL; j++) snap[g][j] = S[j];
                }
                cp[g] -= (int)((pos_in >> g) & 1); cn[g] -= (int)((neg_in >> g) & 1);
#pragma unroll
                for (int j = 0; j < EPL; j++) if (lane + 64 * j >= K) snap[g][j] = 0;
            }
            T wp[G], wn[G], deri[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const double wpd = lane_double(lenw, 2 * g + (int)((pos_in >> g) & 1)), wnd = lane_double(lenw, 2 * g + (int)((neg_in >> g) & 1));
                wp[g] = (T)wpd; wn[g] = (T)wnd;
                if (g >= ng) { wp[g] = wn[g] = 0; deri[g] = 0; continue; }
                T sl[EPL];
#pragma unroll
                for (int j = 0; j < EPL; j++) sl[j] = 0;
#pragma unroll
                for (int r = 0; r < LR; r++) {
                    if (r >= np[g]) continue;
                    const T e = __shfl(etal[g], r, 64);
#pragma unroll
                    for (int j = 0; j < EPL; j++) sl[j] += e * px[g][r][j];
                }
                for (int r = LR; r < np[g]; r++) {
                    const int c = uni(__shfl(pathl[g], r, 64)); const T e = __shfl(etal[g], r, 64);
#pragma unroll
                    for (int j = 0; j < EPL; j++) { const int k = lane + 64 * j; sl[j] += e * (k < K ? P[(int64_t)c * K + k] : (T)0); }
                }
                // sp / sn become the two distance vectors from here on.  Their squared lengths grow with the history (|S|^0.8
                // at alpha = -0.6) and x_uij is their difference: formed and subtracted in fp64, or a history of thousands of
                // items costs f32 tables 1e-4 in x_uij
                double qd = 0;
#pragma unroll
                for (int j = 0; j < EPL; j++) {
                    const double spd = snap[g][j] - (((pos_in >> g) & 1) ? (double)pp[g][j] : 0.0), snd = snap[g][j] - (((neg_in >> g) & 1) ? (double)pn[g][j] : 0.0);
                    const double dp = wpd * spd + (double)sl[j] - (double)pp[g][j], dn = wnd * snd + (double)sl[j] - (double)pn[g][j];
                    sp[g][j] = (T)dp; sn[g][j] = (T)dn;
                    qd += dn * dn - dp * dp;
                }
                const T x_uij = bn[g] - bp[g] + (T)group_sum<64>(qd);
                deri[g] = sigm(x_uij);
            }
            // (no store before every tuple of the group is computed: the wait for a later tuple's loaded values would wait for it)
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (g >= ng || lane != 0) continue;
                scat(beta + pos[g], lr * (-deri[g] - breg * bp[g]), bp[g], at_bias && !hp[g]);   // [9]
                scat(beta + neg[g], lr * (deri[g] - breg * bn[g]), bn[g], at_bias && !hn[g]);    // [10]
            }
            // pass 2.  A kept row of position h that is none of the group's positives, negatives or path items receives
            // deri * (wn * dist_neg - wp * dist_pos) from every tuple whose prefix holds h, and its own regulariser term
            // once per such tuple: with that sum formed once per stretch of positions between two prefix ends, one
            // multiply-add per row.  The other rows (found by comparing the row's id with the group's list of such
            // items, one per lane) go through the per-tuple case analysis.
            const bool listed = G * (L + 2) <= 64;
            const int spec = special_items<G>(pathl, pos, neg, L, lane);
            T lrD[EPL], lrreg = 0; int stretch_end = 0;   // the common step of positions [.., stretch_end)
            auto begin_stretch = [&](int h) {
                int nact = 0; stretch_end = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < EPL; j++) lrD[j] = 0;
#pragma unroll
                for (int g = 0; g < G; g++) {
                    if (h >= tpos[g]) continue;
                    nact++; stretch_end = tpos[g] < stretch_end ? tpos[g] : stretch_end;
#pragma unroll
                    for (int j = 0; j < EPL; j++) lrD[j] += deri[g] * (wn[g] * sn[g][j] - wp[g] * sp[g][j]);
                }
#pragma unroll
                for (int j = 0; j < EPL; j++) lrD[j] *= lr;
                lrreg = lr * lam * (T)nact;
            };
            begin_stretch(0);
            int resume_at = -1;   // >= 0: a prefix ended in the round just done, the common step is formed anew for positions from here
            walk.template pass<true>(at_item, lrD, lrreg,
                [&](int, int ids, int hpos, const int (&it)[RIF], const T (&)[RIF][EPL]) {
                const uint64_t valid = ballot64(ids >= 0), inside = ballot64(ids >= 0 && hpos < stretch_end);
                unsigned common = 0, sm = 0;
#pragma unroll
                for (int q = 0; q < RIF; q++) {
                    const int c = it[q];
                    if (c < 0) continue;
                    // the rows past a prefix end go through the case analysis, like the rows of the tuples' own items
                    if (!listed || !((inside >> q) & 1) || ballot64(spec == c) != 0) sm |= 1u << q; else common |= 1u << q;
                }
                resume_at = inside != valid ? __builtin_amdgcn_readlane(hpos, 63 - __builtin_clzll(valid)) + 1 : -1;
                return common | (sm << 16);
            },
                [&](int h, int c, const T (&x)[EPL], T (&d)[EPL]) {
                int nact = 0;
#pragma unroll
                for (int g = 0; g < G; g++) {
                    if (h >= tpos[g]) continue;
                    uint64_t hit = ballot64(pathl[g] == c);
                    const bool inpath = hit != 0;
                    T se = 0;
                    while (hit) { se += __shfl(etal[g], (int)__ffsll((unsigned long long)hit) - 1, 64); hit &= hit - 1; }
                    T a, b;   // delta_g = deri * (a * dist_pos + b * dist_neg) - lambda * P_c
                    if (inpath) {
                        if (c != pos[g]) { if (c == neg[g]) continue; a = -(wp[g] + se); b = wn[g] + se; }   // [1]
                        else { a = 1 - se; b = wn[g] + se; }                                                // [4]
                    } else {
                        if (c != pos[g]) { a = -wp[g]; b = c != neg[g] ? wn[g] : (T)-1; }                    // [2] / [7]
                        else { a = 1; b = wn[g]; }                                                          // [5]
                    }
                    a *= deri[g]; b *= deri[g]; nact++;
#pragma unroll
                    for (int j = 0; j < EPL; j++) d[j] += a * sp[g][j] + b * sn[g][j];
                }
                if (nact == 0) return false;
                const T reg = lam * (T)nact;
#pragma unroll
                for (int j = 0; j < EPL; j++) d[j] = lr * (d[j] - reg * x[j]);
                return true;
            },
                [&]() { if (resume_at >= 0) begin_stretch(resume_at); });
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (g >= ng) continue;
                const bool root = __shfl(pathl[g], 0, 64) == I;   // empty path -> the Root row, REBUS.cpp:414-418
                const bool ap = at_item && !hp[g], an = at_item && !hn[g];
                const bool pin = (pos_in >> g) & 1, nin = (neg_in >> g) & 1;
#pragma unroll
                for (int j = 0; j < EPL; j++) {
                    const int k = lane + 64 * j;
                    if (k >= K) continue;
                    if (root) scat(P + (int64_t)I * K + k, lr * (deri[g] * (-sp[g][j] + sn[g][j]) - lam * proot[j]), proot[j], at_item);   // [3]
                    if (!nin) scat(P + (int64_t)neg[g] * K + k, lr * (deri[g] * (-sn[g][j]) - lam * pn[g][j]), pn[g][j], an);       // [8]
                    if (!pin) scat(P + (int64_t)pos[g] * K + k, lr * (deri[g] * (sp[g][j]) - lam * pp[g][j]), pp[g][j], ap);        // [6]
                }
            }
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

template <int MODEL, class T, int EPL, int G>
int launch_g(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, int64_t max_in_flight, hipStream_t st)
{
    // one group per wavefront at a time: G tuples in flight per wavefront
    int64_t waves = (n + G - 1) / G;
    if (max_in_flight > 0 && waves > (max_in_flight + G - 1) / G) waves = (max_in_flight + G - 1) / G;
    int64_t blocks = (waves + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if ((n + G - 1) / G > 0x7fffffff - 64 * blocks) { rec_set_error("too many tuples in one launch"); return -1; }
    HIP_TRY(hipMemsetAsync(m.work_counter, 0, sizeof(int), st));
    // groups per claim: few enough claims that the counter is not the bottleneck (atomics on one address retire about
    // 16 ns apart: measured 62.7 ms per 4M single-tuple claims), enough claims per wavefront that the launch ends evenly
    const int64_t per_wave = (n + G - 1) / G / (16 * 4 * blocks);
    const int claim = (int)std::min<int64_t>(std::max<int64_t>(per_wave, 1), G >= 8 ? 1 : 8 / G);
    hipLaunchKernelGGL((sgd_history_batch_kernel<MODEL, T, EPL, G>), dim3((unsigned)blocks), dim3(256), 0, st, m, tuples, n, (T)lr, atomic, claim);
    HIP_TRY(hipGetLastError());
    return 0;
}

// the widest group the registers hold at this K that divides the sampler's batch
template <int MODEL, class T>
int launch_k(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, int64_t max_in_flight, int batch, hipStream_t st)
{
    if (batch == 1) {
        if (m.K <= 64) return launch_g<MODEL, T, 1, 1>(m, tuples, n, lr, atomic, max_in_flight, st);
        if (m.K <= 128) return launch_g<MODEL, T, 2, 1>(m, tuples, n, lr, atomic, max_in_flight, st);
        return launch_g<MODEL, T, 4, 1>(m, tuples, n, lr, atomic, max_in_flight, st);
    }
    if (m.K <= 64) {
        if (batch % 8 == 0) return launch_g<MODEL, T, 1, 8>(m, tuples, n, lr, atomic, max_in_flight, st);
        if (batch % 4 == 0) return launch_g<MODEL, T, 1, 4>(m, tuples, n, lr, atomic, max_in_flight, st);
        return launch_g<MODEL, T, 1, 2>(m, tuples, n, lr, atomic, max_in_flight, st);
    }
    if (m.K <= 128) {
        if (batch % 4 == 0) return launch_g<MODEL, T, 2, 4>(m, tuples, n, lr, atomic, max_in_flight, st);
        return launch_g<MODEL, T, 2, 2>(m, tuples, n, lr, atomic, max_in_flight, st);
    }
    return launch_g<MODEL, T, 4, 2>(m, tuples, n, lr, atomic, max_in_flight, st);
}

template <class T>
int dispatch(const DevModel& m, const int4* tuples, int64_t n, double lr, int atomic, int64_t max_in_flight, int batch, hipStream_t st)
{
    switch (m.model) {
    case REC_FOSSIL: return launch_k<REC_FOSSIL, T>(m, tuples, n, lr, atomic, max_in_flight, batch, st);
    case REC_FOSSIL_SIMPLE: return launch_k<REC_FOSSIL_SIMPLE, T>(m, tuples, n, lr, atomic, max_in_flight, batch, st);
    case REC_REBUS: return launch_k<REC_REBUS, T>(m, tuples, n, lr, atomic, max_in_flight, batch, st);
    default: rec_set_error("not a history model"); return -1;
    }
}
}  // namespace

int launch_sgd_history(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, int atomic, int64_t max_in_flight,
                               int batch, hipStream_t st)
{
    if (n <= 0) return 0;
    if (batch != 1 && batch != 2 && batch != 4 && batch != 8) { rec_set_error("user_batch must be 1, 2, 4 or 8"); return -1; }
    return precision == REC_F64 ? dispatch<double>(m, tuples, n, lr, atomic, max_in_flight, batch, st)
                                : dispatch<float>(m, tuples, n, lr, atomic, max_in_flight, batch, st);
}

int launch_apply_global_history(const DevModel& m, int precision, double lr, double cap, hipStream_t st)
{
    if (!(m.model == REC_FOSSIL || m.model == REC_FOSSIL_SIMPLE)) return 0;
    if (precision == REC_F64) hipLaunchKernelGGL(apply_global_history_kernel<double>, dim3(1), dim3(64), 0, st, m, lr, cap);
    else hipLaunchKernelGGL(apply_global_history_kernel<float>, dim3(1), dim3(64), 0, st, m, (float)lr, (float)cap);
    HIP_TRY(hipGetLastError());
    return 0;
}
