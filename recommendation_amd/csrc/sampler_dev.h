// One tuple of the device sampler as a function of (seed, epoch, step): shared by sampler_kernel (sampler.hip) and by the
// update kernel that draws its own tuples (sgd_wide.hip), so that both see the same tuple for the same step.
#pragma once
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

#define MAX_TRIES 4096  // every rejection loop ends after this many draws; owners whose set leaves almost nothing to
                        // draw (a user who clicked nearly the whole catalog) then take the item of uniform rank among the
                        // non-members of their sorted set - the distribution the reference's unbounded loop converges to

#define NEG_STREAM if (neg_share > 1) rng = Rng(seed ^ 0x8CB92BA72F3D8DD7ull, epoch, step / (uint64_t)neg_share)
template <int MODEL>
__device__ __forceinline__ int4 sample_tuple(const DevModel& m, uint64_t step, uint64_t seed, uint64_t epoch, int user_batch, int neg_share)
{
    Rng rng(seed, epoch, step);
    // neg_share > 1: the steps of an aligned group of neg_share draw their negative candidates from one stream, each
    // accepting the first candidate its own tuple allows - nearly always the same item for the whole group (whose
    // rows the update kernel then scatters once), while every tuple's negative stays uniform over its own allowed set
    uint32_t r_owner = rng.next();
    // user_batch > 1: steps g*user_batch .. (g+1)*user_batch-1 share one owner draw, everything else stays per step
    if (user_batch > 1) { Rng ro(seed ^ 0xD1B54A32D192ED03ull, epoch, step / (uint64_t)user_batch); r_owner = ro.next(); }
    const int pick = (int)bounded(r_owner, (uint32_t)m.n_elig);
    const int e = m.elig_identity ? pick : m.elig[pick];   // every owner eligible: the list is 0..n-1
    int4 t = make_int4(e, 0, 0, 0);
    if constexpr (MODEL == REC_MC) {
        // MC.cpp:80-89, 139-147: source item with successors, a successor, a non-successor
        const uint4 rec = *reinterpret_cast<const uint4*>(m.owner_rec + (int64_t)e * 16);
        const int64_t a = (int64_t)(((uint64_t)rec.y << 32) | rec.x); const int cnt = (int)rec.z;
        const int pos = m.succ_item[a + bounded(rng.next(), (uint32_t)cnt)];
        // the successor set is only consulted when the filter cannot rule the candidate out
        auto successor = [&](int item) {
            if (!filter_may_contain(m.owner_rec, e, item)) return false;
            const int64_t s0 = m.sset_off[e];
            return sorted_contains(m.sset_item + s0, (int)(m.sset_off[e + 1] - s0), item);
        };
        NEG_STREAM; int neg = -1;
        for (int tr = 0; tr < MAX_TRIES && neg < 0; tr++) { const int c = (int)bounded(rng.next(), (uint32_t)m.I); if (!successor(c)) neg = c; }
        if (neg < 0) { const int64_t s0 = m.sset_off[e]; const int ns = (int)(m.sset_off[e + 1] - s0); neg = kth_nonmember(m.sset_item + s0, ns, (int)bounded(rng.next(), (uint32_t)(m.I - ns))); }
        t = make_int4(e, pos, neg, 0);
    } else {
        const uint4 rec = *reinterpret_cast<const uint4*>(m.owner_rec + (int64_t)e * 16);
        const int64_t so = (int64_t)(((uint64_t)rec.y << 32) | rec.x); const int nu = (int)rec.z;
        const int32_t* seq = m.seq_item + so;
        // the clicked list is only consulted when the filter cannot rule the candidate out
        auto clicked = [&](int item) {
            if (!filter_may_contain(m.owner_rec, e, item)) return false;
            const int64_t c0 = m.clk_off[e];
            return sorted_contains(m.clk_item + c0, (int)(m.clk_off[e + 1] - c0), item);
        };
        auto unclicked_of_rank = [&](uint32_t r) {
            const int64_t c0 = m.clk_off[e]; const int nc = (int)(m.clk_off[e + 1] - c0);
            return kth_nonmember(m.clk_item + c0, nc, (int)bounded(r, (uint32_t)(m.I - nc)));
        };
        if constexpr (MODEL == REC_BPRMF) {
            // BPRMF.cpp:74-83, 135-143
            const int pos = seq[bounded(rng.next(), (uint32_t)nu)];
            NEG_STREAM; int neg = -1;
            for (int tr = 0; tr < MAX_TRIES && neg < 0; tr++) { const int c = (int)bounded(rng.next(), (uint32_t)m.I); if (!clicked(c)) neg = c; }
            if (neg < 0) neg = unclicked_of_rank(rng.next());
            t = make_int4(e, pos, neg, 0);
        } else if constexpr (MODEL == REC_FPMC || MODEL == REC_PRME || MODEL == REC_FOSSIL_SIMPLE || MODEL == REC_HRM_AVG ||
                             MODEL == REC_HRM_MAX) {
            // FPMC.cpp:140-151, PRME.cpp:120-132, FossilSimple.cpp:253-264, HRM_avg.cpp:153-159 + 171-174
            const int p = (int)bounded(rng.next(), (uint32_t)(nu - 1));
            const int prev = seq[p], pos = seq[p + 1];
            NEG_STREAM; int neg = 0;
            for (int tr = 0; tr < MAX_TRIES; tr++) {
                neg = (int)bounded(rng.next(), (uint32_t)m.I);
                if (neg != pos && !(MODEL == REC_FOSSIL_SIMPLE && neg == prev)) break;
            }
            t = make_int4(e, prev, pos, neg);
        } else if constexpr (MODEL == REC_TRANSREC || MODEL == REC_TRANSREC_L1) {
            // TransRec.cpp:188-199
            const int p = (int)bounded(rng.next(), (uint32_t)(nu - 1));
            NEG_STREAM; int neg = -1;
            for (int tr = 0; tr < MAX_TRIES && neg < 0; tr++) { const int c = (int)bounded(rng.next(), (uint32_t)m.I); if (!clicked(c)) neg = c; }
            if (neg < 0) neg = unclicked_of_rank(rng.next());
            t = make_int4(e, seq[p], seq[p + 1], neg);
        } else if constexpr (MODEL == REC_FOSSIL) {
            // Fossil.cpp:282-293: target position 1..nu-1, negative outside {pos} + the <= L previous items
            const int tp = 1 + (int)bounded(rng.next(), (uint32_t)(nu - 1));
            const int pos = seq[tp]; const int lo = tp - m.L > 0 ? tp - m.L : 0;
            NEG_STREAM; int neg = 0;
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
            NEG_STREAM; int neg = 0;
            for (int tr = 0; tr < MAX_TRIES; tr++) {
                neg = (int)bounded(rng.next(), (uint32_t)m.I);
                bool bad = neg == pos;
                for (int j = 0; j < np && !bad; j++) bad = path[j] == neg;
                if (!bad) break;
            }
            t = make_int4(e, tp - 1, neg, 0);
        }
    }
    return t;
}
#undef NEG_STREAM
