// Shared between the translation units of librecsys_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/recsys_engine.h"

#define REC_MAX_TABLES 6
#define REC_MAX_EPL 4   // elements per lane: K <= 64 * REC_MAX_EPL
#define REC_MAX_K (64 * REC_MAX_EPL)
#define REC_MAX_L 32

// Everything a kernel needs to know about the model and the corpus; passed by value.
struct DevModel {
    int model, U, I, K, KK, L;
    int user_begin, user_end;           // the engine's own users: only their rows, sequences, click sets and paths are resident
    double lambda, bias_reg, relation_reg, alpha, df;
    void* tab[REC_MAX_TABLES];          // device tables in the reference's flat order, rows contiguous
    const int64_t* seq_off;             // [U+1] training sequences
    const int32_t* seq_item;
    const int32_t* test_item;           // [U]
    const int32_t* val_item;            // [U]
    const int64_t* clk_off;             // [U+1] sorted distinct training items per user
    const int32_t* clk_item;
    const uint8_t* first_occ;           // [E] 1 where a sequence position is the first occurrence of its item
    const double* len_pow;              // history models: [longest sequence + 2] c^alpha (REBUS) / c^-0.2 (Fossil, FossilSimple), 0 at c = 0
    const int32_t* kept_rank;           // REBUS: [E] first occurrences among the positions of the user's sequence before this one
    const int32_t* kept_pair;           // REBUS: [E][2] per user the {item, position in the sequence} of those positions in order, then {0, INT_MAX}
    const int64_t* path_off;            // REBUS matched paths per context
    const int32_t* path_item;
    const double* eta;                  // (L+1) x L damping weights
    const int32_t* elig;                // users (MC: source items) the sampler may draw
    int n_elig;
    int elig_identity;                  // 1: elig[i] == i for every i
    const int64_t* succ_off;            // MC: successors of each item with multiplicity ...
    const int32_t* succ_item;
    const int64_t* sset_off;            // ... and as sorted distinct sets
    const int32_t* sset_item;
    int* work_counter;                  // next unclaimed tuple group of the running history launch
    int pair_min_hist;                  // history kernels: a group of tuples shares one walk only when its user has at least this many training clicks
    void* gacc;                         // accumulators for rows every tuple touches (TransRec r, Fossil WT, FossilSimple alpha)
    const int32_t* eval_users;          // users of this engine's shard that have a test item
    int n_eval;
    // [U or I][16] one 64-byte record per sampling owner (user; MC: source item), so that a draw touches one line for all
    // of it: words 0-1 the owner's offset in seq_item (MC: succ_item), word 2 its length there, words 4-15 a 384-bit
    // membership filter of the set a drawn negative must avoid (clicked items; MC: successors; all clear when unused)
    const uint32_t* owner_rec;
    const int4* ev;                     // [n_ev] sampling events ordered by positive item: (owner, aux, positive, -), see launch_event_sampler
    const float* ev_w;                  // [n_ev] 1 / (number of events of the owner)
    int64_t n_ev;
    // pairs of steps that share their positive item: the events of item i are ev[ev_off[i], ev_off[i+1]); slot k of that
    // range is taken with probability ev_alias_p[k], else slot ev_alias_i[k] (Walker's alias method over ev_w)
    const int64_t* ev_off;              // [I+1]
    const float* ev_alias_p;            // [n_ev]
    const int32_t* ev_alias_i;          // [n_ev] slot inside the item's range
    const int32_t* train_count;         // [I] occurrences of each item in the training sequences (MostPopular)
    const int32_t* pop_order;           // [I] the items by descending train_count (ties by item id): the column order of the top-k sweep
    // [I+1] 0 = ordinary item, 1 = a "hot" item (see rec_set_option "hot_count"): its rows get plain stores where
    // the others get float atomics
    const uint8_t* hot;
};

// table indices, in the flat order of each model's getParametersFromVector / parametersToFlatVector
enum { BPRMF_BETA = 0, BPRMF_GU = 1, BPRMF_GI = 2 };
enum { MC_GI = 0, MC_ETA = 1 };
enum { FPMC_GU = 0, FPMC_GI = 1, FPMC_KAPPA = 2, FPMC_ETA = 3 };
enum { PRME_GU = 0, PRME_GI = 1, PRME_ETA = 2 };
enum { HRM_U = 0, HRM_V = 1 };
enum { TR_BETA = 0, TR_H = 1, TR_R = 2, TR_r = 3 };
enum { FS_ALPHA = 0, FS_ALPHA_U = 1, FS_BETA = 2, FS_UI = 3, FS_V = 4 };
enum { FO_BETA = 0, FO_WT = 1, FO_WTU = 2, FO_UI = 3, FO_V = 4 };
enum { RB_BETA = 0, RB_P = 1 };

void rec_set_error(const std::string& msg);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            rec_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)

// ---- launchers implemented in the kernel translation units (all enqueue on `st`) ----
int launch_sgd_sequential(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, hipStream_t st);
int launch_sgd_hogwild(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, int atomic,
                       hipStream_t st);
// Owner mode of the Hogwild kernel: `partitioned` holds the launch's tuples ordered by the slice of the catalog their
// negative item lies in (launch_partition_by_slice), qoff[s] .. qoff[s+1] being slice s; workgroup s applies them and
// keeps the negatives' deltas in LDS accumulators, moved to memory once per flush_at additions.
bool sgd_owner_supported(int model);
int sgd_owner_slices();
int sgd_owner_slice_rows(const DevModel& m);
int sgd_owner_neg_field(int model);   // component of the tuple that holds the negative item (2 = z, 3 = w)
int launch_sgd_owner(const DevModel& m, int precision, const int4* partitioned, const int64_t* qoff, int64_t n, double lr, int atomic,
                     int flush_at, hipStream_t st);
// The wide form of the Hogwild kernel (sgd_wide.hip): f32 tables, K and KK multiples of 4; qoff as for launch_sgd_owner
// (nullptr: plain mode on the tuples as they come)
bool sgd_wide_supported(const DevModel& m, int precision);
int sgd_wide_owner_slices(const DevModel& m);
int launch_sgd_wide(const DevModel& m, const int4* tuples, const int64_t* qoff, int slices, int64_t n, double lr, int atomic, int flush_at, hipStream_t st);
// the same kernel drawing its own tuples: tuple i of the launch is step first_step + i of (seed, epoch), what launch_sampler
// would have written to tuples[i] (independent draws: no shared users, negatives or positives)
int launch_sgd_wide_sampled(const DevModel& m, int64_t n, int64_t first_step, uint64_t seed, uint64_t epoch, double lr, int atomic, hipStream_t st);
// Stable-by-block partition of n tuples by (component `field`) / slice_rows into `slices` ranges of dst; qoff[slices+1]
// (device) receives the range starts; slices = 256, 512 or 1024.  scratch: partition_scratch_bytes(n, slices).
size_t partition_scratch_bytes(int64_t n, int slices);
int launch_partition_by_slice(const int4* src, int4* dst, int64_t n, int field, int slice_rows, int slices, void* scratch,
                              int64_t* qoff, hipStream_t st, int tile = 0);
// TransRec after a launch: the gradient sum of the global translation r folded in, and (project) every item row back onto
// the unit ball (TransRec.cpp:245-258); launch_project_rows: the second part alone
int launch_apply_global(const DevModel& m, int precision, double lr, int64_t n_tuples, double cap, hipStream_t st, bool project = true);
int launch_project_rows(const DevModel& m, int precision, hipStream_t st);
// max_in_flight: upper bound on the tuples being worked on at the same time (0 = as many as the chip holds).
// batch (1, 2, 4, 8): the tuples come in aligned groups of `batch` consecutive tuples of one user; one wavefront
// walks a group's shared history once.
int launch_sgd_history(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, int atomic, int64_t max_in_flight,
                       int batch, hipStream_t st);
int launch_apply_global_history(const DevModel& m, int precision, double lr, double cap, hipStream_t st);
// user_batch > 1: every aligned group of user_batch consecutive steps shares its owner (user) draw;
// neg_share > 1: every aligned group of neg_share consecutive steps shares its stream of negative candidates
int launch_sampler(const DevModel& m, int4* tuples, int64_t n, int64_t first_step, uint64_t seed, uint64_t epoch,
                   hipStream_t st, int user_batch = 1, int neg_share = 1);
int launch_convert_in(const double* src, void* dst, int64_t n, int precision, hipStream_t st);
int launch_convert_out(const void* src, double* dst, int64_t n, int precision, hipStream_t st);
int launch_param_init(const DevModel& m, int precision, uint64_t seed, hipStream_t st);
// Tuples grouped by positive item and the update kernel that turns each run of equal positives into one
// scatter (models without a row shared by all tuples and without a read-time projection).
bool sgd_runs_supported(int model);
// How long a run of equal positives stays private before it is scattered: `shared` for an item whose run
// continues in a neighbouring segment, `hot` for an item that fills whole segments (scattered with atomics if
// hot_atomic, else with plain stores).
struct RunPolicy { int shared, hot, hot_atomic; };
// segment: batches walked by one lane group (0 = chosen from n)
int launch_sgd_runs(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, int atomic, int64_t segment,
                    RunPolicy rp, hipStream_t st);
// Event-major sampling: every sampling event (an owner's position) independently yields Poisson(n_expected *
// P(event)) tuples, written in event order, i.e. grouped by positive item.  *n_out_dev (device, int64) receives
// the number of tuples drawn (its mean is n_expected); at most cap are written.  scratch: event_scratch_bytes().
// Steps 2p and 2p+1 of the launch as a pair: the first is drawn as launch_sampler draws it, the second is another
// event of the first one's positive item, drawn with the events' own weights (so that its (owner, position) has the
// same distribution as the first one's), and both take their negative from one stream of candidates.
int launch_pair_sampler(const DevModel& m, int4* tuples, int64_t n, int64_t first_step, uint64_t seed, uint64_t epoch, hipStream_t st);
size_t event_scratch_bytes(int64_t n_ev);
int launch_event_sampler(const DevModel& m, int4* tuples, int64_t cap, int64_t n_expected, int64_t first_step, uint64_t chunk_no,
                         uint64_t seed, uint64_t epoch, void* scratch, int64_t* n_out_dev, hipStream_t st);


// Evaluation of the users in m.eval_users against the whole catalog (n_subset = 0) or against the n_subset items
// listed in subset_host (host memory, repetitions allowed: every occurrence is one scored candidate).
// counts_host: [U][3] (count_val, count_test, max; -1 for users not evaluated); xpos_host (optional): [U][2]
// (x_test, x_val).
// user_draws > 0: instead of a sweep, every user is scored against user_draws candidate draws of its own (keyed by
// draw_seed, the user and the draw index; draws_host, optional, [U][user_draws] receives them, -1 for users not evaluated).
// f32 screen of the full-catalog sweep (f32 tables): on (1) / off (0) / unchanged (-1); columns a query may leave to fp64 (cap > 0)
void eval_configure_screen(int on, int cap);
void eval_configure_ranges(int ranges);   // ranges of the catalog per query block in the screen (0 = chosen from the number of query blocks)
void eval_screen_stats(int* mode, int64_t* open);   // of the last full-catalog sweep: 0 fp64 / 1 screen / 2 screen overflowed; undecided columns ("profile" on)
int eval_run(const DevModel& m, int precision, int32_t* counts_host, double* xpos_host, hipStream_t st,
             double* score_ms, int64_t* score_launches, bool profile, const int32_t* subset_host, int n_subset,
             int user_draws = 0, uint64_t draw_seed = 0, int32_t* draws_host = nullptr);

// The k best-scored items of the validation and of the test context of every listed user:
// items_host / scores_host are [n_users][2][k] (context 0 = validation, 1 = test; -1 / -inf past the end).
// users_host == nullptr: the first n_users users of m.eval_users; scores_host may be nullptr.
int eval_topk(const DevModel& m, int precision, const int32_t* users_host, int n_users, int k, int32_t* items_host,
              double* scores_host, hipStream_t st);
void eval_configure_topk(int fused);                  // 1 (default): lists come out of the f16 sweep where it applies; 0: always the score-matrix path
void eval_topk_stats(int64_t* lists, double* ms);     // of the last eval_topk: lists the fused path produced, and its device time

#ifdef __HIPCC__
// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float rexp(float x) { return expf(x); }
__device__ __forceinline__ double rexp(double x) { return exp(x); }
__device__ __forceinline__ float rpow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double rpow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float rsqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ double rsqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float rabs(float x) { return fabsf(x); }
__device__ __forceinline__ double rabs(double x) { return fabs(x); }

// value of the lane a DPP control selects inside each row of 16 lanes: quad_perm [1,0,3,2] = 0xB1 (lane ^ 1),
// quad_perm [2,3,0,1] = 0x4E (lane ^ 2), row_half_mirror = 0x141 (7 - lane inside 8), row_mirror = 0x140 (15 - lane)
template <int CTRL> __device__ __forceinline__ float dpp_move(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_move(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// sum over the lanes of an aligned group of W lanes (W = 2 .. 64); every lane gets the sum.  Inside a row of 16 lanes
// the partners are reached by DPP modifiers of the additions themselves (no LDS crossbar, no wait): after the steps
// lane ^ 1 and lane ^ 2 a quad is uniform, so mirroring inside 8 and inside 16 lanes reaches the other half each time.
template <int W, class T>
__device__ __forceinline__ T group_sum_dpp(T v)
{
    if constexpr (W >= 2) v += dpp_move<0xB1>(v);
    if constexpr (W >= 4) v += dpp_move<0x4E>(v);
    if constexpr (W >= 8) v += dpp_move<0x141>(v);
    if constexpr (W >= 16) v += dpp_move<0x140>(v);
    if constexpr (W >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (W >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// the same through the LDS crossbar alone (ds_bpermute), highest distance first
template <int W, class T>
__device__ __forceinline__ T group_sum(T v)
{
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Philox4x32-10 (Salmon et al. 2011): counter-based, so a tuple depends only on (seed, epoch, step)
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0; key.y += W1;
    }
    return ctr;
}

// uniform integer in [0, n) from 32 random bits (multiply-shift)
__device__ __forceinline__ uint32_t bounded(uint32_t r, uint32_t n) { return __umulhi(r, n); }

// One-hash 384-bit filter in front of a sorted set: a clear bit proves absence with one load
__host__ __device__ __forceinline__ uint32_t filter_slot(int32_t item) { return (uint32_t)(((uint64_t)((uint32_t)item * 2654435761u) * 384u) >> 32); }
__device__ __forceinline__ bool filter_may_contain(const uint32_t* __restrict__ rec, int owner, int32_t item)
{
    const uint32_t h = filter_slot(item);
    return (rec[(int64_t)owner * 16 + 4 + (h >> 5)] >> (h & 31)) & 1u;
}

// is `x` in the sorted array a[0..n) ?
__device__ __forceinline__ bool sorted_contains(const int32_t* a, int n, int32_t x)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        int32_t v = a[mid];
        if (v < x) lo = mid + 1; else hi = mid;
    }
    return lo < n && a[lo] == x;
}
// the k-th (0-based) integer of [0, ...) that is not in the sorted array a[0..n) of distinct values.  With k uniform
// in [0, I - n) it is a uniform draw from the complement of the set: what a rejection loop converges to, without
// the loop (used when that loop has run out of tries).
__device__ __forceinline__ int kth_nonmember(const int32_t* a, int n, int k)
{
    // the number of non-members below a[i] is a[i] - i (non-decreasing in i): find the first i with a[i] - i > k
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] - mid > k) hi = mid; else lo = mid + 1;
    }
    return k + lo;
}
#endif
