// C ABI of librecsys_amd.so: engine life cycle, parameter transfer, dispatch to the kernels.
#include "common.h"
#include <cmath>
#include <algorithm>
#include <math.h>
#include <string.h>
#include <mutex>
#include <vector>

static thread_local std::string g_err;
void rec_set_error(const std::string& msg) { g_err = msg; }
static bool history_model(int model);
extern "C" const char* rec_last_error(void) { return g_err.c_str(); }

struct TableDesc {
    int64_t host_off, rows, cols, dev_off;  // offsets in elements; dev_off: where the table's first RESIDENT row sits in W
    int is_user;
    int64_t row0, nrows;                    // resident rows [row0, row0 + nrows): all of them, or the engine's users' rows
};

struct rec_engine {
    rec_model_desc desc;
    int U = 0, I = 0;
    int64_t E = 0, nw = 0, dev_elems = 0, steps_per_epoch = 0;
    int64_t tuples_drawn = 0;  // tuples applied by rec_sgd_epoch so far
    int elem_size = 4;
    std::vector<TableDesc> tables;
    DevModel dm;
    hipStream_t st = nullptr;
    void* W = nullptr;       // device parameters (all tables)
    void* bestW = nullptr;   // copyBestModel snapshot
    double* stage = nullptr; // f64 staging for upload / download
    int64_t stage_elems = 0;
    int4* tuples = nullptr;  // two tuple buffers of `tuples_cap` each: chunk c+1 is sampled while chunk c is applied
    void* group_scratch = nullptr;  // block sums of the event-major sampler
    int64_t* n_drawn_dev = nullptr; // its tuple count
    int64_t tuples_cap = 0;
    hipStream_t st2 = nullptr;                 // sampler stream
    hipEvent_t ev_sampled[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    bool consumed_valid[2] = {false, false};
    std::vector<void*> owned;
    std::vector<int32_t> train_count_host;  // occurrences of every item in the training sequences (hot-item selection)
    std::vector<int64_t> seq_off_host;  // host copy of the training sequence offsets (replay argument checks)
    // options
    int64_t chunk = 1 << 26;
    int overlap = 0;     // 1: the sampler of chunk c+1 runs on a second stream next to the update of chunk c
    int64_t global_window = 1 << 18;
    int64_t global_cap = 16;    // a folded-in row moves by at most this many steps' worth of learning rate per launch
    int64_t stale_window = 0;   // tuples per launch so that a row rarely receives more than a couple of stale updates (0 = no limit)
    int atomic = 0x1ff;  // bit r: row r of a tuple is scattered with float atomics; bit 8: the biases
    int profile = 0;
    RunPolicy run_policy = {128, 32, 0};
    int64_t run_segment = 0;  // batches per lane group in the run-combining kernel (0 = automatic)
    int event_phases = 0; // event-major sampler: > 0 = events ordered by that many item ranges of equal weight, shuffled inside a range
    int pos_share = 1;   // 2: steps 2p, 2p+1 of an epoch also share their positive item (launch_pair_sampler)
    int neg_share = 1;   // consecutive steps of an epoch that share their stream of negative candidates (group models)
    int user_batch = 1;  // consecutive steps of an epoch that share their user draw (history models)
    int wide = 1;        // 1: the wide form of the Hogwild kernel (sgd_wide.hip) where it applies
    int fused_sampler = 1;   // 1: that kernel draws its own tuples where every step's draws are independent (no sampler launch, no tuple buffer)
    int owner = 0;       // 1: launches of >= owner_min tuples are partitioned by the negative's slice of the catalog and applied in owner mode
    int owner_flush = 8; // additions after which an LDS accumulator row is moved to memory
    int64_t owner_per_item = 8;   // owner mode: a launch holds at most this many tuples per item of the catalog
    int64_t owner_min = 1 << 20;
    int part_tile = 0;   // tuples the partition sorts in LDS at a time (0: its default)
    int owner_slices = 0;   // slices of the catalog in owner mode (0: as few as keep every accumulator in LDS)
    void* part_scratch = nullptr; size_t part_scratch_bytes = 0;
    int64_t* qoff_dev = nullptr;
    int group = 0;       // 1: run-combining update kernel on the tuples as they come; 2: also sample by event, which yields chunks grouped by positive item
    // hip-event statistics: 0 update (owner mode: partition + update kernel), 1 score, 2 sampler, 3 the owner mode's partition
    double ms[4] = {0, 0, 0, 0};
    int64_t launches[4] = {0, 0, 0, 0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[4];
    std::vector<void*> round_base;   // rec_shards_begin_round: a copy of every table that is not indexed by user
    int project_after_launch = 1;    // TransRec: every item row is projected onto the unit ball after each update launch
};

// Tuple staging of rec_sgd_replay_static: lives in the code object, so that entry point never allocates.  One copy per
// GPU the library runs on (1 MiB each, held until the process ends); its address on a device is looked up once, in
// rec_create, with that device current.  In host code the array's name is only a handle for that lookup.
#define REC_STATIC_MAX_TUPLES 65536
static __device__ __align__(256) int4 scratch[REC_STATIC_MAX_TUPLES];
#define REC_MAX_DEVICES 64
static int4* g_scratch_ptr[REC_MAX_DEVICES];   // per device: where `scratch` lives there (nullptr = not looked up yet)
static std::mutex g_scratch_mutex;

// with device `dev` current
static int lookup_scratch(int dev)
{
    if (dev < 0 || dev >= REC_MAX_DEVICES) { rec_set_error("device ordinal beyond the static scratch table"); return -1; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev]) return 0;
    void* ptr = nullptr;
    hipError_t rc = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (rc != hipSuccess || !ptr) { rec_set_error(std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(rc)); return -2; }
    g_scratch_ptr[dev] = (int4*)ptr;
    return 0;
}

static int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

#define ENG_HIP(expr)                                                                 \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            rec_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));         \
            return -2;                                                                \
        }                                                                             \
    } while (0)

template <class T>
static int upload(rec_engine* e, const std::vector<T>& h, const T** dev)
{
    void* p = nullptr;
    ENG_HIP(hipMalloc(&p, std::max<size_t>(h.size() * sizeof(T), 8)));
    e->owned.push_back(p);
    if (!h.empty()) ENG_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = (const T*)p;
    return 0;
}

// uploads h[first, first + count) and returns the address element 0 WOULD have: kernels keep indexing with global
// positions, only [first, first + count) may be touched
template <class T>
static int upload_range(rec_engine* e, const std::vector<T>& h, int64_t first, int64_t count, const T** dev)
{
    void* p = nullptr;
    ENG_HIP(hipMalloc(&p, (size_t)count * sizeof(T) + 64));   // (kernels that clamp an index instead of branching may read one element past an empty range)
    e->owned.push_back(p);
    ENG_HIP(hipMemset(p, 0, (size_t)count * sizeof(T) + 64));
    if (count > 0) ENG_HIP(hipMemcpy(p, h.data() + first, (size_t)count * sizeof(T), hipMemcpyHostToDevice));
    *dev = (const T*)p - first;
    return 0;
}

static void build_tables(rec_engine* e)
{
    const int64_t U = e->U, I = e->I, K = e->desc.K, KK = e->desc.KK, L = e->desc.L;
    std::vector<TableDesc>& t = e->tables;
    // a table indexed by user only holds the rows of the engine's own users [user_begin, user_end)
    const int64_t ub = e->desc.user_begin, ue = e->desc.user_end;
    auto add = [&](int64_t rows, int64_t cols, int is_user) { t.push_back({0, rows, cols, 0, is_user, is_user ? ub : 0, is_user ? ue - ub : rows}); };
    switch (e->desc.model) {
    case REC_BPRMF: add(I, 1, 0); add(U, K, 1); add(I, K, 0); break;                       // BPRMF.cpp:49-61
    case REC_MC: add(I, K, 0); add(I, K, 0); break;                                        // MC.cpp:59-67
    case REC_FPMC: add(U, K, 1); add(I, K, 0); add(I, KK, 0); add(I, KK, 0); break;        // FPMC.cpp:46-63
    case REC_PRME: add(U, K, 1); add(I, K, 0); add(I, KK, 0); break;                       // PRME.cpp:43-56
    case REC_HRM_AVG: case REC_HRM_MAX: add(U, K, 1); add(I, K, 0); break;                 // HRM_avg.cpp:46-56
    case REC_TRANSREC: case REC_TRANSREC_L1: add(I, 1, 0); add(I, K, 0); add(U, K, 1); add(1, K, 0); break;  // TransRec.cpp:61-77
    case REC_FOSSIL_SIMPLE: add(1, 1, 0); add(U, 1, 1); add(I, 1, 0); add(I, K, 0); add(I, K, 0); break;     // FossilSimple.cpp:56-74
    case REC_FOSSIL: add(I, 1, 0); add(1, L, 0); add(U, L, 1); add(I, K, 0); add(I, K, 0); break;            // Fossil.cpp:53-73
    case REC_REBUS: add(I + 1, 1, 0); add(I + 1, K, 0); break;                             // REBUS.cpp:185-193
    default: break;  // MP has no parameters
    }
    int64_t ho = 0, dv = 0;
    for (auto& d : t) {
        d.host_off = ho; d.dev_off = dv;
        ho += d.rows * d.cols;
        dv = align_up(dv + d.nrows * d.cols, 256 / e->elem_size);  // every table starts on a 256-byte boundary
    }
    e->nw = ho; e->dev_elems = std::max<int64_t>(dv, 64);
}

extern "C" int rec_create(const rec_corpus* c, const rec_model_desc* d, rec_engine** out)
{
    if (!c || !d || !out) { rec_set_error("null argument"); return -1; }
    if (d->model < REC_MP || d->model > REC_REBUS) { rec_set_error("unknown model"); return -1; }
    if (d->model != REC_MP && (d->K < 1 || d->K > REC_MAX_K)) { rec_set_error("K must be in 1..256"); return -1; }
    if (d->model == REC_FOSSIL && (d->L < 1 || d->L > REC_MAX_L)) { rec_set_error("L must be in 1..32"); return -1; }
    if (d->model == REC_REBUS && (!c->path_off || !c->path_item)) { rec_set_error("REBUS needs the matched paths"); return -1; }
    if (d->model == REC_REBUS && (d->L < 1 || d->L > REC_MAX_L)) { rec_set_error("L must be in 1..32"); return -1; }
    int ndev = 0;
    ENG_HIP(hipGetDeviceCount(&ndev));
    if (d->device < 0 || d->device >= ndev) { rec_set_error("no such HIP device"); return -1; }
    ENG_HIP(hipSetDevice(d->device));
    if (int rc = lookup_scratch(d->device)) return rc;

    rec_engine* e = new rec_engine();
    e->desc = *d;
    if (e->desc.KK <= 0) e->desc.KK = e->desc.K;
    if (e->desc.L <= 0) e->desc.L = 1;
    if (e->desc.dataset_factor <= 0) e->desc.dataset_factor = 10;
    e->U = c->n_users; e->I = c->n_items; e->E = c->seq_off[c->n_users];
    e->elem_size = d->precision == REC_F64 ? 8 : 4;
    int ub = d->user_begin, ue = d->user_end;
    if (ub == 0 && ue == 0) ue = e->U;
    if (ub < 0 || ue > e->U || ub > ue) { delete e; rec_set_error("bad user range"); return -1; }
    e->desc.user_begin = ub; e->desc.user_end = ue;
    build_tables(e);

    const int U = e->U, I = e->I; const int64_t E = e->E;
    const int L = e->desc.L;
    // ---- host-side derived structures
    std::vector<int64_t> seq_off(c->seq_off, c->seq_off + U + 1);
    e->seq_off_host = seq_off;
    std::vector<int32_t> seq_item(c->seq_item, c->seq_item + E);
    std::vector<int32_t> test_item(c->test_item, c->test_item + U), val_item(c->val_item, c->val_item + U);
    for (int64_t p = 0; p < E; p++)
        if (seq_item[p] < 0 || seq_item[p] >= I) { delete e; rec_set_error("item id out of range in seq_item"); return -1; }
    // clicked_per_user (model.hpp:42-44) as sorted distinct arrays, and first-occurrence flags
    std::vector<int64_t> clk_off(U + 1, 0);
    std::vector<int32_t> clk_item; clk_item.reserve(E);
    std::vector<uint8_t> first_occ(E, 0);
    std::vector<int32_t> train_count(I, 0);
    {
        std::vector<std::pair<int32_t, int64_t>> tmp;
        for (int u = 0; u < U; u++) {
            tmp.clear();
            for (int64_t p = seq_off[u]; p < seq_off[u + 1]; p++) { tmp.push_back({seq_item[p], p}); train_count[seq_item[p]]++; }
            std::sort(tmp.begin(), tmp.end());
            for (size_t i = 0; i < tmp.size(); i++)
                if (i == 0 || tmp[i].first != tmp[i - 1].first) { clk_item.push_back(tmp[i].first); first_occ[tmp[i].second] = 1; }
            clk_off[u + 1] = (int64_t)clk_item.size();
        }
    }
    // users (MC: source items) each sampler may draw, and the number of updateFactors calls per epoch
    std::vector<int32_t> elig, eval_users;
    std::vector<int64_t> succ_off, sset_off; std::vector<int32_t> succ_item, sset_item;
    e->steps_per_epoch = 0;
    for (int u = 0; u < U; u++) e->steps_per_epoch += seq_off[u + 1] - seq_off[u];  // num_pos_events, model.hpp:47-51
    if (d->model == REC_MC) {
        // item_to_vec / item_to_set, MC.cpp:5-22
        std::vector<std::vector<int32_t>> sv(I);
        for (int u = 0; u < U; u++)
            for (int64_t p = seq_off[u]; p + 1 < seq_off[u + 1]; p++) sv[seq_item[p]].push_back(seq_item[p + 1]);
        succ_off.assign(I + 1, 0); sset_off.assign(I + 1, 0);
        e->steps_per_epoch = 0;
        for (int i = 0; i < I; i++) {
            succ_item.insert(succ_item.end(), sv[i].begin(), sv[i].end());
            succ_off[i + 1] = (int64_t)succ_item.size();
            std::vector<int32_t> s = sv[i];
            std::sort(s.begin(), s.end()); s.erase(std::unique(s.begin(), s.end()), s.end());
            sset_item.insert(sset_item.end(), s.begin(), s.end());
            sset_off[i + 1] = (int64_t)sset_item.size();
            e->steps_per_epoch += (int64_t)sv[i].size();  // num_pos_trans
            // MC.cpp:84; an engine that holds a share of the users samples the same share of the source items (its sampler
            // is driven by items, MC.cpp:80-89, so this is how N engines split an epoch's transitions)
            const int64_t ib = (int64_t)I * ub / std::max(1, U), ie = (int64_t)I * ue / std::max(1, U);
            if (!s.empty() && (int)s.size() != I && i >= ib && i < ie) elig.push_back(i);
        }
    } else {
        for (int u = ub; u < ue; u++) {
            const int64_t nu = seq_off[u + 1] - seq_off[u], nd = clk_off[u + 1] - clk_off[u];
            bool ok = false;
            switch (d->model) {
            case REC_BPRMF: ok = nd > 0 && nd < I; break;                                   // BPRMF.cpp:78
            case REC_HRM_AVG: case REC_HRM_MAX: ok = nd >= 2; break;                         // HRM_avg.cpp:130
            // TransRec.cpp:166; a user who clicked the whole catalog has no negative (the reference's draw would never end)
            case REC_TRANSREC: case REC_TRANSREC_L1: ok = nd >= 2 && nd < I; break;
            default: ok = nu >= 2; break;                                                   // FPMC.cpp:81, PRME.cpp:75, Fossil.cpp:123, REBUS.cpp:275
            }
            if (ok) elig.push_back(u);
        }
    }
    // per-owner sampling record (common.h): where the owner's events are, and the filter in front of the set a negative
    // is tested against (sampler.hip)
    std::vector<uint32_t> owner_rec;
    {
        const bool mc = d->model == REC_MC;
        const bool filtered = d->model == REC_BPRMF || d->model == REC_TRANSREC || d->model == REC_TRANSREC_L1 || mc;
        const int owners = mc ? I : U;
        owner_rec.assign((size_t)owners * 16, 0u);
        for (int o = 0; o < owners; o++) {
            uint32_t* r = &owner_rec[(size_t)o * 16];
            const int64_t off = mc ? succ_off[o] : seq_off[o], len = (mc ? succ_off[o + 1] : seq_off[o + 1]) - off;
            r[0] = (uint32_t)off; r[1] = (uint32_t)((uint64_t)off >> 32); r[2] = (uint32_t)len;
            if (!filtered) continue;
            const int64_t b0 = mc ? sset_off[o] : clk_off[o], b1 = mc ? sset_off[o + 1] : clk_off[o + 1];
            for (int64_t p = b0; p < b1; p++) {
                const uint32_t h = filter_slot(mc ? sset_item[p] : clk_item[p]);
                r[4 + (h >> 5)] |= 1u << (h & 31);
            }
        }
    }
    for (int u = ub; u < ue; u++) if (test_item[u] >= 0) eval_users.push_back(u);
    for (int u : eval_users)
        if (seq_off[u + 1] - seq_off[u] < 1 || val_item[u] < 0 || val_item[u] >= I || test_item[u] >= I) {
            delete e; rec_set_error("a user with a test item needs a validation item and at least one training item"); return -1;
        }
    std::vector<double> eta((size_t)(L + 1) * L, 0.0);
    for (int n = 0; n <= L; n++) {
        // REBUS.cpp:112-120; cumWeibull of common.hpp:92-99 is exp(-(x/7)^2)
        double res = 0;
        for (int i = 0; i < n; i++) res += exp(1 - (1 - exp(-pow(i / 7.0, 2.0))));
        for (int r = 0; r < n; r++) eta[(size_t)n * L + r] = exp(1 - (1 - exp(-pow(r / 7.0, 2.0)))) / res;
    }

    // ---- device state
    DevModel& m = e->dm;
    memset(&m, 0, sizeof m);
    m.user_begin = ub; m.user_end = ue;
    m.model = d->model; m.U = U; m.I = I; m.K = e->desc.K; m.KK = e->desc.KK; m.L = L;
    m.lambda = d->lambda; m.bias_reg = d->bias_reg; m.relation_reg = d->relation_reg; m.alpha = d->alpha;
    m.df = (double)e->desc.dataset_factor;
    int rc = 0;
#define UP(vec, field) if (!rc) rc = upload(e, vec, &m.field)
    // Per-user data is resident for the engine's own users only (the offsets, 16 bytes per user, are kept whole):
    // sequences, click sets, first-occurrence flags, sampling records, REBUS paths.  MC's sampling records are per item.
#define UPR(vec, first, count, field) if (!rc) rc = upload_range(e, vec, (int64_t)(first), (int64_t)(count), &m.field)
    UP(seq_off, seq_off); UP(test_item, test_item); UP(val_item, val_item); UP(clk_off, clk_off);
    UPR(seq_item, seq_off[ub], seq_off[ue] - seq_off[ub], seq_item);
    UPR(first_occ, seq_off[ub], seq_off[ue] - seq_off[ub], first_occ);
    if (history_model(d->model)) {
        // the weight of a history of c rows, |S|^alpha (REBUS.cpp:320-321) / |S|^-0.2 (Fossil.cpp:155-156), for every length there is
        int64_t longest = 0;
        for (int u = 0; u < U; u++) longest = std::max(longest, seq_off[u + 1] - seq_off[u]);
        std::vector<double> len_pow((size_t)longest + 2, 0.0);
        const double ex = d->model == REC_REBUS ? d->alpha : -0.2;
        for (size_t c = 1; c < len_pow.size(); c++) len_pow[c] = pow((double)c, ex);
        UP(len_pow, len_pow);
    }
    if (d->model == REC_REBUS) {
        // the history a REBUS step walks is the first occurrences of a prefix (REBUS.cpp:305-318): those positions alone, packed
        std::vector<int32_t> kept_pair((size_t)2 * (seq_off[ue] - seq_off[ub])), kept_rank((size_t)(seq_off[ue] - seq_off[ub]));
        const int64_t base = seq_off[ub];
        for (int u = ub; u < ue; u++) {
            int64_t w = seq_off[u];
            for (int64_t p = seq_off[u]; p < seq_off[u + 1]; p++) {
                kept_rank[p - base] = (int32_t)(w - seq_off[u]);
                if (first_occ[p]) { kept_pair[2 * (w - base)] = seq_item[p]; kept_pair[2 * (w - base) + 1] = (int32_t)(p - seq_off[u]); w++; }
            }
            for (; w < seq_off[u + 1]; w++) { kept_pair[2 * (w - base)] = 0; kept_pair[2 * (w - base) + 1] = 0x7fffffff; }
        }
        if (!rc) rc = upload_range(e, kept_pair, 0, (int64_t)kept_pair.size(), &m.kept_pair);
        if (!rc) m.kept_pair -= 2 * base;
        if (!rc) rc = upload_range(e, kept_rank, 0, (int64_t)kept_rank.size(), &m.kept_rank);
        if (!rc) m.kept_rank -= base;
    }
    UPR(clk_item, clk_off[ub], clk_off[ue] - clk_off[ub], clk_item);
    UP(elig, elig); UP(eval_users, eval_users);
    e->train_count_host = train_count;
    UP(train_count, train_count); UP(eta, eta);
    {
        std::vector<int32_t> pop_order(I);
        for (int i = 0; i < I; i++) pop_order[i] = i;
        std::stable_sort(pop_order.begin(), pop_order.end(), [&](int32_t a, int32_t b) { return train_count[a] > train_count[b]; });
        UP(pop_order, pop_order);
    }
    if (d->model == REC_MC) { UP(owner_rec, owner_rec); }
    else { UPR(owner_rec, (int64_t)ub * 16, (int64_t)(ue - ub) * 16, owner_rec); }
    if (d->model == REC_MC) { UP(succ_off, succ_off); UP(succ_item, succ_item); UP(sset_off, sset_off); UP(sset_item, sset_item); }
    if (d->model == REC_REBUS && !rc) {
        const int64_t nctx = E + 2 * (int64_t)U;
        std::vector<int64_t> po(c->path_off, c->path_off + nctx + 1);
        std::vector<int32_t> pi(c->path_item, c->path_item + po[nctx]);
        for (int64_t x = 0; x < nctx && !rc; x++)
            if (po[x + 1] - po[x] > L) { rec_set_error("a matched path is longer than L"); rc = -1; }
        // contexts of user u: [seq_off[u] + 2u, seq_off[u+1] + 2(u+1))
        const int64_t c0 = seq_off[ub] + 2 * (int64_t)ub, c1 = seq_off[ue] + 2 * (int64_t)ue;
        UPR(po, c0, c1 - c0 + 1, path_off); UPR(pi, po[c0], po[c1] - po[c0], path_item);
    }
#undef UP
#undef UPR
    m.n_elig = (int)elig.size(); m.n_eval = (int)eval_users.size(); m.n_ev = 0;
    m.elig_identity = 1;
    for (size_t i = 0; i < elig.size() && m.elig_identity; i++) if (elig[i] != (int32_t)i) m.elig_identity = 0;
    auto fail = [&](int code) { rec_destroy(e); return code; };
    if (rc) return fail(rc);
    if (hipStreamCreate(&e->st) != hipSuccess || hipStreamCreate(&e->st2) != hipSuccess) { rec_set_error("hipStreamCreate failed"); return fail(-2); }
    for (int b = 0; b < 2; b++)
        if (hipEventCreateWithFlags(&e->ev_sampled[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_consumed[b], hipEventDisableTiming) != hipSuccess) { rec_set_error("hipEventCreate failed"); return fail(-2); }
    if (hipMalloc(&e->W, e->dev_elems * e->elem_size) != hipSuccess || hipMalloc(&e->bestW, e->dev_elems * e->elem_size) != hipSuccess) {
        rec_set_error("hipMalloc of the parameter tables failed"); return fail(-2);
    }
    (void)hipMemset(e->W, 0, e->dev_elems * e->elem_size);
    (void)hipMemset(e->bestW, 0, e->dev_elems * e->elem_size);
    // a table's base is where its row 0 would be: kernels index with global row numbers
    for (size_t t = 0; t < e->tables.size(); t++)
        m.tab[t] = (char*)e->W + (e->tables[t].dev_off - e->tables[t].row0 * e->tables[t].cols) * e->elem_size;
    void* g = nullptr;
    if (hipMalloc(&g, 8 * (REC_MAX_K + 2 * REC_MAX_L + 8)) != hipSuccess) { rec_set_error("hipMalloc failed"); return fail(-2); }
    e->owned.push_back(g); (void)hipMemset(g, 0, 8 * (REC_MAX_K + 2 * REC_MAX_L + 8));
    m.gacc = g;
    void* wc = nullptr;
    if (hipMalloc(&wc, 64) != hipSuccess) { rec_set_error("hipMalloc failed"); return fail(-2); }
    e->owned.push_back(wc); (void)hipMemset(wc, 0, 64);
    m.work_counter = (int*)wc;
    void* hot = nullptr;
    if (hipMalloc(&hot, I + 64) != hipSuccess) { rec_set_error("hipMalloc failed"); return fail(-2); }
    e->owned.push_back(hot); (void)hipMemset(hot, 0, I + 64);
    m.hot = (const uint8_t*)hot;
    // All tuples of one launch may be computed from the same parameter values.  With a small catalog a launch is
    // therefore kept short enough that an item row receives about two updates per launch on average; with a
    // large catalog the number of tuples the chip has in flight (tens of thousands) is below that bound anyway.
    {
        double rows = 2;
        switch (d->model) {
        case REC_MC: case REC_HRM_AVG: case REC_HRM_MAX: case REC_TRANSREC: case REC_TRANSREC_L1: rows = 3; break;
        case REC_FPMC: case REC_PRME: rows = 5; break;
        case REC_FOSSIL: case REC_FOSSIL_SIMPLE: case REC_REBUS: rows = 2 + (double)E / std::max(1, U); break;
        default: break;
        }
        // Updates of one row that may be in flight together: 2 for the group kernels (validated against the reference's
        // trajectories on ML-5); 16 for the history kernel, whose tuples spread a small step over their user's whole
        // history - on a 100k-user x 20k-item corpus Fossil and REBUS follow the same AUC trajectory (within 0.002)
        // with 64 tuples in flight as with 8192.
        const bool hist_model = d->model == REC_FOSSIL || d->model == REC_FOSSIL_SIMPLE || d->model == REC_REBUS;
        const int64_t w = (int64_t)((hist_model ? 16.0 : 2.0) * I / rows);
        // tuples a launch has in flight at most: 2048 resident blocks, 256 / (lanes per tuple) groups each, B tuples per group
        const int kmax = std::max(e->desc.K, e->desc.KK);
        const int lanes = kmax <= 16 ? 16 : kmax <= 32 ? 32 : 64;
        const int per_group = kmax > 128 ? 1 : kmax > 64 ? 2 : (d->model == REC_FPMC || d->model == REC_PRME) ? 2 : 4;
        const int64_t in_flight = hist_model ? 8192 : 2048LL * (256 / lanes) * per_group;
        e->stale_window = w >= in_flight ? 0 : std::max<int64_t>(16, w);
    }
    // Fossil's global weights move by lr/10 per step with a small curvature: a larger budget than TransRec's r
    if (d->model == REC_FOSSIL || d->model == REC_FOSSIL_SIMPLE) {
        e->global_cap = 4096;
        // A launch is at most 1/32 of an epoch (and 2^18 .. 2^22 tuples): every launch ends on its longest history, so short
        // launches lose to their tails.  1M users x 200k items x 100M clicks, launches of 2^18 / 2^20 / 2^21 / 2^22 tuples:
        // 0.70 / 0.61 / - / 0.59 s per epoch with four tuples per walk, - / 0.400 / 0.386 / 0.380 s with eight; test AUC after
        // one to four epochs equal within its seed-to-seed spread of 0.002 for 2^18, 2^20 and 2^22
        // (scripts/exp_fossil_window.py).  Shorter launches are no better for the global row either: on the 12k-user corpus
        // (600k steps per epoch) launches of 4096 / 18 750 / 65 536 / 262 144 tuples end 0.014 / 0.0045 / 0.0003 / 0.0001 below
        // the reference's best test AUC (scripts/exp_mid_variants.py Fossil "global_window=...").
        while (e->global_window < (1 << 22) && e->global_window * 2 * 32 <= e->steps_per_epoch) e->global_window *= 2;
        // An epoch shorter than a launch would move the global rows once per EPOCH, by global_cap steps' worth: 4.4 times less
        // than the reference's per-step updates over the 18 097 steps of an ML-5 epoch, and FossilSimple ended 0.004 of test AUC
        // ABOVE the reference's mean of five seeds.  Such an epoch runs as four launches instead: ML-5 at 4524 tuples a launch
        // ends within 0.001 (FossilSimple) and 0.002 (Fossil, L = 3) of that mean; two launches of 9048 leave Fossil 0.005
        // below it, and one launch with twice the budget 0.006 below (scripts/exp_converged_key.py MODEL 2 global_window=...).
        if (e->steps_per_epoch < e->global_window) e->global_window = std::max<int64_t>(e->global_cap, e->steps_per_epoch / 4);
    }
    // TransRec's global translation r barely depends on the launch length (200k x 50k corpus: the same AUC trajectory
    // within 0.003 for launches of 2^16 .. 3*10^7 tuples), and longer launches lose less to ramp-up and tails
    if (d->model == REC_TRANSREC || d->model == REC_TRANSREC_L1) e->global_window = 1 << 22;
    // scatter policy: exact sums for small problems, the hybrid for large ones (see rec_set_option)
    if (int orc = rec_set_option(e, "update_policy", e->steps_per_epoch >= (1 << 22) ? 1 : 0)) return fail(orc);
    // The history models walk a user's history once for two tuples of that user.  Measured on a 200k-user corpus with
    // ~100 clicks per user: 1.3x (REBUS) / 1.8x (Fossil, FossilSimple) the throughput; early-stopped test AUC equal within
    // run-to-run noise (lr 0.01: 0.8125 vs 0.8124; lr 0.05, REBUS: 0.7794 vs 0.7789).  Both tuples step from the same
    // parameters, which on users with a handful of clicks (where the two tuples mostly coincide) doubles the step: on
    // the bundled ML-5 corpus (<= 3 training clicks per user) REBUS at lr 0.05 then diverges in about 1 run of 10.  So
    // pairs are only the default where histories are long; "user_batch" sets it either way.  Whatever the mean, a user
    // with fewer than "pair_min_history" (8) training clicks never shares a walk: the kernel applies that user's tuples
    // one after the other.
    const double mean_history = U > 0 ? (double)E / (double)U : 0.0;
    e->user_batch = (history_model(d->model) && e->steps_per_epoch >= (1 << 22) && mean_history >= 16.0) ? 2 : 1;
    // Where histories are long, more tuples share a walk: the float atomics of the walk, one row per history row and GROUP,
    // are what bounds the kernels.  Fossil / FossilSimple (every step rewrites the user's whole training sequence): four
    // tuples where the mean history is >= 32 clicks, eight where it is >= 64; REBUS (the tuples of a group end at different
    // positions, the walk is the longest prefix): four from a mean of 32.  1M users x 200k items x 100M clicks (mean 98),
    // ms per epoch for two / four / eight tuples per walk (before the table of weights, which took another 40 .. 125 ms off the
    // defaults): Fossil 1230 / 612 / 438, FossilSimple - / 633 / 364, REBUS 713 / 565 / 531.  On a 5.9M-step corpus with that catalog the best-validation test AUC ends, against the sequential fp64
    // run: Fossil +0.0025 / +0.0017 / +0.0007, FossilSimple +0.0014 / +0.0009 / -0.0005, REBUS +0.0005 / +0.0008 / +0.0014
    // (HIT@50 -0.0009 / -0.0004 / -0.006); tests/test_gpu_seq_anchor.py holds the default to 0.005.
    if (e->user_batch == 2 && mean_history >= 32.0)
        e->user_batch = (d->model != REC_REBUS && mean_history >= 64.0) ? 8 : 4;
    e->dm.pair_min_hist = 8;
    // Owner mode ("owner_neg": launches partitioned by the negative's slice of the catalog, the negatives' deltas summed in
    // LDS by the workgroup that owns the slice) removes half of the float atomics, which bound the update kernel, but it
    // is off by default: within a launch an item's steps in the negative's role come when its workgroup gets to them,
    // not when they were drawn, and reach memory a few steps late.  BPRMF K = 64 on the 1M x 200k x 100M corpus, test AUC
    // after 3 epochs against 0.7313 with atomics per tuple: 0.7307 / 0.7298 / 0.7285 / 0.7278 / 0.7188 for launches of
    // 1 / 2 / 4 / 8 / 16 tuples per item, 0.658 (HIT@50 0.02 against 0.10) for one launch per epoch; the launches short
    // enough to follow are too short to be faster (46 ms per epoch at 8 per item against 47).
    e->owner = 0;
    e->neg_share = 1;
    *out = e;
    return 0;
}

extern "C" int rec_destroy(rec_engine* e)
{
    if (!e) return 0;
    (void)hipSetDevice(e->desc.device);
    if (e->st) (void)hipStreamSynchronize(e->st);
    if (e->st2) (void)hipStreamSynchronize(e->st2);
    for (int b = 0; b < 2; b++) { if (e->ev_sampled[b]) (void)hipEventDestroy(e->ev_sampled[b]); if (e->ev_consumed[b]) (void)hipEventDestroy(e->ev_consumed[b]); }
    if (e->st2) (void)hipStreamDestroy(e->st2);
    for (int w = 0; w < 4; w++) for (auto& p : e->pending[w]) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (void* p : e->owned) (void)hipFree(p);
    if (e->W) (void)hipFree(e->W);
    if (e->bestW) (void)hipFree(e->bestW);
    if (e->stage) (void)hipFree(e->stage);
    if (e->tuples) (void)hipFree(e->tuples);
    if (e->part_scratch) (void)hipFree(e->part_scratch);
    if (e->qoff_dev) (void)hipFree(e->qoff_dev);
    if (e->st) (void)hipStreamDestroy(e->st);
    delete e;
    return 0;
}

extern "C" int rec_sync(rec_engine* e)
{
    if (!e) { rec_set_error("null engine"); return -1; }
    ENG_HIP(hipSetDevice(e->desc.device));
    ENG_HIP(hipStreamSynchronize(e->st2));
    ENG_HIP(hipStreamSynchronize(e->st));
    return 0;
}

extern "C" int64_t rec_nw(const rec_engine* e) { return e ? e->nw : -1; }
extern "C" int64_t rec_steps_per_epoch(const rec_engine* e) { return e ? e->steps_per_epoch : -1; }
extern "C" int64_t rec_tuples_drawn(const rec_engine* e) { return e ? e->tuples_drawn : -1; }

static int ensure_stage(rec_engine* e, int64_t elems)
{
    if (e->stage_elems >= elems) return 0;
    if (e->stage) (void)hipFree(e->stage);
    e->stage = nullptr; e->stage_elems = 0;
    ENG_HIP(hipMalloc((void**)&e->stage, sizeof(double) * elems));
    e->stage_elems = elems;
    return 0;
}

// tables are moved through an f64 staging buffer in slices, so that W never needs a second full-size copy
static const int64_t STAGE_ELEMS = 1 << 24;

extern "C" int rec_params_upload(rec_engine* e, const double* W, int64_t nw)
{
    if (!e || !W) { rec_set_error("null argument"); return -1; }
    if (nw != e->nw) { rec_set_error("NW not match."); return -1; }  // model.cpp:213-216
    ENG_HIP(hipSetDevice(e->desc.device));
    if (int rc = ensure_stage(e, std::min<int64_t>(STAGE_ELEMS, std::max<int64_t>(nw, 1)))) return rc;
    for (auto& t : e->tables) {
        // the resident rows of the table (a user table of a sharded engine: its own users' rows; the others are ignored)
        const int64_t n = t.nrows * t.cols, h0 = t.host_off + t.row0 * t.cols;
        for (int64_t o = 0; o < n; o += e->stage_elems) {
            const int64_t len = std::min(e->stage_elems, n - o);
            ENG_HIP(hipMemcpyAsync(e->stage, W + h0 + o, sizeof(double) * len, hipMemcpyHostToDevice, e->st));
            if (int rc = launch_convert_in(e->stage, (char*)e->W + (t.dev_off + o) * e->elem_size, len, e->desc.precision, e->st)) return rc;
            ENG_HIP(hipStreamSynchronize(e->st));
        }
    }
    return 0;
}

extern "C" int rec_params_download(rec_engine* e, double* W, int64_t nw)
{
    if (!e || !W) { rec_set_error("null argument"); return -1; }
    if (nw != e->nw) { rec_set_error("NW not match."); return -1; }
    ENG_HIP(hipSetDevice(e->desc.device));
    if (int rc = ensure_stage(e, std::min<int64_t>(STAGE_ELEMS, std::max<int64_t>(nw, 1)))) return rc;
    for (auto& t : e->tables) {
        // the resident rows only: rows of other engines' users keep whatever W holds there
        const int64_t n = t.nrows * t.cols, h0 = t.host_off + t.row0 * t.cols;
        for (int64_t o = 0; o < n; o += e->stage_elems) {
            const int64_t len = std::min(e->stage_elems, n - o);
            if (int rc = launch_convert_out((char*)e->W + (t.dev_off + o) * e->elem_size, e->stage, len, e->desc.precision, e->st)) return rc;
            ENG_HIP(hipMemcpyAsync(W + h0 + o, e->stage, sizeof(double) * len, hipMemcpyDeviceToHost, e->st));
            ENG_HIP(hipStreamSynchronize(e->st));
        }
    }
    return 0;
}

extern "C" int rec_params_read(rec_engine* e, int64_t first, int64_t n, double* out)
{
    if (!e || (!out && n > 0)) { rec_set_error("null argument"); return -1; }
    if (first < 0 || n < 0 || first + n > e->nw) { rec_set_error("parameter range out of bounds"); return -1; }
    ENG_HIP(hipSetDevice(e->desc.device));
    ENG_HIP(hipStreamSynchronize(e->st));
    if (int rc = ensure_stage(e, std::min<int64_t>(STAGE_ELEMS, std::max<int64_t>(e->nw, 1)))) return rc;
    for (auto& t : e->tables) {
        const int64_t h0 = t.host_off + t.row0 * t.cols;   // first resident element of the table in the flat vector
        if (first < h0 && first + n > t.host_off && t.row0 > 0) { rec_set_error("parameter range holds rows of users that are not resident in this engine"); return -1; }
        if (first + n > h0 + t.nrows * t.cols && first < t.host_off + t.rows * t.cols && t.nrows != t.rows) { rec_set_error("parameter range holds rows of users that are not resident in this engine"); return -1; }
        const int64_t a = std::max(first, h0), b = std::min(first + n, h0 + t.nrows * t.cols);
        for (int64_t o = a; o < b; o += e->stage_elems) {
            const int64_t len = std::min(e->stage_elems, b - o);
            if (int rc = launch_convert_out((char*)e->W + (t.dev_off + (o - h0)) * e->elem_size, e->stage, len, e->desc.precision, e->st)) return rc;
            ENG_HIP(hipMemcpyAsync(out + (o - first), e->stage, sizeof(double) * len, hipMemcpyDeviceToHost, e->st));
            ENG_HIP(hipStreamSynchronize(e->st));
        }
    }
    return 0;
}

extern "C" int rec_params_init(rec_engine* e, uint64_t seed)
{
    if (!e) { rec_set_error("null engine"); return -1; }
    ENG_HIP(hipSetDevice(e->desc.device));
    if (int rc = launch_param_init(e->dm, e->desc.precision, seed, e->st)) return rc;
    ENG_HIP(hipStreamSynchronize(e->st));
    return 0;
}

extern "C" int rec_snapshot_best(rec_engine* e)
{
    if (!e) { rec_set_error("null engine"); return -1; }
    ENG_HIP(hipSetDevice(e->desc.device));
    ENG_HIP(hipMemcpyAsync(e->bestW, e->W, e->dev_elems * e->elem_size, hipMemcpyDeviceToDevice, e->st));
    ENG_HIP(hipStreamSynchronize(e->st));
    return 0;
}

extern "C" int rec_restore_best(rec_engine* e)
{
    if (!e) { rec_set_error("null engine"); return -1; }
    ENG_HIP(hipSetDevice(e->desc.device));
    ENG_HIP(hipMemcpyAsync(e->W, e->bestW, e->dev_elems * e->elem_size, hipMemcpyDeviceToDevice, e->st));
    ENG_HIP(hipStreamSynchronize(e->st));
    return 0;
}

static int ensure_tuples(rec_engine* e, int64_t n)
{
    ENG_HIP(hipStreamSynchronize(e->st2));
    if (e->tuples_cap >= n) return 0;
    ENG_HIP(hipStreamSynchronize(e->st));
    if (e->tuples) (void)hipFree(e->tuples);
    e->tuples = nullptr; e->tuples_cap = 0;
    ENG_HIP(hipMalloc((void**)&e->tuples, sizeof(int4) * n * 2));
    e->tuples_cap = n;
    e->consumed_valid[0] = e->consumed_valid[1] = false;
    return 0;
}

// Sampling events ordered by positive item (launch_event_sampler): one per (owner, position) the tuple-major sampler
// can pick, with the owner's 1/count as weight.  Built on first use from the device copies of the corpus - the
// default (tuple-major) path never needs them.
static int ensure_events(rec_engine* e)
{
    if (e->dm.n_ev > 0) return 0;
    if (!sgd_runs_supported(e->desc.model)) { rec_set_error("the event-major sampler is not available for this model"); return -1; }
    const int I = e->I, U = e->U;
    const bool mc = e->desc.model == REC_MC, bpr = e->desc.model == REC_BPRMF;
    DevModel& m = e->dm;
    std::vector<int32_t> elig(m.n_elig);
    ENG_HIP(hipMemcpy(elig.data(), m.elig, sizeof(int32_t) * elig.size(), hipMemcpyDeviceToHost));
    const int n_owner = mc ? I : U;
    std::vector<int64_t> off(n_owner + 1);
    ENG_HIP(hipMemcpy(off.data(), mc ? m.succ_off : m.seq_off, sizeof(int64_t) * off.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> item(off[n_owner]);
    if (mc) ENG_HIP(hipMemcpy(item.data(), m.succ_item, sizeof(int32_t) * item.size(), hipMemcpyDeviceToHost));
    else {
        // only the sequences of the engine's own users are resident (and only they are eligible)
        const int64_t a = off[e->desc.user_begin], b = off[e->desc.user_end];
        if (b > a) ENG_HIP(hipMemcpy(item.data() + a, m.seq_item + a, sizeof(int32_t) * (size_t)(b - a), hipMemcpyDeviceToHost));
    }
    auto n_events = [&](int o) -> int64_t { return (mc || bpr) ? off[o + 1] - off[o] : off[o + 1] - off[o] - 1; };
    auto event_of = [&](int o, int64_t x) -> int4 {
        if (mc || bpr) return make_int4(o, 0, item[off[o] + x], 0);
        return make_int4(o, item[off[o] + x], item[off[o] + x + 1], 0);
    };
    std::vector<int64_t> start(I + 1, 0);
    int64_t total = 0;
    for (int o : elig) { const int64_t c = n_events(o); total += c; for (int64_t x = 0; x < c; x++) start[event_of(o, x).z + 1]++; }
    if (total <= 0) { rec_set_error("no sampling events"); return -1; }
    for (int i = 0; i < I; i++) start[i + 1] += start[i];
    std::vector<int4> ev(total); std::vector<float> ev_w(total);
    for (int o : elig) {
        const int64_t c = n_events(o); const float w = 1.0f / (float)c;
        for (int64_t x = 0; x < c; x++) { const int4 r = event_of(o, x); const int64_t at = start[r.z]++; ev[at] = r; ev_w[at] = w; }
    }
    // The events of one item are kept in random order.  In owner order every popular item would visit the
    // users in the same sequence, and the lane groups working on different items would reach the same user
    // at the same moment, launch after launch.
    uint64_t x64 = 0x2545F4914F6CDD1Dull;
    auto rnd = [&](uint64_t bound) { x64 ^= x64 << 13; x64 ^= x64 >> 7; x64 ^= x64 << 17; return (x64 >> 11) % bound; };
    auto shuffle = [&](int64_t b0, int64_t b1) {
        for (int64_t k = b1 - 1; k > b0; k--) {
            const int64_t j = b0 + (int64_t)rnd((uint64_t)(k - b0 + 1));
            std::swap(ev[k], ev[j]); std::swap(ev_w[k], ev_w[j]);
        }
    };
    if (e->event_phases > 0) {
        for (int p = 0; p < e->event_phases; p++) shuffle(total * p / e->event_phases, total * (p + 1) / e->event_phases);
    } else {
        int64_t b0 = 0;
        for (int i = 0; i < I; i++) { shuffle(b0, start[i]); b0 = start[i]; }   // start[i] is now the end of item i's events
    }
    if (int rc = upload(e, ev, &m.ev)) return rc;
    if (int rc = upload(e, ev_w, &m.ev_w)) return rc;
    // per item: where its events are, and an alias table over their weights (Vose's construction)
    {
        std::vector<int64_t> ev_off(I + 1, 0);
        for (int i = 0; i < I; i++) ev_off[i + 1] = start[i];   // start[i] is the end of item i's events
        std::vector<float> ap(total, 1.0f); std::vector<int32_t> ai(total, 0);
        std::vector<double> scaled; std::vector<int32_t> small, large;
        for (int i = 0; i < I; i++) {
            const int64_t a0 = ev_off[i], cnt = ev_off[i + 1] - a0;
            if (cnt <= 0) continue;
            double sum = 0;
            for (int64_t k = 0; k < cnt; k++) sum += ev_w[a0 + k];
            scaled.resize(cnt); small.clear(); large.clear();
            for (int64_t k = 0; k < cnt; k++) { scaled[k] = ev_w[a0 + k] * (double)cnt / sum; (scaled[k] < 1.0 ? small : large).push_back((int32_t)k); ai[a0 + k] = (int32_t)k; }
            while (!small.empty() && !large.empty()) {
                const int32_t sidx = small.back(), lidx = large.back(); small.pop_back();
                ap[a0 + sidx] = (float)scaled[sidx]; ai[a0 + sidx] = lidx;
                scaled[lidx] -= 1.0 - scaled[sidx];
                if (scaled[lidx] < 1.0) { large.pop_back(); small.push_back(lidx); }
            }
            for (int32_t k : small) ap[a0 + k] = 1.0f;
            for (int32_t k : large) ap[a0 + k] = 1.0f;
        }
        if (int rc = upload(e, ev_off, &m.ev_off)) return rc;
        if (int rc = upload(e, ap, &m.ev_alias_p)) return rc;
        if (int rc = upload(e, ai, &m.ev_alias_i)) return rc;
    }
    ENG_HIP(hipMalloc(&e->group_scratch, event_scratch_bytes(total)));
    e->owned.push_back(e->group_scratch);
    ENG_HIP(hipMalloc((void**)&e->n_drawn_dev, 64));
    e->owned.push_back(e->n_drawn_dev);
    m.n_ev = total;
    return 0;
}

static int ensure_partition_scratch(rec_engine* e, int64_t n, int slices)
{
    const size_t need = partition_scratch_bytes(n, slices);
    if (e->part_scratch_bytes < need) {
        ENG_HIP(hipStreamSynchronize(e->st));
        if (e->part_scratch) (void)hipFree(e->part_scratch);
        e->part_scratch = nullptr; e->part_scratch_bytes = 0;
        ENG_HIP(hipMalloc(&e->part_scratch, need));
        e->part_scratch_bytes = need;
    }
    if (!e->qoff_dev) { ENG_HIP(hipMalloc((void**)&e->qoff_dev, sizeof(int64_t) * (1024 + 8))); }
    return 0;
}

// Hogwild launches over `n` tuples.  Models with a row that every tuple touches (TransRec's r,
// Fossil's WT, FossilSimple's alpha) keep that row constant during a launch and fold its summed
// gradient in afterwards, so their launches are kept short (`global_window` tuples).
static bool history_model(int model) { return model == REC_FOSSIL || model == REC_FOSSIL_SIMPLE || model == REC_REBUS; }

// sampled (tuples == nullptr): the update kernel draws tuple i of this call itself, as step sampled->first_step + i of
// (seed, epoch); only where can_fuse_sampler() holds
struct SampledSource { int64_t first_step; uint64_t seed, epoch; };
static bool can_fuse_sampler(const rec_engine* e);

static int launch_update(rec_engine* e, const int4* tuples, int64_t n, double lr, bool grouped = false, int user_batch = 1, const SampledSource* sampled = nullptr)
{
    const int model = e->desc.model;
    const bool history = model == REC_FOSSIL || model == REC_FOSSIL_SIMPLE || model == REC_REBUS;
    const bool global_row = model == REC_TRANSREC || model == REC_TRANSREC_L1 || model == REC_FOSSIL || model == REC_FOSSIL_SIMPLE;
    int64_t window = global_row ? e->global_window : n;
    // Small catalogs: the history kernel bounds the tuples in flight inside one launch (one tuple per wavefront, so
    // fewer wavefronts); the group kernels are launched in windows of that many tuples.
    if (!history && e->stale_window > 0 && e->stale_window < window) window = e->stale_window;
    // Owner mode: workgroup s applies the launch's tuples whose negative lies in slice s, so within a launch an item's steps
    // in the negative's role come when its workgroup gets to them, not when they were drawn.  Over a whole epoch (490
    // such steps per item on the 200k-item corpus, each moving the item's bias by lr / 4 against as many in the positive's
    // role) a workgroup that runs ahead or has a shorter list leaves its items too high for the rest of the launch, and
    // the ranking falls apart (test AUC after one epoch 0.65 against 0.727, HIT@50 0.0003 against 0.09).  Launches of at
    // most owner_per_item tuples per item keep that imbalance to a few steps per item.
    const bool owner_able = e->owner && !history && !grouped && e->group == 0 && !e->overlap && sgd_owner_supported(model);
    const int64_t owner_window = std::max<int64_t>(4096, e->owner_per_item * (int64_t)e->dm.I);
    const bool owner_on = owner_able && owner_window >= e->owner_min && n >= e->owner_min;
    if (owner_on && owner_window < window) window = owner_window;
    if (window < n && (user_batch > 1 || e->neg_share > 1 || e->pos_share > 1) && window > 8) window &= ~(int64_t)7;   // launches start on a group boundary
    for (int64_t o = 0; o < n; o += window) {
        const int64_t len = std::min(window, n - o);
        // owner mode: needs room for the partitioned copy, which is the second half of the engine's tuple buffer
        const bool use_wide = e->wide && !history && e->neg_share == 1 && e->pos_share == 1 && sgd_wide_supported(e->dm, e->desc.precision);
        const bool in_own_buffer = !sampled && e->tuples && tuples + o >= e->tuples && tuples + o + len <= e->tuples + e->tuples_cap;
        if (sampled) {
            if (int rc = launch_sgd_wide_sampled(e->dm, len, sampled->first_step + o, sampled->seed, sampled->epoch, lr, e->atomic, e->st)) return rc;
            if (global_row) {
                if (int rc = launch_apply_global(e->dm, e->desc.precision, lr, len, (double)e->global_cap, e->st, e->project_after_launch != 0)) return rc;
            }
            continue;
        }
        if (owner_on && in_own_buffer && len >= e->owner_min) {
            const int slices = !use_wide ? sgd_owner_slices() : e->owner_slices ? e->owner_slices : sgd_wide_owner_slices(e->dm);
            const int slice_rows = (e->dm.I + slices - 1) / slices;
            if (int rc = ensure_partition_scratch(e, len, slices)) return rc;
            int4* dst = (int4*)(tuples + o) + e->tuples_cap;
            hipEvent_t pa = nullptr, pb = nullptr;
            if (e->profile) { ENG_HIP(hipEventCreate(&pa)); ENG_HIP(hipEventCreate(&pb)); ENG_HIP(hipEventRecord(pa, e->st)); }
            int rc = launch_partition_by_slice(tuples + o, dst, len, sgd_owner_neg_field(model), slice_rows, slices,
                                               e->part_scratch, e->qoff_dev, e->st, e->part_tile);
            if (e->profile) { ENG_HIP(hipEventRecord(pb, e->st)); e->pending[3].push_back({pa, pb}); }
            if (!rc) rc = use_wide ? launch_sgd_wide(e->dm, dst, e->qoff_dev, slices, len, lr, e->atomic, e->owner_flush, e->st)
                                   : launch_sgd_owner(e->dm, e->desc.precision, dst, e->qoff_dev, len, lr, e->atomic, e->owner_flush, e->st);
            if (rc) return rc;
            if (global_row) {
                if ((rc = launch_apply_global(e->dm, e->desc.precision, lr, len, (double)e->global_cap, e->st, e->project_after_launch != 0))) return rc;
            }
            continue;
        }
        int rc = history ? launch_sgd_history(e->dm, e->desc.precision, tuples + o, len, lr, e->atomic, e->stale_window, user_batch, e->st)
                 : ((grouped || e->group == 1) && sgd_runs_supported(model)) ? launch_sgd_runs(e->dm, e->desc.precision, tuples + o, len, lr, e->atomic, e->run_segment, e->run_policy, e->st)
                 : use_wide ? launch_sgd_wide(e->dm, tuples + o, nullptr, 0, len, lr, e->atomic, 1, e->st)
                         : launch_sgd_hogwild(e->dm, e->desc.precision, tuples + o, len, lr, e->atomic | ((e->neg_share >= 4 ? 2 : (e->neg_share == 2 || e->pos_share == 2) ? 1 : 0) << 16), e->st);
        if (rc) return rc;
        if (global_row) {
            if ((rc = launch_apply_global(e->dm, e->desc.precision, lr, len, (double)e->global_cap, e->st, e->project_after_launch != 0))) return rc;
            if ((rc = launch_apply_global_history(e->dm, e->desc.precision, lr, (double)e->global_cap, e->st))) return rc;
        }
    }
    return 0;
}

// The default update of the group models (wide kernel, every step its own draws, one stream) needs no tuple buffer: the
// kernel draws what the sampler would have written ("fused_sampler" 1, the default).
static bool can_fuse_sampler(const rec_engine* e)
{
    return e->fused_sampler && !history_model(e->desc.model) && e->wide && e->neg_share == 1 && e->pos_share == 1 && e->group == 0 && !e->overlap &&
           !e->owner && sgd_wide_supported(e->dm, e->desc.precision);
}

// ---- HIP-event bookkeeping: events are recorded around a launch and read back lazily
static int ev_begin(rec_engine* e, int which, hipEvent_t* a, hipEvent_t* b)
{
    if (!e->profile) return 0;
    ENG_HIP(hipEventCreate(a)); ENG_HIP(hipEventCreate(b));
    ENG_HIP(hipEventRecord(*a, e->st));
    (void)which;
    return 0;
}
static int ev_end(rec_engine* e, int which, hipEvent_t a, hipEvent_t b)
{
    if (!e->profile) return 0;
    ENG_HIP(hipEventRecord(b, e->st));
    e->pending[which].push_back({a, b});
    return 0;
}

static int check_tuples(const rec_engine* e, const int32_t* t, int64_t n)
{
    // operands must fit the tables before a kernel dereferences them
    const int I = e->I; const int model = e->desc.model;
    // users: only this engine's own (the rows and sequences of the others are not resident)
    const int U0 = e->desc.user_begin, U = e->desc.user_end;
    for (int64_t i = 0; i < n; i++) {
        const int32_t a = t[4 * i], b = t[4 * i + 1], c = t[4 * i + 2], d = t[4 * i + 3];
        bool ok = true;
        switch (model) {
        case REC_BPRMF: ok = a >= U0 && a < U && b >= 0 && b < I && c >= 0 && c < I; break;
        case REC_MC: ok = a >= 0 && a < I && b >= 0 && b < I && c >= 0 && c < I; break;
        case REC_FOSSIL: case REC_REBUS: ok = a >= U0 && a < U && c >= 0 && c < I; break;   // the position b is checked by the caller
        default: ok = a >= U0 && a < U && b >= 0 && b < I && c >= 0 && c < I && d >= 0 && d < I; break;
        }
        if (!ok) { rec_set_error("tuple " + std::to_string(i) + " is out of range (or names a user of another engine)"); return -1; }
    }
    return 0;
}

// dev_buf: device room for the n tuples (the engine's own chunk buffer, or the static scratch); timed: HIP events
// around every launch when "profile" is on (creating events allocates)
static int replay_tuples(rec_engine* e, double lr, const int32_t* tuples, int64_t n, int mode, int4* dev_buf, bool timed)
{
    if (mode != REC_REPLAY_SEQUENTIAL && mode != REC_REPLAY_HOGWILD) { rec_set_error("unknown replay mode"); return -1; }   // before anything is enqueued
    if (int rc = check_tuples(e, tuples, n)) return rc;
    if (e->desc.model == REC_FOSSIL || e->desc.model == REC_REBUS) {
        // target positions must lie inside the user's training sequence
        const std::vector<int64_t>& so = e->seq_off_host;
        for (int64_t i = 0; i < n; i++) {
            const int u = tuples[4 * i]; const int64_t nu = so[u + 1] - so[u];
            const int64_t tp = tuples[4 * i + 1] + (e->desc.model == REC_REBUS ? 1 : 0);
            if (tp < 1 || tp >= nu) { rec_set_error("tuple " + std::to_string(i) + ": target position out of range"); return -1; }
        }
    }
    ENG_HIP(hipMemcpyAsync(dev_buf, tuples, sizeof(int4) * n, hipMemcpyHostToDevice, e->st));
    hipEvent_t a = nullptr, b = nullptr;
    if (mode == REC_REPLAY_SEQUENTIAL) {
        if (int rc = launch_sgd_sequential(e->dm, e->desc.precision, dev_buf, n, lr, e->st)) return rc;
        ENG_HIP(hipStreamSynchronize(e->st));
    } else if (mode == REC_REPLAY_HOGWILD) {
        // the batched history kernel takes every aligned group of user_batch tuples as one user's: tuples that do not
        // come that way are walked one by one
        int ub = history_model(e->desc.model) ? e->user_batch : 1;
        int64_t chunk = e->chunk;
        for (int64_t i = 0; i < n && ub > 1; i++)
            if (tuples[4 * i] != tuples[4 * (i - i % ub)]) ub = 1;
        if (ub > 1 || e->neg_share > 1 || e->pos_share > 1) chunk = std::max<int64_t>(8, chunk & ~(int64_t)7);
        for (int64_t o = 0; o < n; o += chunk) {
            const int64_t len = std::min(chunk, n - o);
            if (timed) { if (int rc = ev_begin(e, 0, &a, &b)) return rc; }
            if (int rc = launch_update(e, dev_buf + o, len, lr, false, ub)) return rc;
            if (timed) { if (int rc = ev_end(e, 0, a, b)) return rc; }
        }
        // the host buffer may be reused by the caller: the copy above must have completed
        ENG_HIP(hipStreamSynchronize(e->st));
    } else {
        rec_set_error("unknown replay mode"); return -1;
    }
    return 0;
}

extern "C" int rec_sgd_replay(rec_engine* e, double lr, const int32_t* tuples, int64_t n, int mode)
{
    if (!e || (!tuples && n > 0)) { rec_set_error("null argument"); return -1; }
    if (e->desc.model == REC_MP) { rec_set_error("MostPopular has nothing to train"); return -1; }
    if (n <= 0) return 0;
    ENG_HIP(hipSetDevice(e->desc.device));
    if (int rc = ensure_tuples(e, n)) return rc;
    return replay_tuples(e, lr, tuples, n, mode, e->tuples, true);
}

extern "C" int rec_sgd_replay_static(rec_engine* e, double lr, const int32_t* tuples, int64_t n, int mode)
{
    if (!e || (!tuples && n > 0)) { rec_set_error("null argument"); return -1; }
    if (e->desc.model == REC_MP) { rec_set_error("MostPopular has nothing to train"); return -1; }
    if (n <= 0) return 0;
    if (n > REC_STATIC_MAX_TUPLES) { rec_set_error("rec_sgd_replay_static takes at most " + std::to_string(REC_STATIC_MAX_TUPLES) + " tuples per call"); return -1; }
    ENG_HIP(hipSetDevice(e->desc.device));
    int4* buf = g_scratch_ptr[e->desc.device];   // looked up by rec_create on this device
    if (!buf) { rec_set_error("the static scratch of this device has not been looked up"); return -2; }
    // work enqueued earlier on either stream may still read the engine's own buffers, never this one; calls on one
    // engine do not overlap, and this call returns only after its kernels have finished
    ENG_HIP(hipStreamSynchronize(e->st2));
    return replay_tuples(e, lr, tuples, n, mode, buf, false);
}

extern "C" int rec_sgd_epoch(rec_engine* e, double lr, int64_t n_steps, uint64_t seed, uint64_t epoch)
{
    if (!e) { rec_set_error("null engine"); return -1; }
    if (e->desc.model == REC_MP) { rec_set_error("MostPopular has nothing to train"); return -1; }
    if (n_steps <= 0) return 0;
    ENG_HIP(hipSetDevice(e->desc.device));
    const int ub = history_model(e->desc.model) ? e->user_batch : 1;
    int64_t chunk = std::min(e->chunk, n_steps);
    if (ub > 1 || e->neg_share > 1 || e->pos_share > 1) chunk = (chunk + 7) & ~(int64_t)7;   // chunks start on a group boundary
    // event-major chunks have a Poisson-distributed length: room for 8 standard deviations
    const bool grouped = e->group >= 2 && sgd_runs_supported(e->desc.model) && (e->stale_window == 0 || e->stale_window >= chunk);
    const bool paired = !grouped && e->pos_share == 2 && sgd_runs_supported(e->desc.model);
    if (grouped || paired) { if (int rc = ensure_events(e)) return rc; }
    const int64_t cap = grouped ? chunk + 8 * (int64_t)std::sqrt((double)chunk) + 1024 : chunk;
    const bool fused = can_fuse_sampler(e);
    if (fused) {
        if (e->dm.n_elig <= 0) { rec_set_error("no user (or source item) is eligible for sampling"); return -1; }
        for (int64_t o = 0; o < n_steps; o += chunk) {
            const int64_t len = std::min(chunk, n_steps - o);
            const SampledSource src = {o, seed, epoch};
            hipEvent_t a = nullptr, b = nullptr;
            if (int rc = ev_begin(e, 0, &a, &b)) return rc;
            if (int rc = launch_update(e, nullptr, len, lr, false, 1, &src)) return rc;
            if (int rc = ev_end(e, 0, a, b)) return rc;
            e->tuples_drawn += len;
        }
        return 0;
    }
    if (e->tuples_cap < cap) { if (int rc = ensure_tuples(e, cap)) return rc; }
    hipEvent_t a = nullptr, b = nullptr;
    // By default the sampler of a chunk runs on the update stream, ahead of its update: sampling chunk c+1 next to
    // the update of chunk c (option "overlap") was measured slower, both kernels being bound by random memory traffic.
    hipStream_t ss = e->overlap ? e->st2 : e->st;
    int c = 0;
    for (int64_t o = 0; o < n_steps; o += chunk, c++) {
        int64_t len = std::min(chunk, n_steps - o);
        const int buf = e->overlap ? (c & 1) : 0;
        int4* tb = e->tuples + (int64_t)buf * e->tuples_cap;
        // the buffer must have been consumed by the update that last read it
        if (e->overlap && e->consumed_valid[buf]) ENG_HIP(hipStreamWaitEvent(ss, e->ev_consumed[buf], 0));
        if (e->profile) { ENG_HIP(hipEventCreate(&a)); ENG_HIP(hipEventCreate(&b)); ENG_HIP(hipEventRecord(a, ss)); }
        if (grouped) {
            if (int rc = launch_event_sampler(e->dm, tb, e->tuples_cap, len, o, (uint64_t)c, seed, epoch, e->group_scratch, e->n_drawn_dev, ss)) return rc;
        } else if (paired) {
            if (int rc = launch_pair_sampler(e->dm, tb, len, o, seed, epoch, ss)) return rc;
        } else {
            if (int rc = launch_sampler(e->dm, tb, len, o, seed, epoch, ss, ub, history_model(e->desc.model) ? 1 : e->neg_share)) return rc;
        }
        if (e->profile) { ENG_HIP(hipEventRecord(b, ss)); e->pending[2].push_back({a, b}); }
        if (grouped) {
            // the chunk's length is known once it has been drawn
            int64_t drawn = 0;
            ENG_HIP(hipMemcpyAsync(&drawn, e->n_drawn_dev, sizeof drawn, hipMemcpyDeviceToHost, ss));
            ENG_HIP(hipStreamSynchronize(ss));
            len = std::min(drawn, e->tuples_cap);
        }
        if (e->overlap) {
            ENG_HIP(hipEventRecord(e->ev_sampled[buf], ss));
            ENG_HIP(hipStreamWaitEvent(e->st, e->ev_sampled[buf], 0));
        }
        if (int rc = ev_begin(e, 0, &a, &b)) return rc;
        if (int rc = launch_update(e, tb, len, lr, grouped && e->group == 2, ub)) return rc;
        if (int rc = ev_end(e, 0, a, b)) return rc;
        if (e->overlap) { ENG_HIP(hipEventRecord(e->ev_consumed[buf], e->st)); e->consumed_valid[buf] = true; }
        e->tuples_drawn += len;
    }
    return 0;
}

extern "C" int rec_sample_events(rec_engine* e, int64_t n_expected, uint64_t seed, uint64_t epoch, int64_t cap, int32_t* out, int64_t* n_out)
{
    if (!e || !out || !n_out) { rec_set_error("null argument"); return -1; }
    if (n_expected <= 0 || cap <= 0) { rec_set_error("n_expected and cap must be positive"); return -1; }
    ENG_HIP(hipSetDevice(e->desc.device));
    if (int rc = ensure_events(e)) return rc;
    if (int rc = ensure_tuples(e, cap)) return rc;
    if (int rc = launch_event_sampler(e->dm, e->tuples, cap, n_expected, 0, 0, seed, epoch, e->group_scratch, e->n_drawn_dev, e->st)) return rc;
    int64_t drawn = 0;
    ENG_HIP(hipMemcpyAsync(&drawn, e->n_drawn_dev, sizeof drawn, hipMemcpyDeviceToHost, e->st));
    ENG_HIP(hipStreamSynchronize(e->st));
    *n_out = std::min(drawn, cap);
    ENG_HIP(hipMemcpy(out, e->tuples, sizeof(int4) * (size_t)*n_out, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int rec_owner_slices(rec_engine* e)
{
    if (!e) { rec_set_error("null argument"); return -1; }
    if (!sgd_owner_supported(e->desc.model)) return 0;
    return !(e->wide && sgd_wide_supported(e->dm, e->desc.precision)) ? sgd_owner_slices() : e->owner_slices ? e->owner_slices : sgd_wide_owner_slices(e->dm);
}

extern "C" int rec_partition_tuples(rec_engine* e, const int32_t* tuples, int64_t n, int32_t slices, int32_t* out, int64_t* qoff_out)
{
    if (!e || !tuples || !out || !qoff_out) { rec_set_error("null argument"); return -1; }
    if (!sgd_owner_supported(e->desc.model)) { rec_set_error("the owner mode is not available for this model"); return -1; }
    if (slices != 256 && slices != 512 && slices != 1024) { rec_set_error("slices must be 256, 512 or 1024"); return -1; }
    if (n <= 0) return 0;
    ENG_HIP(hipSetDevice(e->desc.device));
    if (int rc = check_tuples(e, tuples, n)) return rc;
    if (int rc = ensure_tuples(e, n)) return rc;
    if (int rc = ensure_partition_scratch(e, n, slices)) return rc;
    ENG_HIP(hipMemcpyAsync(e->tuples, tuples, sizeof(int4) * n, hipMemcpyHostToDevice, e->st));
    if (int rc = launch_partition_by_slice(e->tuples, e->tuples + e->tuples_cap, n, sgd_owner_neg_field(e->desc.model), (e->dm.I + slices - 1) / slices,
                                           slices, e->part_scratch, e->qoff_dev, e->st, e->part_tile)) return rc;
    ENG_HIP(hipMemcpyAsync(out, e->tuples + e->tuples_cap, sizeof(int4) * n, hipMemcpyDeviceToHost, e->st));
    ENG_HIP(hipMemcpyAsync(qoff_out, e->qoff_dev, sizeof(int64_t) * (slices + 1), hipMemcpyDeviceToHost, e->st));
    ENG_HIP(hipStreamSynchronize(e->st));
    return 0;
}

extern "C" int rec_sample_tuples(rec_engine* e, int64_t n, uint64_t seed, uint64_t epoch, int32_t* out)
{
    if (!e || !out) { rec_set_error("null argument"); return -1; }
    if (n <= 0) return 0;
    ENG_HIP(hipSetDevice(e->desc.device));
    if (int rc = ensure_tuples(e, n)) return rc;
    if (e->pos_share == 2 && sgd_runs_supported(e->desc.model)) {
        if (int rc = ensure_events(e)) return rc;
        if (int rc = launch_pair_sampler(e->dm, e->tuples, n, 0, seed, epoch, e->st)) return rc;
    } else if (int rc = launch_sampler(e->dm, e->tuples, n, 0, seed, epoch, e->st, history_model(e->desc.model) ? e->user_batch : 1, history_model(e->desc.model) ? 1 : e->neg_share)) return rc;
    ENG_HIP(hipMemcpyAsync(out, e->tuples, sizeof(int4) * n, hipMemcpyDeviceToHost, e->st));
    ENG_HIP(hipStreamSynchronize(e->st));
    return 0;
}

// model.cpp:49-97 on per-user counts, in user order (the reference's summation order)
static void reduce_metrics(const rec_engine* e, const int32_t* counts, int topn, double sums[8])
{
    for (int i = 0; i < 8; i++) sums[i] = 0;
    for (int u = e->desc.user_begin; u < e->desc.user_end; u++) {
        const int cv = counts[3 * (int64_t)u], ct = counts[3 * (int64_t)u + 1], mx = counts[3 * (int64_t)u + 2];
        if (mx < 0) continue;
        const double at = 1.0 * ct / mx;
        sums[0] += 1.0 * cv / mx; sums[1] += at;
        int rank = mx - cv + 1; sums[4] += 1.0 / rank; sums[2] += (rank <= topn);
        rank = mx - ct + 1; sums[5] += 1.0 / rank; sums[3] += (rank <= topn);
        sums[6] += at * at; sums[7] += 1;
    }
}

static int eval_counts(rec_engine* e, std::vector<int32_t>& counts, double* xpos, const int32_t* subset = nullptr, int n_subset = 0,
                       int user_draws = 0, uint64_t draw_seed = 0, int32_t* draws = nullptr)
{
    ENG_HIP(hipSetDevice(e->desc.device));
    ENG_HIP(hipStreamSynchronize(e->st));
    counts.resize((size_t)e->U * 3);
    return eval_run(e->dm, e->desc.precision, counts.data(), xpos, e->st, &e->ms[1], &e->launches[1], e->profile != 0, subset, n_subset,
                    user_draws, draw_seed, draws);
}

static void metrics_from_counts(const rec_engine* e, const std::vector<int32_t>& counts, int topn, rec_metrics* out)
{
    double s[8];
    reduce_metrics(e, counts.data(), topn, s);
    const double n = s[7];
    out->n_users = (int64_t)n;
    out->auc_val = s[0] / n; out->auc_test = s[1] / n; out->hit_val = s[2] / n; out->hit_test = s[3] / n;
    out->mrr_val = s[4] / n; out->mrr_test = s[5] / n;
    // two-pass variance as model.cpp:91-97
    double var = 0;
    for (int u = e->desc.user_begin; u < e->desc.user_end; u++) {
        const int mx = counts[3 * (size_t)u + 2];
        if (mx < 0) continue;
        const double dlt = 1.0 * counts[3 * (size_t)u + 1] / mx - out->auc_test;
        var += dlt * dlt;
    }
    out->var_test = var / n;
}

extern "C" int rec_eval_sampled(rec_engine* e, int topn, int32_t n_sampled, uint64_t seed, rec_metrics* out, int32_t* counts_out,
                                int32_t* sampled_out)
{
    if (!e || !out) { rec_set_error("null argument"); return -1; }
    if (n_sampled < 1) { rec_set_error("n_sampled must be positive"); return -1; }
    // one draw of n_sampled items, with replacement, shared by all users (splitmix64)
    std::vector<int32_t> sub(n_sampled);
    uint64_t x = seed + 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < n_sampled; i++) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        sub[i] = (int32_t)((z >> 11) % (uint64_t)e->I);
    }
    std::vector<int32_t> counts;
    if (int rc = eval_counts(e, counts, nullptr, sub.data(), n_sampled)) return rc;
    // a user whose sampled candidates are all excluded has no estimate: leave it out, as a division by max = 0 would
    for (int u = 0; u < e->U; u++)
        if (counts[3 * (size_t)u + 2] == 0) counts[3 * (size_t)u] = counts[3 * (size_t)u + 1] = counts[3 * (size_t)u + 2] = -1;
    metrics_from_counts(e, counts, topn, out);
    if (counts_out) memcpy(counts_out, counts.data(), sizeof(int32_t) * counts.size());
    if (sampled_out) memcpy(sampled_out, sub.data(), sizeof(int32_t) * sub.size());
    return 0;
}

extern "C" int rec_eval_sampled_per_user(rec_engine* e, int topn, int32_t n_sampled, uint64_t seed, rec_metrics* out, int32_t* counts_out,
                                         int32_t* draws_out)
{
    if (!e || !out) { rec_set_error("null argument"); return -1; }
    if (n_sampled < 1 || n_sampled > (1 << 20)) { rec_set_error("n_sampled must be in 1..2^20"); return -1; }
    if (draws_out && (int64_t)e->U * n_sampled > ((int64_t)1 << 31)) { rec_set_error("draws_out: more than 2^31 draws; pass NULL"); return -1; }
    std::vector<int32_t> counts;
    if (int rc = eval_counts(e, counts, nullptr, nullptr, 0, n_sampled, seed, draws_out)) return rc;
    // a user whose draws were all skipped has no estimate (the reference divides by max = 0 there): left out
    for (int u = 0; u < e->U; u++)
        if (counts[3 * (size_t)u + 2] == 0) counts[3 * (size_t)u] = counts[3 * (size_t)u + 1] = counts[3 * (size_t)u + 2] = -1;
    metrics_from_counts(e, counts, topn, out);
    if (counts_out) memcpy(counts_out, counts.data(), sizeof(int32_t) * counts.size());
    return 0;
}

extern "C" int rec_eval(rec_engine* e, int topn, rec_metrics* out, int32_t* counts_out, double* xpos_out)
{
    if (!e || !out) { rec_set_error("null argument"); return -1; }
    std::vector<int32_t> counts;
    if (int rc = eval_counts(e, counts, xpos_out)) return rc;
    metrics_from_counts(e, counts, topn, out);
    if (counts_out) memcpy(counts_out, counts.data(), sizeof(int32_t) * counts.size());
    return 0;
}

extern "C" int rec_eval_sums(rec_engine* e, int topn, double sums[8])
{
    if (!e || !sums) { rec_set_error("null argument"); return -1; }
    std::vector<int32_t> counts;
    if (int rc = eval_counts(e, counts, nullptr)) return rc;
    reduce_metrics(e, counts.data(), topn, sums);
    return 0;
}

extern "C" int64_t rec_eval_user_count(const rec_engine* e) { return e ? e->dm.n_eval : -1; }

extern "C" int rec_eval_topk(rec_engine* e, const int32_t* users, int32_t n_users, int32_t k, int32_t* items_out, double* scores_out)
{
    if (!e || !items_out) { rec_set_error("null argument"); return -1; }
    if (n_users <= 0) return 0;
    ENG_HIP(hipSetDevice(e->desc.device));
    ENG_HIP(hipStreamSynchronize(e->st));
    if (users) {
        std::vector<int32_t> test(e->U);
        ENG_HIP(hipMemcpy(test.data(), e->dm.test_item, sizeof(int32_t) * e->U, hipMemcpyDeviceToHost));
        for (int i = 0; i < n_users; i++)
            if (users[i] < e->desc.user_begin || users[i] >= e->desc.user_end || test[users[i]] < 0) { rec_set_error("user " + std::to_string(users[i]) + " has no validation / test item"); return -1; }
    } else if (n_users > e->dm.n_eval) {
        rec_set_error("n_users exceeds the " + std::to_string(e->dm.n_eval) + " users that have a test item"); return -1;
    }
    return eval_topk(e->dm, e->desc.precision, users, n_users, k, items_out, scores_out, e->st);
}

// ------------------------------------------------------------------------------------------------
// User-sharded training inside one process: N engines with consecutive user ranges, on one device (all of them) or on
// N different devices.  A round = every shard applies its share of tuples to its own replica of the item tables, then
//     table = base + sum_r (table_r - base)          (base: the common value at the start of the round)
// is formed and written to every replica: rec_shards_begin_round keeps the base, rec_shards_combine forms the sum -
// in one kernel when the shards share a device, else as delta_r = table_r - base, one RCCL all-reduce (sum) per table
// over the shards' devices, table_r = base + sum.  RCCL is loaded when the first multi-device combine needs it.
#include <dlfcn.h>
// The few RCCL names the multi-device branch uses, declared here so that a build (and a single-GPU run) needs neither the
// RCCL headers nor the library: the values are those of rccl.h (ncclSuccess 0, ncclSum 0, ncclFloat32 7, ncclFloat64 8).
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
enum { ncclSuccess = 0, ncclSum = 0, ncclFloat32 = 7, ncclFloat64 = 8 };
namespace {
struct ShardPtrs { void* tab[16]; };
template <class T>
__global__ void shard_combine_local_kernel(ShardPtrs p, const T* __restrict__ base, int n_shards, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T b = base[i];
        T s = b;
        for (int r = 0; r < n_shards; r++) s += ((const T*)p.tab[r])[i] - b;
        for (int r = 0; r < n_shards; r++) ((T*)p.tab[r])[i] = s;
    }
}
template <class T>
__global__ void shard_axpy_kernel(T* __restrict__ tab, const T* __restrict__ base, T sign, int64_t n)   // tab += sign * base
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) tab[i] += sign * base[i];
}
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<int> devs; std::vector<ncclComm_t> comms;   // the communicators of the last device list (kept until the process ends)
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
int rccl_comms(const std::vector<int>& devs, ncclComm_t** out)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    Rccl& r = g_rccl;
    if (!r.lib) {
        r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { rec_set_error(std::string("shards on several devices need RCCL: ") + dlerror()); return -2; }
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
            rec_set_error("librccl lacks an expected symbol"); (void)dlclose(r.lib); r.lib = nullptr; return -2;
        }
    }
    if (r.devs != devs) {
        // another device list than last time: the old communicators go first
        for (ncclComm_t c : r.comms) if (c) (void)r.CommDestroy(c);
        r.devs.clear();
        r.comms.assign(devs.size(), nullptr);
        const ncclResult_t rc = r.CommInitAll(r.comms.data(), (int)devs.size(), devs.data());
        if (rc != ncclSuccess) { rec_set_error(std::string("ncclCommInitAll: ") + r.GetErrorString(rc)); r.comms.clear(); return -2; }
        r.devs = devs;
    }
    *out = r.comms.data();
    return 0;
}
int check_shards(rec_engine* const* shards, int n, bool* one_device)
{
    if (!shards || n < 1 || n > 16) { rec_set_error("1..16 shards"); return -1; }
    for (int r = 0; r < n; r++) {
        if (!shards[r]) { rec_set_error("null shard"); return -1; }
        if (shards[r]->nw != shards[0]->nw || shards[r]->desc.model != shards[0]->desc.model || shards[r]->desc.precision != shards[0]->desc.precision ||
            shards[r]->tables.size() != shards[0]->tables.size()) { rec_set_error("the shards are not engines of one model"); return -1; }
    }
    bool same = true, distinct = true;
    for (int r = 0; r < n; r++)
        for (int q = 0; q < r; q++) { if (shards[r]->desc.device != shards[q]->desc.device) same = false; else distinct = false; }
    if (n > 1 && !same && !distinct) { rec_set_error("shards must share one device or sit on one device each"); return -1; }
    *one_device = same;
    return 0;
}
}  // namespace

extern "C" int rec_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { rec_set_error("hipGetDeviceCount failed"); return -1; }
    return n;
}

extern "C" int rec_shards_begin_round(rec_engine* const* shards, int n)
{
    bool one_device;
    if (int rc = check_shards(shards, n, &one_device)) return rc;
    // The projection of TransRec's item rows is not additive: done by every shard after its launches it would enter the
    // sum of deltas once per shard (a row outside the ball is pulled back N times over and overshoots).  The shards'
    // launches only add gradient steps (rows are projected as they are read); the combined table is projected once.
    for (int r = 0; r < n; r++) shards[r]->project_after_launch = 0;
    // shards on one device hold the same values between rounds: the first one's copy serves all of them
    for (int r = 0; r < (one_device ? 1 : n); r++) {
        rec_engine* e = shards[r];
        ENG_HIP(hipSetDevice(e->desc.device));
        if (e->round_base.empty())
            for (const TableDesc& t : e->tables) {
                void* p = nullptr;
                if (!t.is_user) { ENG_HIP(hipMalloc(&p, (size_t)std::max<int64_t>(t.rows * t.cols, 1) * e->elem_size)); e->owned.push_back(p); }
                e->round_base.push_back(p);
            }
        for (size_t t = 0; t < e->tables.size(); t++)
            if (!e->tables[t].is_user)
                ENG_HIP(hipMemcpyAsync(e->round_base[t], (char*)e->W + e->tables[t].dev_off * e->elem_size, (size_t)(e->tables[t].rows * e->tables[t].cols) * e->elem_size,
                                       hipMemcpyDeviceToDevice, e->st));
    }
    // the base has been taken before any shard's stream starts the round (shards on one device share the first one's copy)
    for (int r = 0; r < (one_device ? 1 : n); r++) {
        ENG_HIP(hipSetDevice(shards[r]->desc.device));
        ENG_HIP(hipStreamSynchronize(shards[r]->st));
    }
    return 0;
}

extern "C" int rec_shards_combine(rec_engine* const* shards, int n)
{
    bool one_device;
    if (int rc = check_shards(shards, n, &one_device)) return rc;
    if (n == 1) return 0;
    rec_engine* e0 = shards[0];
    const bool f32 = e0->desc.precision == REC_F32;
    for (int r = 0; r < (one_device ? 1 : n); r++)
        if (shards[r]->round_base.empty()) { rec_set_error("rec_shards_combine without rec_shards_begin_round"); return -1; }
    if (one_device) {
        ENG_HIP(hipSetDevice(e0->desc.device));
        for (int r = 0; r < n; r++) ENG_HIP(hipStreamSynchronize(shards[r]->st));
        for (size_t t = 0; t < e0->tables.size(); t++) {
            if (e0->tables[t].is_user) continue;
            const int64_t cnt = e0->tables[t].rows * e0->tables[t].cols;
            ShardPtrs p = {};
            for (int r = 0; r < n; r++) p.tab[r] = (char*)shards[r]->W + shards[r]->tables[t].dev_off * e0->elem_size;
            const unsigned blocks = (unsigned)std::min<int64_t>(4096, (cnt + 255) / 256);
            if (f32) hipLaunchKernelGGL(shard_combine_local_kernel<float>, dim3(blocks), dim3(256), 0, e0->st, p, (const float*)e0->round_base[t], n, cnt);
            else hipLaunchKernelGGL(shard_combine_local_kernel<double>, dim3(blocks), dim3(256), 0, e0->st, p, (const double*)e0->round_base[t], n, cnt);
        }
        ENG_HIP(hipGetLastError());
        ENG_HIP(hipStreamSynchronize(e0->st));
        for (int r = 0; r < n; r++) { if (int rc = launch_project_rows(shards[r]->dm, shards[r]->desc.precision, shards[r]->st)) return rc; }
        for (int r = 0; r < n; r++) ENG_HIP(hipStreamSynchronize(shards[r]->st));
        return 0;
    }
    std::vector<int> devs(n);
    for (int r = 0; r < n; r++) devs[r] = shards[r]->desc.device;
    ncclComm_t* comms = nullptr;
    if (int rc = rccl_comms(devs, &comms)) return rc;
    for (size_t t = 0; t < e0->tables.size(); t++) {
        if (e0->tables[t].is_user) continue;
        const int64_t cnt = e0->tables[t].rows * e0->tables[t].cols;
        const unsigned blocks = (unsigned)std::min<int64_t>(4096, (cnt + 255) / 256);
        for (int pass = 0; pass < 3; pass++) {
            if (pass == 1) { const ncclResult_t rc = g_rccl.GroupStart(); if (rc != ncclSuccess) { rec_set_error(std::string("ncclGroupStart: ") + g_rccl.GetErrorString(rc)); return -2; } }
            for (int r = 0; r < n; r++) {
                rec_engine* e = shards[r];
                ENG_HIP(hipSetDevice(e->desc.device));
                void* tab = (char*)e->W + e->tables[t].dev_off * e->elem_size;
                if (pass == 1) {
                    const ncclResult_t rc = g_rccl.AllReduce(tab, tab, (size_t)cnt, f32 ? ncclFloat32 : ncclFloat64, ncclSum, comms[r], e->st);
                    if (rc != ncclSuccess) { rec_set_error(std::string("ncclAllReduce: ") + g_rccl.GetErrorString(rc)); (void)g_rccl.GroupEnd(); return -2; }
                } else {
                    // pass 0: table -= base (this shard's delta); pass 2: table += base
                    if (f32) hipLaunchKernelGGL(shard_axpy_kernel<float>, dim3(blocks), dim3(256), 0, e->st, (float*)tab, (const float*)e->round_base[t], pass == 0 ? -1.f : 1.f, cnt);
                    else hipLaunchKernelGGL(shard_axpy_kernel<double>, dim3(blocks), dim3(256), 0, e->st, (double*)tab, (const double*)e->round_base[t], pass == 0 ? -1.0 : 1.0, cnt);
                    ENG_HIP(hipGetLastError());
                }
            }
            if (pass == 1) { const ncclResult_t rc = g_rccl.GroupEnd(); if (rc != ncclSuccess) { rec_set_error(std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(rc)); return -2; } }
        }
    }
    for (int r = 0; r < n; r++) {
        ENG_HIP(hipSetDevice(shards[r]->desc.device));
        if (int rc = launch_project_rows(shards[r]->dm, shards[r]->desc.precision, shards[r]->st)) return rc;
    }
    for (int r = 0; r < n; r++) { ENG_HIP(hipSetDevice(shards[r]->desc.device)); ENG_HIP(hipStreamSynchronize(shards[r]->st)); }
    return 0;
}

extern "C" int rec_project_rows(rec_engine* e)
{
    if (!e) { rec_set_error("null engine"); return -1; }
    ENG_HIP(hipSetDevice(e->desc.device));
    if (int rc = launch_project_rows(e->dm, e->desc.precision, e->st)) return rc;
    ENG_HIP(hipStreamSynchronize(e->st));
    return 0;
}

extern "C" int rec_num_tables(const rec_engine* e) { return e ? (int)e->tables.size() : -1; }

extern "C" int rec_table_info(const rec_engine* e, int t, void** dev_ptr, int64_t* rows, int64_t* cols, int* is_user_table)
{
    if (!e || t < 0 || t >= (int)e->tables.size()) { rec_set_error("no such table"); return -1; }
    if (dev_ptr) *dev_ptr = (char*)e->W + (e->tables[t].dev_off - e->tables[t].row0 * e->tables[t].cols) * e->elem_size;
    if (rows) *rows = e->tables[t].rows;
    if (cols) *cols = e->tables[t].cols;
    if (is_user_table) *is_user_table = e->tables[t].is_user;
    return 0;
}

extern "C" int rec_set_option(rec_engine* e, const char* name, int64_t value)
{
    if (!e || !name) { rec_set_error("null argument"); return -1; }
    const std::string s(name);
    if (s == "chunk") { if (value < 1) { rec_set_error("chunk must be positive"); return -1; } e->chunk = value; }
    else if (s == "atomic") e->atomic = value == 1 ? 0x1ff : (int)value;  // 1 = every row; else a mask: bit r = row r of the tuple, bit 8 = biases
    else if (s == "profile") e->profile = value != 0;
    else if (s == "overlap") {
        ENG_HIP(hipSetDevice(e->desc.device));
        ENG_HIP(hipStreamSynchronize(e->st)); ENG_HIP(hipStreamSynchronize(e->st2));
        e->overlap = value != 0; e->consumed_valid[0] = e->consumed_valid[1] = false;
    }
    else if (s == "stale_window") { if (value < 0) { rec_set_error("stale_window must be >= 0"); return -1; } e->stale_window = value; }
    else if (s == "global_cap") { if (value < 1) { rec_set_error("global_cap must be positive"); return -1; } e->global_cap = value; }
    else if (s == "global_window") { if (value < 1) { rec_set_error("global_window must be positive"); return -1; } e->global_window = value; }
    else if (s == "update_policy") {
        // 0 exact: every row and bias is scattered with float atomics (sums of all contributions);
        // 1 hybrid: rows indexed by user and the item biases are stored plainly (a user is rarely in flight twice;
        //   a 4-byte bias atomic costs a whole 64-byte memory-side request and, measured, buys no accuracy), item
        //   rows use atomics, except for the few items drawn so often (>= 1/1024 of the positives) that their
        //   atomics would serialize on one address - those are stored plainly too;
        // 2 plain: classic racy Hogwild stores everywhere
        e->group = 0;
        if (value == 0) { e->atomic = 0x1ff; return rec_set_option(e, "hot_count", 0); }
        if (value == 2) { e->atomic = 0; return rec_set_option(e, "hot_count", 0); }
        if (value != 1) { rec_set_error("update_policy must be 0, 1 or 2"); return -1; }
        e->atomic = e->desc.model == REC_MC ? 0x0ff : 0x0fe;
        return rec_set_option(e, "hot_count", std::max<int64_t>(1, e->steps_per_epoch / 1024));
    }
    else if (s == "run_shared") { if (value < 1) { rec_set_error("run_shared must be positive"); return -1; } e->run_policy.shared = (int)value; }
    else if (s == "run_hot") { if (value < 1) { rec_set_error("run_hot must be positive"); return -1; } e->run_policy.hot = (int)value; }
    else if (s == "run_hot_atomic") e->run_policy.hot_atomic = value != 0;
    else if (s == "event_phases") { if (value < 0 || e->dm.n_ev > 0) { rec_set_error("event_phases must be >= 0 and set before the first event-major epoch"); return -1; } e->event_phases = (int)value; }
    else if (s == "share_positive") {
        if (value != 1 && value != 2) { rec_set_error("share_positive must be 1 or 2"); return -1; }
        if (value == 2 && (!sgd_runs_supported(e->desc.model) || e->event_phases > 0)) { rec_set_error("share_positive 2 is not available for this model (or with event_phases)"); return -1; }
        e->pos_share = (int)value;
    }
    else if (s == "share_negative") { if (value != 1 && value != 2 && value != 4) { rec_set_error("share_negative must be 1, 2 or 4"); return -1; } e->neg_share = (int)value; }
    else if (s == "project_after_launch") e->project_after_launch = value != 0;
    else if (s == "pair_min_history") { if (value < 0 || value > (1 << 20)) { rec_set_error("pair_min_history must be in 0..2^20"); return -1; } e->dm.pair_min_hist = (int)value; }
    else if (s == "user_batch") { if (value != 1 && value != 2 && value != 4 && value != 8) { rec_set_error("user_batch must be 1, 2, 4 or 8"); return -1; } e->user_batch = (int)value; }
    else if (s == "fused_sampler") { if (value != 0 && value != 1) { rec_set_error("fused_sampler must be 0 or 1"); return -1; } e->fused_sampler = (int)value; }
    else if (s == "wide") { if (value != 0 && value != 1) { rec_set_error("wide must be 0 or 1"); return -1; } e->wide = (int)value; }
    else if (s == "owner_neg") { if (value != 0 && value != 1) { rec_set_error("owner_neg must be 0 or 1"); return -1; } e->owner = (int)value; }
    else if (s == "owner_flush") { if (value < 1 || value > 65536) { rec_set_error("owner_flush must be in 1..65536"); return -1; } e->owner_flush = (int)value; }
    else if (s == "owner_per_item") { if (value < 1) { rec_set_error("owner_per_item must be positive"); return -1; } e->owner_per_item = value; }
    else if (s == "eval_screen") { if (value < 0 || value > 2) { rec_set_error("eval_screen must be 0, 1 or 2"); return -1; } eval_configure_screen((int)value, -1); }
    else if (s == "topk_fused") { if (value != 0 && value != 1) { rec_set_error("topk_fused must be 0 or 1"); return -1; } eval_configure_topk((int)value); }
    else if (s == "eval_ranges") { if (value < 0 || value > 64) { rec_set_error("eval_ranges must be in 0..64"); return -1; } eval_configure_ranges((int)value); }
    else if (s == "eval_screen_cap") { if (value < 0 || value > 4096) { rec_set_error("eval_screen_cap must be in 0..4096"); return -1; } eval_configure_screen(-1, (int)value); }
    else if (s == "owner_slices") { if (value != 0 && value != 256 && value != 512 && value != 1024) { rec_set_error("owner_slices must be 0, 256, 512 or 1024"); return -1; } e->owner_slices = (int)value; }
    else if (s == "part_tile") { if (value != 0 && value != 4096 && value != 7168) { rec_set_error("part_tile must be 0, 4096 or 7168"); return -1; } e->part_tile = (int)value; }
    else if (s == "owner_min") { if (value < 1) { rec_set_error("owner_min must be positive"); return -1; } e->owner_min = value; }
    else if (s == "run_segment") { if (value < 0) { rec_set_error("run_segment must be >= 0"); return -1; } e->run_segment = value; }
    else if (s == "group_by_item") { if (value < 0 || value > 3) { rec_set_error("group_by_item must be 0..3"); return -1; } e->group = (int)value; }
    else if (s == "hot_count") {
        // items with at least `value` training occurrences are "hot" (0 = none)
        ENG_HIP(hipSetDevice(e->desc.device));
        ENG_HIP(hipStreamSynchronize(e->st));
        const int I = e->I;
        std::vector<uint8_t> flags((size_t)I + 1, 0);
        if (value > 0)
            for (int i = 0; i < I; i++) if (e->train_count_host[i] >= value) flags[i] = 1;
        ENG_HIP(hipMemcpy((void*)e->dm.hot, flags.data(), flags.size(), hipMemcpyHostToDevice));
    }
    else { rec_set_error("unknown option " + s); return -1; }
    return 0;
}

extern "C" int rec_kernel_stats(rec_engine* e, int which, double* total_ms, int64_t* launches)
{
    if (!e || which < 0 || which > 6) { rec_set_error("bad argument"); return -1; }
    if (which == 6) {
        if (total_ms) *total_ms = (double)e->user_batch;
        if (launches) *launches = e->global_window;
        return 0;
    }
    ENG_HIP(hipSetDevice(e->desc.device));
    ENG_HIP(hipStreamSynchronize(e->st));
    if (which == 4) {
        int mode = 0; int64_t open = 0;
        eval_screen_stats(&mode, &open);
        if (total_ms) *total_ms = (double)open;
        if (launches) *launches = mode;
        return 0;
    }
    if (which == 5) {
        int64_t lists = 0; double ms = 0;
        eval_topk_stats(&lists, &ms);
        if (total_ms) *total_ms = ms;
        if (launches) *launches = lists;
        return 0;
    }
    for (auto& p : e->pending[which]) {
        float ms = 0;
        ENG_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        e->ms[which] += ms; e->launches[which] += 1;
        (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
    }
    e->pending[which].clear();
    if (total_ms) *total_ms = e->ms[which];
    if (launches) *launches = e->launches[which];
    e->ms[which] = 0; e->launches[which] = 0;
    return 0;
}
