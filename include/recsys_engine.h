/* C ABI of the MI355X engine for the BPR-SGD step and the full-catalog evaluation of the
 * sequential recommenders of visiativ-innovation/recommendation.
 *
 * The reference has no FFI: its seam is the C++ class API (src/model.hpp:9-92 and the per-model
 * classes in src/models/).  Each entry point below names the reference member(s) it replaces.
 * Plain pointers and sizes only; every function returns 0 on success and a negative code on
 * failure, with a message available from rec_last_error().  Nothing throws across this boundary.
 *
 * Threading: calls on one rec_engine must not overlap.  rec_sgd_epoch and rec_sgd_replay in
 * Hogwild mode only enqueue work on the engine's private HIP stream; every other call waits
 * for that stream first, and rec_sync waits explicitly. */
#ifndef RECSYS_ENGINE_H
#define RECSYS_ENGINE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rec_engine rec_engine;

/* CLI model names of src/main.cpp:188-212 */
enum rec_model {
    REC_MP = 0, REC_BPRMF = 1, REC_MC = 2, REC_FPMC = 3, REC_PRME = 4, REC_HRM_AVG = 5,
    REC_HRM_MAX = 6, REC_TRANSREC = 7, REC_TRANSREC_L1 = 8, REC_FOSSIL_SIMPLE = 9,
    REC_FOSSIL = 10, REC_REBUS = 11
};

enum rec_precision { REC_F32 = 0, REC_F64 = 1 };

/* How rec_sgd_replay applies the tuples */
enum rec_replay_mode {
    REC_REPLAY_SEQUENTIAL = 0, /* deterministic batch-1: one wavefront, tuples strictly in order,
                                  in-place semantics of the reference's updateFactors */
    REC_REPLAY_HOGWILD = 1     /* one wavefront (or sub-wavefront lane group) per tuple, float-atomic scatter */
};

/* What the model constructor leaves behind (src/model.hpp:12-53): the post-split training
 * sequences corp->pos_per_user (items only, chronological) and test_per_user / val_per_user.
 * All arrays are copied by rec_create; no host pointer is kept. */
typedef struct {
    int32_t n_users, n_items;
    const int64_t* seq_off;   /* [n_users+1] */
    const int32_t* seq_item;  /* [seq_off[n_users]] */
    const int32_t* test_item; /* [n_users], -1 = user without validation/test items */
    const int32_t* val_item;  /* [n_users] */
    /* REBUS only (NULL otherwise): the matched path of REBUS::findPathStars for every context.
     * The context of user u with a prefix of c items (c = 1..n_u+1) has index
     * seq_off[u] + 2*u + (c-1); c <= n_u-1 are the training positions (histo_user,
     * REBUS.cpp:132-153), c = n_u the validation and c = n_u+1 the test context
     * (REBUS.cpp:758-782).  rec_rebus_build_paths produces it. */
    const int64_t* path_off;  /* [seq_off[n_users] + 2*n_users + 1] */
    const int32_t* path_item;
} rec_corpus;

/* Constructor arguments of the model classes plus the constants the reference hard-codes */
typedef struct {
    int32_t model;            /* enum rec_model */
    int32_t K, KK, L;         /* KK = K in src/main.cpp:197,203 */
    double lambda, bias_reg, relation_reg;
    double alpha;             /* PRME alpha / REBUS alpha_up (CLI argument 10) */
    int32_t dataset_factor;   /* Fossil.hpp:22; 0 selects the reference's 10 */
    int32_t precision;        /* enum rec_precision: storage and arithmetic type of the SGD step */
    int32_t device;           /* HIP device ordinal */
    int32_t user_begin, user_end; /* users this engine samples and evaluates (MC samples the same share of the source items);
                                     0,0 = all.  Rows indexed by user, sequences, click sets and REBUS paths of other users
                                     are not kept on the device, and tuples naming them are rejected. */
} rec_model_desc;

/* Outputs of model::AUC (src/model.cpp:3-98) */
typedef struct {
    double auc_val, auc_test, hit_val, hit_test, mrr_val, mrr_test, var_test;
    int64_t n_users; /* users with a test item in [user_begin, user_end) */
} rec_metrics;

const char* rec_last_error(void);

/* replaces: model ctor's outputs + <Model>::init's allocation (e.g. BPRMF.cpp:3-9) */
int rec_create(const rec_corpus* corpus, const rec_model_desc* desc, rec_engine** out);
int rec_destroy(rec_engine* e);
int rec_sync(rec_engine* e);

/* NW of the model (e.g. BPRMF.cpp:5); the flat order of W is the reference's and is the saved-model format */
int64_t rec_nw(const rec_engine* e);

/* replaces: direct writes/reads of model::W (src/model.hpp:60-62; saveModel/loadModel go through it) */
int rec_params_upload(rec_engine* e, const double* W, int64_t nw);
int rec_params_download(rec_engine* e, double* W, int64_t nw);

/* a slice W[first, first+n) of the flat parameter vector (what the reference reads as W[i], e.g. the
 * "r = ..." / "decay_wt_0 = ..." progress lines of TransRec.cpp:112-124 and Fossil.cpp:260) */
int rec_params_read(rec_engine* e, int64_t first, int64_t n, double* out);

/* replaces: <Model>::init's random fill (e.g. BPRMF.cpp:11-21, TransRec.cpp:14-30) with a
 * counter-based generator of the same distribution */
int rec_params_init(rec_engine* e, uint64_t seed);

/* replaces: model::copyBestModel (src/model.cpp:169-174) and the "W <- bestW" loop of train() (e.g. BPRMF.cpp:189-191) */
int rec_snapshot_best(rec_engine* e);
int rec_restore_best(rec_engine* e);

/* replaces: n consecutive <Model>::updateFactors calls (e.g. BPRMF.cpp:85-105) on the given
 * tuples (4 x int32 each):
 *   BPRMF (u,pos,neg,-)  MC (item,pos,neg,-)  FPMC/PRME/FossilSimple/HRM_* (u,prev,pos,neg)
 *   TransRec* (u,x,y,yn)  Fossil (u,t,neg,-) and REBUS (u,t-1,neg,-) with target seq[u][t] */
int rec_sgd_replay(rec_engine* e, double lr, const int32_t* tuples, int64_t n, int mode);

/* No counterpart in the reference.  rec_sgd_replay for callers that cannot allocate device memory (harnesses that forbid
 * hipMalloc, tight serving loops that must not touch the allocator): the same kernels, with the tuples staged in a fixed
 * 1 MiB device array that is part of the library's code object instead of a buffer grown on demand.  It calls no
 * hipMalloc / hipFree and creates no HIP events ("profile" does not time it); like rec_sgd_replay it copies the tuples
 * from the caller's host memory and waits for its kernels, so it is not meant to be captured into a HIP graph.
 * n <= 65536 tuples per call; larger n, and an unknown mode, are rejected before anything is enqueued.  Every engine on
 * one GPU shares that GPU's array: calls (of any engines) that overlap on one GPU race.  Results are those of
 * rec_sgd_replay on the same tuples, except with "owner_neg" 1, whose partition pass needs the engine's own buffers and is
 * not taken here (the plain Hogwild kernel applies the tuples instead). */
int rec_sgd_replay_static(rec_engine* e, double lr, const int32_t* tuples, int64_t n, int mode);

/* replaces: <Model>::oneiteration / oneIteration (e.g. BPRMF.cpp:107-150): n_steps tuples drawn on
 * the device by a counter-based generator keyed by (seed, epoch, step), applied Hogwild-style */
int rec_sgd_epoch(rec_engine* e, double lr, int64_t n_steps, uint64_t seed, uint64_t epoch);

/* With "group_by_item" = 2 (off by default: item-sorted order slows convergence) a chunk is drawn by event instead
 * of by tuple: every (user, position) the samplers above can pick yields Poisson(n * P(position)) tuples.  The
 * tuples of a chunk are the same i.i.d. draws, grouped by positive item, and their number is Poisson(n) instead
 * of exactly n.  rec_tuples_drawn: the number of tuples rec_sgd_epoch has applied since rec_create. */
int64_t rec_tuples_drawn(const rec_engine* e);

/* the event-major sampler alone: up to cap tuples (4 x int32) to host memory, *n_out of them (mean n_expected) */
int rec_sample_events(rec_engine* e, int64_t n_expected, uint64_t seed, uint64_t epoch, int64_t cap, int32_t* tuples_out, int64_t* n_out);

/* the sampler alone (tests, tuple capture): writes n tuples (4 x int32) to host memory */
int rec_sample_tuples(rec_engine* e, int64_t n, uint64_t seed, uint64_t epoch, int32_t* tuples_out);

/* No counterpart in the reference: the partition step of the update's owner mode ("owner_neg") alone, for inspection.
 * The n tuples ordered by the slice of the catalog their negative item lies in (`slices` = 256, 512 or 1024 slices of
 * ceil(n_items / slices) consecutive items): out receives the tuples, qoff_out[0..slices] where each slice starts in it.
 * rec_owner_slices: the number of slices this engine's update launches use (0: the model has no owner mode). */
int rec_partition_tuples(rec_engine* e, const int32_t* tuples, int64_t n, int32_t slices, int32_t* out, int64_t* qoff_out);
int rec_owner_slices(rec_engine* e);

/* number of updateFactors calls per epoch: num_pos_events (src/model.hpp:47-51), MC: num_pos_trans (MC.cpp:19-22) */
int64_t rec_steps_per_epoch(const rec_engine* e);

/* replaces: model::AUC / Fossil::AUC / REBUS::AUC and model::MultipleMetrics (src/model.cpp:3-98,
 * 239-329).  counts_out (optional): per user count_val, count_test, max (-1,-1,-1 when not
 * evaluated).  xpos_out (optional): per user x_test, x_val. */
int rec_eval(rec_engine* e, int topn, rec_metrics* out, int32_t* counts_out, double* xpos_out);

/* replaces: model::sampleAUC / Fossil::sampleAUC / REBUS::sampleAUC (src/model.cpp:100-167, Fossil.cpp:480-564,
 * REBUS.cpp:625-730) with the reference's sampling: n_sampled candidate draws PER USER (the reference: 2000 from the
 * global rand() stream; here draw d of user u is a counter-based hash of (seed, u, d), uniform over the catalog, with
 * replacement).  A draw that hits a clicked or held-out item is skipped, not replaced (max <= n_sampled); the others
 * are compared under the validation and under the test context.  One wavefront per user gathers and scores its own
 * draws in fp64.  counts_out (optional): as rec_eval, max = the user's number of valid draws (users whose draws were all
 * skipped: -1,-1,-1 and left out of the means).  draws_out (optional): [n_users][n_sampled] the item of every draw
 * (-1 in rows of users that are not evaluated). */
int rec_eval_sampled_per_user(rec_engine* e, int topn, int32_t n_sampled, uint64_t seed, rec_metrics* out, int32_t* counts_out,
                              int32_t* draws_out);

/* The same estimate from ONE draw of n_sampled items, with replacement, shared by all users - computed by the kernels of
 * rec_eval on gathered item rows.  Its mean is the same; its item-sampling error does not average out over users.  Draws that hit a clicked or held-out item are
 * skipped, not replaced (max <= n_sampled), as in the reference.  counts_out (optional): as rec_eval;
 * sampled_out (optional): the n_sampled item ids. */
int rec_eval_sampled(rec_engine* e, int topn, int32_t n_sampled, uint64_t seed, rec_metrics* out, int32_t* counts_out,
                     int32_t* sampled_out);

/* The ranking behind HIT@k (src/model.cpp:32-58): for every listed user, the k best-scored items among the
 * candidates of model::AUC plus the context's positive, for the validation (index 0) and the test (index 1)
 * context.  items_out / scores_out: [n_users][2][k], best first (score descending, item id ascending among equal
 * scores); -1 / -inf when fewer than k exist.  k <= 256.  users = NULL: the first n_users users that have a test item,
 * in ascending order (rec_eval reports how many there are).  scores_out may be NULL.
 * With f32 tables and k <= 64 the lists come out of the full-catalog sweep itself: every query keeps the columns whose
 * 16-bit-pipe score is within a proven bound of its running k-th best (one wavefront selects per query), the survivors
 * are scored in fp64 and ordered exactly - any number of users, no score matrix ("topk_fused" 1, the default).
 * TransRec_L1 (either table precision, k <= 64, K <= 160): a query's list holds (score, item) against its running k-th
 * best, appended to and selected from by the wavefront that scores it - scores of the |.|_1 sweep on fixed-point operands
 * (one v_sad_u32 per term, within a proven bound of the exact ones; what the lists still hold at the end is scored in fp64),
 * or the exact fp64 scores of the fp64 sweep ("eval_screen" 0, operands that are not finite).
 * Otherwise (f64 tables, k > 64, MostPopular, "topk_fused" 0, or a query with more near-ties than its list holds) the
 * scores of a batch of users are written out [rows][n_items] in fp64 and selected from there. */
int rec_eval_topk(rec_engine* e, const int32_t* users, int32_t n_users, int32_t k, int32_t* items_out, double* scores_out);

/* users of this engine's range that model::AUC evaluates (those with a test item, model.cpp:22): what rec_metrics.n_users
 * reports, without an evaluation */
int64_t rec_eval_user_count(const rec_engine* e);

/* Partial sums of rec_eval over this engine's users, for a sum all-reduce across ranks:
 * {sum AUC_val, AUC_test, HIT_val, HIT_test, MRR_val, MRR_test, sum AUC_test^2, n_users} */
int rec_eval_sums(rec_engine* e, int topn, double sums[8]);

/* replaces: REBUS::findPathStars over every context (REBUS.cpp:132-153, 484-524, 758-782).
 * Labels of a 96-FSUB file as integer sequences, oldest item first ("Root" omitted):
 * lab_off[n_labels+1], lab_item[].  Call with path_item = NULL to obtain the size. Host only. */
int64_t rec_rebus_build_paths(int32_t n_users, int32_t n_items, const int64_t* seq_off,
                              const int32_t* seq_item, const int32_t* val_item, int64_t n_labels,
                              const int64_t* lab_off, const int32_t* lab_item, int64_t* path_off,
                              int32_t* path_item);

/* No counterpart in the reference, whose 96-FSUB label files are precomputed and only read (REBUS::loadPST,
 * REBUS.cpp:3-87): the labels of such a file for the given training sequences - every contiguous run of 1..L
 * items (oldest first) occurring at least min_count times - as integer sequences ("Root" omitted), in the layout
 * rec_rebus_build_paths takes.  Returns the number of labels and sets *n_label_items; call with lab_off =
 * lab_item = NULL to obtain the sizes.  Host only. */
int64_t rec_fsub_mine(int32_t n_users, int32_t n_items, const int64_t* seq_off, const int32_t* seq_item, int32_t L,
                      int64_t min_count, int64_t* lab_off, int32_t* lab_item, int64_t* n_label_items);

/* Device tables, for collectives issued by the caller (RCCL through torch.distributed):
 * table t of the model in the reference's flat order.  *dev_ptr points to rows x cols elements
 * of the engine's precision, contiguous.  is_user_table = 1 for tables indexed by user: of those only the rows of the
 * engine's own users [user_begin, user_end) are resident (dev_ptr is where row 0 would be; dev_ptr + user_begin * cols
 * is the first row that exists), and rec_params_upload / rec_params_download move only those rows. */
int rec_num_tables(const rec_engine* e);
int rec_table_info(const rec_engine* e, int t, void** dev_ptr, int64_t* rows, int64_t* cols, int* is_user_table);

/* No counterpart in the reference (one process, one thread).  User-sharded training inside one process: `shards` are n
 * engines of one model created with consecutive user ranges (rec_model_desc.user_begin / user_end), all on one device or
 * one per device (1 <= n <= 16).  A round: rec_shards_begin_round keeps the common value of every table that is not
 * indexed by user; each shard applies its share of the tuples (rec_sgd_epoch); rec_shards_combine sets every replica to
 * base + sum over the shards of (table - base).  Shards on different devices are combined by one RCCL all-reduce (sum)
 * per table over xGMI (librccl is loaded at the first such call), shards on one device by one kernel.  Both calls wait
 * for the shards' streams.  rec_device_count: HIP devices visible to the process. */
int rec_device_count(void);
int rec_shards_begin_round(rec_engine* const* shards, int n);
int rec_shards_combine(rec_engine* const* shards, int n);

/* replaces: TransRec::normalization (TransRec.cpp:245-258) over every item row: rows longer than 1 are scaled back onto
 * the unit ball (nothing to do for the other models).  The update launches do it themselves unless
 * "project_after_launch" is 0 - which sharded training sets, because the projection is not additive: each shard's launches
 * then only add gradient steps, and the combined table is projected once (rec_shards_combine does it; a caller that
 * combines the tables itself, e.g. with RCCL through torch.distributed, calls this afterwards). */
int rec_project_rows(rec_engine* e);

/* Tuning knobs (value 0/1 unless noted):
 *   "owner_neg"     BPRMF / MC / FPMC / PRME / HRM / TransRec, default 0: 1 = a launch of at least "owner_min" tuples is
 *                   first partitioned by the slice of the catalog its negative items lie in, and workgroup s of the update
 *                   applies slice s, summing the negatives' deltas in LDS and moving a row's sum to memory with one
 *                   atomic row per "owner_flush" additions (default 8) instead of one per tuple.  Half the float atomics,
 *                   but an item's steps in the negative's role then come late and out of order within a launch, which
 *                   costs accuracy in proportion to the launch length: launches hold at most "owner_per_item" tuples
 *                   per item of the catalog (default 8; test AUC -0.0035 after 3 epochs on a 1M-user x 200k-item corpus,
 *                   -0.0006 at 1, -0.07 with one launch per epoch)
 *   "owner_flush", "owner_min", "owner_per_item"   see above (defaults 8, 2^20 and 8)
 *   "owner_slices"  slices of the catalog in owner mode: 0 (default) = the fewest of 256 / 512 / 1024 with which every
 *                   item of a slice has its accumulator row in LDS; "part_tile" 0 / 4096 / 7168 = tuples the partition
 *                   sorts in LDS at a time
 *   "wide"          1 (default) = f32 tables with K and KK multiples of 4 are updated by the form of the Hogwild kernel that
 *                   holds four consecutive components per lane (16-byte row accesses, four tuples per wave-instruction at
 *                   K = 64); 0 = always the one-dword-per-lane form (which also serves every other K, f64 and the shared
 *                   negatives / positives)
 *   "fused_sampler" 1 (default) = where the wide kernel applies and every step's draws are independent ("share_negative" 1,
 *                   "share_positive" 1, "group_by_item" 0, "overlap" 0, "owner_neg" 0), rec_sgd_epoch launches no sampler: the
 *                   update kernel draws tuple i of the epoch itself - the tuple rec_sample_tuples reports for step i of
 *                   (seed, epoch) - and no tuple is written to or read from memory; 0 = sampler launch + update launch
 *   "update_policy" 0 = float atomics on every row, 1 = hybrid (plain stores for user rows, biases and the few
 *                   items drawn most often, atomics on the other item rows), 2 = plain racy stores
 *   "atomic"        the same as a bit mask: bit r = row r of the tuple, bit 8 = biases
 *   "hot_count"     items with at least this many training occurrences are stored plainly (0 = none)
 *   "group_by_item" 0 = off, 1 = the update kernel combines runs of equal positive items into one scatter,
 *                   2 = 1 + chunks are drawn by event, which groups them by positive item
 *   "run_segment"   batches one lane group of the run-combining kernel walks (0 = automatic)
 *   "run_shared", "run_hot", "run_hot_atomic"  how many tuples a run of equal positives stays in registers when its
 *                   item is shared with a neighbouring lane group / fills whole segments, and whether the latter is
 *                   scattered with atomics (default 128, 32, 0)
 *   "chunk"         tuples sampled per sampler launch (default 2^26)
 *   "overlap"       1 = sample chunk c+1 on a second stream while chunk c is applied (default 0: measured slower)
 *   "event_phases"  event-major sampler only, before its first use: n > 0 orders the events by n item ranges of equal weight,
 *                   shuffled inside a range (experiment knob: how much item locality in time the SGD tolerates)
 *   "share_negative" BPRMF / MC / FPMC / PRME / HRM / TransRec: 2 (or 4) = that many consecutive steps of an epoch take their negative from one
 *                   stream of candidates (each accepting the first one its own tuple allows, so nearly always the same item)
 *                   and the update scatters that item's rows once for all of them.  Default 1 = independent draws, the reference's sampling.
 *                   Measured for 2: 1.2x the throughput, test AUC 0.002-0.003 lower after the first epochs of a 100M-interaction
 *                   corpus, HIT@50 1-3 % lower (relative)
 *   "share_positive" BPRMF / MC / FPMC / PRME / HRM: 2 = step 2p+1 of an epoch is another event of step 2p's positive item, drawn
 *                   with the events' own weights (every step keeps the reference's distribution; the two share the positive and
 *                   the negative, and the update scatters both items' rows once).  Default 1.  Measured: 1.15x the throughput
 *                   of the default, test AUC 0.0007 and HIT@50 2 % (relative) lower at convergence
 *   "user_batch"    Fossil / FossilSimple / REBUS: 1, 2, 4 or 8 consecutive steps of an epoch share their user draw and one
 *                   wavefront walks that user's history once for all of them (default when an epoch is >= 4M steps: 2 where the
 *                   mean training sequence has >= 16 clicks, 4 where it has >= 32, 8 for Fossil / FossilSimple where it has >= 64; else 1: with very
 *                   short histories the tuples of a group mostly coincide and the summed step can diverge);
 *                   (rec_sgd_replay in Hogwild mode uses it when its tuples come grouped the same way)
 *   "pair_min_history"  a group of "user_batch" tuples shares one walk (all of them stepping from the parameters the group
 *                   found) only when its user has at least this many training clicks (default 8); the tuples of a user
 *                   with fewer are applied one after the other by the same wavefront, each seeing the one before it
 *   "stale_window"  tuples per update launch for small catalogs (0 = no limit)
 *   "global_window" tuples per update launch for models with a row every tuple touches (default 2^22 for TransRec; for
 *                   Fossil / FossilSimple 1/32 of an epoch within 2^18 .. 2^22,
 *                   a quarter of an epoch shorter than 2^18 steps)
 *   "global_cap"    such a row moves by at most this many steps' worth of learning rate per launch
 *   "eval_screen"   with f32 tables the full-catalog sweep of rec_eval runs on a matrix pipe narrower than fp64 and decides every
 *                   comparison it can prove from a rounding bound; the columns within the bound of a threshold are decided in
 *                   fp64, so the counts are those of the fp64 sweep.  1 (default): the 16-bit pipe, every operand as two f16
 *                   halves under power-of-two scales (three MFMAs per 32 contraction indices); 2: the FP32 matrix pipe; 0: the
 *                   fp64 sweep.  Tables or queries that are not finite always take the fp64 sweep.  (process-wide)
 *                   TransRec_L1 (either table precision; 1 or 2): its |.|_1 sweep runs on fixed-point operands, one v_sad_u32
 *                   per term, within a proven quantisation bound; lists of 64 undecided columns per query, and the whole
 *                   sweep in fp64 if one overflows.
 *   "eval_screen_cap"  columns per query the screen may leave to fp64 (default 0 = n_items / 512 within 128 .. 1024, f32 screen:
 *                   max(64, n_items / 2048)); queries that need more are swept again in fp64 - on their own when they are
 *                   few (at most 1/8 of the queries), else the whole sweep moves to the next narrower margin
 *   "project_after_launch"  TransRec: 1 (default) = every update launch ends with the projection of all item rows onto the unit
 *                   ball; 0 = the caller does it (rec_project_rows), see there
 *   "eval_ranges"   ranges the screen cuts the catalog into, each swept by its own blocks (0, the default: one range
 *                   unless the query blocks alone would not fill the chip evenly - few users per engine; then up to 16).  The
 *                   counts do not depend on it.  (process-wide)
 *   "topk_fused"    1 (default) = rec_eval_topk takes its lists from the f16 sweep where that applies; 0 = always from the
 *                   fp64 score matrix.  (process-wide)
 *   "profile"       time every update / scoring / sampler launch with HIP events */
int rec_set_option(rec_engine* e, const char* name, int64_t value);

/* HIP-event statistics gathered while "profile" = 1: which = 0 SGD update (in owner mode: the partition and the update
 * kernel together), 1 eval scoring kernel, 2 sampler, 3 the owner mode's partition alone.  Resets the counters.
 * which = 4 (not reset): how the last full-catalog sweep ran - *launches = 0 fp64, 1 the selected screen held, 2 fp64 took
 * over (many queries overflowed their lists of undecided columns, or the operands are not finite / out of range), 3 the f16
 * screen overflowed and the f32 screen, tried next, held, 4 a screen held except for a few queries (at most 1/8 of them) that
 * overflowed their lists and were swept again on their own in fp64; *total_ms = the number of columns the screen left to
 * fp64, over all queries.
 * which = 5 (not reset): the last rec_eval_topk - *launches = (user, context) lists produced by the fused sweep,
 * *total_ms = device time of that part (HIP events around its kernels).
 * which = 6 (not reset, no device call): the launch shape in effect - *total_ms = "user_batch", *launches = "global_window". */
int rec_kernel_stats(rec_engine* e, int which, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif
