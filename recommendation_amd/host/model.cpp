// Host side of the model classes (see model.hpp).  Everything that is not a hand-over to the engine
// is orchestration the reference also does on the host: the split, the train() loop with its
// validation cadence and early stopping, the progress lines, the saved-model file.
#include "model.hpp"
#include "model_io.hpp"
#include <zlib.h>
#include <sys/time.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

using namespace std;

static double clock_()
{
    timeval tim; gettimeofday(&tim, NULL);
    return tim.tv_sec + (tim.tv_usec / 1000000.0);
}

// ---------------------------------------------------------------------------------------- model
model::model(corpus* corp) : NW(0), W(0), bestW(0), corp(corp), eng(0), rounds_per_epoch(1), epoch_counter(0), best_on_device(false)
{
    nUsers = corp->nUsers; nItems = corp->nItems; nClicks = corp->nClicks;
    clicked_per_user.resize(nUsers);
    // leave out the last two clicks of every user with at least three (src/model.hpp:22-45)
    for (int u = 0; u < nUsers; u++) {
        vector<pair<int, int> >& p = corp->pos_per_user[u];
        if (p.size() < 3) {
            test_per_user.push_back(make_pair(-1, -1));
            teststamp_per_user.push_back(make_pair(-1, -1));
            val_per_user.push_back(make_pair(-1, -1));
        } else {
            int item_test = p.back().first, test_stamp = p.back().second; p.pop_back();
            int item_val = p.back().first, val_stamp = p.back().second; p.pop_back();
            int item_prev = p.back().first;
            test_per_user.push_back(make_pair(item_test, item_val));
            teststamp_per_user.push_back(make_pair(test_stamp, val_stamp));
            val_per_user.push_back(make_pair(item_val, item_prev));
        }
        vector<int>& c = clicked_per_user[u];
        for (size_t i = 0; i < p.size(); i++) c.push_back(p[i].first);
        sort(c.begin(), c.end());
        c.erase(unique(c.begin(), c.end()), c.end());
    }
    num_pos_events = 0;
    for (int u = 0; u < nUsers; u++) num_pos_events += (int)corp->pos_per_user[u].size();
    std::cout << "num_pos_events :" << num_pos_events << "\n\n";
}

model::~model() { destroyEngine(); }

void model::check(int rc) const
{
    if (rc != 0) { printf("recsys engine: %s\n", rec_last_error()); exit(1); }   // the reference's convention: print and exit(1)
}

void model::flattenCorpus(vector<int64_t>& seq_off, vector<int32_t>& seq_item, vector<int32_t>& test_item, vector<int32_t>& val_item) const
{
    seq_off.assign(nUsers + 1, 0);
    test_item.resize(nUsers); val_item.resize(nUsers);
    for (int u = 0; u < nUsers; u++) {
        const vector<pair<int, int> >& p = corp->pos_per_user[u];
        for (size_t i = 0; i < p.size(); i++) seq_item.push_back(p[i].first);
        seq_off[u + 1] = (int64_t)seq_item.size();
        test_item[u] = test_per_user[u].first; val_item[u] = val_per_user[u].first;
    }
    if (seq_item.empty()) seq_item.push_back(0);
}

void model::createEngine(rec_model_desc desc, const vector<int64_t>* path_off, const vector<int32_t>* path_item)
{
    vector<int64_t> seq_off; vector<int32_t> seq_item, test_item, val_item;
    flattenCorpus(seq_off, seq_item, test_item, val_item);
    rec_corpus c;
    c.n_users = nUsers; c.n_items = nItems;
    c.seq_off = seq_off.data(); c.seq_item = seq_item.data(); c.test_item = test_item.data(); c.val_item = val_item.data();
    c.path_off = path_off ? path_off->data() : 0; c.path_item = path_item ? path_item->data() : 0;
    const char* p = getenv("RECSYS_PRECISION");   // "f64" keeps the reference's arithmetic type on the device
    desc.precision = (p && strcmp(p, "f64") == 0) ? REC_F64 : REC_F32;
    const char* dv = getenv("RECSYS_DEVICE");
    desc.device = dv ? atoi(dv) : 0;
    best_on_device = false;
    userless_tuples = desc.model == REC_MC;
    // RECSYS_SHARDS=N: user-sharded training inside this process (N <= 16).  Shard r holds users [U r / N, U (r+1) / N)
    // and sits on GPU r when the process sees at least N GPUs (RECSYS_SHARDS_ONE_DEVICE=1: all on RECSYS_DEVICE).
    const char* ns = getenv("RECSYS_SHARDS");
    int n_shards = ns ? atoi(ns) : 1;
    if (n_shards < 1) n_shards = 1;
    if (n_shards > 16) n_shards = 16;
    if (n_shards > nUsers) n_shards = nUsers > 0 ? nUsers : 1;
    const char* od = getenv("RECSYS_SHARDS_ONE_DEVICE");
    const bool spread = n_shards > 1 && !(od && atoi(od) != 0) && rec_device_count() >= n_shards;
    shards.assign(n_shards, (rec_engine*)0);
    for (int r = 0; r < n_shards; r++) {
        rec_model_desc d = desc;
        d.user_begin = n_shards > 1 ? (int)((int64_t)nUsers * r / n_shards) : 0;
        d.user_end = n_shards > 1 ? (int)((int64_t)nUsers * (r + 1) / n_shards) : 0;
        if (spread) d.device = r;
        check(rec_create(&c, &d, &shards[r]));
    }
    eng = shards[0];
    NW = rec_nw(eng);
    // rounds per epoch: an item row receives about 4 updates per round (every step updates a positive's and a negative's
    // row); longer rounds cost accuracy, measured against the reference's trajectories
    const int64_t steps = rec_steps_per_epoch(eng);
    rounds_per_epoch = n_shards > 1 ? (int)std::max<int64_t>(1, (steps + 2 * (int64_t)nItems - 1) / (2 * (int64_t)std::max(nItems, 1))) : 1;
    const char* rp = getenv("RECSYS_SHARD_ROUNDS");
    if (rp && atoi(rp) > 0) rounds_per_epoch = atoi(rp);
    if (n_shards > 1)
        printf("  \"shards\": %d, \"devices\": %d, \"rounds per epoch\": %d\n", n_shards, spread ? n_shards : 1, rounds_per_epoch);
    W = new double[NW > 0 ? NW : 1];
    bestW = new double[NW > 0 ? NW : 1];
}

void model::destroyEngine()
{
    for (size_t r = 0; r < shards.size(); r++) if (shards[r]) rec_destroy(shards[r]);
    shards.clear();
    eng = 0;
    best_on_device = false;
}

// one updateFactors call; with shards: on the shard that holds the user's rows (MC, whose tuples name no user: the first),
// as a round of its own so that every replica of the item tables receives the step
void model::replayOne(const int32_t* t, double learn_rate)
{
    if (shards.size() == 1) { check(rec_sgd_replay(eng, learn_rate, t, 1, REC_REPLAY_SEQUENTIAL)); return; }
    check(rec_shards_begin_round(shards.data(), (int)shards.size()));
    check(rec_sgd_replay(userless_tuples ? shards[0] : shardOf(t[0]), learn_rate, t, 1, REC_REPLAY_SEQUENTIAL));
    check(rec_shards_combine(shards.data(), (int)shards.size()));
}

void model::initParams() { for (size_t r = 0; r < shards.size(); r++) check(rec_params_init(shards[r], 0)); }

rec_engine* model::shardOf(int user) const
{
    const int n = (int)shards.size();
    for (int r = 0; r < n; r++)
        if (user < (int)((int64_t)nUsers * (r + 1) / n)) return shards[r];
    return shards[n - 1];
}

// every shard moves the tables that are not indexed by user (the same values everywhere) and its own users' rows
void model::syncToHost() { if (NW > 0) for (size_t r = 0; r < shards.size(); r++) check(rec_params_download(shards[r], W, NW)); }
void model::syncToDevice() { if (NW > 0) for (size_t r = 0; r < shards.size(); r++) check(rec_params_upload(shards[r], W, NW)); }

void model::epoch(double learn_rate)
{
    const int n = (int)shards.size();
    if (n == 1) {
        check(rec_sgd_epoch(eng, learn_rate, rec_steps_per_epoch(eng), 0, epoch_counter++));
        check(rec_sync(eng));
        return;
    }
    const int64_t per_round = rec_steps_per_epoch(eng) / ((int64_t)n * rounds_per_epoch);
    for (int k = 0; k < rounds_per_epoch; k++) {
        check(rec_shards_begin_round(shards.data(), n));
        // a different tuple stream for every (epoch, round, shard); the launches only enqueue, so shards on different GPUs run side by side
        for (int r = 0; r < n; r++)
            check(rec_sgd_epoch(shards[r], learn_rate, per_round, 0, (epoch_counter * rounds_per_epoch + k) * n + r));
        check(rec_shards_combine(shards.data(), n));
    }
    epoch_counter++;
}

// the metrics of all shards' users from the 8 sums of each (variance as E[x^2] - mean^2: rounding apart, model.cpp:91-97)
static void sharded_metrics(const std::vector<rec_engine*>& shards, int topn, rec_metrics* m, int (*checker)(int))
{
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t r = 0; r < shards.size(); r++) {
        double part[8];
        checker(rec_eval_sums(shards[r], topn, part));
        for (int i = 0; i < 8; i++) s[i] += part[i];
    }
    const double n = s[7];
    m->n_users = (int64_t)n;
    m->auc_val = s[0] / n; m->auc_test = s[1] / n; m->hit_val = s[2] / n; m->hit_test = s[3] / n;
    m->mrr_val = s[4] / n; m->mrr_test = s[5] / n; m->var_test = s[6] / n - m->auc_test * m->auc_test;
}
static int exit_on_error(int rc)
{
    if (rc != 0) { printf("engine error: %s\n", rec_last_error()); exit(1); }
    return rc;
}

void model::AUC(double* AUC_val, double* AUC_test, double* HIT_val, double* HIT_test, double* MRR_val, double* MRR_test, double* var)
{
    rec_metrics m;
    if (shards.size() > 1) sharded_metrics(shards, 50, &m, exit_on_error);
    else check(rec_eval(eng, 50, &m, 0, 0));
    *AUC_val = m.auc_val; *AUC_test = m.auc_test; *HIT_val = m.hit_val; *HIT_test = m.hit_test;
    *MRR_val = m.mrr_val; *MRR_test = m.mrr_test; *var = m.var_test;
}

// src/model.cpp:100-167 estimates the validation AUC from 2000 candidate draws per user because the exact value costs
// the reference minutes.  Here the exact full-catalog value costs less than an epoch, so it is what train() compares
// (no sampling noise in the early-stopping decision).  RECSYS_SAMPLED_VALIDATION=1 selects the reference's estimate
// instead (2000 draws per user, fresh every call); =2 the cheaper variant with one draw of 2000 items shared by all users.
void model::sampleAUC(double* AUC_val, double* AUC_test, double* var)
{
    rec_metrics m;
    const char* sampled = getenv("RECSYS_SAMPLED_VALIDATION");
    const int mode = (sampled && shards.size() == 1) ? atoi(sampled) : 0;   // sharded runs validate on the exact value
    if (shards.size() > 1) sharded_metrics(shards, 50, &m, exit_on_error);
    else if (mode == 1) check(rec_eval_sampled_per_user(eng, 50, 2000, validation_calls++, &m, 0, 0));
    else if (mode == 2) check(rec_eval_sampled(eng, 50, 2000, validation_calls++, &m, 0, 0));
    else check(rec_eval(eng, 50, &m, 0, 0));
    *AUC_val = m.auc_val; *AUC_test = m.auc_test; *var = m.var_test;
}

void model::MultipleMetrics(int topN, bool sample, double& valid)
{
    rec_metrics m;
    if (shards.size() > 1) {
        sharded_metrics(shards, topN, &m, exit_on_error);
    } else if (sample && nUsers > 50000) {
        // src/model.cpp:260: above 50000 users each valid user is kept with probability ~50000 / nUsers (same rand() test);
        // every user is scored on the device, the subsample is taken from the per-user counts
        std::vector<int32_t> counts((size_t)nUsers * 3);
        check(rec_eval(eng, topN, &m, counts.data(), 0));
        double s[6] = {0, 0, 0, 0, 0, 0};
        int num_user = 0;
        for (int u = 0; u < nUsers; u++) {
            const int cv = counts[3 * (size_t)u], ct = counts[3 * (size_t)u + 1], mx = counts[3 * (size_t)u + 2];
            if (mx < 0) continue;
            if (rand() % nUsers > 50000) continue;
            int rank = mx - ct + 1;
            s[0] += 1.0 * ct / mx; s[1] += 1.0 / rank; s[2] += (rank <= topN);
            rank = mx - cv + 1;
            s[3] += 1.0 * cv / mx; s[4] += 1.0 / rank; s[5] += (rank <= topN);
            num_user++;
        }
        m.n_users = num_user;
        m.auc_test = s[0] / num_user; m.mrr_test = s[1] / num_user; m.hit_test = s[2] / num_user;
        m.auc_val = s[3] / num_user; m.mrr_val = s[4] / num_user; m.hit_val = s[5] / num_user;
    } else {
        check(rec_eval(eng, topN, &m, 0, 0));
    }
    printf("\n\n#Users = %d, VALID: AUC = %f, MRR = %f, HR@%d = %f\n", (int)m.n_users, m.auc_val, m.mrr_val, topN, m.hit_val);
    printf("#Users = %d, TEST:  AUC = %f, MRR = %f, HR@%d = %f\n", (int)m.n_users, m.auc_test, m.mrr_test, topN, m.hit_test);
    valid = m.hit_val;
}

void model::copyBestModel()
{
    for (size_t r = 0; r < shards.size(); r++) check(rec_snapshot_best(shards[r]));
    best_on_device = true;
}

void model::saveModel(const char* path)
{
    // src/model.cpp:176-192 writes bestW, which train() has just copied into W
    syncToHost();
    for (int64_t w = 0; w < NW; w++) bestW[w] = W[w];
    write_model_file(path, bestW, NW);
    printf("\nModel saved to %s.\n", path);
}

void model::loadModel(const char* path)
{
    // src/model.cpp:195-232
    printf("\n  loading parameters from %s.\n", path);
    read_model_file(path, W, NW);
    syncToDevice();
}

string model::toString() { return "Empty Model!"; }

void model::trainLoop(int iterations, double learn_rate, int cadence, int first_validation_after, int patience, bool verbose_validation)
{
    printf("%s", ("\n<<< " + toString() + " >>>\n\n").c_str());
    double bestValidAUC = -1;
    int best_iter = 0;
    for (int iter = 1; iter <= iterations; iter++) {
        double l_dlStart = clock_();
        beforeEpoch(iter);
        epoch(learn_rate);
        printf("Iter: %d, took %f\n", iter, clock_() - l_dlStart);
        fflush(stdout);
        afterEpoch(iter);
        if (iter % cadence == 0 && iter > first_validation_after) {
            if (verbose_validation) printf("Validation ...\n");
            double valid, test, var;
            sampleAUC(&valid, &test, &var);
            printf("[Valid AUC = %f], Test AUC = %f, Test var = %f\n", valid, test, var);
            if (verbose_validation) {
                printf(" SampleAUC() --> Previous best : %f, current : %f \n", bestValidAUC, valid);
                printf(" SampleAUC() --> ite : %d, best_iter + 300 : %d \n", iter, best_iter + 300);
            }
            fflush(stdout);
            if (bestValidAUC < valid) {
                bestValidAUC = valid; best_iter = iter;
                copyBestModel();
            } else if (iter > best_iter + patience) {
                printf("Overfitted. Exiting... \n");
                break;
            }
        }
    }
    // "copy back best parameters" (e.g. BPRMF.cpp:189-191).  Without any validation the reference copies an
    // uninitialised bestW (undefined behaviour); here the trained parameters are kept instead.
    if (best_on_device) for (size_t r = 0; r < shards.size(); r++) check(rec_restore_best(shards[r]));
    double AUC_val, AUC_test, HIT_val, HIT_test, MRR_val, MRR_test, var;
    AUC(&AUC_val, &AUC_test, &HIT_val, &HIT_test, &MRR_val, &MRR_test, &var);
    printf("\n\n <<< %s >>> Test AUC = %f, Val AUC = %f, Test var = %f\n", toString().c_str(), AUC_test, AUC_val, var);
    printf("\n\n <<< %s >>> Test HIT50 = %f, Val HIT50 = %f\n", toString().c_str(), HIT_test, HIT_val);
    printf("\n\n <<< %s >>> Test MRR = %f, Val MRR = %f\n", toString().c_str(), MRR_test, MRR_val);
}

static rec_model_desc make_desc(int model, int K, int KK, int L, double lambda, double bias_reg, double relation_reg, double alpha, int df)
{
    rec_model_desc d;
    memset(&d, 0, sizeof d);
    d.model = model; d.K = K; d.KK = KK; d.L = L; d.lambda = lambda; d.bias_reg = bias_reg; d.relation_reg = relation_reg;
    d.alpha = alpha; d.dataset_factor = df;
    return d;
}

#define REPLAY1(a, b, c, d)                                                          \
    do {                                                                             \
        int32_t t_[4] = {(int32_t)(a), (int32_t)(b), (int32_t)(c), (int32_t)(d)};    \
        replayOne(t_, learn_rate);                                                   \
    } while (0)

// ---------------------------------------------------------------------------------------- MostPopular
MostPopular::MostPopular(corpus* corp) : model(corp)
{
    createEngine(make_desc(REC_MP, 0, 0, 1, 0, 0, 0, 0, 0));
}

// ---------------------------------------------------------------------------------------- BPRMF
void BPRMF::init() { createEngine(make_desc(REC_BPRMF, K, K, 1, lambda, bias_reg, 0, 0, 0)); initParams(); }
void BPRMF::cleanUp() { destroyEngine(); delete[] W; delete[] bestW; W = bestW = 0; }
void BPRMF::train(int iterations, double learn_rate) { trainLoop(iterations, learn_rate, 50, 0, 300, false); }
void BPRMF::updateFactors(int user_id, int pos_item_id, int neg_item_id, double learn_rate) { REPLAY1(user_id, pos_item_id, neg_item_id, 0); }
string BPRMF::toString() { char s[1000]; sprintf(s, "BPR-MF__K_%d_lambda_%f_biasReg_%f", K, lambda, bias_reg); return s; }

// ---------------------------------------------------------------------------------------- MC
void MC::init() { createEngine(make_desc(REC_MC, K, K, 1, lambda, 0, 0, 0, 0)); initParams(); }
void MC::cleanUp() { destroyEngine(); delete[] W; delete[] bestW; W = bestW = 0; }
void MC::train(int iterations, double learn_rate) { trainLoop(iterations, learn_rate, 50, 0, 1000, false); }   // MC.cpp:182
void MC::updateFactors(int item_id, int pos_item_id, int neg_item_id, double learn_rate) { REPLAY1(item_id, pos_item_id, neg_item_id, 0); }
string MC::toString() { char s[100]; sprintf(s, "MC__K_%d_lambda_%f", K, lambda); return s; }

// ---------------------------------------------------------------------------------------- FPMC
void FPMC::init() { createEngine(make_desc(REC_FPMC, K, KK, 1, lambda, 0, 0, 0, 0)); initParams(); }
void FPMC::cleanUp() { destroyEngine(); delete[] W; delete[] bestW; W = bestW = 0; }
void FPMC::train(int iterations, double learn_rate) { trainLoop(iterations, learn_rate, 50, 0, 300, false); }
void FPMC::updateFactors(int user_id, int item_id, int pos_item_id, int neg_item_id, double learn_rate) { REPLAY1(user_id, item_id, pos_item_id, neg_item_id); }
string FPMC::toString() { char s[100]; sprintf(s, "FPMC__K_%d_KK_%d_lambda_%f", K, KK, lambda); return s; }

// ---------------------------------------------------------------------------------------- PRME
void PRME::init() { createEngine(make_desc(REC_PRME, K, KK, 1, lambda, 0, 0, alpha, 0)); initParams(); }
void PRME::cleanUp() { destroyEngine(); delete[] W; delete[] bestW; W = bestW = 0; }
void PRME::train(int iterations, double learn_rate) { trainLoop(iterations, learn_rate, 50, 0, 300, false); }
void PRME::updateFactors(int user_id, int item_id, int pos_item_id, int neg_item_id, double learn_rate) { REPLAY1(user_id, item_id, pos_item_id, neg_item_id); }
string PRME::toString() { char s[100]; sprintf(s, "PRME__K_%d_KK_%d_alpha_%f_lambda_%f", K, KK, alpha, lambda); return s; }

// ---------------------------------------------------------------------------------------- HRM
void HRM_avg::init() { createEngine(make_desc(modelId(), K, K, 1, lambda, 0, 0, 0, 0)); initParams(); }
void HRM_avg::cleanUp() { destroyEngine(); delete[] W; delete[] bestW; W = bestW = 0; }
void HRM_avg::train(int iterations, double learn_rate) { trainLoop(iterations, learn_rate, 50, 0, 300, false); }
void HRM_avg::updateFactors(int user, int prev_item, int item, double learn_rate)
{
    int neg_item;
    do { neg_item = rand() % nItems; } while (neg_item == item);   // HRM_avg.cpp:171-174
    REPLAY1(user, prev_item, item, neg_item);
}
string HRM_avg::toString() { char s[100]; sprintf(s, "HRM_AvgPooling_BPR_K_%d_lambda_%f", K, lambda); return s; }
string HRM_max::toString() { char s[100]; sprintf(s, "HRM_MaxPooling_BPR_K_%d_lambda_%f", K, lambda); return s; }

// ---------------------------------------------------------------------------------------- TransRec
void TransRec::init() { createEngine(make_desc(modelId(), K, K, 1, lambda, bias_reg, relation_reg, 0, 0)); initParams(); }
void TransRec::cleanUp() { destroyEngine(); delete[] W; delete[] bestW; W = bestW = 0; }
void TransRec::train(int iterations, double learn_rate) { trainLoop(iterations, learn_rate, 50, 0, 300, false); }
void TransRec::updateFactors(int user, int x, int y, int yn, double learn_rate) { REPLAY1(user, x, y, yn); }
void TransRec::afterEpoch(int iter)
{
    // "Observe if parameters are healthy", TransRec.cpp:111-124; W = [beta:I][H:I*K][R:U*K][r:K]
    if (iter % 50 != 0) return;
    vector<double> v(K);
    const int64_t r_off = (int64_t)nItems + (int64_t)nItems * K + (int64_t)nUsers * K;
    check(rec_params_read(eng, r_off, K, v.data()));
    printf("r = ");
    for (int k = 0; k < K; k++) printf("%f  ", v[k]);
    printf("\n");
    int rand_u = rand() % nUsers;
    check(rec_params_read(shardOf(rand_u), (int64_t)nItems + (int64_t)nItems * K + (int64_t)rand_u * K, K, v.data()));
    printf("R[u] = ");
    for (int k = 0; k < K; k++) printf("%f  ", v[k]);
    printf("\n");
}
string TransRec::toString() { char s[200]; sprintf(s, "TransRec__K_%d_lambda_%f_relationReg_%f_biasReg_%f", K, lambda, relation_reg, bias_reg); return s; }
string TransRec_L1::toString() { char s[200]; sprintf(s, "TransRec_L1__K_%d_lambda_%f_relationReg_%f_biasReg_%f", K, lambda, relation_reg, bias_reg); return s; }

// ---------------------------------------------------------------------------------------- FossilSimple
void FossilSimple::init() { createEngine(make_desc(REC_FOSSIL_SIMPLE, K, K, 1, lambda, bias_reg, 0, 0, 0)); initParams(); }
void FossilSimple::cleanUp() { destroyEngine(); delete[] W; delete[] bestW; W = bestW = 0; }
void FossilSimple::train(int iterations, double learn_rate) { trainLoop(iterations, learn_rate, 50, 0, 500, false); }   // FossilSimple.cpp:302
void FossilSimple::updateFactors(int user_id, int item_id, int pos_item_id, int neg_item_id, double learn_rate) { REPLAY1(user_id, item_id, pos_item_id, neg_item_id); }
string FossilSimple::toString() { char s[100]; sprintf(s, "FossilSimple__K_%d_lambda_%.2f_biasReg_%.2f", K, lambda, bias_reg); return s; }

// ---------------------------------------------------------------------------------------- Fossil
void Fossil::init() { createEngine(make_desc(REC_FOSSIL, K, K, L, lambda, bias_reg, 0, 0, dataset_factor)); initParams(); }
void Fossil::cleanUp() { destroyEngine(); delete[] W; delete[] bestW; W = bestW = 0; }
void Fossil::train(int iterations, double learn_rate) { trainLoop(iterations, learn_rate, 50, 0, 300, false); }
void Fossil::beforeEpoch(int)
{
    // the progress line at the top of every Fossil::oneiteration, Fossil.cpp:260; W = [beta:I][WT:L]...
    double wt[2] = {0, 0};
    check(rec_params_read(eng, nItems, L < 2 ? L : 2, wt));
    printf("decay_wt_0 = %f, decay_wt_1 = %f\n", wt[0], wt[1]);
}
void Fossil::oneiteration(double learn_rate)
{
    beforeEpoch(0);
    epoch(learn_rate);
}
void Fossil::updateFactors(int user_id, vector<int>& prev_items, int pos_item_id, int neg_item_id, double learn_rate)
{
    // find the target position the arguments describe (Fossil.cpp:266-272 builds them from it)
    const vector<pair<int, int> >& p = corp->pos_per_user[user_id];
    for (int t = 1; t < (int)p.size(); t++) {
        if (p[t].first != pos_item_id) continue;
        const int np = t < L ? t : L;
        if ((int)prev_items.size() != np) continue;
        bool same = true;
        for (int ind = 0; ind < np && same; ind++) same = p[t - 1 - ind].first == prev_items[ind];
        if (same) { REPLAY1(user_id, t, neg_item_id, 0); return; }
    }
    printf("Fossil::updateFactors: prev_items do not precede item %d in the sequence of user %d\n", pos_item_id, user_id);
    exit(1);
}
string Fossil::toString() { char s[100]; sprintf(s, "Fossil__L_%d_K_%d_lambda_%.2f_biasReg_%.2f", L, K, lambda, bias_reg); return s; }

// ---------------------------------------------------------------------------------------- REBUS
int REBUS::loadPST(int minCount, int L, const char* data_path)
{
    printf(" -- Load Sequences Started %s -- \n\n", toString().c_str());
    this->minCount = minCount; this->L = L;
    // REBUS.cpp:8-26: second path component of the click file, minus its 4-character extension
    vector<string> parts; { stringstream ss(data_path); string item; while (getline(ss, item, '/')) parts.push_back(item); }
    if (parts.size() < 2 || parts[1].size() < 4) { printf("Can't derive the FSUB file name from %s.\n", data_path); return 1; }
    const string stem = parts[1].substr(0, parts[1].size() - 4);
    string file;
    if (type_seq == "fsub") file = "96-FSUB/" + stem + "_root_fsub_minCount_" + to_string(minCount) + "_L_" + to_string(L) + ".txt";
    else if (type_seq == "fseq") file = "97-FSEQ/" + stem + "_root_fseq_minCount_" + to_string(minCount) + "_L_" + to_string(L) + ".txt";
    else if (type_seq == "fseq_bide") file = "95-FSEQ_BIDE/" + stem + "_root_fseq_bide_minCount_" + to_string(minCount) + "_L_" + to_string(L) + ".txt";
    else file = "98-PST/" + stem + "_root_minCount_" + to_string(minCount) + "_L_" + to_string(L) + ".txt";
    std::cout << "ss : " << file << "\n";
    gzFile in = gzopen(file.c_str(), "rb");
    if (!in) { printf("Can't read clicks from %s.\n\n\n\n", file.c_str()); return 1; }
    lab_off.assign(1, 0); lab_item.clear();
    int count_i = 0, count_not_i = 0;
    char line[1 << 16];
    while (gzgets(in, line, sizeof line)) {
        char* p = line;
        while (*p && isspace((unsigned char)*p)) p++;
        char* e = p; while (*e && !isspace((unsigned char)*e)) e++;
        *e = 0;
        if (!*p) continue;
        if (strcmp(p, "Root") == 0) { count_not_i++; continue; }   // never a candidate of findPathStars
        if (strchr(p, '-')) count_not_i++; else count_i++;
        while (p && *p) { lab_item.push_back(atoi(p)); p = strchr(p, '-'); if (p) p++; }
        lab_off.push_back((int64_t)lab_item.size());
    }
    gzclose(in);
    std::cout << "  size of nodes_label" << (lab_off.size() - 1 + (count_not_i > 0)) << '\n';
    std::cout << "  count_not_i :" << count_not_i << '\n';
    std::cout << "  count_i :" << count_i << '\n';
    std::cout << "  nItems " << nItems << '\n';
    nb_items_plus_root = nItems + 1;
    std::cout << "  nb_items_plus_root " << nb_items_plus_root << '\n';
    return 0;
}

void REBUS::init()
{
    printf(" --Init Started %s -- \n\n", toString().c_str());
    vector<int64_t> seq_off; vector<int32_t> seq_item, test_item, val_item;
    flattenCorpus(seq_off, seq_item, test_item, val_item);
    vector<int64_t> path_off((size_t)seq_off[nUsers] + 2 * (size_t)nUsers + 1);
    const int64_t n = rec_rebus_build_paths(nUsers, nItems, seq_off.data(), seq_item.data(), val_item.data(), (int64_t)lab_off.size() - 1,
                                            lab_off.data(), lab_item.data(), path_off.data(), 0);
    if (n < 0) check(-1);
    vector<int32_t> path_item((size_t)(n > 0 ? n : 1));
    rec_rebus_build_paths(nUsers, nItems, seq_off.data(), seq_item.data(), val_item.data(), (int64_t)lab_off.size() - 1, lab_off.data(),
                          lab_item.data(), path_off.data(), path_item.data());
    createEngine(make_desc(REC_REBUS, K, K, L, lambda, bias_reg, 0, alpha_up, 0), &path_off, &path_item);
    printf(" --NW : %lld -- \n\n", (long long)NW);
    initParams();
    printf(" --Init REBUS ended -- \n\n");
}
void REBUS::cleanUp() { destroyEngine(); delete[] W; delete[] bestW; W = bestW = 0; }
void REBUS::train(int iterations, double learn_rate)
{
    printf(" --Train Strated %s -- \n\n", toString().c_str());
    trainLoop(iterations, learn_rate, 100, start_auc_test, 300, true);   // REBUS.cpp:578-594
}
void REBUS::updateFactors(int user_id, int rand_num, int neg_item_id, double learn_rate) { REPLAY1(user_id, rand_num, neg_item_id, 0); }
vector<int> REBUS::findPathStars(vector<int> prev_items)
{
    // one pseudo-user whose training sequence is the prefix; its validation context (a prefix of all n items) is the answer
    const int n = (int)prev_items.size();
    vector<int> out;
    if (n == 0) { out.push_back(nItems); return out; }
    int64_t seq_off[2] = {0, n}; int32_t val = 0;
    vector<int32_t> seq(prev_items.begin(), prev_items.end());
    vector<int64_t> po(n + 3);
    vector<int32_t> pi((size_t)(n + 2) * (L + 1) + 4);
    rec_rebus_build_paths(1, nItems, seq_off, seq.data(), &val, (int64_t)lab_off.size() - 1, lab_off.data(), lab_item.data(), po.data(), pi.data());
    for (int64_t i = po[n - 1]; i < po[n]; i++) out.push_back(pi[i]);
    return out;
}
string REBUS::toString()
{
    char s[250];
    sprintf(s, "REBUS_K_%d_minCount_%d_L_%d_lambda_%f_biasReg_%f_typeSeq_%s_findPathStar_%d_alphaUP%f", K, minCount, L, lambda, bias_reg,
            type_seq.c_str(), find_path_stars, alpha_up);
    return s;
}
