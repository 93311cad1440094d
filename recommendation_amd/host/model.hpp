// Host-side model classes with the public surface of the reference (src/model.hpp:9-92 and
// src/models/*.hpp): same constructors, same method names, same printed lines, same saved-model
// format.  The hot methods - oneiteration / oneIteration, updateFactors, AUC, sampleAUC,
// MultipleMetrics, copyBestModel - hand over to the HIP engine through include/recsys_engine.h.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>
#include "../../include/recsys_engine.h"
#include "corpus.hpp"

class model
{
public:
    model(corpus* corp);              // leave-last-two split, src/model.hpp:12-53 (pops the two items off the corpus)
    virtual ~model();

    /* Model parameters */
    int64_t NW;                       // the reference holds an int (src/model.hpp:60); 64 bits here so that the count never narrows
    double* W;                        // host copy; the device copy is authoritative between syncToHost() calls
    double* bestW;

    /* Corpus related */
    corpus* corp;
    int nUsers, nItems, nClicks;
    std::vector<std::vector<int> > clicked_per_user;   // sorted distinct training items (the reference holds unordered_sets)
    std::vector<std::pair<int, int> > val_per_user;    // (val item, previous item)
    std::vector<std::pair<int, int> > test_per_user;   // (test item, val item)
    std::vector<std::pair<int, int> > teststamp_per_user;
    int num_pos_events;

    virtual void sampleAUC(double* AUC_val, double* AUC_test, double* var);
    uint64_t validation_calls = 0;  // seeds the item draw of sampleAUC
    virtual void AUC(double* AUC_val, double* AUC_test, double* HIT_val, double* HIT_test, double* MRR_val, double* MRR_test, double* var);
    virtual void copyBestModel();
    virtual void saveModel(const char* path);
    virtual void loadModel(const char* path);
    virtual std::string toString();
    void MultipleMetrics(int topN, bool sample, double& valid);

    /* engine plumbing */
    void syncToHost();                // device parameters -> W
    void syncToDevice();              // W -> device parameters
    rec_engine* eng;                  // the engine (with RECSYS_SHARDS > 1: the first shard)
    // RECSYS_SHARDS=N: N engines with consecutive user ranges (one per GPU when the process sees N of them, else all on
    // one), trained in rounds of local steps + one combination of the item tables per round (rec_shards_combine)
    std::vector<rec_engine*> shards;
    int rounds_per_epoch;

protected:
    void initParams();                // <Model>::init's random fill, the same values on every shard
    rec_engine* shardOf(int user) const;
    void replayOne(const int32_t* tuple, double learn_rate);
    bool userless_tuples;             // MC: a tuple's first component is an item
    // creates the engine for `desc` (model id, K, ... filled by the subclass) and allocates W / bestW
    void createEngine(rec_model_desc desc, const std::vector<int64_t>* path_off = nullptr, const std::vector<int32_t>* path_item = nullptr);
    void destroyEngine();
    void flattenCorpus(std::vector<int64_t>& seq_off, std::vector<int32_t>& seq_item, std::vector<int32_t>& test_item,
                       std::vector<int32_t>& val_item) const;
    // the shared train() loop of the reference (e.g. src/models/BPRMF.cpp:152-203)
    void trainLoop(int iterations, double learn_rate, int cadence, int first_validation_after, int patience, bool verbose_validation);
    virtual void epoch(double learn_rate);
    virtual void beforeEpoch(int iter) {}
    virtual void afterEpoch(int iter) {}
    void check(int rc) const;
    unsigned long long epoch_counter;
    bool best_on_device;
};

// ---- one class per reference model; constructor arguments as in src/models/*.hpp

class MostPopular : public model
{
public:
    MostPopular(corpus* corp);
    std::string toString() { return "MostPopular"; }
};

#define DECLARE_COMMON(NAME)                                         \
    void init();                                                     \
    void cleanUp();                                                  \
    void train(int iterations, double learn_rate);                   \
    std::string toString();

class BPRMF : public model
{
public:
    BPRMF(corpus* corp, int K, double lambda, double bias_reg) : model(corp), K(K), lambda(lambda), bias_reg(bias_reg) {}
    DECLARE_COMMON(BPRMF)
    void oneiteration(double learn_rate) { epoch(learn_rate); }
    virtual void updateFactors(int user_id, int pos_item_id, int neg_item_id, double learn_rate);
    int K; double lambda, bias_reg;
};

class MC : public model
{
public:
    MC(corpus* corp, int K, double lambda) : model(corp), K(K), lambda(lambda) {}
    DECLARE_COMMON(MC)
    void oneiteration(double learn_rate) { epoch(learn_rate); }
    virtual void updateFactors(int item_id, int pos_item_id, int neg_item_id, double learn_rate);
    int K; double lambda;
};

class FPMC : public model
{
public:
    FPMC(corpus* corp, int K, int KK, double lambda) : model(corp), K(K), KK(KK), lambda(lambda) {}
    DECLARE_COMMON(FPMC)
    void oneiteration(double learn_rate) { epoch(learn_rate); }
    virtual void updateFactors(int user_id, int item_id, int pos_item_id, int neg_item_id, double learn_rate);
    int K, KK; double lambda;
};

class PRME : public model
{
public:
    PRME(corpus* corp, int K, int KK, double alpha, double lambda) : model(corp), K(K), KK(KK), alpha(alpha), lambda(lambda) {}
    DECLARE_COMMON(PRME)
    void oneiteration(double learn_rate) { epoch(learn_rate); }
    virtual void updateFactors(int user_id, int item_id, int pos_item_id, int neg_item_id, double learn_rate);
    int K, KK; double alpha, lambda;
};

class HRM_avg : public model
{
public:
    HRM_avg(corpus* corp, int K, double lambda) : model(corp), K(K), lambda(lambda) {}
    DECLARE_COMMON(HRM_avg)
    void oneIteration(double learn_rate) { epoch(learn_rate); }
    // the reference draws the negative inside (src/models/HRM_avg.cpp:171-174); here it is drawn on the host the same way
    void updateFactors(int user, int prev_item, int item, double learn_rate);
    int K; double lambda;
protected:
    virtual int modelId() const { return REC_HRM_AVG; }
};

class HRM_max : public HRM_avg
{
public:
    HRM_max(corpus* corp, int K, double lambda) : HRM_avg(corp, K, lambda) {}
    std::string toString();
protected:
    int modelId() const { return REC_HRM_MAX; }
};

class TransRec : public model
{
public:
    TransRec(corpus* corp, int K, double lambda, double relation_reg, double bias_reg)
        : model(corp), K(K), lambda(lambda), relation_reg(relation_reg), bias_reg(bias_reg) {}
    DECLARE_COMMON(TransRec)
    void oneIteration(double learn_rate) { epoch(learn_rate); }
    void updateFactors(int user, int x, int y, int yn, double learn_rate);
    int K; double lambda, relation_reg, bias_reg;
protected:
    virtual int modelId() const { return REC_TRANSREC; }
    void afterEpoch(int iter);
};

class TransRec_L1 : public TransRec
{
public:
    TransRec_L1(corpus* corp, int K, double lambda, double relation_reg, double bias_reg) : TransRec(corp, K, lambda, relation_reg, bias_reg) {}
    std::string toString();
protected:
    int modelId() const { return REC_TRANSREC_L1; }
};

class FossilSimple : public model
{
public:
    FossilSimple(corpus* corp, int K, double lambda, double bias_reg) : model(corp), K(K), lambda(lambda), bias_reg(bias_reg) {}
    DECLARE_COMMON(FossilSimple)
    void oneiteration(double learn_rate) { epoch(learn_rate); }
    void updateFactors(int user_id, int item_id, int pos_item_id, int neg_item_id, double learn_rate);
    int K; double lambda, bias_reg;
};

class Fossil : public model
{
public:
    Fossil(corpus* corp, int L, int K, double lambda, double bias_reg, int dataset_factor = 10)
        : model(corp), L(L), K(K), lambda(lambda), bias_reg(bias_reg), dataset_factor(dataset_factor) {}
    DECLARE_COMMON(Fossil)
    void oneiteration(double learn_rate);
    // prev_items must be the (at most L) items before the target position, newest first, as oneiteration builds them
    void updateFactors(int user_id, std::vector<int>& prev_items, int pos_item_id, int neg_item_id, double learn_rate);
    int L, K; double lambda, bias_reg; int dataset_factor;
protected:
    void beforeEpoch(int iter);
};

class REBUS : public model
{
public:
    REBUS(corpus* corp, int K, double lambda, double bias_reg, double alpha_up)
        : model(corp), minCount(0), K(K), L(1), lambda(lambda), bias_reg(bias_reg), type_seq("fsub"), find_path_stars(true),
          nb_stars(0), nb_items_plus_root(0), start_auc_test(49), alpha_up(alpha_up) {}
    int loadPST(int minCount, int L, const char* data_path);   // src/models/REBUS.cpp:3-87
    DECLARE_COMMON(REBUS)
    void oneiteration(double learn_rate) { epoch(learn_rate); }
    // tuple form of src/models/REBUS.cpp:540-554: the target is position rand_num + 1 of the user's training sequence
    void updateFactors(int user_id, int rand_num, int neg_item_id, double learn_rate);
    std::vector<int> findPathStars(std::vector<int> prev_items);
    int minCount, K, L; double lambda, bias_reg;
    std::string type_seq; bool find_path_stars; int nb_stars;
    int nb_items_plus_root, start_auc_test; double alpha_up;
    std::vector<int64_t> lab_off; std::vector<int32_t> lab_item;   // the label set as integer sequences, oldest item first
};
