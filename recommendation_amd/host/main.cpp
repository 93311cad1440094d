// ./train with the reference's 13 positional arguments (src/main.cpp:142-231) on the HIP engine.
#include <cstdio>
#include <cstdlib>
#include <string>
#include "corpus.hpp"
#include "model.hpp"

using namespace std;

// With 0 iterations and an existing model file the run scores that file instead of training: the reference
// has loadModel (src/model.cpp:195-232) but no command line that reaches it.
template <class M>
static void run(M& md, int iterations, const char* model_path, bool multiple_metrics)
{
    md.init();
    const string saved = string(model_path) + "__" + md.toString() + ".txt";
    if (iterations == 0) {
        if (FILE* f = fopen(saved.c_str(), "r")) {
            fclose(f);
            md.loadModel(saved.c_str());
            double AUC_val, AUC_test, HIT_val, HIT_test, MRR_val, MRR_test, var;
            md.AUC(&AUC_val, &AUC_test, &HIT_val, &HIT_test, &MRR_val, &MRR_test, &var);
            printf("\n\n <<< %s >>> Test AUC = %f, Val AUC = %f, Test var = %f\n", md.toString().c_str(), AUC_test, AUC_val, var);
            printf("\n\n <<< %s >>> Test HIT50 = %f, Val HIT50 = %f\n", md.toString().c_str(), HIT_test, HIT_val);
            if (multiple_metrics) { double valid; md.MultipleMetrics(50, false, valid); }
            md.cleanUp();
            return;
        }
    }
    md.train(iterations, 0.05);
    if (multiple_metrics) { double valid; md.MultipleMetrics(50, false, valid); }
    md.saveModel((string(model_path) + "__" + md.toString() + ".txt").c_str());
    md.cleanUp();
}

int main(int argc, char** argv)
{
    srand(0);
    if (argc != 14) {
        printf(" Parameters as following: \n");
        printf(" 1. Click triples path \n");
        printf(" 2. user min \n");
        printf(" 3. item min \n");
        printf(" 4. Min occurence for substring (MinCount) \n");
        printf(" 5. Max size for substring (L) \n");
        printf(" 6. Latent Feature Dimension (K) \n");
        printf(" 7. lambda (L2-norm regularizer) \n");
        printf(" 8. bias_reg (L2-norm regularizer for bias terms) \n");
        printf(" 9. relation_reg (L2-norm regularizer for USER translation vectors, only for TransRec \n");
        printf(" 10. alpha (for PRME and REBUS) \n");
        printf(" 11. Maximum number of iterations \n");
        printf(" 12. Model path \n");
        printf(" 13. Model name \n\n");
        exit(1);
    }
    char* data_path = argv[1];
    int user_min = atoi(argv[2]), item_min = atoi(argv[3]), minCount = atoi(argv[4]), L = atoi(argv[5]), K = atoi(argv[6]);
    double lambda = atof(argv[7]), bias_reg = atof(argv[8]), relation_reg = atof(argv[9]), alpha = atof(argv[10]);
    int iter = atoi(argv[11]);
    char* model_path = argv[12];
    string type_model = argv[13];

    corpus corp;
    corp.loadData(data_path, user_min, item_min);

    if (type_model == "REBUS") {
        REBUS md(&corp, K, lambda, bias_reg, alpha);
        if (md.loadPST(minCount, L, data_path) == 0) run(md, iter, model_path, false);
        else printf("No substrings found for the dataset %s with minCount : %d and L : %d\n", data_path, minCount, L);
    } else if (type_model == "MP") {
        MostPopular md(&corp);
        double valid; md.MultipleMetrics(50, false, valid);
    } else if (type_model == "BPRMF") { BPRMF md(&corp, K, lambda, bias_reg); run(md, iter, model_path, true);
    } else if (type_model == "MC") { MC md(&corp, K, lambda); run(md, iter, model_path, true);
    } else if (type_model == "FPMC") { FPMC md(&corp, K, K, lambda); run(md, iter, model_path, true);
    } else if (type_model == "HRM_max") { HRM_max md(&corp, K, lambda); run(md, iter, model_path, true);
    } else if (type_model == "HRM_avg") { HRM_avg md(&corp, K, lambda); run(md, iter, model_path, true);
    } else if (type_model == "PRME") { PRME md(&corp, K, K, alpha, lambda); run(md, iter, model_path, true);
    } else if (type_model == "TransRec_L1") { TransRec_L1 md(&corp, K, lambda, relation_reg, bias_reg); run(md, iter, model_path, true);
    } else if (type_model == "TransRec") { TransRec md(&corp, K, lambda, relation_reg, bias_reg); run(md, iter, model_path, true);
    } else if (type_model == "FossilSimple") { FossilSimple md(&corp, K, lambda, bias_reg); run(md, iter, model_path, false);
    } else if (type_model == "Fossil") { Fossil md(&corp, L, K, lambda, bias_reg); run(md, iter, model_path, false);
    } else {
        printf("Unknown model !!! \n");
        printf("Possible models: \n  - MP \n  - BPRMF \n  - MC \n  - FPMC \n  - HRM_max \n  - HRM_avg \n  - PRME \n  - TransRec_L1 \n  - TransRec \n  - Fossil \n  - REBUS \n");
    }
    printf("}\n");
    return 0;
}
