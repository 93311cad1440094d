// fsub_mine: writes the 96-FSUB label file REBUS::loadPST (src/models/REBUS.cpp:3-87) reads.
// The reference ships such files precomputed for its bundled data sets and holds no code that
// produces them; this tool fills that gap for other click files.
//
//   fsub_mine <click file> <userMin> <itemMin> <minCount> <L> [output file]
//
// The default output name is the one loadPST derives from its arguments:
// 96-FSUB/<second path component of the click file minus 4 characters>_root_fsub_minCount_<m>_L_<L>.txt
#include <cstdio>
#include <cstdlib>
#include <string>
#include <sstream>
#include <vector>
#include "corpus.hpp"
#include "recsys_engine.h"

int main(int argc, char** argv)
{
    if (argc < 6) {
        printf(" Parameters as following: \n");
        printf(" 1. Click triples path\n 2. user min\n 3. item min\n 4. minCount\n 5. L\n 6. (optional) output file\n");
        return 1;
    }
    const char* data_path = argv[1];
    const int userMin = atoi(argv[2]), itemMin = atoi(argv[3]), minCount = atoi(argv[4]), L = atoi(argv[5]);
    corpus corp;
    corp.loadData(data_path, userMin, itemMin);
    // training sequences: the last two clicks of every user with at least three are held out (src/model.hpp:22-45)
    std::vector<int64_t> seq_off(corp.nUsers + 1, 0);
    std::vector<int32_t> seq_item;
    for (int u = 0; u < corp.nUsers; u++) {
        const auto& s = corp.pos_per_user[u];
        const size_t keep = s.size() >= 3 ? s.size() - 2 : s.size();
        for (size_t i = 0; i < keep; i++) seq_item.push_back(s[i].first);
        seq_off[u + 1] = (int64_t)seq_item.size();
    }
    std::string out;
    if (argc > 6) out = argv[6];
    else {
        std::vector<std::string> parts; { std::stringstream ss(data_path); std::string item; while (getline(ss, item, '/')) parts.push_back(item); }
        if (parts.size() < 2 || parts[1].size() < 4) { printf("Can't derive the FSUB file name from %s.\n", data_path); return 1; }
        out = "96-FSUB/" + parts[1].substr(0, parts[1].size() - 4) + "_root_fsub_minCount_" + std::to_string(minCount) + "_L_" + std::to_string(L) + ".txt";
    }
    FILE* f = fopen(out.c_str(), "w");
    if (!f) { printf("Can't write %s.\n", out.c_str()); return 1; }
    if (minCount == 1 && L == 1) {
        // the bundled *_minCount_1_L_1 files list every item id and no Root
        for (int i = 0; i < corp.nItems; i++) fprintf(f, "%d\n", i);
        fclose(f);
        printf("%d labels written to %s\n", corp.nItems, out.c_str());
        return 0;
    }
    int64_t total = 0;
    const int64_t n = rec_fsub_mine(corp.nUsers, corp.nItems, seq_off.data(), seq_item.data(), L, minCount, nullptr, nullptr, &total);
    if (n < 0) { printf("%s\n", rec_last_error()); return 1; }
    std::vector<int64_t> lab_off(n + 1); std::vector<int32_t> lab_item(total > 0 ? total : 1);
    rec_fsub_mine(corp.nUsers, corp.nItems, seq_off.data(), seq_item.data(), L, minCount, lab_off.data(), lab_item.data(), &total);
    fprintf(f, "Root\n");
    for (int64_t i = 0; i < n; i++) {
        for (int64_t p = lab_off[i]; p < lab_off[i + 1]; p++) fprintf(f, p == lab_off[i] ? "%d" : "-%d", lab_item[p]);
        fputc('\n', f);
    }
    fclose(f);
    printf("%lld labels written to %s\n", (long long)(n + 1), out.c_str());
    return 0;
}
