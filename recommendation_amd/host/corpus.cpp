// Two passes over the click file as src/corpus.cpp:20-226: count per user / item, then keep the
// clicks whose user and item reach the minimum counts, giving ids in order of first appearance
// (the 96-FSUB label files are written in these ids), and sort every user's clicks by time.
// The file is read through zlib (plain text passes through gzread unchanged) and split in place,
// without a stringstream per line.
#include "corpus.hpp"
#include <zlib.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace {
struct Click { std::string user, item; int time; };

// calls f(user, item, time) for every line; false when the file cannot be opened
bool for_each_click(const char* path, bool csv, const std::function<void(const char*, const char*, int)>& f)
{
    gzFile in = gzopen(path, "rb");
    if (!in) return false;
    gzbuffer(in, 1 << 20);
    std::vector<char> line(1 << 16);
    while (gzgets(in, line.data(), (int)line.size())) {
        char* p = line.data();
        char* tok[4] = {nullptr, nullptr, nullptr, nullptr};
        int nt = 0;
        if (csv) {
            // the reference takes the first whitespace-delimited word of the line and splits it at commas
            while (*p == ' ' || *p == '\t') p++;
            char* end = p;
            while (*end && !isspace((unsigned char)*end)) end++;
            *end = 0;
            while (nt < 4 && p) { tok[nt++] = p; char* c = strchr(p, ','); if (c) { *c = 0; p = c + 1; } else p = nullptr; }
        } else {
            while (nt < 4) {
                while (*p && isspace((unsigned char)*p)) p++;
                if (!*p) break;
                tok[nt++] = p;
                while (*p && !isspace((unsigned char)*p)) p++;
                if (*p) *p++ = 0;
            }
        }
        if (nt < 4) continue;
        f(tok[0], tok[1], atoi(tok[3]));
    }
    gzclose(in);
    return true;
}
}  // namespace

void corpus::loadData(const char* clickFile, int userMin, int itemMin)
{
    nItems = 0; nUsers = 0; nClicks = 0;
    loadClicks(clickFile, userMin, itemMin);
    printf("\n  \"nUsers\": %d, \"nItems\": %d, \"nClicks\": %d\n", nUsers, nItems, nClicks);
}

void corpus::loadClicks(const char* clickFile, int userMin, int itemMin)
{
    printf("  Loading clicks from %s, userMin = %d  itemMin = %d ", clickFile, userMin, itemMin);
    std::string type = clickFile;
    type = type.size() >= 4 ? type.substr(type.size() - 4) : type;
    printf("\nclickFile_type %s\n", type.c_str());
    const bool csv = type == ".csv";

    std::unordered_map<std::string, int> uCounts, iCounts;
    int nRead = 0;
    bool ok = for_each_click(clickFile, csv, [&](const char* u, const char* i, int) {
        uCounts[u]++; iCounts[i]++;
        if (++nRead % 100000 == 0) { printf("."); fflush(stdout); }
    });
    if (!ok) { printf("Can't read clicks from %s.\n", clickFile); exit(1); }
    printf("\n  First pass: #users = %d, #items = %d, #clicks = %d\n", (int)uCounts.size(), (int)iCounts.size(), nRead);

    nRead = 0;
    for_each_click(clickFile, csv, [&](const char* u, const char* i, int t) {
        if (++nRead % 100000 == 0) { printf("."); fflush(stdout); }
        if (uCounts[u] < userMin || iCounts[i] < itemMin) return;
        nClicks++;
        auto fi = itemIds.find(i);
        if (fi == itemIds.end()) { rItemIds[nItems] = i; fi = itemIds.emplace(i, nItems++).first; }
        auto fu = userIds.find(u);
        if (fu == userIds.end()) { rUserIds[nUsers] = u; fu = userIds.emplace(u, nUsers++).first; pos_per_user.emplace_back(); }
        pos_per_user[fu->second].push_back(std::make_pair(fi->second, t));
    });

    printf("\n  Sorting clicks for each users ");
#pragma omp parallel for schedule(dynamic, 1024)
    for (int u = 0; u < nUsers; u++) {
        // the reference's std::sort is unstable on equal timestamps (src/corpus.cpp:219); a stable sort is one of its valid outcomes
        std::stable_sort(pos_per_user[u].begin(), pos_per_user[u].end(),
                         [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second < b.second; });
    }
    printf("\n");
}
