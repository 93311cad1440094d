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
#include <sys/stat.h>

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

// Binary image of a loaded corpus next to the click file (RECSYS_CORPUS_CACHE=1): the parse of a 100M-line file
// dominates start-up, and the ids - which the 96-FSUB files refer to - must not change between runs.
namespace {
const char CACHE_MAGIC[8] = {'R', 'E', 'C', 'C', 'O', 'R', 'P', '1'};
std::string cache_name(const char* clickFile, int userMin, int itemMin)
{
    return std::string(clickFile) + ".u" + std::to_string(userMin) + "i" + std::to_string(itemMin) + ".recbin";
}
bool newer_than(const std::string& a, const char* b)
{
    struct stat sa, sb;
    return stat(a.c_str(), &sa) == 0 && stat(b, &sb) == 0 && sa.st_mtime >= sb.st_mtime;
}
void put_str(FILE* f, const std::string& s) { const int32_t n = (int32_t)s.size(); fwrite(&n, 4, 1, f); fwrite(s.data(), 1, n, f); }
bool get_str(FILE* f, std::string& s)
{
    int32_t n; if (fread(&n, 4, 1, f) != 1 || n < 0 || n > (1 << 20)) return false;
    s.resize(n); return n == 0 || fread(&s[0], 1, n, f) == (size_t)n;
}
}  // namespace

bool corpus::loadCache(const std::string& path)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[8]; int32_t hdr[3];
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, CACHE_MAGIC, 8) == 0 && fread(hdr, 4, 3, f) == 3;
    if (ok) {
        nUsers = hdr[0]; nItems = hdr[1]; nClicks = hdr[2];
        pos_per_user.assign(nUsers, {});
        for (int u = 0; u < nUsers && ok; u++) {
            int32_t n; ok = fread(&n, 4, 1, f) == 1 && n >= 0;
            if (!ok) break;
            pos_per_user[u].resize(n);
            ok = n == 0 || fread(pos_per_user[u].data(), sizeof(std::pair<int, int>), n, f) == (size_t)n;
        }
        std::string s;
        for (int u = 0; u < nUsers && ok; u++) { ok = get_str(f, s); if (ok) { userIds[s] = u; rUserIds[u] = s; } }
        for (int i = 0; i < nItems && ok; i++) { ok = get_str(f, s); if (ok) { itemIds[s] = i; rItemIds[i] = s; } }
    }
    fclose(f);
    if (!ok) { pos_per_user.clear(); userIds.clear(); itemIds.clear(); rUserIds.clear(); rItemIds.clear(); nUsers = nItems = nClicks = 0; }
    return ok;
}

void corpus::saveCache(const std::string& path) const
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return;
    const int32_t hdr[3] = {nUsers, nItems, nClicks};
    fwrite(CACHE_MAGIC, 1, 8, f); fwrite(hdr, 4, 3, f);
    for (int u = 0; u < nUsers; u++) {
        const int32_t n = (int32_t)pos_per_user[u].size();
        fwrite(&n, 4, 1, f);
        if (n) fwrite(pos_per_user[u].data(), sizeof(std::pair<int, int>), n, f);
    }
    for (int u = 0; u < nUsers; u++) put_str(f, rUserIds.at(u));
    for (int i = 0; i < nItems; i++) put_str(f, rItemIds.at(i));
    fclose(f);
}

void corpus::loadData(const char* clickFile, int userMin, int itemMin)
{
    nItems = 0; nUsers = 0; nClicks = 0;
    const char* env = getenv("RECSYS_CORPUS_CACHE");
    const bool use_cache = env && atoi(env) != 0;
    const std::string cache = cache_name(clickFile, userMin, itemMin);
    if (use_cache && newer_than(cache, clickFile) && loadCache(cache)) {
        printf("  Loading clicks from cache %s\n", cache.c_str());
    } else {
        loadClicks(clickFile, userMin, itemMin);
        if (use_cache) saveCache(cache);
    }
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
        // std::sort with the reference's comparator on the same input order (src/corpus.cpp:4-7, 219): it is unstable
        // on equal timestamps, and with the same libstdc++ introsort the permutation of tied clicks is the reference's
        std::sort(pos_per_user[u].begin(), pos_per_user[u].end(),
                  [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second < b.second; });
    }
    printf("\n");
}
