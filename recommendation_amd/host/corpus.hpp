// Click-file loader with the public surface of the reference's `corpus` (src/corpus.hpp:5-27).
#pragma once
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

class corpus
{
public:
    corpus() : nUsers(0), nItems(0), nClicks(0) {}

    std::vector<std::vector<std::pair<int, int> > > pos_per_user;  // per user: (item, time), sorted by time

    int nUsers;
    int nItems;
    int nClicks;

    std::unordered_map<std::string, int> userIds;
    std::unordered_map<std::string, int> itemIds;
    std::unordered_map<int, std::string> rUserIds;
    std::unordered_map<int, std::string> rItemIds;

    // "user item rating time" per line (".csv": comma separated), plain or gzip (src/corpus.cpp:9-226)
    void loadData(const char* clickFile, int userMin, int itemMin);

private:
    void loadClicks(const char* clickFile, int userMin, int itemMin);
    bool loadCache(const std::string& path);
    void saveCache(const std::string& path) const;
};
