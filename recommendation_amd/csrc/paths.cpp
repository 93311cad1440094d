// Host-side REBUS context builder: the matched path of REBUS::findPathStars (REBUS.cpp:484-524)
// for every training position and for the validation / test contexts of every user
// (REBUS.cpp:132-153, 758-782), produced as one CSR.
//
// The label set of a 96-FSUB file (REBUS.cpp:36-55) is held as an integer trie over the labels read
// newest item first: a candidate "x-seq" is the child x of the node of "seq", and it is a hit only
// when that child is itself the end of a label.  Replaces the std::map<string,int> lookups on
// freshly concatenated strings of the reference.
#include <stdint.h>
#include <omp.h>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/recsys_engine.h"

void rec_set_error(const std::string& msg);

namespace {
struct Trie {
    std::unordered_map<uint64_t, int32_t> edge;  // (node << 32 | item) -> child node
    std::vector<uint8_t> is_label;               // per node
    int max_len = 0;
    Trie() : is_label(1, 0) {}
    void insert(const int32_t* it, int len)
    {
        int32_t node = 0;
        for (int p = len - 1; p >= 0; p--) {  // newest first
            const uint64_t key = ((uint64_t)(uint32_t)node << 32) | (uint32_t)it[p];
            auto f = edge.find(key);
            if (f == edge.end()) {
                const int32_t child = (int32_t)is_label.size();
                is_label.push_back(0);
                edge.emplace(key, child);
                node = child;
            } else node = f->second;
        }
        if (len > 0) { is_label[node] = 1; if (len > max_len) max_len = len; }
    }
    // node of "item-<sequence of node>" when that is a label, else -1
    int32_t extend(int32_t node, int32_t item) const
    {
        auto f = edge.find(((uint64_t)(uint32_t)node << 32) | (uint32_t)item);
        if (f == edge.end() || !is_label[f->second]) return -1;
        return f->second;
    }
};

// path of the context whose prefix is seq[0..c) (position n of the prefix is the validation item)
int walk(const Trie& tr, const int32_t* seq, int n, int32_t val, int c, int32_t n_items, int32_t* out)
{
    int plen = 0; int32_t node = 0;
    for (int t = c - 1; t >= 0 && plen < tr.max_len; t--) {
        const int32_t it = t < n ? seq[t] : val;
        const int32_t nx = tr.extend(node, it);
        if (nx >= 0) { node = nx; if (out) out[plen] = it; plen++; }  // a miss is skipped ("star"), the walk goes on
    }
    if (plen == 0) { if (out) out[0] = n_items; plen = 1; }          // empty path -> Root, REBUS.cpp:518-521
    return plen;
}
}  // namespace

extern "C" int64_t rec_rebus_build_paths(int32_t n_users, int32_t n_items, const int64_t* seq_off, const int32_t* seq_item,
                                         const int32_t* val_item, int64_t n_labels, const int64_t* lab_off,
                                         const int32_t* lab_item, int64_t* path_off, int32_t* path_item)
{
    if (!seq_off || !seq_item || !val_item || !lab_off || !path_off) { rec_set_error("null argument"); return -1; }
    Trie tr;
    for (int64_t i = 0; i < n_labels; i++) tr.insert(lab_item + lab_off[i], (int)(lab_off[i + 1] - lab_off[i]));
    // pass 1: lengths
    const int64_t nctx = seq_off[n_users] + 2 * (int64_t)n_users;
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t u = 0; u < n_users; u++) {
        const int32_t* seq = seq_item + seq_off[u]; const int n = (int)(seq_off[u + 1] - seq_off[u]);
        const int64_t base = seq_off[u] + 2 * (int64_t)u;
        for (int c = 1; c <= n + 2; c++) {
            const bool used = c <= n - 1 || (val_item[u] >= 0 && c <= n + 1);
            path_off[base + c] = used ? walk(tr, seq, n, val_item[u], c, n_items, nullptr) : 0;  // length, shifted by one
        }
    }
    // exclusive scan (path_off[i+1] currently holds the length of context i)
    int64_t run = 0;
    path_off[0] = 0;
    for (int64_t i = 0; i < nctx; i++) { run += path_off[i + 1]; path_off[i + 1] = run; }
    if (!path_item) return run;
#pragma omp parallel for schedule(dynamic, 256)
    for (int32_t u = 0; u < n_users; u++) {
        const int32_t* seq = seq_item + seq_off[u]; const int n = (int)(seq_off[u + 1] - seq_off[u]);
        const int64_t base = seq_off[u] + 2 * (int64_t)u;
        for (int c = 1; c <= n + 2; c++)
            if (path_off[base + c] > path_off[base + c - 1]) walk(tr, seq, n, val_item[u], c, n_items, path_item + path_off[base + c - 1]);
    }
    return run;
}
