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
#include <algorithm>
#include <parallel/algorithm>
#include <unordered_map>
#include <vector>
#include "../../include/recsys_engine.h"

void rec_set_error(const std::string& msg);

namespace {
// Labels as a trie over item ids, newest item first.  Edges live in an open-addressing table; every node also
// carries a 256-bit filter of the items that extend it, so that the long scans of findPathStars (which skips
// every item that does not extend the current label and keeps going to the start of the history) reject most
// items with one bit test.
struct Trie {
    std::unordered_map<uint64_t, int32_t> build;  // (node << 32 | item) -> child node, while inserting
    std::vector<uint8_t> is_label;                // per node
    std::vector<uint64_t> keys; std::vector<int32_t> vals; uint64_t mask = 0;   // frozen table
    std::vector<uint64_t> filter;                 // [nodes][4]
    int max_len = 0;
    Trie() : is_label(1, 0) {}
    static uint64_t mix(uint64_t k) { k *= 0x9E3779B97F4A7C15ull; return k ^ (k >> 29); }
    void insert(const int32_t* it, int len)
    {
        int32_t node = 0;
        for (int p = len - 1; p >= 0; p--) {  // newest first
            const uint64_t key = ((uint64_t)(uint32_t)node << 32) | (uint32_t)it[p];
            auto f = build.find(key);
            if (f == build.end()) {
                const int32_t child = (int32_t)is_label.size();
                is_label.push_back(0);
                build.emplace(key, child);
                node = child;
            } else node = f->second;
        }
        if (len > 0) { is_label[node] = 1; if (len > max_len) max_len = len; }
    }
    void freeze()
    {
        uint64_t cap = 16;
        while (cap < 2 * build.size() + 16) cap <<= 1;
        mask = cap - 1;
        keys.assign(cap, ~(uint64_t)0); vals.assign(cap, -1);
        filter.assign(is_label.size() * 4, 0);
        for (const auto& kv : build) {
            if (!is_label[kv.second]) continue;   // only label nodes can ever be returned
            uint64_t h = mix(kv.first) & mask;
            while (keys[h] != ~(uint64_t)0) h = (h + 1) & mask;
            keys[h] = kv.first; vals[h] = kv.second;
            const uint32_t node = (uint32_t)(kv.first >> 32), bit = (uint32_t)mix((uint32_t)kv.first) & 255;
            filter[(size_t)node * 4 + (bit >> 6)] |= (uint64_t)1 << (bit & 63);
        }
        std::unordered_map<uint64_t, int32_t>().swap(build);
    }
    // node of "item-<sequence of node>" when that is a label, else -1
    int32_t extend(int32_t node, int32_t item) const
    {
        const uint32_t bit = (uint32_t)mix((uint32_t)item) & 255;
        if (!((filter[(size_t)node * 4 + (bit >> 6)] >> (bit & 63)) & 1)) return -1;
        const uint64_t key = ((uint64_t)(uint32_t)node << 32) | (uint32_t)item;
        for (uint64_t h = mix(key) & mask;; h = (h + 1) & mask) {
            if (keys[h] == key) return vals[h];
            if (keys[h] == ~(uint64_t)0) return -1;
        }
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
    tr.freeze();
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

// ------------------------------------------------------------------------------------------------
// Label set of a 96-FSUB file.  The reference ships these files precomputed and holds no code that
// writes them; REBUS::loadPST (REBUS.cpp:3-87) only reads them.  Their content, checked against the
// bundled ML-5 files: every contiguous run of 1..L items of the training sequences (oldest item
// first) that occurs at least min_count times, counted over all users and positions.
// Mined level by level: a run of length l is a frequent run of length l-1 plus one item, so a
// position only needs the node id of the frequent run that starts there (-1: none) and the count
// map is keyed by (that node, next item).  A run can only be frequent if the run that starts one
// position later (its suffix) is frequent as well, which prunes most candidates.
namespace {
struct FsubNode { int64_t parent; int32_t item; int32_t len; };
}

extern "C" int64_t rec_fsub_mine(int32_t n_users, int32_t n_items, const int64_t* seq_off, const int32_t* seq_item, int32_t L,
                                 int64_t min_count, int64_t* lab_off, int32_t* lab_item, int64_t* n_label_items)
{
    if (!seq_off || !seq_item || !n_label_items) { rec_set_error("null argument"); return -1; }
    if (L < 1 || min_count < 1) { rec_set_error("L and min_count must be positive"); return -1; }
    const int64_t E = seq_off[n_users];
    std::vector<FsubNode> nodes;
    std::vector<int64_t> cur(E, -1), nxt(E, -1);
    // level 1
    {
        std::vector<int64_t> cnt(n_items, 0), id(n_items, -1);
        for (int64_t p = 0; p < E; p++) {
            if (seq_item[p] < 0 || seq_item[p] >= n_items) { rec_set_error("item id out of range in seq_item"); return -1; }
            cnt[seq_item[p]]++;
        }
        for (int32_t i = 0; i < n_items; i++)
            if (cnt[i] >= min_count) { id[i] = (int64_t)nodes.size(); nodes.push_back({-1, i, 1}); }
        for (int64_t p = 0; p < E; p++) cur[p] = id[seq_item[p]];
    }
    for (int32_t len = 2; len <= L; len++) {
        if (nodes.size() >= ((uint64_t)1 << 31)) { rec_set_error("too many labels"); return -1; }
        // candidates of this length as (node of the length-1 prefix, next item) keys; sorted, equal keys are adjacent
        std::vector<uint64_t> key(E);
        const uint64_t NONE = ~(uint64_t)0;
#pragma omp parallel for schedule(static, 4096)
        for (int32_t u = 0; u < n_users; u++)
            for (int64_t p = seq_off[u]; p < seq_off[u + 1]; p++)
                key[p] = (p + len > seq_off[u + 1] || cur[p] < 0 || cur[p + 1] < 0)
                             ? NONE : (((uint64_t)cur[p] << 32) | (uint32_t)seq_item[p + len - 1]);
        std::vector<uint64_t> sorted(key);
        __gnu_parallel::sort(sorted.begin(), sorted.end());
        std::vector<uint64_t> freq;   // frequent keys, ascending
        for (int64_t i = 0; i < E && sorted[i] != NONE;) {
            int64_t j = i + 1;
            while (j < E && sorted[j] == sorted[i]) j++;
            if (j - i >= min_count) freq.push_back(sorted[i]);
            i = j;
        }
        if (freq.empty()) break;
        const int64_t base = (int64_t)nodes.size();
        for (uint64_t k : freq) nodes.push_back({(int64_t)(k >> 32), (int32_t)(uint32_t)k, len});
#pragma omp parallel for schedule(static, 65536)
        for (int64_t p = 0; p < E; p++) {
            nxt[p] = -1;
            if (key[p] == NONE) continue;
            auto it = std::lower_bound(freq.begin(), freq.end(), key[p]);
            if (it != freq.end() && *it == key[p]) nxt[p] = base + (it - freq.begin());
        }
        cur.swap(nxt);
    }
    int64_t total = 0;
    for (const FsubNode& n : nodes) total += n.len;
    *n_label_items = total;
    if (lab_off && lab_item) {
        int64_t at = 0;
        for (size_t i = 0; i < nodes.size(); i++) {
            lab_off[i] = at;
            int64_t w = at + nodes[i].len - 1;
            for (int64_t n = (int64_t)i; n >= 0; n = nodes[n].parent) lab_item[w--] = nodes[n].item;
            at += nodes[i].len;
        }
        lab_off[nodes.size()] = at;
    }
    return (int64_t)nodes.size();
}
