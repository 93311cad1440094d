// corpus_dump: what corpus::loadData (src/corpus.cpp:9-226) leaves behind, as text on stdout:
//   nUsers nItems nClicks, then one line per user: n item time item time ...
// Lets a click file be checked against the reference's loader (ids by first appearance, per-user order after its
// unstable sort by time) without a GPU.
//
//   corpus_dump <click file> <userMin> <itemMin>
#include <cstdio>
#include <cstdlib>
#include "corpus.hpp"

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: corpus_dump <click file> <userMin> <itemMin>\n"); return 1; }
    corpus corp;
    FILE* out = stdout;
    stdout = stderr;   // the loader's progress lines go to stderr, the dump alone to stdout
    corp.loadData(argv[1], atoi(argv[2]), atoi(argv[3]));
    fprintf(out, "%d %d %d\n", corp.nUsers, corp.nItems, corp.nClicks);
    for (int u = 0; u < corp.nUsers; u++) {
        fprintf(out, "%d", (int)corp.pos_per_user[u].size());
        for (auto& p : corp.pos_per_user[u]) fprintf(out, " %d %d", p.first, p.second);
        fputc('\n', out);
    }
    return 0;
}
