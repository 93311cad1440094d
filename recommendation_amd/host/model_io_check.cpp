// model_io_check <n values> <scratch file> [compare]: writes n pseudo-random parameters with write_model_file, reads
// them back with read_model_file and checks every value against its own 6-decimal rounding; with `compare`, also
// writes the file the reference's per-element fprintf loop (src/model.cpp:176-192) produces and compares the bytes.
// Prints "write <s> read <s> bytes <n>".
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "model_io.hpp"

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: model_io_check <n> <file> [compare]\n"); return 2; }
    const long long n = atoll(argv[1]);
    const char* path = argv[2];
    std::vector<double> W((size_t)n), R((size_t)n);
    unsigned long long x = 88172645463325252ull;
    for (long long i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const double u = (double)(x >> 11) / 9007199254740992.0;
        // mostly parameter-sized values, some tiny, some large, some negative, exact ties of the sixth decimal
        W[i] = i % 97 == 0 ? (u - 0.5) * 1e9 : i % 89 == 0 ? (u - 0.5) * 1e-7 : i % 83 == 0 ? 0.0000005 * (double)(x % 1000) : (u - 0.5) * 4.0;
    }
    auto t0 = std::chrono::steady_clock::now();
    write_model_file(path, W.data(), n);
    auto t1 = std::chrono::steady_clock::now();
    read_model_file(path, R.data(), n);
    auto t2 = std::chrono::steady_clock::now();
    for (long long i = 0; i < n; i++) {
        char a[400], b[400];
        snprintf(a, sizeof a, "%f", W[i]);
        if (R[i] != strtod(a, nullptr)) { snprintf(b, sizeof b, "%f", R[i]); printf("value %lld: wrote %s read %s\n", i, a, b); return 1; }
    }
    FILE* f = fopen(path, "rb"); fseek(f, 0, SEEK_END); const long long bytes = ftell(f); fclose(f);
    if (argc > 3 && strcmp(argv[3], "compare") == 0) {
        std::string ref = std::string(path) + ".ref";
        FILE* g = fopen(ref.c_str(), "w");
        fprintf(g, "{\n"); fprintf(g, "  \"NW\": %d,\n", (int)n); fprintf(g, "  \"W\": [");
        for (long long w = 0; w < n; w++) { fprintf(g, "%f", W[w]); if (w < n - 1) fprintf(g, ", "); }
        fprintf(g, "]\n"); fprintf(g, "}\n"); fclose(g);
        FILE* a = fopen(path, "rb"); FILE* b = fopen(ref.c_str(), "rb");
        int ca, cb; long long pos = 0;
        do { ca = fgetc(a); cb = fgetc(b); if (ca != cb) { printf("byte %lld differs\n", pos); return 1; } pos++; } while (ca != EOF);
        fclose(a); fclose(b); remove(ref.c_str());
    }
    printf("write %.3f read %.3f bytes %lld\n", std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), bytes);
    return 0;
}
