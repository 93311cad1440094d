#include "model_io.hpp"
#include <omp.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// Every value goes through the C library's own "%f" (snprintf), so the digits are those fprintf writes; the slices
// are formatted by all cores into private buffers and written out in order.
void write_model_file(const char* path, const double* W, int64_t NW)
{
    FILE* f = fopen(path, "w");
    if (!f) { printf("An error occurred while opening file %s.\n", path); exit(1); }
    fprintf(f, "{\n");
    fprintf(f, "  \"NW\": %lld,\n", (long long)NW);
    fprintf(f, "  \"W\": [");
    const int64_t SLICE = 1 << 16;                 // values per buffer
    const int64_t nslices = (NW + SLICE - 1) / SLICE;
    const int T = omp_get_max_threads();
    for (int64_t s0 = 0; s0 < nslices; s0 += T) {
        const int64_t cnt = nslices - s0 < T ? nslices - s0 : T;
        std::vector<std::string> out(cnt);
#pragma omp parallel for schedule(static, 1)
        for (int64_t k = 0; k < cnt; k++) {
            const int64_t a = (s0 + k) * SLICE, b = a + SLICE < NW ? a + SLICE : NW;
            std::string& o = out[k];
            o.reserve((size_t)(b - a) * 12);
            char tmp[400];
            for (int64_t w = a; w < b; w++) {
                const int len = snprintf(tmp, sizeof tmp, "%f", W[w]);
                o.append(tmp, (size_t)len);
                if (w < NW - 1) o.append(", ", 2);
            }
        }
        for (int64_t k = 0; k < cnt; k++)
            if (fwrite(out[k].data(), 1, out[k].size(), f) != out[k].size()) { printf("An error occurred while writing file %s.\n", path); exit(1); }
    }
    fprintf(f, "]\n");
    fprintf(f, "}\n");
    fclose(f);
}

// The reference reads line by line and extracts `NW` and the W values with operator>> (src/model.cpp:204-228); here the
// file is read whole, the W line cut at commas into one range per thread, and every range parsed with strtod.
void read_model_file(const char* path, double* W, int64_t NW)
{
    FILE* f = fopen(path, "rb");
    if (!f) { printf("Can't read init solution from %s.\n", path); exit(1); }
    fseek(f, 0, SEEK_END);
    const long long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)size + 1);
    if (size > 0 && fread(buf.data(), 1, (size_t)size, f) != (size_t)size) { printf("Can't read init solution from %s.\n", path); exit(1); }
    fclose(f);
    buf[(size_t)size] = 0;
    const char* p = strstr(buf.data(), "\"NW\":");
    if (p) {
        const long long nw = strtoll(p + 5, nullptr, 10);
        if (nw != NW) { printf("NW not match."); exit(1); }
    }
    const char* w = strstr(buf.data(), "\"W\":");
    if (!w) { if (NW > 0) { printf("Read W[] error."); exit(1); } return; }
    const char* a = strchr(w, '[');
    const char* end = a ? strchr(a, ']') : nullptr;
    if (!a || !end) { printf("Read W[] error."); exit(1); }
    a++;
    if (NW == 0) return;
    const int T = omp_get_max_threads();
    // range t starts after the first comma at or behind its byte offset
    std::vector<const char*> start(T + 1, end);
    start[0] = a;
    for (int t = 1; t < T; t++) {
        const char* q = a + (end - a) * (long long)t / T;
        const char* c = (const char*)memchr(q, ',', (size_t)(end - q));
        start[t] = c ? c + 1 : end;
    }
    for (int t = 1; t <= T; t++) if (start[t] < start[t - 1]) start[t] = start[t - 1];
    std::vector<int64_t> count(T + 1, 0);
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < T; t++) {
        int64_t c = 0;
        for (const char* q = start[t]; q < start[t + 1]; q++) c += *q == ',';
        // a range that holds any value holds one more value than separators, unless its last value ends with the comma
        // that opens the next range
        bool any = false;
        for (const char* q = start[t]; q < start[t + 1] && !any; q++) any = *q != ' ' && *q != ',';
        const bool closed = start[t + 1] > start[t] && start[t + 1] < end + 1 && start[t + 1] != end && start[t + 1][-1] == ',';
        count[t + 1] = any ? c + (closed ? 0 : 1) : 0;
    }
    for (int t = 0; t < T; t++) count[t + 1] += count[t];
    if (count[T] != NW) { printf("Read W[] error."); exit(1); }
    int bad = 0;
#pragma omp parallel for schedule(static, 1) reduction(+ : bad)
    for (int t = 0; t < T; t++) {
        const char* q = start[t];
        for (int64_t i = count[t]; i < count[t + 1]; i++) {
            char* e2 = nullptr;
            W[i] = strtod(q, &e2);
            if (e2 == q) { bad++; break; }
            q = e2;
            while (q < end && (*q == ',' || *q == ' ')) q++;
        }
    }
    if (bad) { printf("Read W[] error."); exit(1); }
}
