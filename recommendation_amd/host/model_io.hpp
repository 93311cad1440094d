// The saved-model text file of src/model.cpp:176-232, byte for byte:
//   {\n  "NW": <NW>,\n  "W": [<%f>, <%f>, ...]\n}\n
// written and parsed in parallel slices so that a 77M-parameter model (0.8 GB of text) takes seconds, not minutes.
#pragma once
#include <cstdint>

// prints the reference's messages and exits with 1 on failure, as saveModel / loadModel do
void write_model_file(const char* path, const double* W, int64_t NW);
void read_model_file(const char* path, double* W, int64_t NW);
