// A hipcc-compiled caller of the reference's third export, genasm_gpu::ascii_to_twobit_strings (src/genasm_gpu.hpp:9), in the
// shape of the reference's own test of it (src/tests.cu:582-647: managed arrays of pointers and lengths, one launch of 32
// workgroups of 32 threads, a synchronise) — compiled against include/compat/ only.  Prints the packed bytes of every string as
// hex, one line each; the Python side (tests/test_reference_callers.py) holds them to the layout of src/genasm_gpu.cu:640-669.
#include "genasm_gpu.hpp"

#include <cstdio>
#include <string>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 3;                                                                   \
        }                                                                               \
    } while (0)

int main(int argc, char** argv)
{
    std::vector<std::string> inputs;
    for (int i = 1; i < argc; i++) inputs.push_back(std::string(argv[i]) == "-" ? std::string() : std::string(argv[i]));
    const size_t n = inputs.size();
    char **ascii_strings = nullptr, **twobit_strings = nullptr;
    long long* string_lengths = nullptr;
    CHECK(hipMallocManaged(&ascii_strings, sizeof(char*) * (n + 1)));
    CHECK(hipMallocManaged(&twobit_strings, sizeof(char*) * (n + 1)));
    CHECK(hipMallocManaged(&string_lengths, sizeof(long long) * (n + 1)));
    for (size_t i = 0; i < n; i++) {
        CHECK(hipMallocManaged(ascii_strings + i, inputs[i].size() + 1));
        CHECK(hipMallocManaged(twobit_strings + i, (inputs[i].size() + 3) / 4 + 1));
        string_lengths[i] = (long long)inputs[i].size();
        for (size_t j = 0; j < inputs[i].size(); j++) ascii_strings[i][j] = inputs[i][j];
        twobit_strings[i][(inputs[i].size() + 3) / 4] = (char)0x5a;                 // a guard byte behind every output
    }
    genasm_gpu::ascii_to_twobit_strings<<<32, 32>>>((int)n, string_lengths, ascii_strings, twobit_strings);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    for (size_t i = 0; i < n; i++) {
        const size_t nb = (inputs[i].size() + 3) / 4;
        if (twobit_strings[i][nb] != (char)0x5a) {
            std::fprintf(stderr, "string %zu: wrote past its %zu bytes\n", i, nb);
            return 4;
        }
        for (size_t b = 0; b < nb; b++) std::printf("%02x", (unsigned)(unsigned char)twobit_strings[i][b]);
        std::printf("\n");
    }
    return 0;
}
