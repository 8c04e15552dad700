// Batches pipelined over two handles and two streams through the device-pointer layer of the C ABI
// (INTEGRATION.md §4b): the wavefronts of batch k+1 start while the last pairs of batch k finish.
// Every batch is checked against the host-pointer entry point scrg_align_pairs.
//
// Build:  hipcc --offload-arch=gfx950 -std=c++17 -Iinclude examples/pipeline_example.cpp
//               -Lscrooge_amd -lscrooge_amd -Wl,-rpath,$PWD/scrooge_amd -o /tmp/pipeline_example
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "scrooge_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_SCRG(x) do { scrg_status s_ = (x); if (s_ != SCRG_OK) { fprintf(stderr, "%s: %s\n", #x, scrg_status_string(s_)); return 2; } } while (0)

struct Batch {
    std::vector<std::string> texts, reads;
    // device side
    char* d_ascii = nullptr;
    uint64_t* d_seq = nullptr;
    scrg_pair_desc* d_pairs = nullptr;
    uint32_t* d_bad = nullptr;
    uint64_t words = 0;
};

static std::string random_seq(std::mt19937& g, size_t n)
{
    std::string s(n, 'A');
    for (char& c : s) c = "ACGT"[g() & 3];
    return s;
}

int main()
{
    const int n_batches = 4, n_pairs = 4000, len = 1500;
    if (scrg_device_count() <= 0) { fprintf(stderr, "no usable HIP device\n"); return 2; }
    std::mt19937 gen(7);
    scrg_ctx* ctx[2];
    void* stream[2];
    CHECK_SCRG(scrg_stream_create(0, +1, &stream[0]));      // two priorities: two hardware queues
    CHECK_SCRG(scrg_stream_create(0, -1, &stream[1]));
    for (int k = 0; k < 2; k++) {
        CHECK_SCRG(scrg_ctx_create(0, &ctx[k]));
        CHECK_SCRG(scrg_ctx_set_stream(ctx[k], stream[k]));
    }
    // output buffers, one set per lane
    const uint64_t cap = (2 * len + 8 + 15) / 16 * 16;
    scrg_run* d_runs[2];
    int64_t* d_ed[2];
    uint32_t *d_nruns[2], *d_status[2];
    for (int k = 0; k < 2; k++) {
        CHECK_HIP(hipMalloc(&d_runs[k], n_pairs * cap * sizeof(scrg_run)));
        CHECK_HIP(hipMalloc(&d_ed[k], n_pairs * sizeof(int64_t)));
        CHECK_HIP(hipMalloc(&d_nruns[k], n_pairs * sizeof(uint32_t)));
        CHECK_HIP(hipMalloc(&d_status[k], n_pairs * sizeof(uint32_t)));
    }
    std::vector<Batch> batches(n_batches);
    std::vector<std::vector<int64_t>> got(n_batches, std::vector<int64_t>(n_pairs));
    for (int b = 0; b < n_batches; b++) {
        Batch& B = batches[b];
        // reads with ~8 % substitutions against their source text
        const uint64_t tw = (len + 200 + 31) / 32, rw = (len + 31) / 32;
        B.words = (uint64_t)n_pairs * (tw + rw);
        std::vector<char> ascii(B.words * 32, 0);
        std::vector<scrg_pair_desc> desc(n_pairs);
        for (int i = 0; i < n_pairs; i++) {
            std::string t = random_seq(gen, len + 200), r = t.substr(0, len);
            for (char& c : r) if (gen() % 12 == 0) c = "ACGT"[gen() & 3];
            B.texts.push_back(t);
            B.reads.push_back(r);
            memcpy(&ascii[(uint64_t)i * (tw + rw) * 32], t.data(), t.size());
            memcpy(&ascii[((uint64_t)i * (tw + rw) + tw) * 32], r.data(), r.size());
            desc[i] = {(uint64_t)i * (tw + rw) * 32, t.size(), ((uint64_t)i * (tw + rw) + tw) * 32, r.size(), (uint64_t)i * cap, cap};
        }
        CHECK_HIP(hipMalloc(&B.d_ascii, ascii.size()));
        CHECK_HIP(hipMalloc(&B.d_seq, (B.words + SCRG_SEQ_PAD_WORDS) * 8));
        CHECK_HIP(hipMalloc(&B.d_pairs, desc.size() * sizeof(scrg_pair_desc)));
        CHECK_HIP(hipMalloc(&B.d_bad, 4));
        CHECK_HIP(hipMemset(B.d_seq, 0, (B.words + SCRG_SEQ_PAD_WORDS) * 8));
        CHECK_HIP(hipMemset(B.d_bad, 0, 4));
        CHECK_HIP(hipMemcpy(B.d_ascii, ascii.data(), ascii.size(), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(B.d_pairs, desc.data(), desc.size() * sizeof(scrg_pair_desc), hipMemcpyHostToDevice));
    }
    CHECK_HIP(hipDeviceSynchronize());
    // the pipeline: batch b on lane b & 1; results leave the lane's buffers before the lane is reused
    for (int b = 0; b < n_batches; b++) {
        const int k = b & 1;
        Batch& B = batches[b];
        hipStream_t s = static_cast<hipStream_t>(stream[k]);
        CHECK_SCRG(scrg_pack_planar(ctx[k], B.d_ascii, B.words, B.d_seq, B.d_bad));
        CHECK_SCRG(scrg_align_device(ctx[k], nullptr, n_pairs, B.d_seq, B.d_pairs, d_runs[k], d_ed[k], d_nruns[k], d_status[k]));
        CHECK_HIP(hipMemcpyAsync(got[b].data(), d_ed[k], n_pairs * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    }
    CHECK_HIP(hipDeviceSynchronize());
    // check against the host-pointer entry point
    int bad = 0;
    for (int b = 0; b < n_batches; b++) {
        const Batch& B = batches[b];
        std::vector<const char*> tp(n_pairs), rp(n_pairs);
        std::vector<uint64_t> tl(n_pairs), rl(n_pairs);
        for (int i = 0; i < n_pairs; i++) { tp[i] = B.texts[i].data(); tl[i] = B.texts[i].size(); rp[i] = B.reads[i].data(); rl[i] = B.reads[i].size(); }
        scrg_result* res = nullptr;
        CHECK_SCRG(scrg_align_pairs(ctx[0], nullptr, n_pairs, tp.data(), tl.data(), rp.data(), rl.data(), &res));
        for (int i = 0; i < n_pairs; i++) bad += res->edit_distance[i] != got[b][i];
        scrg_result_free(res);
    }
    printf("batches=%d pairs_per_batch=%d mismatches=%d\n", n_batches, n_pairs, bad);
    for (int k = 0; k < 2; k++) { scrg_ctx_destroy(ctx[k]); scrg_stream_destroy(stream[k]); }
    return bad ? 1 : 0;
}
