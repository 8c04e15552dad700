// scrg_api.cpp — host side of the C ABI declared in include/scrooge_amd.h.
//
// Mirrors the staging the reference does around its kernel
// (src/genasm_gpu.cu:890-1065: concatenate + pack sequences, size the CIGAR
// storage, launch, read CIGARs back) with explicit device memory instead of
// managed memory, status codes instead of exit(), and one handle per device
// instead of __managed__ globals.  There is no CPU fallback anywhere in this
// file: without a HIP device every entry point fails with SCRG_ERR_NO_DEVICE.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "genasm_kernels.h"
#include "edit_stream.h"
#include "../../include/scrooge_amd_io.h"

namespace {

std::atomic<int> g_log{0};
std::atomic<int> g_live_ctx{0};          // handles alive: the result pool is emptied when the last one goes
// the work queue is a 32-bit counter and every wavefront over-asks by up to 64 once it is empty (at most 32 wavefronts
// on each of at most 1024 CUs): leave room for that, or the counter could wrap and hand out low indices a second time
constexpr uint64_t kMaxPairsPerLaunch = 0xffffffffull - 64ull * 32ull * 1024ull;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct HostPinned {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

int64_t now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

// (heavy = true: every index is a block of work — a few of them are worth the threads)
template <typename F> void parallel_for(uint64_t n, F f, bool heavy = false)
{
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = hw ? std::min(hw, 16u) : 4u;
    if (heavy) nt = (unsigned)std::min<uint64_t>(nt, n);
    if (n < (heavy ? 2u : 64u) || nt <= 1) {
        for (uint64_t i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<uint64_t> next{0};
    const uint64_t chunk = std::max<uint64_t>(1, n / (nt * 16));
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; k++)
        th.emplace_back([&]() {
            for (;;) {
                uint64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                uint64_t e = std::min(n, b + chunk);
                for (uint64_t i = b; i < e; i++) f(i);
            }
        });
    for (auto& t : th) t.join();
}

}  // namespace

struct scrg_ctx {
    int device = 0;
    int n_cus = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool have_timing = false;
    std::string last_error;

    DevBuf counter;     // work queue head
    DevBuf spill;       // HBM overflow rows of R
    DevBuf stats;       // profiling counters (params.reserved[1] != 0)
    DevBuf sort_ws;     // scrg_decode_edit_stream: pair order by stream length (indices, sorted keys / indices, radix sort scratch)
    // staging used by the host-pointer entry points
    HostPinned h_ascii;
    HostPinned h_desc;   // problem descriptors (pinned: no page faults after the first call, full-rate H2D)
    HostPinned h_out, h_runs, h_off;   // pinned landing zones: per-pair scalars, dense runs, dense offsets (full-rate D2H / H2D;
                                       // results are copied out to the caller's arrays in parallel)
    // a genome kept resident by scrg_genome_set(): the first `genome_words` words of d_seq hold it, packed
    uint64_t genome_len = 0, genome_words = 0;
    bool genome_resident = false;
    DevBuf d_ascii, d_seq, d_pairs, d_runs, d_ed, d_nruns, d_status, d_bad, d_dense_off, d_dense;

    scrg_status fail(scrg_status s, const char* what, hipError_t e = hipSuccess)
    {
        last_error = what;
        if (e != hipSuccess) {
            last_error += ": ";
            last_error += hipGetErrorString(e);
            (void)hipGetLastError();
        }
        if (g_log.load()) fprintf(stderr, "[scrooge_amd] error: %s\n", last_error.c_str());
        return s;
    }
};

#define HIP_TRY(ctx, call)                                                     \
    do {                                                                       \
        hipError_t e__ = (call);                                               \
        if (e__ != hipSuccess)                                                 \
            return (ctx)->fail(e__ == hipErrorOutOfMemory ? SCRG_ERR_OOM : SCRG_ERR_HIP, #call, e__); \
    } while (0)

// Host entry points never let a C++ exception cross the C boundary (std::vector growth on huge batches):
// allocation failures become SCRG_ERR_OOM, anything else SCRG_ERR_INVALID_ARG.
template <typename F> static scrg_status guarded(scrg_ctx* c, F&& f)
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return c ? c->fail(SCRG_ERR_OOM, "host allocation failed") : SCRG_ERR_OOM;
    } catch (...) {
        return c ? c->fail(SCRG_ERR_INVALID_ARG, "unexpected exception") : SCRG_ERR_INVALID_ARG;
    }
}

// Result arrays are recycled: a batch of millions of pairs returns hundreds of MB, and freshly mapped pages cost
// more (first-touch faults) than filling them.  scrg_result_free() parks the big arrays here, the next call of
// similar size takes them back.  At most 12 blocks / 2 GB are kept; everything else goes to malloc/free.
namespace {
struct ResultPool {
    struct Block { void* p; size_t cap; };
    std::mutex mu;
    std::vector<Block> blocks;
    size_t held = 0;
    static constexpr size_t kMinPooled = 1u << 20, kMaxHeld = 2ull << 30, kMaxBlocks = 12;

    void* get(size_t bytes, bool zero)
    {
        void* p = nullptr;
        size_t cap = 0;
        if (bytes >= kMinPooled) {
            std::lock_guard<std::mutex> g(mu);
            size_t best = blocks.size();
            for (size_t i = 0; i < blocks.size(); i++)
                if (blocks[i].cap >= bytes && blocks[i].cap <= 2 * bytes + (64u << 20) &&
                    (best == blocks.size() || blocks[i].cap < blocks[best].cap))
                    best = i;
            if (best != blocks.size()) {
                p = blocks[best].p;
                cap = blocks[best].cap;
                held -= cap;
                blocks.erase(blocks.begin() + (long)best);
            }
        }
        if (!p) {
            cap = bytes >= kMinPooled ? bytes + bytes / 8 : bytes;
            p = malloc(cap + sizeof(size_t) * 2);
            if (!p) return nullptr;
            static_cast<size_t*>(p)[0] = cap;
        }
        void* user = static_cast<char*>(p) + sizeof(size_t) * 2;
        if (zero) memset(user, 0, bytes);
        return user;
    }
    void put(void* user)
    {
        if (!user) return;
        void* p = static_cast<char*>(user) - sizeof(size_t) * 2;
        const size_t cap = static_cast<size_t*>(p)[0];
        if (cap >= kMinPooled) {
            std::lock_guard<std::mutex> g(mu);
            if (blocks.size() < kMaxBlocks && held + cap <= kMaxHeld) {
                blocks.push_back({p, cap});
                held += cap;
                return;
            }
        }
        free(p);
    }
    void trim()
    {
        std::lock_guard<std::mutex> g(mu);
        for (Block& b : blocks) free(b.p);
        blocks.clear();
        held = 0;
    }
};
ResultPool g_pool;
}  // namespace

extern "C" {

void scrg_params_default(scrg_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->W = 64;                // src/genasm_cpu.cpp:7
    p->O = 33;                // src/genasm_cpu.cpp:9
    p->lanes_per_pair = 0;    // 0 = chosen for W: see scrg_params_resolve()
    p->lds_rows = 0;
    p->waves_per_cu = 0;
    p->sort_by_length = 1;
}

const char* scrg_status_string(scrg_status s)
{
    switch (s) {
    case SCRG_OK: return "ok";
    case SCRG_ERR_INVALID_ARG: return "invalid argument";
    case SCRG_ERR_BAD_BASE: return "sequence contains a character other than ACGTacgt";
    case SCRG_ERR_NO_DEVICE: return "no usable HIP device";
    case SCRG_ERR_HIP: return "HIP runtime error";
    case SCRG_ERR_OOM: return "out of memory";
    case SCRG_ERR_CIGAR_OVERFLOW: return "CIGAR arena slice too small";
    default: return "unknown status";
    }
}

void scrg_set_log(int enabled) { g_log.store(enabled ? 1 : 0); }
int scrg_get_log(void) { return g_log.load(); }

int scrg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

scrg_status scrg_ctx_create(int device, scrg_ctx** out)
{
    if (!out) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        (void)hipGetLastError();
        return SCRG_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) return SCRG_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SCRG_ERR_NO_DEVICE;
    scrg_ctx* c = new (std::nothrow) scrg_ctx();
    if (!c) return SCRG_ERR_OOM;
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_stop) != hipSuccess) {
        delete c;
        return SCRG_ERR_HIP;
    }
    c->stream = c->own_stream;
    g_live_ctx.fetch_add(1);
    if (g_log.load())
        fprintf(stderr, "[scrooge_amd] device %d: %s, %d CUs, arch %s\n", device, prop.name, c->n_cus,
                prop.gcnArchName);
    *out = c;
    return SCRG_OK;
}

void scrg_ctx_destroy(scrg_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    c->counter.release();
    c->spill.release();
    c->stats.release();
    c->sort_ws.release();
    c->h_ascii.release();
    c->h_desc.release();
    for (HostPinned* b : {&c->h_out, &c->h_runs, &c->h_off}) b->release();
    for (DevBuf* b : {&c->d_ascii, &c->d_seq, &c->d_pairs, &c->d_runs, &c->d_ed, &c->d_nruns, &c->d_status,
                      &c->d_bad, &c->d_dense_off, &c->d_dense})
        b->release();
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    if (g_live_ctx.fetch_sub(1) == 1) g_pool.trim();      // last handle gone: give the recycled result arrays back
}

scrg_status scrg_ctx_set_stream(scrg_ctx* c, void* hip_stream)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    c->stream = static_cast<hipStream_t>(hip_stream);
    return SCRG_OK;
}

scrg_status scrg_ctx_use_own_stream(scrg_ctx* c)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    c->stream = c->own_stream;
    return SCRG_OK;
}

scrg_status scrg_stream_create(int device, int priority, void** stream)
{
    if (!stream || priority < -1 || priority > 1) return SCRG_ERR_INVALID_ARG;
    hipStream_t s = nullptr;
    if (hipSetDevice(device) != hipSuccess) return SCRG_ERR_NO_DEVICE;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority) != hipSuccess) return SCRG_ERR_HIP;
    *stream = s;
    return SCRG_OK;
}

scrg_status scrg_stream_destroy(void* stream)
{
    if (!stream) return SCRG_OK;
    return hipStreamDestroy(static_cast<hipStream_t>(stream)) == hipSuccess ? SCRG_OK : SCRG_ERR_HIP;
}

const char* scrg_last_error(const scrg_ctx* c) { return c ? c->last_error.c_str() : "null context"; }

// ---------------------------------------------------------------------------
// launch geometry
// ---------------------------------------------------------------------------
static bool resolve_params(const scrg_params* in, scrg_params* p)
{
    scrg_params_default(p);
    if (in) {
        if (in->W) p->W = in->W;
        if (in->O || in->W) p->O = in->O;
        p->lanes_per_pair = in->lanes_per_pair;
        p->lds_rows = in->lds_rows;
        p->waves_per_cu = in->waves_per_cu;
        p->sort_by_length = in->sort_by_length;
        p->text_stride_words = in->text_stride_words;
        p->read_stride_words = in->read_stride_words;
        p->reserved[0] = in->reserved[0];      // ablation switches and profiling counters travel with the parameters
        p->reserved[1] = in->reserved[1];
    }
    // experiment switches: only those that leave the results intact, unless this is an ablation build (genasm_kernels.h)
    if (p->reserved[0] & ~scrg::SCRG_ALLOWED_SWITCHES) return false;
    if (p->text_stride_words == 0) p->text_stride_words = 1;
    if (p->read_stride_words == 0) p->read_stride_words = 1;
    if (p->text_stride_words < 1 || p->read_stride_words < 1) return false;
    if (p->W < 2 || p->W > 256) return false;
    const int tbl = p->W - p->O;
    if (tbl < 1 || p->O < 1) return false;   // O = 0 (no overlap) would let the traceback read the boundary column
    const size_t row_bytes = (size_t)scrg::stored_row_dwords(p->W, tbl) * 4;
    if (p->W > 64) {
        // one pair per lane here too (genasm_lane_mw_kernel.hip: multi-word difference vectors, the table in HBM); the
        // GenASM-row kernel with multi-word entries (genasm_kernel_multiword.hip) stays selectable: slots of 32 or 64
        // lanes; as many rows of R in LDS as fit in about 40 KB per wavefront, the rest of a window's rows go to HBM
        if (p->lanes_per_pair == 0) p->lanes_per_pair = 1;
        if (p->lanes_per_pair != 1 && p->lanes_per_pair != 32 && p->lanes_per_pair != 64) return false;
        if (p->lanes_per_pair == 1) {
            if (p->lds_rows == 0) p->lds_rows = 12;                 // (not used)
            if (p->waves_per_cu == 0) p->waves_per_cu = 8;          // two per SIMD: the kernel needs up to 256 VGPRs
        } else if (p->lds_rows == 0) {
            const size_t fit = (40u << 10) / (row_bytes * (64 / p->lanes_per_pair));
            p->lds_rows = (int32_t)std::min<size_t>(32, std::max<size_t>(4, fit));
        }
    } else {
        // one pair per lane for every W <= 64: genasm_lane_kernel.hip (table in registers) while a window's traceback
        // consumes at most W-O <= 31 characters, genasm_lane_mw_kernel.hip (64-bit rows, table in HBM) beyond
        if (p->lanes_per_pair == 0) p->lanes_per_pair = 1;
        if (p->lds_rows == 0) p->lds_rows = 12;
    }
    // 11 and 12 wavefronts per CU align equally fast (the kernel is issue-bound); 11 leaves VGPRs and LDS on
    // every CU for kernels of other streams (RCCL's gather in bench.py --gpus N).  The LDS footprint caps it.
    if (p->waves_per_cu == 0) p->waves_per_cu = p->lanes_per_pair == 1 ? 16 : 11;
    const int g = p->lanes_per_pair;
    if (g == 1) return p->waves_per_cu >= 1 && p->waves_per_cu <= 32;      // no table in LDS: lds_rows is not used
    if (p->text_stride_words != 1 || p->read_stride_words != 1) return false;   // strided sequences: lane kernel only
    if (!(g == 4 || g == 8 || g == 16 || g == 32 || g == 64)) return false;
    if (p->lds_rows < 1) return false;
    if (p->lds_rows > p->W + 1) p->lds_rows = p->W + 1;
    // one wavefront's table has to fit the 160 KB of a CU
    const size_t per_row = row_bytes * (64 / g);
    if ((size_t)p->lds_rows * per_row > (150u << 10)) p->lds_rows = (int32_t)std::max<size_t>(1, (150u << 10) / per_row);
    if (p->waves_per_cu < 1 || p->waves_per_cu > 32) return false;
    return true;
}

static size_t lds_bytes_for(const scrg_params& p)
{
    if (p.lanes_per_pair == 1 && (p.W > 64 || p.W - p.O > 31))
        return scrg::lane_mw_lds_bytes(p.W - p.O);   // genasm_lane_mw_kernel: CIGAR ring + insertion-run lengths (the table is in HBM)
    if (p.lanes_per_pair == 1) return 64 * (68 + 36 + 32 + 8);  // per lane: CIGAR staging ring (32 runs + 1 dword), insertion-run lengths of a window, Eq table (+ the "no match" word)
    const size_t slots = 64 / p.lanes_per_pair;
    // per slot: CIGAR staging ring (16 dwords) + 1 scratch dword + R rows (+1 dword against bank
    // conflicts); 8 dwords of padding at the end (the traceback's speculative lanes read a little past a
    // row).  A row is 32 DENT dwords, or 64 whole entries when W-O > 31 (the kernel's WIDE variant);
    // see stored_row_dwords() for W > 64.
    return (slots * (17 + (size_t)scrg::slot_stride_dwords(p.W, p.W - p.O, p.lanes_per_pair, p.lds_rows)) + 8) * sizeof(uint32_t);
}

scrg_status scrg_params_resolve(const scrg_params* in, scrg_params* out)
{
    if (!out) return SCRG_ERR_INVALID_ARG;
    return resolve_params(in, out) ? SCRG_OK : SCRG_ERR_INVALID_ARG;
}

scrg_status scrg_query_launch(scrg_ctx* c, const scrg_params* params, int32_t* n_waves, int32_t* pairs_per_wave,
                              int32_t* lds_bytes, int32_t* n_cus)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    scrg_params p;
    if (!resolve_params(params, &p)) return c->fail(SCRG_ERR_INVALID_ARG, "bad scrg_params");
    size_t lds = lds_bytes_for(p);
    int wpc = p.waves_per_cu;
    const size_t lds_cap = 160 * 1024;
    if (lds * wpc > lds_cap) wpc = (int)std::max<size_t>(1, lds_cap / lds);
    if (n_waves) *n_waves = c->n_cus * wpc;
    if (pairs_per_wave) *pairs_per_wave = 64 / p.lanes_per_pair;
    if (lds_bytes) *lds_bytes = (int32_t)lds;
    if (n_cus) *n_cus = c->n_cus;
    return SCRG_OK;
}

// ---------------------------------------------------------------------------
// device-pointer entry points
// ---------------------------------------------------------------------------
scrg_status scrg_pack_planar(scrg_ctx* c, const char* d_ascii, uint64_t n_words, uint64_t* d_planar,
                             uint32_t* d_bad_count)
{
    if (!c || (n_words && (!d_ascii || !d_planar)) || !d_bad_count) return SCRG_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, scrg::launch_pack_planar(d_ascii, n_words, d_planar, d_bad_count, c->n_cus, c->stream));
    return SCRG_OK;
}

scrg_status scrg_pack_planar_groups(scrg_ctx* c, const char* d_ascii, uint64_t n_rows, uint64_t words_per_row,
                                    uint64_t* d_planar, uint32_t* d_bad_count)
{
    if (!c || (n_rows && words_per_row && (!d_ascii || !d_planar)) || !d_bad_count) return SCRG_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, scrg::launch_pack_planar_groups(d_ascii, n_rows, words_per_row, d_planar, d_bad_count, c->n_cus, c->stream));
    return SCRG_OK;
}

static scrg_status align_device_impl(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const uint64_t* d_seq,
                                     const scrg_pair_desc* d_pairs, scrg_run* d_runs, int64_t* d_edit_distance,
                                     uint32_t* d_n_runs, uint32_t* d_pair_status, bool edits, uint32_t* d_run_count = nullptr)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    scrg_params p;
    if (!resolve_params(params, &p)) return c->fail(SCRG_ERR_INVALID_ARG, "bad scrg_params");
    if (edits && p.lanes_per_pair != 1)
        return c->fail(SCRG_ERR_INVALID_ARG, "edit-stream output needs lanes_per_pair = 1, the default "
                                             "(the GenASM-row mappings: scrg_align_device + scrg_encode_edit_stream)");
    if (n_pairs > kMaxPairsPerLaunch) return c->fail(SCRG_ERR_INVALID_ARG, "too many pairs for one launch");
    if (n_pairs && (!d_seq || !d_pairs || !d_runs || !d_edit_distance || !d_n_runs || !d_pair_status))
        return c->fail(SCRG_ERR_INVALID_ARG, "null device pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    c->have_timing = false;
    if (n_pairs == 0) return SCRG_OK;

    int32_t n_waves = 0, ppw = 0, lds = 0;
    scrg_status s = scrg_query_launch(c, &p, &n_waves, &ppw, &lds, nullptr);
    if (s != SCRG_OK) return s;
    // no point in launching more slots than pairs
    const uint64_t need_waves = (n_pairs + ppw - 1) / ppw;
    if ((uint64_t)n_waves > need_waves) n_waves = (int32_t)need_waves;

    HIP_TRY(c, c->counter.ensure(sizeof(uint32_t)));
    const size_t spill_rows = p.W > 64 ? (size_t)p.W + 1 : scrg::SPILL_ROWS;
    const size_t spill_row_dw = scrg::stored_row_dwords(p.W, p.W - p.O);
    const bool lane_mw = p.lanes_per_pair == 1 && (p.W > 64 || p.W - p.O > 31);      // rows of more than 32 bits: genasm_lane_mw_kernel
    if (lane_mw)                         // its window tables: one slab of HBM per wavefront
        HIP_TRY(c, c->spill.ensure((size_t)n_waves * scrg::lane_mw_table_bytes(p.W - p.O)));
    else if (p.lanes_per_pair != 1)      // (genasm_lane_kernel keeps its table in registers: nothing spills)
        HIP_TRY(c, c->spill.ensure((size_t)n_waves * ppw * spill_rows * spill_row_dw * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(c->counter.p, 0, sizeof(uint32_t), c->stream));

    scrg::AlignArgs a;
    a.seq = d_seq;
    a.pairs = d_pairs;
    a.runs = reinterpret_cast<uint16_t*>(d_runs);
    a.ed = d_edit_distance;
    a.n_runs = d_n_runs;
    a.status = d_pair_status;
    a.run_count = d_run_count;
    a.counter = c->counter.as<uint32_t>();
    a.spill = c->spill.as<uint32_t>();
    a.n_pairs = (uint32_t)n_pairs;
    a.W = p.W;
    a.tb_limit = p.W - p.O;
    a.lds_rows = p.lds_rows;
    a.text_stride = (uint32_t)p.text_stride_words;
    a.read_stride = (uint32_t)p.read_stride_words;
    a.debug = p.reserved[0];
    a.stats = nullptr;
    if (params && params->reserved[1]) {
        HIP_TRY(c, c->stats.ensure(12 * sizeof(uint64_t)));
        HIP_TRY(c, hipMemsetAsync(c->stats.p, 0, 12 * sizeof(uint64_t), c->stream));
        a.stats = c->stats.as<uint64_t>();
    }

    HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    if (lane_mw)
        HIP_TRY(c, scrg::launch_align_lane_mw(a, n_waves, (size_t)lds, c->stream, edits));
    else if (p.W > 64)
        HIP_TRY(c, scrg::launch_align_multiword(p.lanes_per_pair, a, n_waves, (size_t)lds, c->stream));
    else if (p.lanes_per_pair == 1)
        HIP_TRY(c, scrg::launch_align_lane(a, n_waves, (size_t)lds, c->stream, edits));
    else
        HIP_TRY(c, scrg::launch_align(p.lanes_per_pair, a, n_waves, (size_t)lds, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->have_timing = true;
    return SCRG_OK;
}

scrg_status scrg_align_device(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const uint64_t* d_seq,
                              const scrg_pair_desc* d_pairs, scrg_run* d_runs, int64_t* d_edit_distance,
                              uint32_t* d_n_runs, uint32_t* d_pair_status)
{
    return align_device_impl(c, params, n_pairs, d_seq, d_pairs, d_runs, d_edit_distance, d_n_runs, d_pair_status, false);
}

scrg_status scrg_align_device_edits(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const uint64_t* d_seq,
                                    const scrg_pair_desc* d_pairs, uint8_t* d_streams, int64_t* d_edit_distance,
                                    uint32_t* d_stream_len, uint32_t* d_pair_status, uint32_t* d_n_runs)
{
    if (reinterpret_cast<uintptr_t>(d_streams) & 31u) return c ? c->fail(SCRG_ERR_INVALID_ARG, "d_streams needs 32-byte alignment") : SCRG_ERR_INVALID_ARG;
    return align_device_impl(c, params, n_pairs, d_seq, d_pairs, reinterpret_cast<scrg_run*>(d_streams), d_edit_distance,
                             d_stream_len, d_pair_status, true, d_n_runs);
}

scrg_status scrg_last_kernel_ms(scrg_ctx* c, float* ms)
{
    if (!c || !ms) return SCRG_ERR_INVALID_ARG;
    *ms = 0.f;
    if (!c->have_timing) return SCRG_OK;
    HIP_TRY(c, hipEventSynchronize(c->ev_stop));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return SCRG_OK;
}

scrg_status scrg_debug_stats(scrg_ctx* c, uint64_t out[12])
{
    if (!c || !out) return SCRG_ERR_INVALID_ARG;
    memset(out, 0, 12 * sizeof(uint64_t));
    if (!c->stats.p) return SCRG_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, c->stats.p, 12 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SCRG_OK;
}

scrg_status scrg_compact_runs(scrg_ctx* c, uint64_t n_pairs, const scrg_pair_desc* d_pairs, const scrg_run* d_runs,
                              const uint32_t* d_n_runs, const uint64_t* d_dense_offset, scrg_run* d_dense)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    if (n_pairs && (!d_pairs || !d_runs || !d_n_runs || !d_dense_offset || !d_dense))
        return c->fail(SCRG_ERR_INVALID_ARG, "null device pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, scrg::launch_compact_runs(n_pairs, d_pairs, reinterpret_cast<const uint16_t*>(d_runs), d_n_runs,
                                         d_dense_offset, reinterpret_cast<uint16_t*>(d_dense), c->n_cus, c->stream));
    return SCRG_OK;
}

scrg_status scrg_compact_runs_packed(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const scrg_pair_desc* d_pairs,
                                     const scrg_run* d_runs, const uint32_t* d_n_runs, const uint64_t* d_dense_offset,
                                     uint8_t* d_packed)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    scrg_params p;
    if (!resolve_params(params, &p)) return c->fail(SCRG_ERR_INVALID_ARG, "bad scrg_params");
    if (p.W - p.O > 63) return c->fail(SCRG_ERR_INVALID_ARG, "packed runs hold counts up to 63: W-O must be <= 63");
    if (n_pairs && (!d_pairs || !d_runs || !d_n_runs || !d_dense_offset || !d_packed))
        return c->fail(SCRG_ERR_INVALID_ARG, "null device pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, scrg::launch_compact_runs_packed(n_pairs, d_pairs, reinterpret_cast<const uint16_t*>(d_runs), d_n_runs,
                                                d_dense_offset, d_packed, c->n_cus, c->stream));
    return SCRG_OK;
}

scrg_status scrg_unpack_runs(scrg_ctx* c, uint64_t n_runs, const uint8_t* d_packed, scrg_run* d_runs)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    if (n_runs && (!d_packed || !d_runs)) return c->fail(SCRG_ERR_INVALID_ARG, "null device pointer");
    if ((reinterpret_cast<uintptr_t>(d_packed) & 3u) || (reinterpret_cast<uintptr_t>(d_runs) & 7u))
        return c->fail(SCRG_ERR_INVALID_ARG, "packed runs need 4-byte, runs 8-byte alignment");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, scrg::launch_unpack_runs(n_runs, d_packed, reinterpret_cast<uint16_t*>(d_runs), c->n_cus, c->stream));
    return SCRG_OK;
}

scrg_status scrg_encode_edit_stream(scrg_ctx* c, uint64_t n_pairs, const scrg_pair_desc* d_pairs, const scrg_run* d_runs,
                                    const uint32_t* d_n_runs, uint8_t* d_stream, uint64_t stream_cap, uint64_t* d_stream_off,
                                    uint32_t* d_stream_len, uint64_t* d_total)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    if (!d_total) return c->fail(SCRG_ERR_INVALID_ARG, "d_total is required");
    if (n_pairs && (!d_pairs || !d_runs || !d_n_runs || !d_stream_off || !d_stream_len || (stream_cap && !d_stream)))
        return c->fail(SCRG_ERR_INVALID_ARG, "null device pointer");
    if (reinterpret_cast<uintptr_t>(d_stream) & 3u) return c->fail(SCRG_ERR_INVALID_ARG, "d_stream needs 4-byte alignment");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, scrg::launch_encode_edits(n_pairs, d_pairs, reinterpret_cast<const uint16_t*>(d_runs), d_n_runs, d_stream,
                                         stream_cap, d_stream_off, d_stream_len, d_total, c->stream));
    return SCRG_OK;
}

scrg_status scrg_decode_edit_stream(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const uint8_t* d_stream,
                                    uint64_t stream_bytes, const uint64_t* d_stream_off, const uint32_t* d_stream_len,
                                    const uint64_t* d_read_len, uint64_t read_len_stride, const uint64_t* d_dense_offset,
                                    scrg_run* d_dense, uint32_t* d_n_runs, uint32_t* d_bad_count)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    scrg_params p;
    if (!resolve_params(params, &p)) return c->fail(SCRG_ERR_INVALID_ARG, "bad scrg_params");
    if (!d_bad_count || (n_pairs && (!d_stream_off || !d_stream_len || !d_read_len || !d_n_runs)) || (stream_bytes && !d_stream))
        return c->fail(SCRG_ERR_INVALID_ARG, "null device pointer");
    if (d_dense && !d_dense_offset) return c->fail(SCRG_ERR_INVALID_ARG, "d_dense needs d_dense_offset");
    // streams are fetched in aligned 16-byte blocks, runs leave in aligned 16-byte stores
    if ((reinterpret_cast<uintptr_t>(d_stream) & 15u) || (reinterpret_cast<uintptr_t>(d_dense) & 15u))
        return c->fail(SCRG_ERR_INVALID_ARG, "d_stream and d_dense need 16-byte alignment");
    HIP_TRY(c, hipSetDevice(c->device));
    // batches that fill the GPU are decoded longest stream first (a wavefront's 64 pairs then finish together)
    void* ws = nullptr;
    size_t temp_bytes = 0;
    if (n_pairs >= 4096 && n_pairs < 0x7fffffffull) {
        temp_bytes = scrg::decode_sort_temp_bytes(n_pairs);
        HIP_TRY(c, c->sort_ws.ensure(3 * n_pairs * sizeof(uint32_t) + 256 + temp_bytes));
        ws = c->sort_ws.p;
    }
    HIP_TRY(c, scrg::launch_decode_edits(n_pairs, (uint32_t)p.W, (uint32_t)p.O, d_stream, stream_bytes, d_stream_off, d_stream_len,
                                         d_read_len, read_len_stride, d_dense_offset, reinterpret_cast<uint16_t*>(d_dense),
                                         d_n_runs, d_bad_count, ws, temp_bytes, c->stream));
    return SCRG_OK;
}

// The device decoder's per-lane state machine (edit_stream.h: decode_lane_step, the code decode_edits_kernel runs in
// every lane) on the host, for ONE pair: same arguments and results as scrg_edit_stream_to_runs.  Exists so that the
// state machine can be held against the plain replay without a GPU (tests/test_edit_stream.py).
scrg_status scrg_edit_stream_to_runs_lane(const scrg_params* params, uint64_t read_len, const uint8_t* stream, uint64_t n_bytes,
                                          scrg_run* runs, uint64_t runs_cap, uint64_t* n_runs)
{
    scrg_params p;
    if (!n_runs || (n_bytes && !stream) || (runs_cap && !runs) || !resolve_params(params, &p)) return SCRG_ERR_INVALID_ARG;
    *n_runs = 0;
    if (read_len > 0x7fffffffull || n_bytes > 0x7fffffffull) return SCRG_ERR_INVALID_ARG;
    scrg::DecodeLane s;
    scrg::decode_lane_init(s, (uint32_t)p.W, (uint32_t)p.O, 0u, (uint32_t)n_bytes, (uint32_t)read_len);
    while (s.aliveM)
        (void)scrg::decode_lane_step(s, s.pos < n_bytes ? (uint32_t)stream[s.pos] : 0u, [&](uint32_t k, uint32_t word) {
            if (k < runs_cap) { runs[k].count = (uint8_t)word; runs[k].op = (char)(word >> 8); }
        });
    if (!scrg::decode_lane_clean(s)) return SCRG_ERR_INVALID_ARG;
    *n_runs = s.n;
    return s.n > runs_cap ? SCRG_ERR_CIGAR_OVERFLOW : SCRG_OK;
}

scrg_status scrg_edit_stream_to_runs(const scrg_params* params, uint64_t read_len, const uint8_t* stream, uint64_t n_bytes,
                                     scrg_run* runs, uint64_t runs_cap, uint64_t* n_runs)
{
    scrg_params p;
    if (!n_runs || (n_bytes && !stream) || (runs_cap && !runs) || !resolve_params(params, &p)) return SCRG_ERR_INVALID_ARG;
    uint64_t k = 0;
    const uint64_t n = scrg::replay_edit_stream(stream, n_bytes, read_len, (uint32_t)p.W, (uint32_t)p.O,
                                                [&](uint32_t op, uint64_t t) {
                                                    if (k < runs_cap) { runs[k].count = (uint8_t)t; runs[k].op = (char)op; }
                                                    k++;
                                                });
    *n_runs = n == ~0ull ? 0 : n;
    if (n == ~0ull) return SCRG_ERR_INVALID_ARG;
    return n > runs_cap ? SCRG_ERR_CIGAR_OVERFLOW : SCRG_OK;
}

scrg_status scrg_runs_to_edit_stream(const scrg_run* runs, uint64_t n_runs, uint8_t* stream, uint64_t stream_cap,
                                     uint64_t* n_bytes)
{
    if (!n_bytes || (n_runs && !runs) || (stream_cap && !stream)) return SCRG_ERR_INVALID_ARG;
    uint64_t k = 0, pend = 0;
    auto put = [&](uint32_t b) {
        if (k < stream_cap) stream[k] = (uint8_t)b;
        k++;
    };
    for (uint64_t r = 0; r < n_runs; r++) {
        const uint32_t op = (uint8_t)runs[r].op, cnt = runs[r].count;
        if (op != '=' && op != 'X' && op != 'I' && op != 'D') return SCRG_ERR_INVALID_ARG;
        if (op == '=' || cnt == 0) { pend += cnt; continue; }
        const uint32_t code = scrg::edit_code_of_char(op) << 6;
        for (uint64_t q = pend >> 6; q; q--) put(0x3F);
        put(code | (uint32_t)(pend & 63u));
        for (uint32_t q = 1; q < cnt; q++) put(code);
        pend = 0;
    }
    *n_bytes = k;
    return k > stream_cap ? SCRG_ERR_CIGAR_OVERFLOW : SCRG_OK;
}

scrg_status scrg_ascii_to_twobit(scrg_ctx* c, uint64_t count, const uint64_t* d_lens, const uint64_t* d_ascii_off,
                                 const char* d_ascii, const uint64_t* d_twobit_off, uint8_t* d_twobit,
                                 uint32_t* d_bad_count)
{
    if (!c || !d_bad_count) return SCRG_ERR_INVALID_ARG;
    if (count && (!d_lens || !d_ascii_off || !d_ascii || !d_twobit_off || !d_twobit))
        return c->fail(SCRG_ERR_INVALID_ARG, "null device pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    // the grid's x extent only needs an upper bound on the string length; take it from the
    // device array without a sync by assuming long strings (extra blocks exit immediately)
    HIP_TRY(c, scrg::launch_ascii_to_twobit(count, d_lens, d_ascii_off, d_ascii, d_twobit_off, d_twobit, d_bad_count,
                                            1 << 16, c->stream));
    return SCRG_OK;
}

// ---------------------------------------------------------------------------
// host-pointer entry points
// ---------------------------------------------------------------------------
void scrg_result_pool_trim(void) { g_pool.trim(); }

void scrg_result_free(scrg_result* r)
{
    if (!r) return;
    g_pool.put(r->edit_distance);
    g_pool.put(r->pair_status);
    g_pool.put(r->run_offset);
    g_pool.put(r->runs);
    g_pool.put(r->cigar_offset);
    g_pool.put(r->cigar_text);
    free(r);
}

namespace {

struct SeqRef {
    const char* p;
    uint64_t len;
    uint64_t word_off;   // first planar word of this sequence
    bool revcomp = false;   // stage the reverse complement (reverse-strand candidates)
};

inline char complement_base(char c)
{
    switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    default: return c;   // left as is: the pack kernel reports it as a bad base
    }
}

struct Problem {         // one (text, read) problem in caller order
    uint64_t text_off, text_len, read_off, read_len;   // base offsets into the planar array
};

// Shared tail of both host entry points: sequences are described by `seqs`
// (each packed once, 32-base aligned), problems by `probs`.
// `resident_words` leading words of d_seq are already packed on the device (a genome kept by scrg_genome_set):
// they are neither staged nor transferred nor packed again; `seqs` then describes the words after them only.
scrg_status run_batch(scrg_ctx* c, const scrg_params& p, std::vector<SeqRef>& seqs, uint64_t total_words,
                      const std::vector<Problem>& probs, scrg_result** out, uint64_t resident_words = 0, bool resident = false)
{
    const int64_t t_begin = now_ns();
    const uint64_t n = probs.size();
    HIP_TRY(c, hipSetDevice(c->device));
    const bool host_timing = getenv("SCRG_HOST_TIMING") != nullptr;
    int64_t t_mark = t_begin;
    auto mark = [&](const char* what) {
        if (!host_timing) return;
        const int64_t t = now_ns();
        fprintf(stderr, "[scrooge_amd host] %-28s %8.3f ms\n", what, (double)(t - t_mark) / 1e6);
        t_mark = t;
    };

    scrg_result* r = static_cast<scrg_result*>(calloc(1, sizeof(scrg_result)));
    if (!r) return c->fail(SCRG_ERR_OOM, "result header");
    r->n_pairs = n;
    auto bail = [&](scrg_status s) {
        scrg_result_free(r);
        return s;
    };
    // (every element of these four is written below; only the terminating entries need the zero)
    r->edit_distance = static_cast<int64_t*>(g_pool.get((n + 1) * sizeof(int64_t), n < 4096));
    r->pair_status = static_cast<uint32_t*>(g_pool.get((n + 1) * sizeof(uint32_t), n < 4096));
    r->run_offset = static_cast<uint64_t*>(g_pool.get((n + 1) * sizeof(uint64_t), n < 4096));
    r->cigar_offset = static_cast<uint64_t*>(g_pool.get((n + 1) * sizeof(uint64_t), n < 4096));
    if (r->edit_distance) r->edit_distance[n] = 0;
    if (r->pair_status) r->pair_status[n] = 0;
    if (r->run_offset) r->run_offset[0] = r->run_offset[n] = 0;
    if (r->cigar_offset) r->cigar_offset[0] = r->cigar_offset[n] = 0;
    if (!r->edit_distance || !r->pair_status || !r->run_offset || !r->cigar_offset)
        return bail(c->fail(SCRG_ERR_OOM, "result arrays"));

    // ---- stage ASCII (32 bytes per planar word, zero padded), H2D, pack ----
    const uint64_t seq_words = total_words + SCRG_SEQ_PAD_WORDS;
    const uint64_t new_words = total_words - resident_words;       // words to stage, transfer and pack in this call
    const size_t ascii_bytes = (size_t)new_words * 32;
    if (ascii_bytes) {
        hipError_t e = c->h_ascii.ensure(ascii_bytes);
        if (e != hipSuccess) return bail(c->fail(SCRG_ERR_OOM, "pinned staging buffer", e));
        char* h = static_cast<char*>(c->h_ascii.p);
        // sequences of up to 4 MB are copied whole, one per index; longer ones (a chromosome) are cut into pieces of
        // 4 MB so that all threads share them
        struct Piece { uint64_t seq, from, to; };
        const uint64_t PIECE = 4u << 20;
        auto copy_piece = [&](const Piece& pc) {
            const SeqRef& q = seqs[pc.seq];
            char* dst = h + (q.word_off - resident_words) * 32;
            const uint64_t data_to = std::min(pc.to, q.len);
            if (pc.from < data_to) {
                if (!q.revcomp) memcpy(dst + pc.from, q.p + pc.from, data_to - pc.from);
                else
                    for (uint64_t k = pc.from; k < data_to; k++) dst[k] = complement_base(q.p[q.len - 1 - k]);
            }
            if (pc.to > std::max(pc.from, q.len)) memset(dst + std::max(pc.from, q.len), 0, pc.to - std::max(pc.from, q.len));
        };
        std::vector<Piece> pieces;                      // of the long sequences only
        for (uint64_t s = 0; s < seqs.size(); s++) {
            const uint64_t span = ((seqs[s].len + 31) / 32) * 32;
            if (span > PIECE)
                for (uint64_t a = 0; a < span; a += PIECE) pieces.push_back({s, a, std::min(span, a + PIECE)});
        }
        parallel_for(pieces.size(), [&](uint64_t i) { copy_piece(pieces[i]); }, true);
        parallel_for(seqs.size(), [&](uint64_t s) {
            const uint64_t span = ((seqs[s].len + 31) / 32) * 32;
            if (span <= PIECE && span) copy_piece(Piece{s, 0, span});
        });
    }
    hipError_t e;
    if (resident_words && (size_t)seq_words * 8 > c->d_seq.cap) {
        // the sequence array has to grow: keep the resident genome (device-to-device) instead of packing it again
        DevBuf bigger;
        if ((e = bigger.ensure((size_t)seq_words * 8)) != hipSuccess) return bail(c->fail(SCRG_ERR_OOM, "device sequence buffer", e));
        if ((e = hipMemcpyAsync(bigger.p, c->d_seq.p, resident_words * 8, hipMemcpyDeviceToDevice, c->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
            bigger.release();
            return bail(c->fail(SCRG_ERR_HIP, "moving the resident genome", e));
        }
        c->d_seq.release();
        c->d_seq = bigger;
    }
    if (!resident) c->genome_resident = false;     // d_seq is about to be overwritten from word 0
    if ((e = c->d_ascii.ensure(ascii_bytes + 32)) != hipSuccess || (e = c->d_seq.ensure(seq_words * 8)) != hipSuccess ||
        (e = c->d_bad.ensure(4)) != hipSuccess)
        return bail(c->fail(SCRG_ERR_OOM, "device sequence buffers", e));
    mark("result arrays + staging copy");
    const int64_t t_pack0 = now_ns();
    if ((e = hipMemsetAsync(c->d_bad.p, 0, 4, c->stream)) != hipSuccess ||
        (e = hipMemsetAsync(c->d_seq.as<uint64_t>() + total_words, 0, SCRG_SEQ_PAD_WORDS * 8, c->stream)) != hipSuccess)
        return bail(c->fail(SCRG_ERR_HIP, "memset", e));
    if (ascii_bytes) {
        if ((e = hipMemcpyAsync(c->d_ascii.p, c->h_ascii.p, ascii_bytes, hipMemcpyHostToDevice, c->stream)) != hipSuccess)
            return bail(c->fail(SCRG_ERR_HIP, "H2D ascii", e));
        scrg_status s = scrg_pack_planar(c, c->d_ascii.as<char>(), new_words, c->d_seq.as<uint64_t>() + resident_words,
                                         c->d_bad.as<uint32_t>());
        if (s != SCRG_OK) return bail(s);
    }
    uint32_t bad = 0;
    if ((e = hipMemcpyAsync(&bad, c->d_bad.p, 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(c->stream)) != hipSuccess)
        return bail(c->fail(SCRG_ERR_HIP, "pack", e));
    r->pack_ns = now_ns() - t_pack0;
    mark("H2D + pack kernel");
    if (bad) return bail(c->fail(SCRG_ERR_BAD_BASE, "input contains characters other than ACGTacgt"));

    // ---- problem descriptors, longest read first (src/tests.cu:375-377) ----
    std::vector<uint32_t> order(n);
    std::atomic<int> unsorted{0};
    {
        // identity, and at the same time: is the batch already in issue order?  (read sets of one length, or sorted
        // ones, need no sort: 4 M comparisons instead of 90 M)
        const uint64_t BLK = 1u << 16, nb = (n + BLK - 1) / BLK;
        parallel_for(nb, [&](uint64_t b) {
            bool ok = true;
            for (uint64_t k = b * BLK; k < std::min(n, (b + 1) * BLK); k++) {
                order[k] = (uint32_t)k;
                if (k && probs[k - 1].read_len < probs[k].read_len) ok = false;
            }
            if (!ok) unsorted.store(1, std::memory_order_relaxed);
        }, true);
    }
    if (p.sort_by_length) {
        const bool sorted = unsorted.load() == 0;
        if (!sorted)
            std::stable_sort(order.begin(), order.end(),
                             [&](uint32_t x, uint32_t y) { return probs[x].read_len > probs[y].read_len; });
    }
    mark("  order");
    if (hipError_t eh = c->h_desc.ensure(std::max<uint64_t>(n, 1) * sizeof(scrg_pair_desc)); eh != hipSuccess)
        return bail(c->fail(SCRG_ERR_OOM, "pinned descriptor buffer", eh));
    scrg_pair_desc* const desc = static_cast<scrg_pair_desc*>(c->h_desc.p);
    // slices: same bound as the reference's GPU list sizing (2*read_len, genasm_gpu.cu:906-911), whole 32-byte
    // pieces; offsets by a two-level prefix sum (blocks of 64 k pairs in parallel)
    uint64_t arena = 0;
    {
        const uint64_t BLK = 1u << 16, nb = (n + BLK - 1) / BLK;
        std::vector<uint64_t> block_sum(nb + 1, 0);
        parallel_for(nb, [&](uint64_t b) {
            uint64_t acc = 0;
            for (uint64_t k = b * BLK; k < std::min(n, (b + 1) * BLK); k++) {
                const Problem& q = probs[order[k]];
                scrg_pair_desc& d = desc[k];
                d.text_off = q.text_off;
                d.text_len = q.text_len;
                d.read_off = q.read_off;
                d.read_len = q.read_len;
                d.cigar_cap = (2 * q.read_len + 8 + 15) & ~(uint64_t)15;
                d.cigar_off = acc;
                acc += d.cigar_cap;
            }
            block_sum[b + 1] = acc;
        }, true);
        for (uint64_t b = 0; b < nb; b++) block_sum[b + 1] += block_sum[b];
        arena = block_sum[nb];
        parallel_for(nb, [&](uint64_t b) {
            if (block_sum[b])
                for (uint64_t k = b * BLK; k < std::min(n, (b + 1) * BLK); k++) desc[k].cigar_off += block_sum[b];
        }, true);
    }
    mark("  build descriptors");
    if (n == 0) {
        r->runs = static_cast<scrg_run*>(g_pool.get(sizeof(scrg_run), true));
        r->cigar_text = static_cast<char*>(g_pool.get(1, true));
        r->total_ns = now_ns() - t_begin;
        *out = r;
        return SCRG_OK;
    }
    if ((e = c->d_pairs.ensure(n * sizeof(scrg_pair_desc))) != hipSuccess ||
        (e = c->d_runs.ensure(arena * sizeof(scrg_run))) != hipSuccess || (e = c->d_ed.ensure(n * 8)) != hipSuccess ||
        (e = c->d_nruns.ensure(n * 4)) != hipSuccess || (e = c->d_status.ensure(n * 4)) != hipSuccess ||
        (e = c->d_dense_off.ensure(n * 8)) != hipSuccess)
        return bail(c->fail(SCRG_ERR_OOM, "device result buffers", e));
    if ((e = hipMemcpyAsync(c->d_pairs.p, desc, n * sizeof(scrg_pair_desc), hipMemcpyHostToDevice, c->stream)) !=
        hipSuccess)
        return bail(c->fail(SCRG_ERR_HIP, "H2D descriptors", e));

    mark("descriptors (sort, build, H2D)");
    // ---- the timed region of the reference: kernel + sync (genasm_gpu.cu:939-944) ----
    scrg_params pd = p;                   // this path packs contiguously, whatever the caller's device-layout strides say
    pd.text_stride_words = pd.read_stride_words = 1;
    scrg_status s = scrg_align_device(c, &pd, n, c->d_seq.as<uint64_t>(), c->d_pairs.as<scrg_pair_desc>(),
                                      c->d_runs.as<scrg_run>(), c->d_ed.as<int64_t>(), c->d_nruns.as<uint32_t>(),
                                      c->d_status.as<uint32_t>());
    if (s != SCRG_OK) return bail(s);
    float ms = 0.f;
    s = scrg_last_kernel_ms(c, &ms);
    if (s != SCRG_OK) return bail(s);
    r->kernel_ns = (int64_t)((double)ms * 1e6);
    if (g_log.load() && ms > 0.f)   // the reference's log line, genasm_gpu.cu:949-951
        fprintf(stderr, "core algorithm ran at %lld aligns/second\n", (long long)((double)n * 1000.0 / ms));

    mark("align kernel");
    // ---- read back: per-pair scalars, then the compacted runs ----
    // (pinned landing zone: a D2H into pageable memory runs at a fraction of the link rate)
    if ((e = c->h_out.ensure(n * 16)) != hipSuccess) return bail(c->fail(SCRG_ERR_OOM, "pinned result buffer", e));
    int64_t* const ed = static_cast<int64_t*>(c->h_out.p);
    uint32_t* const nr = reinterpret_cast<uint32_t*>(ed + n);
    uint32_t* const st = nr + n;
    if ((e = hipMemcpyAsync(ed, c->d_ed.p, n * 8, hipMemcpyDeviceToHost, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(nr, c->d_nruns.p, n * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(st, c->d_status.p, n * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(c->stream)) != hipSuccess)
        return bail(c->fail(SCRG_ERR_HIP, "D2H scalars", e));
    mark("  D2H scalars");

    // dense layout in caller order
    if ((e = c->h_off.ensure(n * 8)) != hipSuccess) return bail(c->fail(SCRG_ERR_OOM, "pinned offset buffer", e));
    uint64_t* const dense_off_sorted = static_cast<uint64_t*>(c->h_off.p);
    {
        // run_offset[i] = exclusive prefix sum of the (capped) run counts in caller order: counts are scattered into
        // run_offset itself, then a two-level scan (blocks of 64 k in parallel)
        parallel_for(n, [&](uint64_t k) {
            r->run_offset[order[k]] = std::min<uint64_t>(nr[k], desc[k].cigar_cap);
        });
        const uint64_t BLK = 1u << 16, nb = (n + BLK - 1) / BLK;
        std::vector<uint64_t> block_sum(nb + 1, 0);
        parallel_for(nb, [&](uint64_t b) {
            uint64_t acc = 0;
            for (uint64_t i = b * BLK; i < std::min(n, (b + 1) * BLK); i++) {
                const uint64_t cnt = r->run_offset[i];
                r->run_offset[i] = acc;
                acc += cnt;
            }
            block_sum[b + 1] = acc;
        }, true);
        for (uint64_t b = 0; b < nb; b++) block_sum[b + 1] += block_sum[b];
        parallel_for(nb, [&](uint64_t b) {
            if (block_sum[b])
                for (uint64_t i = b * BLK; i < std::min(n, (b + 1) * BLK); i++) r->run_offset[i] += block_sum[b];
        }, true);
        r->run_offset[n] = block_sum[nb];
        parallel_for(n, [&](uint64_t k) { dense_off_sorted[k] = r->run_offset[order[k]]; });
    }
    const uint64_t total_runs = r->run_offset[n];
    mark("  dense offsets");
    r->runs = static_cast<scrg_run*>(g_pool.get((total_runs + 1) * sizeof(scrg_run), false));
    if (!r->runs) return bail(c->fail(SCRG_ERR_OOM, "runs"));
    if (total_runs) {
        if ((e = c->d_dense.ensure(total_runs * sizeof(scrg_run))) != hipSuccess)
            return bail(c->fail(SCRG_ERR_OOM, "dense runs", e));
        if ((e = hipMemcpyAsync(c->d_dense_off.p, dense_off_sorted, n * 8, hipMemcpyHostToDevice, c->stream)) !=
            hipSuccess)
            return bail(c->fail(SCRG_ERR_HIP, "H2D offsets", e));
        s = scrg_compact_runs(c, n, c->d_pairs.as<scrg_pair_desc>(), c->d_runs.as<scrg_run>(), c->d_nruns.as<uint32_t>(),
                              c->d_dense_off.as<uint64_t>(), c->d_dense.as<scrg_run>());
        if (s != SCRG_OK) return bail(s);
        if ((e = c->h_runs.ensure(total_runs * sizeof(scrg_run))) != hipSuccess)
            return bail(c->fail(SCRG_ERR_OOM, "pinned run buffer", e));
        if ((e = hipMemcpyAsync(c->h_runs.p, c->d_dense.p, total_runs * sizeof(scrg_run), hipMemcpyDeviceToHost, c->stream)) !=
                hipSuccess ||
            (e = hipStreamSynchronize(c->stream)) != hipSuccess)
            return bail(c->fail(SCRG_ERR_HIP, "D2H runs", e));
        mark("  compaction + D2H runs");
        const uint64_t bytes = total_runs * sizeof(scrg_run), CH = 1u << 20;
        parallel_for((bytes + CH - 1) / CH, [&](uint64_t i) {
            memcpy(reinterpret_cast<char*>(r->runs) + i * CH, static_cast<const char*>(c->h_runs.p) + i * CH,
                   std::min<uint64_t>(CH, bytes - i * CH));
        }, true);
    }

    mark("  copy runs out");
    std::atomic<int> any_overflow{0};
    parallel_for(n, [&](uint64_t k) {
        r->edit_distance[order[k]] = ed[k];
        r->pair_status[order[k]] = st[k] ? (uint32_t)SCRG_ERR_CIGAR_OVERFLOW : (uint32_t)SCRG_OK;
        if (st[k]) any_overflow.store(1, std::memory_order_relaxed);
    });
    const scrg_status worst = any_overflow.load() ? SCRG_ERR_CIGAR_OVERFLOW : SCRG_OK;

    // ---- "%d%c" text, as genasm_cpu.cpp:387-403 ----
    {
        // (decimal digits of a count: from a table — the low bytes of `word` are the digits, `len` of them)
        struct DigitLut {
            uint32_t word[256];
            uint8_t len[256];
            DigitLut()
            {
                for (unsigned v = 0; v < 256; v++) {
                    char d[4];
                    const int l = snprintf(d, sizeof d, "%u", v);
                    uint32_t w = 0;
                    for (int k = 0; k < l; k++) w |= (uint32_t)(uint8_t)d[k] << (8 * k);
                    word[v] = w;
                    len[v] = (uint8_t)l;
                }
            }
        };
        static const DigitLut dig;
        uint64_t acc = 0;
        {
            // length of every pair's text and its offset within its block in one pass over the runs, then the block sums
            const uint64_t BLK = 1u << 13, nb = (n + BLK - 1) / BLK;
            std::vector<uint64_t> block_sum(nb + 1, 0);
            parallel_for(nb, [&](uint64_t b) {
                uint64_t a = 0;
                for (uint64_t i = b * BLK; i < std::min(n, (b + 1) * BLK); i++) {
                    uint64_t chars = 0;
                    for (uint64_t k = r->run_offset[i]; k < r->run_offset[i + 1]; k++) chars += dig.len[r->runs[k].count] + 1u;
                    r->cigar_offset[i] = a;
                    a += chars + 1;
                }
                block_sum[b + 1] = a;
            }, true);
            for (uint64_t b = 0; b < nb; b++) block_sum[b + 1] += block_sum[b];
            parallel_for(nb, [&](uint64_t b) {
                if (block_sum[b])
                    for (uint64_t i = b * BLK; i < std::min(n, (b + 1) * BLK); i++) r->cigar_offset[i] += block_sum[b];
            }, true);
            acc = block_sum[nb];
        }
        r->cigar_offset[n] = acc;
        mark("  text sizes");
        r->cigar_text = static_cast<char*>(g_pool.get(acc + 1, false));
        if (!r->cigar_text) return bail(c->fail(SCRG_ERR_OOM, "cigar text"));
        parallel_for(n, [&](uint64_t i) {
            char* w = r->cigar_text + r->cigar_offset[i];
            const uint64_t k0 = r->run_offset[i], k1 = r->run_offset[i + 1];
            // every run but the last: one 4-byte store (digits + op; the one or two bytes too many land where the next
            // run of the same pair is written afterwards); the last run byte by byte — the next pair may be another thread's
            for (uint64_t k = k0; k + 1 < k1; k++) {
                const unsigned cnt = r->runs[k].count, l = dig.len[cnt];
                const uint32_t v = dig.word[cnt] | ((uint32_t)(uint8_t)r->runs[k].op << (8 * l));
                memcpy(w, &v, 4);
                w += l + 1;
            }
            if (k1 > k0) {
                const unsigned cnt = r->runs[k1 - 1].count, l = dig.len[cnt];
                for (unsigned q = 0; q < l; q++) *w++ = (char)(dig.word[cnt] >> (8 * q));
                *w++ = r->runs[k1 - 1].op;
            }
            *w = '\0';
        });
    }
    mark("CIGAR text");
    r->total_ns = now_ns() - t_begin;
    *out = r;
    if (worst != SCRG_OK) c->fail(worst, "at least one pair overflowed its CIGAR slice (see pair_status)");
    return worst;
}

}  // namespace

static scrg_status align_pairs_impl(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const char* const* texts,
                             const uint64_t* text_lens, const char* const* queries, const uint64_t* query_lens,
                             scrg_result** out)
{
    if (!c || !out) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_pairs && (!texts || !text_lens || !queries || !query_lens))
        return c->fail(SCRG_ERR_INVALID_ARG, "null input array");
    scrg_params p;
    if (!resolve_params(params, &p)) return c->fail(SCRG_ERR_INVALID_ARG, "bad scrg_params");
    if (n_pairs > kMaxPairsPerLaunch) return c->fail(SCRG_ERR_INVALID_ARG, "too many pairs");

    std::vector<SeqRef> seqs(2 * n_pairs);
    std::vector<Problem> probs(n_pairs);
    uint64_t w = 0;
    for (uint64_t i = 0; i < n_pairs; i++) {
        if ((text_lens[i] && !texts[i]) || (query_lens[i] && !queries[i]))
            return c->fail(SCRG_ERR_INVALID_ARG, "null sequence pointer");
        if (query_lens[i] > 0x7fffffffull) return c->fail(SCRG_ERR_INVALID_ARG, "read longer than 2^31-1");
        seqs[2 * i] = {texts[i], text_lens[i], w, false};
        probs[i].text_off = w * 32;
        probs[i].text_len = text_lens[i];
        w += (text_lens[i] + 31) / 32;
        seqs[2 * i + 1] = {queries[i], query_lens[i], w, false};
        probs[i].read_off = w * 32;
        probs[i].read_len = query_lens[i];
        w += (query_lens[i] + 31) / 32;
    }
    return run_batch(c, p, seqs, w, probs, out);
}

scrg_status scrg_align_pairs(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const char* const* texts,
                             const uint64_t* text_lens, const char* const* queries, const uint64_t* query_lens,
                             scrg_result** out)
{
    return guarded(c, [&] { return align_pairs_impl(c, params, n_pairs, texts, text_lens, queries, query_lens, out); });
}

static scrg_status align_mapping_impl(scrg_ctx* c, const scrg_params* params, const char* genome,
                                        uint64_t genome_len, uint64_t n_reads, const char* const* reads,
                                        const uint64_t* read_lens, const uint64_t* cand_offsets,
                                        const uint64_t* cand_start, const uint8_t* cand_reverse, scrg_result** out,
                                        bool resident = false)
{
    if (!c || !out) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    if (resident) {                          // the genome scrg_genome_set() left packed at the front of d_seq
        if (!c->genome_resident) return c->fail(SCRG_ERR_INVALID_ARG, "no resident genome: call scrg_genome_set first");
        genome = nullptr;
        genome_len = c->genome_len;
    }
    if ((!resident && genome_len && !genome) || (n_reads && (!reads || !read_lens)) || !cand_offsets)
        return c->fail(SCRG_ERR_INVALID_ARG, "null input array");
    scrg_params p;
    if (!resolve_params(params, &p)) return c->fail(SCRG_ERR_INVALID_ARG, "bad scrg_params");
    const uint64_t n_pairs = cand_offsets[n_reads];
    if (n_pairs > kMaxPairsPerLaunch) return c->fail(SCRG_ERR_INVALID_ARG, "too many pairs");
    if (n_pairs && !cand_start) return c->fail(SCRG_ERR_INVALID_ARG, "null candidate array");

    // genome and every read are packed exactly once (README.md:83 of the reference asks for this);
    // a read with reverse-strand candidates is additionally staged reverse-complemented, once
    std::vector<SeqRef> seqs;
    seqs.reserve(1 + n_reads);
    std::vector<Problem> probs(n_pairs);
    uint64_t w = 0;
    if (!resident) seqs.push_back({genome, genome_len, 0, false});
    w += (genome_len + 31) / 32;
    for (uint64_t r = 0; r < n_reads; r++) {
        if (read_lens[r] && !reads[r]) return c->fail(SCRG_ERR_INVALID_ARG, "null read pointer");
        if (read_lens[r] > 0x7fffffffull) return c->fail(SCRG_ERR_INVALID_ARG, "read longer than 2^31-1");
        if (cand_offsets[r + 1] < cand_offsets[r]) return c->fail(SCRG_ERR_INVALID_ARG, "cand_offsets not monotone");
        bool any_fwd = false, any_rev = false;
        for (uint64_t k = cand_offsets[r]; k < cand_offsets[r + 1]; k++) {
            if (cand_reverse && cand_reverse[k]) any_rev = true;
            else any_fwd = true;
        }
        const uint64_t words = (read_lens[r] + 31) / 32;
        uint64_t fwd_off = 0, rev_off = 0;
        if (any_fwd || !any_rev) {
            seqs.push_back({reads[r], read_lens[r], w, false});
            fwd_off = w * 32;
            w += words;
        }
        if (any_rev) {
            seqs.push_back({reads[r], read_lens[r], w, true});
            rev_off = w * 32;
            w += words;
        }
        for (uint64_t k = cand_offsets[r]; k < cand_offsets[r + 1]; k++) {
            if (cand_start[k] > genome_len) return c->fail(SCRG_ERR_INVALID_ARG, "candidate past end of genome");
            // text = genome suffix from start_in_reference (genasm_cpu.cpp:512-514)
            probs[k].text_off = cand_start[k];
            probs[k].text_len = genome_len - cand_start[k];
            probs[k].read_off = (cand_reverse && cand_reverse[k]) ? rev_off : fwd_off;
            probs[k].read_len = read_lens[r];
        }
    }
    return run_batch(c, p, seqs, w, probs, out, resident ? c->genome_words : 0, resident);
}

// scrg_genome_set: stage, transfer and pack a genome once; it stays at the front of the handle's sequence
// array until another genome is set, scrg_genome_clear() is called or a call that brings its own sequences
// (scrg_align_pairs, scrg_align_mapping) reuses the array.
static scrg_status genome_set_impl(scrg_ctx* c, const char* genome, uint64_t genome_len)
{
    if (!c || (genome_len && !genome)) return SCRG_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    c->genome_resident = false;
    const uint64_t words = (genome_len + 31) / 32;
    const size_t bytes = (size_t)words * 32;
    hipError_t e;
    if ((e = c->h_ascii.ensure(bytes + 32)) != hipSuccess) return c->fail(SCRG_ERR_OOM, "pinned staging buffer", e);
    char* const h = static_cast<char*>(c->h_ascii.p);
    const uint64_t PIECE = 4u << 20;
    parallel_for((bytes + PIECE - 1) / PIECE, [&](uint64_t i) {
        const uint64_t a = i * PIECE, b = std::min<uint64_t>(bytes, a + PIECE), d = std::min(b, genome_len);
        if (a < d) memcpy(h + a, genome + a, d - a);
        if (b > std::max(a, genome_len)) memset(h + std::max(a, genome_len), 0, b - std::max(a, genome_len));
    }, true);
    if ((e = c->d_ascii.ensure(bytes + 32)) != hipSuccess || (e = c->d_seq.ensure((words + SCRG_SEQ_PAD_WORDS) * 8)) != hipSuccess ||
        (e = c->d_bad.ensure(4)) != hipSuccess)
        return c->fail(SCRG_ERR_OOM, "device sequence buffers", e);
    HIP_TRY(c, hipMemsetAsync(c->d_bad.p, 0, 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_seq.as<uint64_t>() + words, 0, SCRG_SEQ_PAD_WORDS * 8, c->stream));
    if (bytes) {
        HIP_TRY(c, hipMemcpyAsync(c->d_ascii.p, h, bytes, hipMemcpyHostToDevice, c->stream));
        scrg_status s = scrg_pack_planar(c, c->d_ascii.as<char>(), words, c->d_seq.as<uint64_t>(), c->d_bad.as<uint32_t>());
        if (s != SCRG_OK) return s;
    }
    uint32_t bad = 0;
    HIP_TRY(c, hipMemcpyAsync(&bad, c->d_bad.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (bad) return c->fail(SCRG_ERR_BAD_BASE, "genome contains characters other than ACGTacgt");
    c->genome_len = genome_len;
    c->genome_words = words;
    c->genome_resident = true;
    return SCRG_OK;
}

scrg_status scrg_align_mapping_stranded(scrg_ctx* c, const scrg_params* params, const char* genome,
                                        uint64_t genome_len, uint64_t n_reads, const char* const* reads,
                                        const uint64_t* read_lens, const uint64_t* cand_offsets,
                                        const uint64_t* cand_start, const uint8_t* cand_reverse, scrg_result** out)
{
    return guarded(c, [&] {
        return align_mapping_impl(c, params, genome, genome_len, n_reads, reads, read_lens, cand_offsets, cand_start,
                                  cand_reverse, out);
    });
}

scrg_status scrg_genome_set(scrg_ctx* c, const char* genome, uint64_t genome_len)
{
    return guarded(c, [&] { return genome_set_impl(c, genome, genome_len); });
}

void scrg_genome_clear(scrg_ctx* c)
{
    if (c) c->genome_resident = false;
}

scrg_status scrg_align_mapping_resident(scrg_ctx* c, const scrg_params* params, uint64_t n_reads, const char* const* reads,
                                        const uint64_t* read_lens, const uint64_t* cand_offsets, const uint64_t* cand_start,
                                        const uint8_t* cand_reverse, scrg_result** out)
{
    return guarded(c, [&] {
        return align_mapping_impl(c, params, nullptr, 0, n_reads, reads, read_lens, cand_offsets, cand_start, cand_reverse,
                                  out, true);
    });
}

scrg_status scrg_align_mapping(scrg_ctx* c, const scrg_params* params, const char* genome, uint64_t genome_len,
                               uint64_t n_reads, const char* const* reads, const uint64_t* read_lens,
                               const uint64_t* cand_offsets, const uint64_t* cand_start, scrg_result** out)
{
    return scrg_align_mapping_stranded(c, params, genome, genome_len, n_reads, reads, read_lens, cand_offsets,
                                       cand_start, nullptr, out);
}

}  // extern "C"
