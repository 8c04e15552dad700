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
#include "scrg_internal.h"
#include "../../include/scrooge_amd_io.h"

namespace {

std::atomic<int> g_log{0};
std::atomic<int> g_live_ctx{0};          // handles alive: the result pool is emptied when the last one goes
// the work queue is a 32-bit counter and every wavefront over-asks by up to 64 once it is empty (at most 32 wavefronts
// on each of at most 1024 CUs): leave room for that, or the counter could wrap and hand out low indices a second time
constexpr uint64_t kMaxPairsPerLaunch = 0xffffffffull - 64ull * 32ull * 1024ull;

using scrg_int::DevBuf;
using scrg_int::HostPinned;
using scrg_int::now_ns;
using scrg_int::parallel_for;

}  // namespace

struct scrg_ctx {
    int device = 0;
    int n_cus = 0;
    bool counted = true;          // false: a handle the library made for itself (host path slots)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool have_timing = false;
    std::string last_error;

    DevBuf counter;     // work queue head
    DevBuf spill;       // HBM overflow rows of R
    DevBuf stats;       // profiling counters (params.reserved[1] != 0)
    DevBuf sort_ws;     // scrg_decode_edit_stream: pair order by stream length (indices, sorted keys / indices, radix sort scratch)
    void* host_state = nullptr;   // the pipelined host-pointer path's buffers, streams and resident genome (scrg_host.cpp), made on first use

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

using scrg_int::g_pool;

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

int scrg_abi_version(void) { return SCRG_ABI_VERSION; }

int scrg_build_flags(void)
{
    int f = 0;
#ifdef SCRG_STATS
    f |= SCRG_BUILD_STATS;
#endif
#ifdef SCRG_ABLATE
    f |= SCRG_BUILD_ABLATE;
#endif
#ifdef SCRG_SELECT
    f |= SCRG_BUILD_SELECT;
#endif
    return f;
}

int scrg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

static scrg_status ctx_create_impl(int device, scrg_ctx** out, bool counted)
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
    c->counted = counted;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_stop) != hipSuccess) {
        delete c;
        return SCRG_ERR_HIP;
    }
    c->stream = c->own_stream;
    if (counted) g_live_ctx.fetch_add(1);
    if (counted && g_log.load())
        fprintf(stderr, "[scrooge_amd] device %d: %s, %d CUs, arch %s\n", device, prop.name, c->n_cus,
                prop.gcnArchName);
    *out = c;
    return SCRG_OK;
}

scrg_status scrg_ctx_create(int device, scrg_ctx** out) { return ctx_create_impl(device, out, true); }

void scrg_ctx_destroy(scrg_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    c->counter.release();
    c->spill.release();
    c->stats.release();
    c->sort_ws.release();
    if (c->host_state) scrg_host::state_free(c->host_state);
    c->host_state = nullptr;
    (void)hipSetDevice(c->device);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    const bool counted = c->counted;
    delete c;
    if (counted && g_live_ctx.fetch_sub(1) == 1) g_pool.trim();      // the caller's last handle gone: give the recycled result arrays back
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
        p->outputs = in->outputs;
        p->reserved[0] = in->reserved[0];      // ablation switches and profiling counters travel with the parameters
        p->reserved[1] = in->reserved[1];
        p->stranded = in->stranded;
    }
    if (p->stranded != 0 && p->stranded != 1) return false;
    // experiment switches: only those that leave the results intact, unless this is an ablation build (genasm_kernels.h)
    if (p->reserved[0] & ~scrg::SCRG_ALLOWED_SWITCHES) return false;
    if (p->reserved[1] && !scrg::SCRG_HAVE_STATS) return false;       // the kernels' counters exist in -DSCRG_STATS builds only
    if (p->outputs < SCRG_OUT_ALL || p->outputs > SCRG_OUT_RUNS) return false;
    if (p->text_stride_words == 0) p->text_stride_words = 1;
    if (p->read_stride_words == 0) p->read_stride_words = 1;
    if (p->text_stride_words < 1 || p->read_stride_words < 1) return false;
    if (p->W < 2 || p->W > 256) return false;
    const int tbl = p->W - p->O;
    // O = 0 (no overlap: the reference's special case src/genasm_cpu.cpp:104-110, reached by its O sweep for W < 32,
    // scripts/profile.py:92-93): a window's traceback may consume all W characters.  The one-pair-per-lane kernels serve it
    // (a table column holds the steps OUT of it, so column W-1 needs nothing of the boundary column); the GenASM-row
    // mappings, whose traceback reads R[i+1], do not.
    if (tbl < 1 || p->O < 0) return false;
    if (p->O == 0 && p->lanes_per_pair > 1) return false;
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
        // 32 <= W-O <= 63: genasm_lane_wide_kernel.hip, whose table takes 128 registers: two wavefronts per SIMD
        if (p->lanes_per_pair == 1 && p->waves_per_cu == 0 && scrg::lane_wide_serves(p->W, tbl) && !SCRG_SEL(p->reserved[0], scrg::SCRG_SWITCH_MW_TABLE))
            p->waves_per_cu = 8;
    }
    // 11 and 12 wavefronts per CU align equally fast (the kernel is issue-bound); 11 leaves VGPRs and LDS on
    // every CU for kernels of other streams (RCCL's gather in bench.py --gpus N).  The LDS footprint caps it.
    if (p->waves_per_cu == 0) p->waves_per_cu = p->lanes_per_pair == 1 ? 16 : 11;
    const int g = p->lanes_per_pair;
    // reverse-strand pairs from one packed copy of the read: the one-pair-per-lane kernels (every W / O)
    if (p->stranded && g != 1) return false;
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
    if (p.lanes_per_pair == 1 && scrg::lane_wide_serves(p.W, p.W - p.O) && !SCRG_SEL(p.reserved[0], scrg::SCRG_SWITCH_MW_TABLE))
        return scrg::lane_wide_lds_bytes(p.W);       // genasm_lane_wide_kernel
    if (p.lanes_per_pair == 1 && scrg::lane_parts_serves(p.W, p.W - p.O) && !SCRG_SEL(p.reserved[0], scrg::SCRG_SWITCH_MW_TABLE))
        return scrg::lane_parts_lds_bytes(p.W);      // genasm_lane_parts_kernel
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
    // One pair per lane, runs output, the default table (W <= 64, W-O <= 31): a launch of at most one wavefront per SIMD (65 536
    // pairs on 1024 SIMDs) is bound by the chain of a pair's windows, not by issue slots — it runs with a window's work split
    // over a producer and a consumer wavefront (genasm_lane_split_kernel: 1.93 -> 1.50 ms for 25 k ... 50 k x 10 kb pairs, the
    // chunks of the host entry points included).  From two wavefronts on some SIMDs on the split gains nothing (100 k pairs:
    // 2.51 vs 2.52 ms: those SIMDs are the last to finish either way), and launches that fill the GPU or overlap with others —
    // and edit-stream output — keep the one-wavefront kernel (3 % fewer instructions).  (Test build, -DSCRG_SELECT: reserved[0] = 512 / 1024 force one or the other.)
    bool lane_split = false;
    // A caller that set scrg_params.waves_per_cu itself (co-resident work sized from scrg_query_launch, a work queue shorter than
    // the batch) gets exactly that geometry: the split form, which has its own (two workgroups of eight wavefronts per CU), is
    // then only taken when asked for.
    const bool user_waves = params && params->waves_per_cu > 0;
    if (!edits && p.lanes_per_pair == 1 && p.W <= 64 && p.W - p.O <= 31 && !SCRG_SEL(p.reserved[0], scrg::SCRG_SWITCH_NO_SPLIT) &&
        !(params && params->reserved[1])) {
        const uint64_t simds = 4ull * (uint64_t)c->n_cus;
        lane_split = SCRG_SEL(p.reserved[0], scrg::SCRG_SWITCH_SPLIT) || (need_waves <= simds && !user_waves);
        if (lane_split) n_waves = c->n_cus * scrg::LANE_SPLIT_PRODUCERS_PER_CU;
    }
    if ((uint64_t)n_waves > need_waves) n_waves = (int32_t)need_waves;

    HIP_TRY(c, c->counter.ensure(sizeof(uint32_t)));
    const size_t spill_rows = p.W > 64 ? (size_t)p.W + 1 : scrg::SPILL_ROWS;
    const size_t spill_row_dw = scrg::stored_row_dwords(p.W, p.W - p.O);
    const bool lane_wide = p.lanes_per_pair == 1 && scrg::lane_wide_serves(p.W, p.W - p.O) && !SCRG_SEL(p.reserved[0], scrg::SCRG_SWITCH_MW_TABLE);
    const bool lane_parts = !lane_wide && p.lanes_per_pair == 1 && scrg::lane_parts_serves(p.W, p.W - p.O) && !SCRG_SEL(p.reserved[0], scrg::SCRG_SWITCH_MW_TABLE);
    const bool lane_mw = !lane_wide && !lane_parts && p.lanes_per_pair == 1 && (p.W > 64 || p.W - p.O > 31);      // the rest: genasm_lane_mw_kernel
    if (lane_parts)                      // its checkpoints: one slab of HBM per wavefront (workgroups of four)
        HIP_TRY(c, c->spill.ensure((size_t)((n_waves + 3) / 4 * 4) * scrg::lane_parts_checkpoint_bytes(p.W)));
    else if (lane_mw)                    // its window tables: one slab of HBM per wavefront
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
    a.stranded = (uint32_t)p.stranded;
    a.stats = nullptr;
    if (params && params->reserved[1]) {
        HIP_TRY(c, c->stats.ensure(12 * sizeof(uint64_t)));
        HIP_TRY(c, hipMemsetAsync(c->stats.p, 0, 12 * sizeof(uint64_t), c->stream));
        a.stats = c->stats.as<uint64_t>();
    }

    HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    if (lane_wide)
        HIP_TRY(c, scrg::launch_align_lane_wide(a, n_waves, (size_t)lds, c->stream, edits));
    else if (lane_parts)
        HIP_TRY(c, scrg::launch_align_lane_parts(a, n_waves, (size_t)lds, c->stream, edits));
    else if (lane_mw)
        HIP_TRY(c, scrg::launch_align_lane_mw(a, n_waves, (size_t)lds, c->stream, edits));
    else if (p.W > 64)
        HIP_TRY(c, scrg::launch_align_multiword(p.lanes_per_pair, a, n_waves, (size_t)lds, c->stream));
    else if (lane_split)
        HIP_TRY(c, scrg::launch_align_lane_split(a, n_waves, c->stream));
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
    if (!scrg::SCRG_HAVE_STATS)
        return c->fail(SCRG_ERR_INVALID_ARG, "this library was built without -DSCRG_STATS: the kernels have no counters (scripts/ab.sh build stats -DSCRG_STATS)");
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

scrg_status scrg_encode_edit_stream(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const scrg_pair_desc* d_pairs,
                                    const scrg_run* d_runs, const uint32_t* d_n_runs, uint8_t* d_stream, uint64_t stream_cap,
                                    uint64_t* d_stream_off, uint32_t* d_stream_len, uint64_t* d_total)
{
    if (!c) return SCRG_ERR_INVALID_ARG;
    scrg_params p;
    if (!resolve_params(params, &p)) return c->fail(SCRG_ERR_INVALID_ARG, "bad scrg_params");
    if (!d_total) return c->fail(SCRG_ERR_INVALID_ARG, "d_total is required");
    if (n_pairs && (!d_pairs || !d_runs || !d_n_runs || !d_stream_off || !d_stream_len || (stream_cap && !d_stream)))
        return c->fail(SCRG_ERR_INVALID_ARG, "null device pointer");
    if (reinterpret_cast<uintptr_t>(d_stream) & 3u) return c->fail(SCRG_ERR_INVALID_ARG, "d_stream needs 4-byte alignment");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, scrg::launch_encode_edits(n_pairs, (uint32_t)p.W, (uint32_t)p.O, d_pairs, reinterpret_cast<const uint16_t*>(d_runs), d_n_runs, d_stream,
                                         stream_cap, d_stream_off, d_stream_len, d_total, c->stream));
    return SCRG_OK;
}

scrg_status scrg_decode_edit_stream(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const uint8_t* d_stream,
                                    uint64_t stream_bytes, const uint64_t* d_stream_off, const uint32_t* d_stream_len,
                                    const uint64_t* d_read_len, uint64_t read_len_stride, const uint64_t* d_dense_offset,
                                    scrg_run* d_dense, uint64_t dense_capacity, uint32_t* d_n_runs, uint32_t* d_bad_count)
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
    if (n_pairs >= 4096 && n_pairs < 0x7fffffffull && !scrg::decode_by_wavefront(n_pairs, stream_bytes)) {       // (the lane-per-pair decoder only)
        temp_bytes = scrg::decode_sort_temp_bytes(n_pairs);
        HIP_TRY(c, c->sort_ws.ensure(3 * n_pairs * sizeof(uint32_t) + 256 + temp_bytes));
        ws = c->sort_ws.p;
    }
    HIP_TRY(c, scrg::launch_decode_edits(n_pairs, d_stream, stream_bytes, d_stream_off, d_stream_len,
                                         d_read_len, read_len_stride, d_dense_offset, reinterpret_cast<uint16_t*>(d_dense), dense_capacity,
                                         d_n_runs, d_bad_count, ws, temp_bytes, c->stream));
    return SCRG_OK;
}

// The device decoder's per-lane state machine (edit_stream.h: decode_lane_step, the code decode_edits_kernel runs in
// every lane) on the host, for ONE pair: same arguments as scrg_edit_stream_to_runs, same runs; like the device it does
// not look at the window geometry.  Exists so that the state machine can be checked without a GPU (tests/test_edit_stream.py).
scrg_status scrg_edit_stream_to_runs_lane(const scrg_params* params, uint64_t read_len, const uint8_t* stream, uint64_t n_bytes,
                                          scrg_run* runs, uint64_t runs_cap, uint64_t* n_runs)
{
    scrg_params p;
    if (!n_runs || (n_bytes && !stream) || (runs_cap && !runs) || !resolve_params(params, &p)) return SCRG_ERR_INVALID_ARG;
    *n_runs = 0;
    if (read_len > 0x7fffffffull || n_bytes > 0x3fffffffull) return SCRG_ERR_INVALID_ARG;
    scrg::DecodeLane s;
    scrg::decode_lane_init(s, 0u);
    for (uint64_t k = 0; k < n_bytes; k++) {
        scrg::decode_lane_step(s, (uint32_t)stream[k], [&](uint32_t at, uint32_t word) {         // at: byte offset of the run's slot
            if ((at >> 1) < runs_cap) { runs[at >> 1].count = (uint8_t)word; runs[at >> 1].op = (char)(word >> 8); }
        });
        if ((k & 15u) == 15u) scrg::decode_lane_guard(s);
    }
    if (read_len > 0x7fffffffull) return SCRG_ERR_INVALID_ARG;
    if (!scrg::decode_lane_clean(s, n_bytes ? (uint32_t)stream[n_bytes - 1] : 0u, (uint32_t)read_len)) return SCRG_ERR_INVALID_ARG;
    *n_runs = scrg::decode_lane_runs(s);
    return *n_runs > runs_cap ? SCRG_ERR_CIGAR_OVERFLOW : SCRG_OK;
}

scrg_status scrg_edit_stream_to_runs(const scrg_params* params, uint64_t read_len, const uint8_t* stream, uint64_t n_bytes,
                                     scrg_run* runs, uint64_t runs_cap, uint64_t* n_runs)
{
    scrg_params p;
    if (!n_runs || (n_bytes && !stream) || (runs_cap && !runs) || !resolve_params(params, &p)) return SCRG_ERR_INVALID_ARG;
    uint64_t k = 0;
    const uint64_t n = scrg::replay_edit_stream(stream, n_bytes, read_len, (uint32_t)(p.W - p.O),
                                                [&](uint32_t op, uint64_t t) {
                                                    if (k < runs_cap) { runs[k].count = (uint8_t)t; runs[k].op = (char)op; }
                                                    k++;
                                                });
    *n_runs = n == ~0ull ? 0 : n;
    if (n == ~0ull) return SCRG_ERR_INVALID_ARG;
    return n > runs_cap ? SCRG_ERR_CIGAR_OVERFLOW : SCRG_OK;
}

scrg_status scrg_runs_to_edit_stream(const scrg_params* params, const scrg_run* runs, uint64_t n_runs, uint8_t* stream,
                                     uint64_t stream_cap, uint64_t* n_bytes)
{
    scrg_params p;
    if (!n_bytes || (n_runs && !runs) || (stream_cap && !stream) || !resolve_params(params, &p)) return SCRG_ERR_INVALID_ARG;
    uint64_t k = 0;
    const uint64_t n = scrg::encode_runs(n_runs, (uint32_t)(p.W - p.O),
                                         [&](uint64_t r) { return (uint32_t)runs[r].count | (uint32_t)(uint8_t)runs[r].op << 8; },
                                         [&](uint8_t b) {
                                             if (k < stream_cap) stream[k] = b;
                                             k++;
                                         });
    *n_bytes = n == ~0ull ? 0 : n;
    if (n == ~0ull) return SCRG_ERR_INVALID_ARG;
    return n > stream_cap ? SCRG_ERR_CIGAR_OVERFLOW : SCRG_OK;
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

}  // extern "C"

// ---------------------------------------------------------------------------
// host-pointer entry points: thin argument checks in front of the pipeline of scrg_host.cpp
// ---------------------------------------------------------------------------
namespace {

thread_local std::string g_multi_error;
std::mutex g_multi_mu;
std::vector<std::pair<int, void*>> g_multi_states;       // (device, state), in the order they were first asked for
std::vector<char> g_multi_busy;

void* ctx_state(scrg_ctx* c)
{
    if (!c->host_state) c->host_state = scrg_host::state_create(c->device);
    return c->host_state;
}

// checks and the pair -> read table shared by the mapping entry points
scrg_status mapping_batch(uint64_t n_reads, const char* const* reads, const uint64_t* read_lens, const uint64_t* cand_offsets,
                          const uint64_t* cand_start, const uint8_t* cand_reverse, scrg_host::Batch* b, std::vector<uint32_t>* pair_read,
                          std::string* err)
{
    if ((n_reads && (!reads || !read_lens)) || !cand_offsets) { *err = "null input array"; return SCRG_ERR_INVALID_ARG; }
    const uint64_t n_pairs = cand_offsets[n_reads];
    if (n_pairs > kMaxPairsPerLaunch || n_reads > 0xfffffff0ull) { *err = "too many pairs"; return SCRG_ERR_INVALID_ARG; }
    if (n_pairs && !cand_start) { *err = "null candidate array"; return SCRG_ERR_INVALID_ARG; }
    for (uint64_t r = 0; r < n_reads; r++) {
        if (read_lens[r] && !reads[r]) { *err = "null read pointer"; return SCRG_ERR_INVALID_ARG; }
        if (read_lens[r] > 0x7fffffffull) { *err = "read longer than 2^31-1"; return SCRG_ERR_INVALID_ARG; }
        if (cand_offsets[r + 1] < cand_offsets[r]) { *err = "cand_offsets not monotone"; return SCRG_ERR_INVALID_ARG; }
    }
    pair_read->resize(n_pairs);
    parallel_for(n_reads, [&](uint64_t r) {
        for (uint64_t k = cand_offsets[r]; k < cand_offsets[r + 1]; k++) (*pair_read)[k] = (uint32_t)r;
    });
    b->n_pairs = n_pairs;
    b->mapping = true;
    b->reads = reads;
    b->read_lens = read_lens;
    b->cand_start = cand_start;
    b->cand_reverse = cand_reverse;
    b->pair_read = pair_read->data();
    b->n_reads = n_reads;
    return SCRG_OK;
}

scrg_status pairs_batch(uint64_t n_pairs, const char* const* texts, const uint64_t* text_lens, const char* const* queries,
                        const uint64_t* query_lens, scrg_host::Batch* b, std::string* err)
{
    if (n_pairs && (!texts || !text_lens || !queries || !query_lens)) { *err = "null input array"; return SCRG_ERR_INVALID_ARG; }
    if (n_pairs > kMaxPairsPerLaunch) { *err = "too many pairs"; return SCRG_ERR_INVALID_ARG; }
    for (uint64_t i = 0; i < n_pairs; i++) {
        if ((text_lens[i] && !texts[i]) || (query_lens[i] && !queries[i])) { *err = "null sequence pointer"; return SCRG_ERR_INVALID_ARG; }
        if (query_lens[i] > 0x7fffffffull) { *err = "read longer than 2^31-1"; return SCRG_ERR_INVALID_ARG; }
    }
    b->n_pairs = n_pairs;
    b->mapping = false;
    b->texts = texts;
    b->text_lens = text_lens;
    b->reads = queries;
    b->read_lens = query_lens;
    return SCRG_OK;
}

scrg_status ctx_align(scrg_ctx* c, const scrg_params* params, scrg_host::Batch& b, scrg_result** out)
{
    scrg_params p;
    if (!resolve_params(params, &p)) return c->fail(SCRG_ERR_INVALID_ARG, "bad scrg_params");
    void* st = ctx_state(c);
    if (!st) return c->fail(SCRG_ERR_NO_DEVICE, "no usable HIP device for the host path");
    std::string err;
    const scrg_status s = scrg_host::align(&st, 1, p, b, out, &err);
    if (s != SCRG_OK) c->fail(s, err.c_str());
    return s;
}

// the states of a multi-device call: one per listed device (a device listed twice gets two), kept for later calls
scrg_status multi_states(const int32_t* devices, int32_t n_devices, std::vector<void*>* st, std::vector<size_t>* taken)
{
    std::lock_guard<std::mutex> g(g_multi_mu);
    for (int32_t d = 0; d < n_devices; d++) {
        size_t found = g_multi_states.size();
        for (size_t k = 0; k < g_multi_states.size(); k++)
            if (g_multi_states[k].first == devices[d] && !g_multi_busy[k]) { found = k; break; }
        if (found == g_multi_states.size()) {
            void* s = scrg_host::state_create(devices[d]);
            if (!s) {
                for (size_t k : *taken) g_multi_busy[k] = 0;
                return SCRG_ERR_NO_DEVICE;
            }
            g_multi_states.emplace_back(devices[d], s);
            g_multi_busy.push_back(0);
        }
        g_multi_busy[found] = 1;
        taken->push_back(found);
        st->push_back(g_multi_states[found].second);
    }
    return SCRG_OK;
}

void multi_done(const std::vector<size_t>& taken)
{
    std::lock_guard<std::mutex> g(g_multi_mu);
    for (size_t k : taken) g_multi_busy[k] = 0;
}

scrg_status multi_align(const int32_t* devices, int32_t n_devices, const scrg_params* params, scrg_host::Batch& b, scrg_result** out)
{
    scrg_params p;
    if (!resolve_params(params, &p)) { g_multi_error = "bad scrg_params"; return SCRG_ERR_INVALID_ARG; }
    std::vector<void*> st;
    std::vector<size_t> taken;
    scrg_status s = multi_states(devices, n_devices, &st, &taken);
    if (s != SCRG_OK) { g_multi_error = "no usable HIP device"; return s; }
    std::string err;
    s = scrg_host::align(st.data(), (int)st.size(), p, b, out, &err);
    multi_done(taken);
    g_multi_error = s == SCRG_OK ? "" : err;
    return s;
}

template <typename F> scrg_status multi_guarded(F&& f)
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        g_multi_error = "host allocation failed";
        return SCRG_ERR_OOM;
    } catch (...) {
        g_multi_error = "unexpected exception";
        return SCRG_ERR_INVALID_ARG;
    }
}

}  // namespace

extern "C" {

scrg_status scrg_align_pairs(scrg_ctx* c, const scrg_params* params, uint64_t n_pairs, const char* const* texts,
                             const uint64_t* text_lens, const char* const* queries, const uint64_t* query_lens,
                             scrg_result** out)
{
    if (!c || !out) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded(c, [&] {
        scrg_host::Batch b;
        std::string err;
        scrg_status s = pairs_batch(n_pairs, texts, text_lens, queries, query_lens, &b, &err);
        if (s != SCRG_OK) return c->fail(s, err.c_str());
        return ctx_align(c, params, b, out);
    });
}

scrg_status scrg_align_mapping_stranded(scrg_ctx* c, const scrg_params* params, const char* genome,
                                        uint64_t genome_len, uint64_t n_reads, const char* const* reads,
                                        const uint64_t* read_lens, const uint64_t* cand_offsets,
                                        const uint64_t* cand_start, const uint8_t* cand_reverse, scrg_result** out)
{
    if (!c || !out) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded(c, [&] {
        if (genome_len && !genome) return c->fail(SCRG_ERR_INVALID_ARG, "null input array");
        scrg_host::Batch b;
        std::vector<uint32_t> pair_read;
        std::string err;
        scrg_status s = mapping_batch(n_reads, reads, read_lens, cand_offsets, cand_start, cand_reverse, &b, &pair_read, &err);
        if (s != SCRG_OK) return c->fail(s, err.c_str());
        static const char empty_genome[1] = {0};
        b.genome = genome ? genome : empty_genome;          // (the genome is staged by this call, as the reference re-converts it, genasm_cpu.cpp:508)
        b.genome_len = genome_len;
        return ctx_align(c, params, b, out);
    });
}

scrg_status scrg_align_mapping(scrg_ctx* c, const scrg_params* params, const char* genome, uint64_t genome_len,
                               uint64_t n_reads, const char* const* reads, const uint64_t* read_lens,
                               const uint64_t* cand_offsets, const uint64_t* cand_start, scrg_result** out)
{
    return scrg_align_mapping_stranded(c, params, genome, genome_len, n_reads, reads, read_lens, cand_offsets,
                                       cand_start, nullptr, out);
}

// scrg_genome_set: pack (host threads) and transfer a genome once; it stays at the front of the handle's sequence
// array until another genome is set or scrg_genome_clear() is called.
scrg_status scrg_genome_set(scrg_ctx* c, const char* genome, uint64_t genome_len)
{
    if (!c || (genome_len && !genome)) return SCRG_ERR_INVALID_ARG;
    return guarded(c, [&] {
        void* st = ctx_state(c);
        if (!st) return c->fail(SCRG_ERR_NO_DEVICE, "no usable HIP device for the host path");
        std::string err;
        static const char empty_genome[1] = {0};
        const scrg_status s = scrg_host::genome_set(st, genome ? genome : empty_genome, genome_len, &err);
        if (s != SCRG_OK) c->fail(s, err.c_str());
        return s;
    });
}

void scrg_genome_clear(scrg_ctx* c)
{
    if (c && c->host_state) scrg_host::genome_clear(c->host_state);
}

scrg_status scrg_align_mapping_resident(scrg_ctx* c, const scrg_params* params, uint64_t n_reads, const char* const* reads,
                                        const uint64_t* read_lens, const uint64_t* cand_offsets, const uint64_t* cand_start,
                                        const uint8_t* cand_reverse, scrg_result** out)
{
    if (!c || !out) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded(c, [&] {
        if (!c->host_state || !scrg_host::genome_resident(c->host_state, nullptr))
            return c->fail(SCRG_ERR_INVALID_ARG, "no resident genome: call scrg_genome_set first");
        scrg_host::Batch b;
        std::vector<uint32_t> pair_read;
        std::string err;
        scrg_status s = mapping_batch(n_reads, reads, read_lens, cand_offsets, cand_start, cand_reverse, &b, &pair_read, &err);
        if (s != SCRG_OK) return c->fail(s, err.c_str());
        b.genome = nullptr;                                  // the resident one
        return ctx_align(c, params, b, out);
    });
}

scrg_status scrg_align_pairs_multi(const int32_t* devices, int32_t n_devices, const scrg_params* params, uint64_t n_pairs,
                                   const char* const* texts, const uint64_t* text_lens, const char* const* queries,
                                   const uint64_t* query_lens, scrg_result** out)
{
    if (!out || !devices || n_devices < 1 || n_devices > 64) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    return multi_guarded([&] {
        scrg_host::Batch b;
        std::string err;
        scrg_status s = pairs_batch(n_pairs, texts, text_lens, queries, query_lens, &b, &err);
        if (s != SCRG_OK) { g_multi_error = err; return s; }
        return multi_align(devices, n_devices, params, b, out);
    });
}

scrg_status scrg_align_mapping_multi(const int32_t* devices, int32_t n_devices, const scrg_params* params, const char* genome,
                                     uint64_t genome_len, uint64_t n_reads, const char* const* reads, const uint64_t* read_lens,
                                     const uint64_t* cand_offsets, const uint64_t* cand_start, const uint8_t* cand_reverse,
                                     scrg_result** out)
{
    if (!out || !devices || n_devices < 1 || n_devices > 64 || (genome_len && !genome)) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    return multi_guarded([&] {
        scrg_host::Batch b;
        std::vector<uint32_t> pair_read;
        std::string err;
        scrg_status s = mapping_batch(n_reads, reads, read_lens, cand_offsets, cand_start, cand_reverse, &b, &pair_read, &err);
        if (s != SCRG_OK) { g_multi_error = err; return s; }
        static const char empty_genome[1] = {0};
        b.genome = genome ? genome : empty_genome;
        b.genome_len = genome_len;
        return multi_align(devices, n_devices, params, b, out);
    });
}

scrg_status scrg_host_plan(const scrg_params* params, int32_t n_devices, uint64_t n_pairs, const uint64_t* text_lens,
                           const uint64_t* read_lens, uint32_t* issue_order, uint64_t* chunk_first, uint64_t chunk_cap, uint64_t* n_chunks)
{
    if (n_devices < 1 || !n_chunks || (n_pairs && !read_lens)) return SCRG_ERR_INVALID_ARG;
    return multi_guarded([&] {
        scrg_params p;
        if (!resolve_params(params, &p)) return (scrg_status)SCRG_ERR_INVALID_ARG;
        scrg_host::Batch b;
        std::vector<uint64_t> zeros;
        b.n_pairs = n_pairs;
        b.mapping = false;
        b.read_lens = read_lens;
        if (!text_lens) {
            zeros.assign(n_pairs, 0);
            text_lens = zeros.data();
        }
        b.text_lens = text_lens;
        return scrg_host::plan(p, n_devices, b, issue_order, chunk_first, chunk_cap, n_chunks);
    });
}

scrg_status scrg_pack_planar_host(const char* ascii, uint64_t n_bases, uint64_t* planar, uint64_t stride_words, uint64_t n_words)
{
    if ((n_bases && !ascii) || (n_words && !planar) || 32 * n_words < n_bases) return SCRG_ERR_INVALID_ARG;
    return scrg_host::pack_planar_host(ascii, n_bases, planar, stride_words, n_words) ? SCRG_ERR_BAD_BASE : SCRG_OK;
}

void scrg_multi_release(void)
{
    std::lock_guard<std::mutex> g(g_multi_mu);
    for (auto& ds : g_multi_states) scrg_host::state_free(ds.second);
    g_multi_states.clear();
    g_multi_busy.clear();
    if (g_live_ctx.load() == 0) g_pool.trim();     // no handle of the caller's is alive either: give the recycled result arrays back
}

const char* scrg_multi_last_error(void) { return g_multi_error.c_str(); }

}  // extern "C"

namespace scrg_int {
scrg_status ctx_create_internal(int device, scrg_ctx** out) { return ctx_create_impl(device, out, false); }
}  // namespace scrg_int
