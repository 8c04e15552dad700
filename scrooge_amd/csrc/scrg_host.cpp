// scrg_host.cpp — the host-pointer entry points as a PIPELINE over chunks, streams and GPUs.
//
// What the reference does around its kernel (src/genasm_gpu.cu:890-1065: convert and concatenate the sequences, size the
// CIGAR storage, launch, walk the lists into strings) happens here per CHUNK of a few thousand pairs, four chunks in
// flight per GPU, so that the PCIe transfers, the kernels and the host's own work overlap:
//
//   host   pack the chunk's sequences to 2 bits per base (AVX2: 32 bases per load, two movemasks) straight into pinned
//          memory, in the lane-interleaved layout the align kernel reads best — a quarter of the bytes of ASCII cross PCIe
//   H2D    packed sequences + 8..16 bytes per pair; the 48-byte problem descriptors are built on the device
//   GPU    genasm_lane_kernel, per-pair text lengths, offsets by prefix sum, run compaction, "%d%c" rendering of the text
//   D2H    dense runs, text, offsets, scores of the chunk into pinned staging; two sizes come back first
//   host   the chunk's results are copied to their place in the result arrays (threads), offsets get their base
//
// A batch is cut in issue order (longest read first, src/tests.cu:375-377); chunk k goes to device k mod N, so several
// GPUs share one call (scrg_align_pairs_multi: two host threads, four streams and four buffer sets per device).
// Results are assembled in issue order as chunks finish and put into caller order at the end (nothing to do when the
// batch already was in issue order).  No CPU fallback anywhere: without a device every entry point fails.
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <condition_variable>
#include <cstdio>
#include <numeric>
#include <shared_mutex>

#include "genasm_kernels.h"
#include "host_path.h"
#include "scrg_internal.h"

namespace {

using scrg_int::DevBuf;
using scrg_int::g_pool;
using scrg_int::HostPinned;
using scrg_int::now_ns;
using scrg_int::parallel_for;

constexpr int NSLOT = 4;                                 // chunks in flight per device (HIP has 4 hardware queues per process and device) ...
constexpr int MAXSLOT = 8;                               // ... and twice as many for short reads: their kernels take ~0.2 ms per chunk, what limits such a call is
                                                         // how soon a slot comes back (pack, H2D, kernels, look at the sizes, compaction, text, D2H, collection)
constexpr uint64_t SHORT_READS = 1000;                   // longest read of a call that uses MAXSLOT slots
constexpr size_t NCOLLECT = 2;                            // collect threads per device
constexpr size_t LAG2_DEFAULT = 2;                        // a chunk's sizes are looked at this many chunks after it was launched
constexpr uint64_t GROUP = 64;
constexpr uint64_t SEQ_PAD = 2 * GROUP + 2;              // SCRG_SEQ_PAD_WORDS_STRIDED(64)

// ---------------------------------------------------------------------------------------------------------------
// ASCII -> planar 2-bit on the host.  One uint64 per 32 bases: bit k of the low dword = bit 0 of base k's code, bit k of
// the high dword = bit 1 (A0 C1 G2 T3, src/genasm_cpu.cpp:87-90; lower case too, :462-493).  (c >> 1) & 3 gives
// A0 C1 T2 G3; code = x ^ (x >> 1).  Returns the word; *bad gets a bit for every byte that is not one of ACGTacgt.
// ---------------------------------------------------------------------------------------------------------------
__attribute__((target("avx2"))) inline uint64_t pack32_avx2(const char* p, uint32_t* bad)
{
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
    const uint32_t b1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 6));      // bit 1 of every byte
    const uint32_t b2 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 5));      // bit 2
    const __m256i u = _mm256_and_si256(x, _mm256_set1_epi8((char)0xDF));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('C'))),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('T'))));
    *bad = ~(uint32_t)_mm256_movemask_epi8(ok);
    return ((uint64_t)b2 << 32) | (uint64_t)(b1 ^ b2);
}

inline uint64_t pack32_scalar(const char* p, uint32_t* bad)
{
    uint32_t lo = 0, hi = 0, bd = 0;
    for (unsigned k = 0; k < 32; k++) {
        const unsigned c = (unsigned char)p[k], u = c & 0xDFu;
        const unsigned x0 = (c >> 1) & 1u, x1 = (c >> 2) & 1u;
        lo |= (x0 ^ x1) << k;
        hi |= x1 << k;
        bd |= (u == 'A' || u == 'C' || u == 'G' || u == 'T' ? 0u : 1u) << k;
    }
    *bad = bd;
    return ((uint64_t)hi << 32) | lo;
}

const bool g_have_avx2 = __builtin_cpu_supports("avx2");

inline char complement_base(char c)
{
    switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    default: return c;   // left as is: counted as a bad base
    }
}

// words [w_from, words) of one sequence at dst[w * stride] (zero beyond the sequence); returns true if a bad base was seen
bool pack_sequence(const char* src, uint64_t len, bool revcomp, uint64_t* dst, uint64_t stride, uint64_t words, uint64_t w_from = 0)
{
    bool bad_any = false;
    const uint64_t full = len / 32;
    uint32_t bad = 0;
    if (!revcomp) {
        // (a 64-bases-per-step AVX-512BW version of this loop packs at the same rate on the EPYC 9575F of the GPU boxes: with
        // 16 threads the packing is bound by memory bandwidth, ~150 GB/s of ASCII, not by instructions)
        if (g_have_avx2)
            for (uint64_t w = w_from; w < full; w++) { dst[w * stride] = pack32_avx2(src + 32 * w, &bad); bad_any |= bad != 0; }
        else
            for (uint64_t w = w_from; w < full; w++) { dst[w * stride] = pack32_scalar(src + 32 * w, &bad); bad_any |= bad != 0; }
    }
    char tmp[32];
    for (uint64_t w = revcomp ? w_from : std::max(full, w_from); 32 * w < len; w++) {
        const uint64_t n = std::min<uint64_t>(32, len - 32 * w);
        if (revcomp) for (uint64_t k = 0; k < n; k++) tmp[k] = complement_base(src[len - 1 - (32 * w + k)]);
        else memcpy(tmp, src + 32 * w, n);
        if (n < 32) memset(tmp + n, 'A', 32 - n);
        uint64_t v = g_have_avx2 ? pack32_avx2(tmp, &bad) : pack32_scalar(tmp, &bad);
        if (n < 32) {                        // padding encodes as A = 0 anyway; keep only the real bases' error bits
            bad &= (1u << n) - 1u;
        }
        dst[w * stride] = v;
        bad_any |= bad != 0;
    }
    for (uint64_t w = std::max((len + 31) / 32, w_from); w < words; w++) dst[w * stride] = 0;
    return bad_any;
}

// Eight forward rows of one group of the lane-interleaved layout at once (dst[w * 64 + l], l = 0..7 consecutive lanes):
// the whole words all eight have are packed word by word across the rows, so that every store completes a 64-byte line
// (row by row, consecutive stores of a row are 512 bytes apart); the rest of each row follows row by row.
__attribute__((target("avx2"))) bool pack_rows8_avx2(const char* const* src, const uint64_t* len, uint64_t* dst, uint64_t words)
{
    uint64_t common = ~0ull;
    for (int l = 0; l < 8; l++) common = std::min(common, len[l] / 32);
    uint32_t bad_acc = 0, bad = 0;
    for (uint64_t w = 0; w < common; w++) {
        uint64_t* const line = dst + w * GROUP;
        for (int l = 0; l < 8; l++) {
            line[l] = pack32_avx2(src[l] + 32 * w, &bad);
            bad_acc |= bad;
        }
    }
    bool bad_any = bad_acc != 0;
    for (int l = 0; l < 8; l++) bad_any |= pack_sequence(src[l], len[l], false, dst + l, GROUP, words, common);
    return bad_any;
}

// ---------------------------------------------------------------------------------------------------------------
// per-device state
// ---------------------------------------------------------------------------------------------------------------
struct Slot {
    hipStream_t stream = nullptr;
    scrg_ctx* ctx = nullptr;                 // a handle of the device-pointer layer bound to `stream`
    hipEvent_t ev_tot = nullptr, ev_done = nullptr;
    HostPinned h_seq, h_meta, h_out, h_tot;
    DevBuf d_meta, d_desc, d_slices, d_ed, d_nruns, d_cnt64, d_len64, d_tot, d_dense, d_temp;     // d_ed: the per-pair results (PerPairLayout), d_dense: runs, then text
    // the chunk in flight
    uint64_t n = 0, first = 0;               // pairs, first issue index
    uint64_t tot_runs = 0, tot_text = 0;
    int64_t t_pack_ns = 0;
};

struct DeviceState {
    int device = 0, n_cus = 0;
    DevBuf d_seq;                            // [ genome | slot 0 | slot 1 | slot 2 ]
    uint64_t genome_words = 0, genome_len = 0;
    bool genome_ok = false;                  // d_seq starts with a packed genome
    uint64_t slot_words = 0;
    Slot slot[MAXSLOT];
    std::mutex mu;                           // one call at a time per device state
    // the last error text: written by the launch thread and both collect threads of a call
    std::mutex err_mu;
    std::string err;
    void set_err(const std::string& e)
    {
        std::lock_guard<std::mutex> g(err_mu);
        err = e;
    }
    std::string get_err()
    {
        std::lock_guard<std::mutex> g(err_mu);
        return err;
    }
};

#define HTRY(ds, call)                                                                                   \
    do {                                                                                                 \
        hipError_t e__ = (call);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            (ds)->set_err(std::string(#call) + ": " + hipGetErrorString(e__));                           \
            (void)hipGetLastError();                                                                     \
            return e__ == hipErrorOutOfMemory ? SCRG_ERR_OOM : SCRG_ERR_HIP;                             \
        }                                                                                                \
    } while (0)

// the sequence array: genome first (so that a candidate's text offset is its start_in_reference), then one region per slot
scrg_status ensure_seq(DeviceState* ds, uint64_t genome_words, uint64_t slot_words)
{
    const uint64_t gpad = genome_words ? genome_words + SCRG_SEQ_PAD_WORDS : 0;
    const uint64_t need_slot = std::max(slot_words, ds->slot_words);
    const uint64_t have_g = ds->genome_words ? ds->genome_words + SCRG_SEQ_PAD_WORDS : 0;
    if (gpad == have_g && need_slot == ds->slot_words && ds->d_seq.p) return SCRG_OK;
    HTRY(ds, hipSetDevice(ds->device));
    HTRY(ds, hipDeviceSynchronize());
    DevBuf bigger;
    const uint64_t grow_slot = need_slot > ds->slot_words ? need_slot + need_slot / 4 : need_slot;
    HTRY(ds, bigger.ensure((gpad + MAXSLOT * grow_slot + 8) * sizeof(uint64_t)));
    if (ds->genome_ok && gpad == have_g && gpad)         // keep a resident genome
        HTRY(ds, hipMemcpy(bigger.p, ds->d_seq.p, gpad * sizeof(uint64_t), hipMemcpyDeviceToDevice));
    else
        ds->genome_ok = false;
    ds->d_seq.release();
    ds->d_seq = bigger;
    ds->genome_words = genome_words;
    ds->slot_words = grow_slot;
    return SCRG_OK;
}

// A genome is packed ONCE per call, whatever the number of device states: into one pinned, portable staging buffer
// (host threads), from which every device gets its copy — the H2D copies of all devices run side by side.  The staging
// buffer is kept for the next call while it is small (a 100 Mbp chromosome is 25 MB) and released when it is not (a
// 3 Gbp genome would otherwise pin 0.8 GB of host memory for good).
constexpr size_t GENOME_STAGING_KEEP = 256u << 20;
struct GenomeStaging {
    std::mutex mu;
    HostPinned buf;                    // (registered memory is portable: every device copies from it)
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        release();
        hipError_t e = bytes >= HostPinned::kRegisterMin ? buf.ensure(bytes) : hipErrorInvalidValue;
        if (e == hipSuccess && buf.registered) {
            p = buf.p;
            cap = buf.cap;
            return hipSuccess;
        }
        buf.release();                 // small, or not registered: portable memory from HIP itself
        e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
        if (e == hipSuccess) cap = bytes;
        else p = nullptr;
        return e;
    }
    void release()
    {
        if (buf.p) buf.release();
        else if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};
GenomeStaging g_genome_staging;

scrg_status stage_genome(DeviceState* const* dss, int n_states, const char* genome, uint64_t genome_len)
{
    const uint64_t words = (genome_len + 31) / 32;
    for (int d = 0; d < n_states; d++) {
        dss[d]->genome_ok = false;
        scrg_status s = ensure_seq(dss[d], words, dss[d]->slot_words);
        if (s != SCRG_OK) {
            if (d) dss[0]->set_err(dss[d]->get_err());
            return s;
        }
    }
    DeviceState* const ds = dss[0];
    std::lock_guard<std::mutex> g(g_genome_staging.mu);
    // a staging area larger than GENOME_STAGING_KEEP is given back on EVERY way out of this function (a rejected 3 Gbp genome
    // must not leave 0.8 GB pinned until the next call)
    struct ReleaseOversize {
        ~ReleaseOversize() { if (g_genome_staging.cap > GENOME_STAGING_KEEP) g_genome_staging.release(); }
    } release_oversize;
    HTRY(ds, hipSetDevice(ds->device));
    HTRY(ds, g_genome_staging.ensure((words + SCRG_SEQ_PAD_WORDS) * sizeof(uint64_t)));
    uint64_t* const h = static_cast<uint64_t*>(g_genome_staging.p);
    const uint64_t PIECE = 1u << 15;                    // words per work item: 1 Mbase
    std::atomic<int> bad{0};
    parallel_for((words + PIECE - 1) / PIECE, [&](uint64_t i) {
        const uint64_t w0 = i * PIECE, w1 = std::min(words, w0 + PIECE);
        const uint64_t b0 = 32 * w0, b1 = std::min(genome_len, 32 * w1);
        if (pack_sequence(genome + b0, b1 - b0, false, h + w0, 1, w1 - w0)) bad.store(1, std::memory_order_relaxed);
    }, true);
    for (uint64_t w = words; w < words + SCRG_SEQ_PAD_WORDS; w++) h[w] = 0;
    if (bad.load()) {
        ds->set_err("genome contains characters other than ACGTacgt");
        return SCRG_ERR_BAD_BASE;
    }
    scrg_status st = SCRG_OK;
    int issued = 0;
    for (int d = 0; d < n_states && st == SCRG_OK; d++) {
        hipError_t e = hipSetDevice(dss[d]->device);
        if (e == hipSuccess)
            e = hipMemcpyAsync(dss[d]->d_seq.p, h, (words + SCRG_SEQ_PAD_WORDS) * sizeof(uint64_t), hipMemcpyHostToDevice, dss[d]->slot[0].stream);
        if (e != hipSuccess) {
            ds->set_err(std::string("genome upload: ") + hipGetErrorString(e));
            (void)hipGetLastError();
            st = SCRG_ERR_HIP;
        } else {
            issued = d + 1;
        }
    }
    for (int d = 0; d < issued; d++) {
        (void)hipSetDevice(dss[d]->device);
        const hipError_t e = hipStreamSynchronize(dss[d]->slot[0].stream);
        if (e != hipSuccess && st == SCRG_OK) {
            ds->set_err(std::string("genome upload: ") + hipGetErrorString(e));
            (void)hipGetLastError();
            st = SCRG_ERR_HIP;
        }
        if (st == SCRG_OK) {
            dss[d]->genome_len = genome_len;
            dss[d]->genome_ok = true;
        }
    }
    return st;
}

scrg_status pack_genome(DeviceState* ds, const char* genome, uint64_t genome_len)
{
    DeviceState* one[1] = {ds};
    return stage_genome(one, 1, genome, genome_len);
}

// ---------------------------------------------------------------------------------------------------------------
// a call
// ---------------------------------------------------------------------------------------------------------------
// A chunk's per-pair results on the device: [ed 8n | status 4n (+pad) | run_off 8n | text_off 8n] (what the kernels write and
// read), and what of them crosses PCIe, the "wire": [ed 4n | run count, bit 31 = overflow 4n | text length 4n] at o_wire
// (wire_totals_kernel) — the offsets are made again on the host from the counts (stage 3).
struct PerPairLayout {
    size_t o_st, o_ro, o_to, o_wire, bytes, wire_bytes;
    explicit PerPairLayout(uint64_t n)
        : o_st(8 * n), o_ro((8 * n + 4 * n + 15) & ~(size_t)15), o_to(o_ro + 8 * n), o_wire((o_to + 8 * n + 255) & ~(size_t)255), bytes(o_wire + 12 * n),
          wire_bytes(12 * n) {}
    size_t host_runs() const { return (wire_bytes + 256 + 4095) & ~(size_t)4095; }      // staging area: [wire | runs (then the text)]
};

struct Grow {                          // a result array that the chunks' collectors fill, each chunk at its own offset
    char* p = nullptr;
    size_t cap = 0;
    std::atomic<size_t> hi{0};         // end of the highest region that has been written (chunks may be collected out of order)
};

struct Call {
    const scrg_host::Batch* b = nullptr;
    scrg_params p;
    uint64_t n = 0;
    std::vector<uint32_t> order;       // issue index -> caller index
    bool identity = true;
    std::vector<uint64_t> chunk_first; // issue index of every chunk's first pair, + n at the end
    int want_runs = 1, want_text = 1;

    // results in issue order
    std::shared_mutex grow_mu;
    Grow runs, text;
    uint64_t* iss_run_off = nullptr;   // [n + 1]
    uint64_t* iss_text_off = nullptr;  // [n + 1]
    int64_t* iss_ed = nullptr;
    uint32_t* iss_status = nullptr;
    // chunk totals, published in any order; bases are prefix sums over them
    std::mutex tot_mu;
    std::condition_variable tot_cv;
    std::vector<uint64_t> c_runs, c_text;
    std::vector<char> c_known;
    // status
    std::atomic<int> status{SCRG_OK};
    std::atomic<int> any_overflow{0};
    std::mutex err_mu;
    std::string err;
    std::atomic<int64_t> kernel_ns{0}, pack_ns{0};
    unsigned threads_per_worker = 16;
    int n_slots = NSLOT;               // slots a device uses for this call (make_plan)

    void fail(scrg_status s, const std::string& what)
    {
        int expect = SCRG_OK;
        if (status.compare_exchange_strong(expect, s)) {
            std::lock_guard<std::mutex> g(err_mu);
            err = what;
        }
        // a collector that has just evaluated its wait predicate (under tot_mu) and not blocked yet must not miss this:
        // pass through the mutex before notifying
        { std::lock_guard<std::mutex> g(tot_mu); }
        tot_cv.notify_all();
    }
    bool failed() const { return status.load() != SCRG_OK; }
};

inline uint64_t read_of(const Call& c, uint64_t p) { return c.b->mapping ? c.b->pair_read[p] : p; }

// `hint`: what the whole call is expected to need (from the first chunk that arrives: bytes per pair x pairs + 8 %).
// Writers hold grow_mu shared while they copy and raise g.hi first; a regrow (exclusive) moves everything below g.hi.
bool grow_to(Call& c, Grow& g, size_t need, size_t hint)
{
    {
        std::shared_lock<std::shared_mutex> lk(c.grow_mu);
        if (need <= g.cap) return true;
    }
    std::unique_lock<std::shared_mutex> lk(c.grow_mu);
    if (need <= g.cap) return true;
    // sizes in classes (steps of 2^(1/4)): which chunk arrives first, and with it the hint, differs from call to call, and
    // the pool hands a block back only to a request it fits (a fresh block of 100 MB costs ~10 ms of page faults)
    size_t cap = std::max(std::max(need + need / 8, g.cap + g.cap / 2), hint);
    if (cap > (1u << 20)) {
        size_t cls = (size_t)1 << 20;
        while (cls < cap) cls <<= 1;                       // 2^k >= cap
        const size_t q = cls / 8;                          // candidates: 5/8, 6/8, 7/8, 8/8 of 2^k
        for (size_t m8 = 5; m8 <= 8; m8++)
            if (q * m8 >= cap) { cap = q * m8; break; }
    }
    char* np = static_cast<char*>(g_pool.get(cap, false));
    if (!np) return false;
    const size_t used = std::min(g.hi.load(), g.cap);
    if (g.p && used) {
        const size_t CH = 1u << 22;
        parallel_for((used + CH - 1) / CH, [&](uint64_t i) { memcpy(np + i * CH, g.p + i * CH, std::min(CH, used - i * CH)); }, true);
    }
    g_pool.put(g.p);
    g.p = np;
    g.cap = cap;
    return true;
}

// stage 1: pack the chunk, send it, align it, lay its results out (sizes come back through ev_tot)
scrg_status stage1(DeviceState* ds, Slot& sl, Call& c, uint64_t chunk)
{
    const scrg_host::Batch& b = *c.b;
    const uint64_t first = c.chunk_first[chunk], n = c.chunk_first[chunk + 1] - first;
    sl.first = first;
    sl.n = n;
    const int64_t t0 = now_ns();
    HTRY(ds, hipSetDevice(ds->device));

    // ---- layout of the chunk's sequences: read rows (mapping: one row per run of pairs with the same read), text rows.
    // Reverse-strand candidates (cand_reverse): the one-pair-per-lane kernels take the read's reverse complement on the DEVICE
    // from the one packed copy (scrg_params.stranded: the pair's descriptor carries the strand), so both strands of a read share
    // a row; the GenASM-row mappings get a row of their own, packed reverse-complemented here.
    const bool dev_strand = b.mapping && b.cand_reverse && c.p.lanes_per_pair == 1;
    std::vector<uint32_t> row(b.mapping ? n : 0);
    std::vector<uint64_t> row_pair;                  // a pair that owns each row (its read is the row's content)
    uint64_t max_read = 0, max_text = 0;
    {
        // (blocks of 16 k pairs in parallel: flags "a new row starts here", block counts, then the row numbers)
        const uint64_t BLK = 1u << 14, nb = (n + BLK - 1) / BLK;
        std::vector<uint64_t> blk_rows(nb + 1, 0), blk_mr(nb, 0), blk_mt(nb, 0);
        auto starts_row = [&](uint64_t i) -> bool {
            if (i == 0) return true;
            const uint64_t p = c.order[first + i], q = c.order[first + i - 1];
            return !(b.pair_read[q] == b.pair_read[p] &&
                     (dev_strand || (b.cand_reverse && b.cand_reverse[q]) == (b.cand_reverse && b.cand_reverse[p])));
        };
        parallel_for(nb, [&](uint64_t k) {
            uint64_t cnt = 0, mr = 0, mt = 0;
            for (uint64_t i = k * BLK; i < std::min(n, (k + 1) * BLK); i++) {
                const uint64_t p = c.order[first + i];
                if (b.mapping) {
                    cnt += starts_row(i) ? 1 : 0;
                    mr = std::max<uint64_t>(mr, b.read_lens[b.pair_read[p]]);
                } else {
                    mr = std::max<uint64_t>(mr, b.read_lens[p]);
                    mt = std::max<uint64_t>(mt, b.text_lens[p]);
                }
            }
            blk_rows[k + 1] = cnt;
            blk_mr[k] = mr;
            blk_mt[k] = mt;
        }, true, c.threads_per_worker);
        for (uint64_t k = 0; k < nb; k++) {
            blk_rows[k + 1] += blk_rows[k];
            max_read = std::max(max_read, blk_mr[k]);
            max_text = std::max(max_text, blk_mt[k]);
        }
        if (b.mapping) {
            row_pair.resize(blk_rows[nb]);
            parallel_for(nb, [&](uint64_t k) {
                uint64_t r = blk_rows[k];
                for (uint64_t i = k * BLK; i < std::min(n, (k + 1) * BLK); i++) {
                    if (starts_row(i)) row_pair[r++] = c.order[first + i];
                    row[i] = (uint32_t)(r - 1);
                }
            }, true, c.threads_per_worker);
        }
    }
    const uint64_t n_rows = b.mapping ? row_pair.size() : n;
    const uint64_t rw = std::max<uint64_t>(1, (max_read + 31) / 32), tw = b.mapping ? 0 : std::max<uint64_t>(1, (max_text + 31) / 32);
    const uint64_t r_groups = (n_rows + GROUP - 1) / GROUP, t_groups = b.mapping ? 0 : (n + GROUP - 1) / GROUP;
    const uint64_t read_words = r_groups * GROUP * rw, text_words = t_groups * GROUP * tw;
    const uint64_t seq_words = read_words + text_words + SEQ_PAD;
    if (seq_words > ds->slot_words) {
        ds->set_err("internal: chunk larger than its slot");
        return SCRG_ERR_INVALID_ARG;
    }
    HTRY(ds, sl.h_seq.ensure(seq_words * sizeof(uint64_t)));
    uint64_t* const h = static_cast<uint64_t*>(sl.h_seq.p);

    // ---- pack: one work item per group of 64 rows (the group's block of the interleaved layout is written by one thread;
    // the GenASM-row kernels, lanes_per_pair >= 4, read contiguous rows instead)
    const bool linear = c.p.lanes_per_pair != 1;
    const uint64_t rstride = linear ? 1 : GROUP;
    std::atomic<int> bad{0};
    parallel_for(r_groups + t_groups, [&](uint64_t g) {
        const bool is_text = g >= r_groups;
        const uint64_t gi = is_text ? g - r_groups : g, W = is_text ? tw : rw;
        uint64_t* const blockp = h + (is_text ? read_words : 0) + gi * GROUP * W;
        const uint64_t rows_here = std::min<uint64_t>(GROUP, (is_text ? n : n_rows) - gi * GROUP);
        bool bd = false;
        // the rows of this group: source, length, strand
        const char* rsrc[GROUP];
        uint64_t rlen[GROUP];
        bool rrev[GROUP];
        for (uint64_t l = 0; l < GROUP; l++) {
            rsrc[l] = nullptr;
            rlen[l] = 0;
            rrev[l] = false;
            if (l >= rows_here) continue;
            const uint64_t r = gi * GROUP + l;
            if (is_text) {
                const uint64_t p = c.order[first + r];
                rsrc[l] = b.texts[p];
                rlen[l] = b.text_lens[p];
            } else if (b.mapping) {
                const uint64_t p = row_pair[r], rd = b.pair_read[p];
                rsrc[l] = b.reads[rd];
                rlen[l] = b.read_lens[rd];
                rrev[l] = !dev_strand && b.cand_reverse && b.cand_reverse[p];
            } else {
                const uint64_t p = c.order[first + r];
                rsrc[l] = b.reads[p];
                rlen[l] = b.read_lens[p];
            }
        }
        // short rows (reads of a mapper: a million separate strings) are cache misses, not bandwidth: ask for all of the
        // group's lines before the first one is needed
        if (W <= 16)
            for (uint64_t l = 0; l < rows_here; l++)
                for (uint64_t o = 0; o < rlen[l]; o += 64) __builtin_prefetch(rsrc[l] + o, 0, 0);
        for (uint64_t l0 = 0; l0 < GROUP; l0 += 8) {
            bool plain = !linear && g_have_avx2;
            for (uint64_t l = l0; l < l0 + 8; l++) plain = plain && !rrev[l] && (rlen[l] == 0 || rsrc[l] != nullptr);
            if (plain) {
                static const char none[1] = {0};
                const char* s8[8];
                for (int k = 0; k < 8; k++) s8[k] = rsrc[l0 + k] ? rsrc[l0 + k] : none;
                bd |= pack_rows8_avx2(s8, rlen + l0, blockp + l0, W);
            } else {
                for (uint64_t l = l0; l < l0 + 8; l++)
                    bd |= pack_sequence(rsrc[l] ? rsrc[l] : "", rlen[l], rrev[l], linear ? blockp + l * W : blockp + l, rstride, W);
            }
        }
        if (bd) bad.store(1, std::memory_order_relaxed);
    }, true, c.threads_per_worker);
    for (uint64_t w = read_words + text_words; w < seq_words; w++) h[w] = 0;
    if (bad.load()) {
        ds->set_err("input contains characters other than ACGTacgt");
        return SCRG_ERR_BAD_BASE;
    }

    // ---- per-pair scalars: read length (+ text length | start in the genome and read row)
    const size_t meta_bytes = n * (b.mapping ? 16 : 8) + 64;
    HTRY(ds, sl.h_meta.ensure(meta_bytes));
    uint32_t* const m_rl = static_cast<uint32_t*>(sl.h_meta.p);
    uint32_t* const m_tl = m_rl + n;                          // pairwise: text length | mapping: read row
    uint64_t* const m_st = reinterpret_cast<uint64_t*>(static_cast<char*>(sl.h_meta.p) + ((8 * n + 15) & ~(size_t)15));
    parallel_for(n, [&](uint64_t i) {
        const uint64_t p = c.order[first + i];
        if (b.mapping) {
            m_rl[i] = (uint32_t)b.read_lens[b.pair_read[p]];
            m_tl[i] = row[i] | ((dev_strand && b.cand_reverse[p]) ? 0x80000000u : 0u);      // (bit 31: the reverse complement of the row's read)
            m_st[i] = b.cand_start[p];
        } else {
            m_rl[i] = (uint32_t)b.read_lens[p];
            m_tl[i] = (uint32_t)std::min<uint64_t>(b.text_lens[p], 0xffffffffull);
        }
    }, false, c.threads_per_worker);
    sl.t_pack_ns = now_ns() - t0;

    // ---- device side
    const uint64_t cap = (2 * max_read + 8 + 15) & ~(uint64_t)15;         // runs per slice (src/genasm_gpu.cu:906-911: 2 * read_len)
    const uint64_t slot_index = (uint64_t)(&sl - ds->slot);
    const uint64_t gpad = ds->genome_words ? ds->genome_words + SCRG_SEQ_PAD_WORDS : 0;
    const uint64_t base_word = gpad + slot_index * ds->slot_words;
    uint64_t* const d_seq = ds->d_seq.as<uint64_t>();
    HTRY(ds, sl.d_meta.ensure(meta_bytes));
    HTRY(ds, sl.d_desc.ensure(n * sizeof(scrg_pair_desc)));
    HTRY(ds, sl.d_slices.ensure(n * cap * sizeof(scrg_run)));
    // the four per-pair result arrays sit back to back in one buffer, in the layout of the host staging area: ONE read-back
    // (a read-back of 1-2 MB runs at ~12 GB/s; six of them per chunk were 0.85 ms of 1.3 ms per 250 k mapping pairs)
    const PerPairLayout lay(n);
    HTRY(ds, sl.d_ed.ensure(lay.bytes + 256));
    char* const d_pp = sl.d_ed.as<char>();
    int64_t* const d_ed = reinterpret_cast<int64_t*>(d_pp);
    uint32_t* const d_status = reinterpret_cast<uint32_t*>(d_pp + lay.o_st);
    uint64_t* const d_runoff = reinterpret_cast<uint64_t*>(d_pp + lay.o_ro);
    uint64_t* const d_textoff = reinterpret_cast<uint64_t*>(d_pp + lay.o_to);
    HTRY(ds, sl.d_nruns.ensure(n * 4));
    HTRY(ds, sl.d_cnt64.ensure(n * 8));
    HTRY(ds, sl.d_len64.ensure(n * 8));
    HTRY(ds, sl.d_tot.ensure(16));
    const size_t temp_bytes = scrg::host_scan_temp_bytes(n);
    HTRY(ds, sl.d_temp.ensure(temp_bytes + 256));
    HTRY(ds, sl.h_tot.ensure(16));
    HTRY(ds, hipMemcpyAsync(d_seq + base_word, h, seq_words * sizeof(uint64_t), hipMemcpyHostToDevice, sl.stream));
    HTRY(ds, hipMemcpyAsync(sl.d_meta.p, sl.h_meta.p, meta_bytes, hipMemcpyHostToDevice, sl.stream));
    scrg::HostDescArgs da{};
    da.n = n;
    da.desc = sl.d_desc.as<scrg_pair_desc>();
    da.read_len = sl.d_meta.as<uint32_t>();
    da.text_len = b.mapping ? nullptr : sl.d_meta.as<uint32_t>() + n;
    da.row = b.mapping ? sl.d_meta.as<uint32_t>() + n : nullptr;
    da.start = b.mapping ? reinterpret_cast<const uint64_t*>(sl.d_meta.as<char>() + ((8 * n + 15) & ~(size_t)15)) : nullptr;
    da.genome_len = ds->genome_len;
    da.read_base = base_word;
    da.read_words = rw;
    da.text_base = base_word + read_words;
    da.text_words = tw;
    da.cap = cap;
    da.linear = linear ? 1u : 0u;
    HTRY(ds, scrg::launch_build_desc(da, sl.stream));
    scrg_params pp = c.p;
    pp.read_stride_words = (int32_t)rstride;
    pp.text_stride_words = b.mapping ? 1 : (int32_t)rstride;
    pp.stranded = dev_strand ? 1 : 0;
    scrg_status s = scrg_align_device(sl.ctx, &pp, n, d_seq, sl.d_desc.as<scrg_pair_desc>(), sl.d_slices.as<scrg_run>(), d_ed,
                                      sl.d_nruns.as<uint32_t>(), d_status);
    if (s != SCRG_OK) {
        ds->set_err(scrg_last_error(sl.ctx));
        return s;
    }
    HTRY(ds, scrg::launch_result_layout(n, sl.d_desc.as<scrg_pair_desc>(), sl.d_slices.as<uint16_t>(), sl.d_nruns.as<uint32_t>(), d_ed, d_status,
                                        sl.d_cnt64.as<uint64_t>(), sl.d_len64.as<uint64_t>(), d_runoff, d_textoff, sl.d_tot.as<uint64_t>(),
                                        reinterpret_cast<uint32_t*>(d_pp + lay.o_wire), sl.d_temp.p, temp_bytes, c.want_text, ds->n_cus, sl.stream));
    HTRY(ds, hipMemcpyAsync(sl.h_tot.p, sl.d_tot.p, 16, hipMemcpyDeviceToHost, sl.stream));
    HTRY(ds, hipEventRecord(sl.ev_tot, sl.stream));
    return SCRG_OK;
}

// stage 2: the sizes are known — compact the runs, render the text, bring everything back
scrg_status stage2(DeviceState* ds, Slot& sl, Call& c, uint64_t chunk)
{
    HTRY(ds, hipSetDevice(ds->device));
    HTRY(ds, hipEventSynchronize(sl.ev_tot));
    const uint64_t* const tot = static_cast<const uint64_t*>(sl.h_tot.p);
    sl.tot_runs = tot[0];
    sl.tot_text = tot[1];
    {
        std::lock_guard<std::mutex> g(c.tot_mu);
        c.c_runs[chunk] = sl.tot_runs;
        c.c_text[chunk] = sl.tot_text;
        c.c_known[chunk] = 1;
    }
    c.tot_cv.notify_all();
    float ms = 0.f;
    if (scrg_last_kernel_ms(sl.ctx, &ms) == SCRG_OK) c.kernel_ns.fetch_add((int64_t)((double)ms * 1e6));
    c.pack_ns.fetch_add(sl.t_pack_ns);
    const uint64_t n = sl.n;
    // host staging of a chunk: [wire: ed 4n | run count + overflow bit 4n | text length 4n (absent without text)] [runs 2R (+pad)]
    // [text T] (PerPairLayout: host_runs() follows from wire_bytes); on the device the per-pair arrays are one buffer
    // ([ed 8n | status 4n | run_off 8n | text_off 8n | wire]) and runs + text another: two read-backs per chunk, the first of
    // the wire only
    const PerPairLayout lay(n);
    const size_t o_runs = lay.host_runs(), text_rel = (2 * sl.tot_runs + 15) & ~(size_t)15;
    const size_t o_text = o_runs + text_rel, total = o_text + sl.tot_text + 512;
    HTRY(ds, sl.h_out.ensure(total));
    char* const h = static_cast<char*>(sl.h_out.p);
    HTRY(ds, sl.d_dense.ensure(text_rel + (c.want_text ? sl.tot_text : 0) + 512));
    char* const d_pp = sl.d_ed.as<char>();
    uint64_t* const d_runoff = reinterpret_cast<uint64_t*>(d_pp + lay.o_ro);
    uint64_t* const d_textoff = reinterpret_cast<uint64_t*>(d_pp + lay.o_to);
    uint8_t* const d_text = sl.d_dense.as<uint8_t>() + text_rel;
    scrg_status s = scrg_compact_runs(sl.ctx, n, sl.d_desc.as<scrg_pair_desc>(), sl.d_slices.as<scrg_run>(), sl.d_nruns.as<uint32_t>(),
                                      d_runoff, sl.d_dense.as<scrg_run>());
    if (s != SCRG_OK) {
        ds->set_err(scrg_last_error(sl.ctx));
        return s;
    }
    if (c.want_text)
        HTRY(ds, scrg::launch_render_text(n, sl.d_dense.as<uint16_t>(), d_runoff, sl.d_cnt64.as<uint64_t>(), d_textoff, d_text, ds->n_cus, sl.stream));
    // (whole multiples of 256 bytes to page-aligned host addresses: the buffers have the slack)
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    HTRY(ds, hipMemcpyAsync(h, d_pp + lay.o_wire, up(c.want_text ? 12 * n : 8 * n), hipMemcpyDeviceToHost, sl.stream));
    if (c.want_runs && c.want_text && sl.tot_runs + sl.tot_text)
        HTRY(ds, hipMemcpyAsync(h + o_runs, sl.d_dense.p, up(text_rel + sl.tot_text), hipMemcpyDeviceToHost, sl.stream));
    else if (c.want_runs && sl.tot_runs)
        HTRY(ds, hipMemcpyAsync(h + o_runs, sl.d_dense.p, up(2 * sl.tot_runs), hipMemcpyDeviceToHost, sl.stream));
    else if (c.want_text && sl.tot_text)
        HTRY(ds, hipMemcpyAsync(h + o_text, d_text, up(sl.tot_text), hipMemcpyDeviceToHost, sl.stream));
    HTRY(ds, hipEventRecord(sl.ev_done, sl.stream));
    return SCRG_OK;
}

// stage 3: the chunk's results go to their place in the (issue order) result arrays
scrg_status stage3(DeviceState* ds, Slot& sl, Call& c, uint64_t chunk)
{
    HTRY(ds, hipSetDevice(ds->device));
    HTRY(ds, hipEventSynchronize(sl.ev_done));
    // bases: everything the chunks before this one produced
    uint64_t base_runs = 0, base_text = 0;
    {
        std::unique_lock<std::mutex> lk(c.tot_mu);
        c.tot_cv.wait(lk, [&] {
            if (c.failed()) return true;
            for (uint64_t k = 0; k < chunk; k++)
                if (!c.c_known[k]) return false;
            return true;
        });
        if (c.failed()) return (scrg_status)c.status.load();
        for (uint64_t k = 0; k < chunk; k++) {
            base_runs += c.c_runs[k];
            base_text += c.c_text[k];
        }
    }
    const uint64_t n = sl.n, first = sl.first;
    const PerPairLayout lay(n);
    const size_t o_runs = lay.host_runs();
    const size_t o_text = o_runs + ((2 * sl.tot_runs + 15) & ~(size_t)15);
    const char* const h = static_cast<const char*>(sl.h_out.p);
    const double per_pair_scale = 1.08 * (double)c.n / (double)std::max<uint64_t>(1, n);
    if (c.want_runs && !grow_to(c, c.runs, 2 * (base_runs + sl.tot_runs) + 2, (size_t)(2.0 * (double)sl.tot_runs * per_pair_scale) + 4096))
        return SCRG_ERR_OOM;
    if (c.want_text && !grow_to(c, c.text, base_text + sl.tot_text + 1, (size_t)((double)sl.tot_text * per_pair_scale) + 4096))
        return SCRG_ERR_OOM;
    {
        std::shared_lock<std::shared_mutex> lk(c.grow_mu);
        auto raise = [](std::atomic<size_t>& hi, size_t v) {
            size_t cur = hi.load();
            while (cur < v && !hi.compare_exchange_weak(cur, v)) {}
        };
        if (c.want_runs) raise(c.runs.hi, 2 * (base_runs + sl.tot_runs));
        if (c.want_text) raise(c.text.hi, base_text + sl.tot_text);
        const size_t CH = 1u << 19;
        const size_t rb = c.want_runs ? 2 * sl.tot_runs : 0, tb = c.want_text ? sl.tot_text : 0;
        const uint64_t n_r = (rb + CH - 1) / CH, n_t = (tb + CH - 1) / CH;
        parallel_for(n_r + n_t, [&](uint64_t i) {
            if (i < n_r) memcpy(c.runs.p + 2 * base_runs + i * CH, h + o_runs + i * CH, std::min(CH, rb - i * CH));
            else memcpy(c.text.p + base_text + (i - n_r) * CH, h + o_text + (i - n_r) * CH, std::min(CH, tb - (i - n_r) * CH));
        }, true, c.threads_per_worker);
    }
    // the wire (PerPairLayout): edit distances, run counts with the overflow flag, text lengths; the offsets are their prefix
    // sums — blocks of 16 k pairs: the blocks' sums side by side, their scan, then the offsets inside every block
    const uint32_t* const w_ed = reinterpret_cast<const uint32_t*>(h);
    const uint32_t* const w_cnt = w_ed + n;
    const uint32_t* const w_len = w_cnt + n;
    const uint64_t BLK = 1u << 14, nb = (n + BLK - 1) / BLK;
    std::vector<uint64_t> br(nb + 1, 0), bt(nb + 1, 0);
    parallel_for(nb, [&](uint64_t k) {
        uint64_t ar = 0, at = 0;
        for (uint64_t i = k * BLK; i < std::min(n, (k + 1) * BLK); i++) {
            ar += w_cnt[i] & 0x7fffffffu;
            if (c.want_text) at += w_len[i];
        }
        br[k + 1] = ar;
        bt[k + 1] = at;
    }, true, c.threads_per_worker);
    for (uint64_t k = 0; k < nb; k++) {
        br[k + 1] += br[k];
        bt[k + 1] += bt[k];
    }
    std::atomic<int> ovf{0};
    parallel_for(nb, [&](uint64_t k) {
        uint64_t ar = base_runs + br[k], at = base_text + bt[k];
        bool any = false;
        for (uint64_t i = k * BLK; i < std::min(n, (k + 1) * BLK); i++) {
            const uint32_t cw = w_cnt[i];
            c.iss_ed[first + i] = (int64_t)w_ed[i];
            c.iss_status[first + i] = (cw >> 31) ? (uint32_t)SCRG_ERR_CIGAR_OVERFLOW : (uint32_t)SCRG_OK;
            any |= (cw >> 31) != 0;
            c.iss_run_off[first + i] = ar;
            c.iss_text_off[first + i] = c.want_text ? at : 0;
            ar += cw & 0x7fffffffu;
            if (c.want_text) at += w_len[i];
        }
        if (any) ovf.store(1, std::memory_order_relaxed);
    }, true, c.threads_per_worker);
    if (br[nb] != sl.tot_runs || (c.want_text && bt[nb] != sl.tot_text)) {
        ds->set_err("internal: a chunk's per-pair sizes do not add up to its totals");
        return SCRG_ERR_HIP;
    }
    if (ovf.load()) c.any_overflow.store(1);
    return SCRG_OK;
}

// One device: a LAUNCH thread packs chunk i, sends it and starts its kernels; the sizes of an earlier chunk are looked at
// as soon as they have arrived (without waiting, unless the chunk is LAG2 launches behind) and its compaction, rendering
// and read-back are started; two COLLECT threads put finished chunks into the result arrays.  A slot is reused n_slots
// chunks later, once its previous chunk has been collected.  n_slots is 4 for long reads — the align kernel of a chunk of
// 10 kb reads takes ~2 ms however small the chunk is (~330 dependent window rounds), and HIP has four hardware queues —
// and 8 for short ones, where a chunk's kernels take ~0.5 ms and what limits the call is how soon a slot comes back
// (1 M x 4 mapping pairs: 21 ms with four slots, 16 ms with eight; the read-back alone is 11.5 ms at PCIe rate).
void worker(DeviceState* ds, Call* c, int dev_index, int n_dev)
{
    std::vector<uint64_t> mine;
    const uint64_t n_chunks = c->chunk_first.size() - 1;
    for (uint64_t k = dev_index; k < n_chunks; k += n_dev) mine.push_back(k);
    const size_t m = mine.size();
    const bool timing = getenv("SCRG_HOST_TIMING") != nullptr;
    // a call whose chunks all have a slot launches every one of them before it waits for the first (a chunk's kernel takes
    // ~2 ms for 10 kb reads however small the chunk); a longer call keeps one slot of slack between launch and collection
    const size_t NS = (size_t)c->n_slots;
    const size_t LAG2 = m <= NS ? NS - 1 : (NS > (size_t)NSLOT ? NS - 2 : LAG2_DEFAULT);      // launches a chunk's stage 2 may be behind
    const int64_t tw0 = now_ns();

    std::mutex mu;
    std::condition_variable cv;
    size_t ready = 0;            // chunks (local index) whose read-back has been enqueued: the collector may wait for them
    bool stop = false;
    auto fail_with = [&](scrg_status s) {
        const std::string e = ds->get_err();
        c->fail(s, e.empty() ? std::string(scrg_status_string(s)) : e);
        { std::lock_guard<std::mutex> g(mu); }      // (the launch thread may be between its predicate and its wait on cv)
        cv.notify_all();
    };

    // COLLECT threads: chunk i (local index) is taken by collector i mod NCOLLECT — a chunk's place in the result arrays
    // depends only on the totals of the chunks before it, which are known before it is read back
    std::vector<char> done(m, 0);
    auto collect = [&](size_t which) {
        // (no exception leaves a thread: a std::bad_alloc in here would otherwise be std::terminate)
        try {
        for (size_t i = which; i < m; i += NCOLLECT) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ready > i || stop; });
                if (ready <= i) return;
            }
            const int64_t ta = now_ns();
            scrg_status s = c->failed() ? (scrg_status)c->status.load() : stage3(ds, ds->slot[i % NS], *c, mine[i]);
            if (s != SCRG_OK) fail_with(s);
            if (timing)
                fprintf(stderr, "[scrooge_amd host] dev %d chunk %zu collected at %.3f ms: stage3 %.3f ms\n", dev_index, i, (now_ns() - tw0) / 1e6,
                        (now_ns() - ta) / 1e6);
            {
                std::lock_guard<std::mutex> g(mu);
                done[i] = 1;
            }
            cv.notify_all();
        }
        } catch (const std::bad_alloc&) {
            ds->set_err("host allocation failed while collecting results");
            fail_with(SCRG_ERR_OOM);
        } catch (...) {
            ds->set_err("unexpected exception while collecting results");
            fail_with(SCRG_ERR_INVALID_ARG);
        }
    };
    std::thread collectors[NCOLLECT];
    for (size_t k = 0; k < NCOLLECT; k++) collectors[k] = std::thread(collect, k);

    // LAUNCH thread.  A chunk's sizes are looked at (stage 2) as soon as they have arrived — asked for without waiting after
    // every launch — and waited for only when the chunk is LAG chunks behind the launches or nothing is left to launch.
    size_t next2 = 0;
    bool ok = true;
    auto do_stage2 = [&](size_t k) -> bool {
        const int64_t tb = now_ns();
        scrg_status s = stage2(ds, ds->slot[k % NS], *c, mine[k]);
        if (s != SCRG_OK) {
            fail_with(s);
            return false;
        }
        {
            std::lock_guard<std::mutex> g(mu);
            ready = k + 1;
        }
        cv.notify_all();
        if (timing) fprintf(stderr, "[scrooge_amd host] dev %d chunk %zu stage2 at %.3f ms: %.3f ms\n", dev_index, k, (tb - tw0) / 1e6, (now_ns() - tb) / 1e6);
        return true;
    };
    try {
    for (size_t i = 0; i < m && ok; i++) {
        if (c->failed()) break;
        const int64_t ta = now_ns();
        if (i >= NS) {               // the slot's previous chunk must have been collected (its stage 2 is behind us: LAG < NS)
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done[i - NS] || c->failed(); });
            if (c->failed()) break;
        }
        scrg_status s = stage1(ds, ds->slot[i % NS], *c, mine[i]);
        if (timing)
            fprintf(stderr, "[scrooge_amd host] dev %d chunk %zu stage1 at %.3f ms: %.3f ms (pack %.3f)\n", dev_index, i, (ta - tw0) / 1e6,
                    (now_ns() - ta) / 1e6, ds->slot[i % NS].t_pack_ns / 1e6);
        if (s != SCRG_OK) {
            fail_with(s);
            break;
        }
        while (ok && next2 <= i) {
            const bool must = i + 1 - next2 > LAG2;
            if (!must && hipEventQuery(ds->slot[next2 % NS].ev_tot) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            ok = do_stage2(next2++);
        }
    }
    while (ok && !c->failed() && next2 < m) ok = do_stage2(next2++);
    } catch (const std::bad_alloc&) {
        ds->set_err("host allocation failed while staging a chunk");
        fail_with(SCRG_ERR_OOM);
    } catch (...) {
        ds->set_err("unexpected exception while staging a chunk");
        fail_with(SCRG_ERR_INVALID_ARG);
    }
    {
        std::lock_guard<std::mutex> g(mu);
        stop = true;
    }
    cv.notify_all();
    for (auto& t : collectors) t.join();
    (void)hipSetDevice(ds->device);
    for (Slot& sl : ds->slot) (void)hipStreamSynchronize(sl.stream);        // nothing of this call is in flight when it returns
}

// The plan of a call: the issue order (longest read first, src/tests.cu:375-377; stable, so a batch that is sorted already
// keeps its order) and the chunks it is cut into.  Chunk k is processed by device state k mod n_states.
void make_plan(Call& c, const scrg_params& resolved, int n_states, bool* sorted_issue_out)
{
    const scrg_host::Batch& b = *c.b;
    const uint64_t n = c.n;
    c.order.resize(n);
    std::atomic<int> unsorted{0}, long_reads{0};
    {
        const uint64_t BLK = 1u << 16, nb = (n + BLK - 1) / BLK;
        parallel_for(nb, [&](uint64_t blk) {
            bool ok = true, lng = false;
            for (uint64_t k = blk * BLK; k < std::min(n, (blk + 1) * BLK); k++) {
                c.order[k] = (uint32_t)k;
                const uint64_t len = b.read_lens[read_of(c, k)];
                if (k && b.read_lens[read_of(c, k - 1)] < len) ok = false;
                if (len > SHORT_READS) lng = true;
            }
            if (!ok) unsorted.store(1, std::memory_order_relaxed);
            if (lng) long_reads.store(1, std::memory_order_relaxed);
        }, true);
    }
    c.n_slots = long_reads.load() ? NSLOT : MAXSLOT;
    c.identity = true;
    if (resolved.sort_by_length && unsorted.load()) {
        std::stable_sort(c.order.begin(), c.order.end(),
                         [&](uint32_t x, uint32_t y) { return b.read_lens[read_of(c, x)] > b.read_lens[read_of(c, y)]; });
        c.identity = false;
    }
    // chunks in issue order: whole groups of 64 pairs.  A chunk's align kernel takes ~2 ms for 10 kb reads however few
    // pairs it has, and only NSLOT of them run side by side: a small batch is cut into NSLOT chunks per device, a large one
    // into chunks of at most 32 MB of packed sequence or 256 k pairs.  In sorted issue order the first pair of a chunk has
    // its longest read.
    const bool sorted_issue = resolved.sort_by_length || !unsorted.load();
    // A device's LAST chunk is smaller (0.6 of the others): after the last launch nothing overlaps that chunk's kernel,
    // read-back and collection any more, so it should be the cheapest one (20 k x 10 kb pairs: 6.9 -> 6.6 ms).
    const uint64_t max_words = 4u << 20, want_chunks = (uint64_t)n_states * c.n_slots;
    const uint64_t full_chunks = want_chunks - (uint64_t)n_states;                       // chunks of the full size
    const uint64_t per_full = (uint64_t)((double)n / ((double)full_chunks + 0.6 * (double)n_states));
    const uint64_t target_full = std::min<uint64_t>(1u << 18, std::max<uint64_t>(512, (per_full + GROUP - 1) / GROUP * GROUP));
    const uint64_t rest = n > full_chunks * target_full ? n - full_chunks * target_full : 0;
    const uint64_t target_last = std::min<uint64_t>(1u << 18, std::max<uint64_t>(512, (rest / (uint64_t)n_states + GROUP - 1) / GROUP * GROUP));
    c.chunk_first.clear();
    c.chunk_first.push_back(0);
    uint64_t k = 0;
    while (k < n) {
        const uint64_t target = c.chunk_first.size() - 1 < full_chunks ? target_full : target_last;
        uint64_t e;
        if (sorted_issue && b.mapping) {
            const uint64_t w = std::max<uint64_t>(1, (b.read_lens[read_of(c, c.order[k])] + 31) / 32);
            e = k + std::max<uint64_t>(GROUP, std::min(target, max_words / w / GROUP * GROUP));
        } else {
            uint64_t words = 0;
            e = k;
            while (e < n) {
                const uint64_t p = c.order[e];
                const uint64_t w = (b.read_lens[read_of(c, p)] + 31) / 32 + (b.mapping ? 0 : (b.text_lens[p] + 31) / 32);
                if (e > k && (e - k) % GROUP == 0 && (e - k >= target || words + w > max_words)) break;
                words += w;
                e++;
            }
        }
        e = std::min(e, n);
        c.chunk_first.push_back(e);
        k = e;
    }
    if (sorted_issue_out) *sorted_issue_out = sorted_issue;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// interface (scrg_internal.h)
// ---------------------------------------------------------------------------------------------------------------
namespace scrg_host {

// The issue order and the chunk boundaries a call with these lengths would use (no GPU involved): what the CPU tests check.
scrg_status plan(const scrg_params& resolved, int n_states, const Batch& b, uint32_t* order_out, uint64_t* chunk_first_out,
                 uint64_t chunk_cap, uint64_t* n_chunks_out)
{
    if (n_states < 1 || !n_chunks_out) return SCRG_ERR_INVALID_ARG;
    Call c;
    c.b = &b;
    c.p = resolved;
    c.n = b.n_pairs;
    make_plan(c, resolved, n_states, nullptr);
    const uint64_t nc = c.chunk_first.size() - 1;
    *n_chunks_out = nc;
    if (order_out) memcpy(order_out, c.order.data(), c.n * sizeof(uint32_t));
    if (chunk_first_out) {
        if (chunk_cap < nc + 1) return SCRG_ERR_CIGAR_OVERFLOW;
        memcpy(chunk_first_out, c.chunk_first.data(), (nc + 1) * sizeof(uint64_t));
    }
    return SCRG_OK;
}

// One sequence, ASCII -> planar 2-bit, with the packer the host entry points use (AVX2 or scalar by what the CPU has)
bool pack_planar_host(const char* ascii, uint64_t n_bases, uint64_t* planar, uint64_t stride_words, uint64_t n_words)
{
    return pack_sequence(ascii ? ascii : "", n_bases, false, planar, stride_words ? stride_words : 1, n_words);
}

void* state_create(int device)
{
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return nullptr;
    DeviceState* ds = new (std::nothrow) DeviceState();
    if (!ds) return nullptr;
    ds->device = device;
    ds->n_cus = prop.multiProcessorCount;
    // three streams of different priorities: HIP maps them to different hardware queues, so the kernels of consecutive
    // chunks share the GPU instead of queueing behind each other
    const int prio[MAXSLOT] = {0, -1, 1, 0, 0, -1, 1, 0};
    for (int k = 0; k < MAXSLOT; k++) {
        Slot& sl = ds->slot[k];
        if (hipStreamCreateWithPriority(&sl.stream, hipStreamNonBlocking, prio[k]) != hipSuccess || hipEventCreateWithFlags(&sl.ev_tot, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) != hipSuccess || scrg_int::ctx_create_internal(device, &sl.ctx) != SCRG_OK ||
            scrg_ctx_set_stream(sl.ctx, sl.stream) != SCRG_OK) {
            state_free(ds);
            return nullptr;
        }
    }
    return ds;
}

void state_free(void* state)
{
    DeviceState* ds = static_cast<DeviceState*>(state);
    if (!ds) return;
    (void)hipSetDevice(ds->device);
    (void)hipDeviceSynchronize();
    for (Slot& sl : ds->slot) {
        if (sl.ctx) scrg_ctx_destroy(sl.ctx);
        if (sl.ev_tot) (void)hipEventDestroy(sl.ev_tot);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
        for (HostPinned* hp : {&sl.h_seq, &sl.h_meta, &sl.h_out, &sl.h_tot}) hp->release();
        for (DevBuf* db : {&sl.d_meta, &sl.d_desc, &sl.d_slices, &sl.d_ed, &sl.d_nruns, &sl.d_cnt64, &sl.d_len64, &sl.d_tot, &sl.d_dense, &sl.d_temp})
            db->release();
    }
    ds->d_seq.release();
    delete ds;
    {   // the genome staging buffer is not worth keeping pinned for a process that is letting go of its device states
        std::lock_guard<std::mutex> g(g_genome_staging.mu);
        g_genome_staging.release();
    }
}

scrg_status genome_set(void* state, const char* genome, uint64_t genome_len, std::string* err)
{
    DeviceState* ds = static_cast<DeviceState*>(state);
    if (!ds) return SCRG_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> g(ds->mu);
    ds->set_err("");
    scrg_status s = pack_genome(ds, genome, genome_len);
    if (s != SCRG_OK && err) *err = ds->get_err();
    return s;
}

void genome_clear(void* state)
{
    DeviceState* ds = static_cast<DeviceState*>(state);
    if (!ds) return;
    std::lock_guard<std::mutex> g(ds->mu);
    ds->genome_ok = false;
}

bool genome_resident(void* state, uint64_t* genome_len)
{
    DeviceState* ds = static_cast<DeviceState*>(state);
    if (!ds || !ds->genome_ok) return false;
    if (genome_len) *genome_len = ds->genome_len;
    return true;
}

scrg_status align(void* const* states, int n_states, const scrg_params& resolved, const Batch& b, scrg_result** out, std::string* err)
{
    const int64_t t_begin = now_ns();
    auto set_err = [&](const std::string& e) { if (err) *err = e; };
    if (!out || n_states < 1 || !states) return SCRG_ERR_INVALID_ARG;
    *out = nullptr;
    for (int d = 0; d < n_states; d++)
        if (!states[d]) return SCRG_ERR_NO_DEVICE;
    const uint64_t n = b.n_pairs;

    scrg_result* r = static_cast<scrg_result*>(calloc(1, sizeof(scrg_result)));
    if (!r) return SCRG_ERR_OOM;
    r->n_pairs = n;
    Call c;
    c.b = &b;
    c.p = resolved;
    c.n = n;
    c.want_runs = resolved.outputs != SCRG_OUT_TEXT;
    c.want_text = resolved.outputs != SCRG_OUT_RUNS;
    {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        c.threads_per_worker = std::max(1u, std::min(16u, hw / (unsigned)n_states));
    }
    auto bail = [&](scrg_status s, const std::string& e) {
        set_err(e);
        g_pool.put(c.runs.p);
        g_pool.put(c.text.p);
        g_pool.put(c.iss_run_off);
        g_pool.put(c.iss_text_off);
        g_pool.put(c.iss_ed);
        g_pool.put(c.iss_status);
        scrg_result_free(r);
        return s;
    };

    bool sorted_issue = false;
    make_plan(c, resolved, n_states, &sorted_issue);
    const uint64_t n_chunks = c.chunk_first.size() - 1;
    c.c_runs.assign(n_chunks, 0);
    c.c_text.assign(n_chunks, 0);
    c.c_known.assign(n_chunks, 0);

    // ---- result arrays in issue order
    c.iss_ed = static_cast<int64_t*>(g_pool.get((n + 1) * 8, n < 4096));
    c.iss_status = static_cast<uint32_t*>(g_pool.get((n + 1) * 4, n < 4096));
    c.iss_run_off = static_cast<uint64_t*>(g_pool.get((n + 1) * 8, n < 4096));
    c.iss_text_off = static_cast<uint64_t*>(g_pool.get((n + 1) * 8, n < 4096));
    if (!c.iss_ed || !c.iss_status || !c.iss_run_off || !c.iss_text_off) return bail(SCRG_ERR_OOM, "result arrays");

    // ---- device side: lock the states, size their sequence arrays (the largest chunk decides), genome
    std::vector<DeviceState*> ds(n_states);
    for (int d = 0; d < n_states; d++) ds[d] = static_cast<DeviceState*>(states[d]);
    std::vector<std::unique_lock<std::mutex>> locks;
    {
        std::vector<DeviceState*> uniq(ds.begin(), ds.end());
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        if (uniq.size() != ds.size()) return bail(SCRG_ERR_INVALID_ARG, "a device state may be used once per call");
        for (DeviceState* u : uniq) locks.emplace_back(u->mu);
    }
    uint64_t slot_words = 0;
    {
        std::vector<uint64_t> cw(n_chunks, 0);
        parallel_for(n_chunks, [&](uint64_t k) {
            // upper bound of the chunk's padded layout: rows x the chunk's longest read / text
            uint64_t mr = 0, mt = 0;
            const uint64_t a = c.chunk_first[k], e = c.chunk_first[k + 1];
            if (sorted_issue && b.mapping) mr = b.read_lens[read_of(c, c.order[a])];
            else
                for (uint64_t i = a; i < e; i++) {
                    const uint64_t p = c.order[i];
                    mr = std::max<uint64_t>(mr, b.read_lens[read_of(c, p)]);
                    if (!b.mapping) mt = std::max<uint64_t>(mt, b.text_lens[p]);
                }
            const uint64_t rows = (e - a + GROUP - 1) / GROUP * GROUP;
            cw[k] = rows * std::max<uint64_t>(1, (mr + 31) / 32) + (b.mapping ? 0 : rows * std::max<uint64_t>(1, (mt + 31) / 32)) + SEQ_PAD;
        }, true);
        for (uint64_t w : cw) slot_words = std::max(slot_words, w);
    }
    for (int d = 0; d < n_states; d++) ds[d]->set_err("");
    if (b.mapping && b.genome) {             // packed once, copied to every device side by side
        const scrg_status s = stage_genome(ds.data(), n_states, b.genome, b.genome_len);
        if (s != SCRG_OK) return bail(s, ds[0]->get_err());
    }
    for (int d = 0; d < n_states; d++) {
        scrg_status s = SCRG_OK;
        if (b.mapping && !ds[d]->genome_ok) {
            ds[d]->set_err("no resident genome: call scrg_genome_set first");
            s = SCRG_ERR_INVALID_ARG;
        }
        if (s == SCRG_OK) s = ensure_seq(ds[d], ds[d]->genome_words, slot_words);
        if (s != SCRG_OK) return bail(s, ds[d]->get_err());
    }
    if (b.mapping) {
        const uint64_t glen = ds[0]->genome_len;
        std::atomic<int> past{0};
        parallel_for(n, [&](uint64_t p) { if (b.cand_start[p] > glen) past.store(1, std::memory_order_relaxed); });
        if (past.load()) return bail(SCRG_ERR_INVALID_ARG, "candidate past end of genome");
    }

    const int64_t t_setup = now_ns();
    // ---- run: one worker per device
    if (n) {
        if (n_states == 1) worker(ds[0], &c, 0, 1);
        else {
            std::vector<std::thread> th;
            for (int d = 0; d < n_states; d++) th.emplace_back(worker, ds[d], &c, d, n_states);
            for (auto& t : th) t.join();
        }
    }
    if (c.failed()) return bail((scrg_status)c.status.load(), c.err);
    if (getenv("SCRG_HOST_TIMING"))
        fprintf(stderr, "[scrooge_amd host] %llu pairs, %llu chunks, %d device state(s): set-up %.3f ms, pipeline %.3f ms\n", (unsigned long long)n,
                (unsigned long long)n_chunks, n_states, (t_setup - t_begin) / 1e6, (now_ns() - t_setup) / 1e6);

    uint64_t total_runs = 0, total_text = 0;
    for (uint64_t k = 0; k < n_chunks; k++) {
        total_runs += c.c_runs[k];
        total_text += c.c_text[k];
    }
    c.iss_run_off[n] = total_runs;
    c.iss_text_off[n] = total_text;
    if (!c.runs.p) c.runs.p = static_cast<char*>(g_pool.get(16, true));
    if (!c.text.p) c.text.p = static_cast<char*>(g_pool.get(16, true));

    // ---- caller order
    if (c.identity) {
        r->edit_distance = c.iss_ed;
        r->pair_status = c.iss_status;
        r->run_offset = c.iss_run_off;
        r->cigar_offset = c.iss_text_off;
        r->runs = reinterpret_cast<scrg_run*>(c.runs.p);
        r->cigar_text = c.text.p;
        r->edit_distance[n] = 0;
        r->pair_status[n] = 0;
        c.iss_ed = nullptr; c.iss_status = nullptr; c.iss_run_off = nullptr; c.iss_text_off = nullptr;
        c.runs.p = nullptr; c.text.p = nullptr;
    } else {
        r->edit_distance = static_cast<int64_t*>(g_pool.get((n + 1) * 8, false));
        r->pair_status = static_cast<uint32_t*>(g_pool.get((n + 1) * 4, false));
        r->run_offset = static_cast<uint64_t*>(g_pool.get((n + 1) * 8, false));
        r->cigar_offset = static_cast<uint64_t*>(g_pool.get((n + 1) * 8, false));
        r->runs = static_cast<scrg_run*>(g_pool.get(2 * total_runs + 16, false));
        r->cigar_text = static_cast<char*>(g_pool.get(total_text + 16, false));
        if (!r->edit_distance || !r->pair_status || !r->run_offset || !r->cigar_offset || !r->runs || !r->cigar_text)
            return bail(SCRG_ERR_OOM, "result arrays");
        // per-pair sizes into caller order, prefix sums (two levels, blocks of 64 k pairs in parallel), then the gather
        parallel_for(n, [&](uint64_t k) {
            const uint64_t p = c.order[k];
            r->run_offset[p] = c.iss_run_off[k + 1] - c.iss_run_off[k];
            r->cigar_offset[p] = c.iss_text_off[k + 1] - c.iss_text_off[k];
            r->edit_distance[p] = c.iss_ed[k];
            r->pair_status[p] = c.iss_status[k];
        });
        const uint64_t BLK = 1u << 16, nb = (n + BLK - 1) / BLK;
        std::vector<uint64_t> bs_r(nb + 1, 0), bs_t(nb + 1, 0);
        parallel_for(nb, [&](uint64_t blk) {
            uint64_t ar = 0, at = 0;
            for (uint64_t i = blk * BLK; i < std::min(n, (blk + 1) * BLK); i++) {
                const uint64_t cr = r->run_offset[i], ct = r->cigar_offset[i];
                r->run_offset[i] = ar;
                r->cigar_offset[i] = at;
                ar += cr;
                at += ct;
            }
            bs_r[blk + 1] = ar;
            bs_t[blk + 1] = at;
        }, true);
        for (uint64_t blk = 0; blk < nb; blk++) { bs_r[blk + 1] += bs_r[blk]; bs_t[blk + 1] += bs_t[blk]; }
        parallel_for(nb, [&](uint64_t blk) {
            for (uint64_t i = blk * BLK; i < std::min(n, (blk + 1) * BLK); i++) {
                r->run_offset[i] += bs_r[blk];
                r->cigar_offset[i] += bs_t[blk];
            }
        }, true);
        r->run_offset[n] = total_runs;
        r->cigar_offset[n] = total_text;
        r->edit_distance[n] = 0;
        r->pair_status[n] = 0;
        parallel_for(n, [&](uint64_t k) {
            const uint64_t p = c.order[k];
            if (c.want_runs)
                memcpy(reinterpret_cast<char*>(r->runs) + 2 * r->run_offset[p], c.runs.p + 2 * c.iss_run_off[k], 2 * (c.iss_run_off[k + 1] - c.iss_run_off[k]));
            if (c.want_text)
                memcpy(r->cigar_text + r->cigar_offset[p], c.text.p + c.iss_text_off[k], c.iss_text_off[k + 1] - c.iss_text_off[k]);
        });
        g_pool.put(c.runs.p); g_pool.put(c.text.p);
        g_pool.put(c.iss_ed); g_pool.put(c.iss_status); g_pool.put(c.iss_run_off); g_pool.put(c.iss_text_off);
        c.runs.p = c.text.p = nullptr;
        c.iss_ed = nullptr; c.iss_status = nullptr; c.iss_run_off = nullptr; c.iss_text_off = nullptr;
    }
    if (!c.want_runs) memset(r->run_offset, 0, (n + 1) * sizeof(uint64_t));
    if (!c.want_text) memset(r->cigar_offset, 0, (n + 1) * sizeof(uint64_t));
    r->kernel_ns = c.kernel_ns.load();
    if (scrg_get_log() && r->kernel_ns > 0)   // the reference's log line, src/genasm_gpu.cu:949-951 (kernel time: the sum over the chunks' launches)
        fprintf(stderr, "core algorithm ran at %lld aligns/second\n", (long long)((double)n * 1e9 / (double)r->kernel_ns));
    r->pack_ns = c.pack_ns.load();
    r->total_ns = now_ns() - t_begin;
    if (getenv("SCRG_HOST_TIMING"))
        fprintf(stderr, "[scrooge_amd host] total %.3f ms (results in %s order)\n", r->total_ns / 1e6, c.identity ? "issue = caller" : "caller (permuted)");
    *out = r;
    if (c.any_overflow.load()) {
        set_err("at least one pair overflowed its CIGAR slice (see pair_status)");
        return SCRG_ERR_CIGAR_OVERFLOW;
    }
    return SCRG_OK;
}

}  // namespace scrg_host
