// scrg_internal.h — small host-side helpers shared by the C-ABI translation units (scrg_api.cpp: handles and the
// device-pointer layer; scrg_host.cpp: the pipelined host-pointer entry points).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <sys/mman.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/scrooge_amd.h"

namespace scrg_int {

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

inline int64_t now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A pool of worker threads shared by every parallel_for of the process (threads are not created per call: a host entry
// point runs a dozen short loops per chunk).  parallel_for() cuts [0, n) into pieces that the caller and up to
// max_threads - 1 pool threads take from a shared counter; it returns when all of them are done.  Callable from several
// threads at once (the workers of a multi-GPU call): jobs queue up.
class ThreadPool {
public:
    static ThreadPool& get()
    {
        static ThreadPool pool;
        return pool;
    }
    unsigned size() const { return (unsigned)threads_.size(); }

    template <typename F> void run(uint64_t n, uint64_t chunk, unsigned helpers, F& f)
    {
        struct Job {
            std::atomic<uint64_t> next{0};
            std::atomic<unsigned> active{0};
            uint64_t n, chunk;
            F* f;
        } job;
        job.n = n;
        job.chunk = chunk;
        job.f = &f;
        auto body = [](void* pj) {
            Job* j = static_cast<Job*>(pj);
            for (;;) {
                const uint64_t b = j->next.fetch_add(j->chunk);
                if (b >= j->n) break;
                const uint64_t e = std::min(j->n, b + j->chunk);
                for (uint64_t i = b; i < e; i++) (*j->f)(i);
            }
        };
        helpers = std::min<unsigned>(helpers, size());
        job.active.store(helpers);
        {
            std::lock_guard<std::mutex> g(mu_);
            for (unsigned k = 0; k < helpers; k++) tasks_.push_back(Task{body, &job, &job.active});
        }
        if (helpers) cv_.notify_all();
        body(&job);
        // the helpers may still be inside their last piece (or not have started: then they find nothing to do)
        while (job.active.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }

private:
    struct Task {
        void (*fn)(void*);
        void* arg;
        std::atomic<unsigned>* active;
    };
    ThreadPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned nt = hw ? std::min(hw, 32u) : 4u;
        for (unsigned k = 0; k + 1 < nt; k++) threads_.emplace_back([this] { loop(); });
    }
    ~ThreadPool()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void loop()
    {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !tasks_.empty(); });
                if (stop_ && tasks_.empty()) return;
                t = tasks_.front();
                tasks_.pop_front();
            }
            t.fn(t.arg);
            t.active->fetch_sub(1, std::memory_order_release);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Task> tasks_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
};

// (heavy = true: every index is a block of work — a few of them are worth the threads)
template <typename F> void parallel_for(uint64_t n, F f, bool heavy = false, unsigned max_threads = 16)
{
    unsigned nt = std::min(ThreadPool::get().size() + 1, std::max(1u, max_threads));
    if (heavy) nt = (unsigned)std::min<uint64_t>(nt, n);
    if (n < (heavy ? 2u : 64u) || nt <= 1) {
        for (uint64_t i = 0; i < n; i++) f(i);
        return;
    }
    const uint64_t chunk = std::max<uint64_t>(1, n / (nt * 16));
    ThreadPool::get().run(n, chunk, nt - 1, f);
}

// Host memory the copy engines read and write directly.  Large buffers are ordinary memory — 2 MB aligned, huge-page advised,
// faulted in by all threads — made known to HIP with hipHostRegister: copies to and from it run at the same rate as with
// hipHostMalloc memory (scripts/ubench/d2h_rate.hip: 56 GB/s both ways at any offset and size), and getting it costs a fraction
// (320 MB: 65 ms from hipHostMalloc; 20 ms of page faults spread over the threads + 0.6 ms to register) — what the FIRST call of a
// process pays for its staging areas.  Small ones (and any the registration refuses) come from hipHostMalloc.
struct HostPinned {
    void* p = nullptr;
    size_t cap = 0;
    bool registered = false;
    static constexpr size_t kRegisterMin = 4u << 20, kHugePage = 2u << 20;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        release();
        if (bytes >= kRegisterMin) {
            const size_t total = (bytes + kHugePage - 1) / kHugePage * kHugePage;
            char* const q = static_cast<char*>(aligned_alloc(kHugePage, total));
            if (q) {
                (void)madvise(q, total, MADV_HUGEPAGE);
                const size_t pages = total / 4096, PER = 512;                  // 2 MB per work item
                parallel_for((pages + PER - 1) / PER, [&](uint64_t i) {
                    for (size_t k = i * PER; k < std::min(pages, (i + 1) * PER); k++) *reinterpret_cast<volatile char*>(q + k * 4096) = 0;
                }, true);
                if (hipHostRegister(q, total, hipHostRegisterPortable) == hipSuccess) {
                    p = q;
                    cap = total;
                    registered = true;
                    return hipSuccess;
                }
                (void)hipGetLastError();
                free(q);
            }
        }
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e == hipSuccess) cap = bytes;
        else p = nullptr;
        return e;
    }
    void release()
    {
        if (p) {
            if (registered) {
                (void)hipHostUnregister(p);
                free(p);
            } else {
                (void)hipHostFree(p);
            }
        }
        p = nullptr;
        cap = 0;
        registered = false;
    }
};

// Result arrays are recycled: a batch of millions of pairs returns hundreds of MB, and freshly mapped pages cost
// more (first-touch faults) than filling them.  scrg_result_free() parks the big arrays here, the next call of
// similar size takes them back.  At most 12 blocks / 2 GB are kept, the most recently returned ones (the oldest are freed).
struct ResultPool {
    struct Block { void* p; size_t cap; };
    std::mutex mu;
    std::vector<Block> blocks;
    size_t held = 0;
    static constexpr size_t kMinPooled = 1u << 20, kMaxHeld = 2ull << 30, kMaxBlocks = 12, kHuge = 2u << 20;

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
            if (cap >= (8u << 20)) {
                // large blocks: 2 MB aligned and marked for transparent huge pages — first-touch faults of a fresh
                // 600 MB result cost more than filling it (150 k faults of 4 KB)
                const size_t total = (cap + sizeof(size_t) * 2 + kHuge - 1) / kHuge * kHuge;
                p = aligned_alloc(kHuge, total);
                if (p) (void)madvise(p, total, MADV_HUGEPAGE);
            } else {
                p = malloc(cap + sizeof(size_t) * 2);
            }
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
        if (cap < kMinPooled || cap > kMaxHeld) {
            free(p);
            return;
        }
        // the newest block stays, the oldest go: a caller that moves from big batches to small ones must not find the
        // pool full of blocks it no longer asks for (a fresh 90 MB array costs 20 ms of first-touch faults per call)
        std::vector<void*> evicted;
        {
            std::lock_guard<std::mutex> g(mu);
            blocks.push_back({p, cap});
            held += cap;
            while (blocks.size() > kMaxBlocks || held > kMaxHeld) {
                held -= blocks.front().cap;
                evicted.push_back(blocks.front().p);
                blocks.erase(blocks.begin());
            }
        }
        for (void* q : evicted) free(q);
    }
    void trim()
    {
        std::lock_guard<std::mutex> g(mu);
        for (Block& b : blocks) free(b.p);
        blocks.clear();
        held = 0;
    }
};
inline ResultPool g_pool;

// A handle for the library's own use (the slots of the host path): like scrg_ctx_create, but not counted among the
// caller's handles — "destroying the last handle trims the result pool" is about the handles the caller made.
scrg_status ctx_create_internal(int device, scrg_ctx** out);


}  // namespace scrg_int

// ---- the pipelined host-pointer path (scrg_host.cpp) behind scrg_align_pairs / scrg_align_mapping* ----
namespace scrg_host {

struct Batch {                       // one call, caller order
    uint64_t n_pairs = 0;
    bool mapping = false;
    // pairwise: pair p = (texts[p], reads[p])
    const char* const* texts = nullptr;
    const uint64_t* text_lens = nullptr;
    // both: reads; mapping: pair p aligns reads[pair_read[p]] against the genome from cand_start[p]
    const char* const* reads = nullptr;
    const uint64_t* read_lens = nullptr;
    const uint64_t* cand_start = nullptr;
    const uint8_t* cand_reverse = nullptr;     // may be null
    const uint32_t* pair_read = nullptr;
    uint64_t n_reads = 0;
    // mapping: the genome (null: the one made resident by genome_set on every device state used)
    const char* genome = nullptr;
    uint64_t genome_len = 0;
};

bool pack_planar_host(const char* ascii, uint64_t n_bases, uint64_t* planar, uint64_t stride_words, uint64_t n_words);     // true: a byte other than ACGTacgt
void* state_create(int device);                 // per-device buffers, streams and handles of the path; nullptr if that fails
void state_free(void* state);
scrg_status genome_set(void* state, const char* genome, uint64_t genome_len, std::string* err);
void genome_clear(void* state);
bool genome_resident(void* state, uint64_t* genome_len);
scrg_status plan(const scrg_params& resolved, int n_states, const Batch& b, uint32_t* order_out, uint64_t* chunk_first_out,
                 uint64_t chunk_cap, uint64_t* n_chunks_out);
// Aligns the batch on the given device states (one or several GPUs), results in caller order.
scrg_status align(void* const* states, int n_states, const scrg_params& resolved, const Batch& b, scrg_result** out, std::string* err);

}  // namespace scrg_host
