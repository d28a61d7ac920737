// pipeline.hip — the device pipeline of the assembly path (gfx950, wave64): device-side views, the
// kernel headers, the device/pinned memory pools and the per-handle Pipeline<W> that sequences them.
//
// Reference rows replaced (the Rust lives in rust/sparrowhawk-asm, NOT IN TREE,
// so citations are to the observable phases in www/src/components/pages/AssemblyPage.vue):
//   a4/a5  preprocess:*:loop      -> count_part.h   k_partition (extract + rolling ntHash + minimiser runs),
//                                                   k_count_partitions (LDS counting), k_ovf_scatter / k_count_buckets
//   a6/a8  histo, :filtering      -> fused into k_count_partitions (k_compact_rows when the fit raises the threshold)
//   a10    assembly:create_graph  -> graph_part.h   k_gp_*, k_graph_local, k_graph_remote
//   a11    assembly:correct_graph -> graph_part.h   k_tip_*, k_bubble, k_apply_removed
//   a12    assembly:collapse_graph-> collapse.h     k_succ_split, k_walk_frags, k_rank_*, k_emit
//   (count_global.h: the first, global-atomic counting path, kept as a cross-check)
// Integer/hash work only: no MFMA anywhere on this path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <thread>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "kmer.h"
#include "pipeline.h"
#include "shard_comm.h"
#include "unitig_graph.h"

namespace shk {

#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess) {                                                           \
            err = std::string(#call) + ": " + hipGetErrorString(_e);                      \
            return -5;                                                                    \
        }                                                                                 \
    } while (0)
// host wait on the pipeline's stream (inside Pipeline<W> methods with `err` in scope): during a collective call of more
// than one rank the wait is the shard layer's watchdog (shard_comm.h: comm_stream_wait) — a peer that left or died cannot
// hold this rank for ever
#define WAIT_STREAM()                                                                     \
    do {                                                                                  \
        n_host_waits_++;                                                                  \
        if (wd_comm_) { if (int _rc = comm_stream_wait(wd_comm_, stream_, err)) return _rc; } \
        else HIPCHK(stream_wait(stream_));                                                \
    } while (0)

static constexpr uint32_t NIL = 0xFFFFFFFFu;
// device-side counters and flags of a pipeline (ctl_): 0-2 count / graph build, 3-6 correction rounds, 5-10 collapse, 8-9
// pass 1, 11-12 / 14 pass 2, 13 shard layer, 16-17 what the FIRST correction round removed (tips, bubbles: read back with
// the collapse's first counters — the round costs no host round trip of its own)
static constexpr int CTL_WORDS = 32;
static constexpr uint64_t EMPTY64 = ~0ull;
static constexpr int MAX_PROBE = 4096;
static constexpr int SPLIT_LOG_DEFAULT = 6;    // one sampled splitter every ~64 oriented nodes (the walk hops over LDS-built fragments: 5 -> 6 measured best)

// ------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------
template <int W> struct KeyArr {               // SoA key storage
    uint64_t *w[W];
    __device__ __forceinline__ Kmer<W> load(uint64_t i) const {
        Kmer<W> x;
#pragma unroll
        for (int j = 0; j < W; j++) x.w[j] = w[j][i];
        return x;
    }
    __device__ __forceinline__ void store(uint64_t i, const Kmer<W> &x) const {
#pragma unroll
        for (int j = 0; j < W; j++) w[j][i] = x.w[j];
    }
};

template <int W> struct CountTable {           // open addressing, linear probing
    KeyArr<W> keys;
    uint32_t *cnt;
    uint32_t *state;                           // W >= 2 only: 0 empty, 1 being written, 2 ready
    uint64_t mask;
};

// Graph membership table, split into GP mini tables by minimiser partition: a k-mer and its graph
// neighbours share their minimiser nine times out of ten, and the rows arrive grouped by partition
// (pass 2 emits partition by partition), so the 8 probes of a node and of the rows around it go to
// one ~32 KB region that stays in the XCD's L2 — instead of 8 scattered 64-byte fabric requests per
// node into one 134 MB table (the flat version: 3.2 GB fetched for 5 M nodes, profiles/r01_s2_start).
struct GraphTable {                            // entry = fingerprint<<32 | node index
    uint64_t *e;
    uint2 *scan = nullptr;                     // [3][scan_n]: every node's minimiser scan as k_gp_count made it (hash states of its first and last gm-mer,
    uint32_t scan_n = 0;                       // the minima without the first / without the last) — k_graph_local reads 24 bytes instead of repeating k + 1 hash steps
    uint8_t *occ;                              // one bit per slot of e[]: occupied (1.5 MB for 5 M nodes: it stays in every XCD's L2, and three of
                                               // four cross-partition questions are about k-mers that do not exist — half of those end at an empty first slot)
    const unsigned long long *off;             // [GP] first slot of the partition's table
    const uint32_t *msk;                       // [GP] its size - 1 (power of two)
    uint32_t gp_mask;                          // GP - 1
    int gm;                                    // minimiser length of the graph partition function
    uint32_t dbg;                              // timing experiments only (SHK_DEBUG_G)
    // sharded assembly (one process per GPU): a k-mer lives on rank ((minimiser hash & cp_mask) % world) — the owner of
    // its counting partition.  world == 1: everything is local.  world_inv = ceil(2^32 / world) (exact quotients below 2^16)
    uint32_t cp_mask = 0, world = 1, rank = 0, world_inv = 0;
    __device__ __forceinline__ uint32_t owner_of(uint32_t minhash) const {
        const uint32_t x = minhash & cp_mask;                  // < 16384
        const uint32_t q = __umulhi(x, world_inv);
        return x - q * world;
    }
};

}  // namespace shk
#include "count_part.h"
namespace shk {

#include "count_global.h"
#include "graph_part.h"
#include "collapse.h"
#include "shard_graph.h"
#include "writer_gpu.h"

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Process-wide cache of device allocations: a handle lives for one preprocess+assemble, and
// hipMalloc/hipFree of its multi-GB buffers cost more than the kernels (measured ~10 ms/step).
// Blocks are returned here on release and reused by the next handle; shk_release_cached_memory()
// gives them back to the driver.
// Peak device memory of a handle (the reference reports peak wasm memory with every assembly: Assembler.ts:69-71,137).
// Every block the pool hands out is charged to the accounting context the calling thread carries (set by the C ABI's entry
// points for the handle they serve) and released from the context it was charged to, whichever thread gives it back.
struct MemAcct {
    std::atomic<uint64_t> cur{0}, peak{0};
    void add(uint64_t b) { const uint64_t c = cur.fetch_add(b) + b; uint64_t p = peak.load(); while (c > p && !peak.compare_exchange_weak(p, c)) {} }
    void sub(uint64_t b) { cur.fetch_sub(b); }
};
static thread_local std::shared_ptr<MemAcct> tls_mem_acct;
std::shared_ptr<void> mem_acct_new() { return std::static_pointer_cast<void>(std::make_shared<MemAcct>()); }
std::shared_ptr<void> mem_acct_set(std::shared_ptr<void> a) {
    std::shared_ptr<void> prev = std::static_pointer_cast<void>(tls_mem_acct);
    tls_mem_acct = std::static_pointer_cast<MemAcct>(a);
    return prev;
}
uint64_t mem_acct_peak(const std::shared_ptr<void> &a) { return a ? std::static_pointer_cast<MemAcct>(a)->peak.load() : 0; }
uint64_t mem_acct_current(const std::shared_ptr<void> &a) { return a ? std::static_pointer_cast<MemAcct>(a)->cur.load() : 0; }
void mem_acct_replay(const int64_t *deltas, size_t n, uint64_t *peak, uint64_t *current) {
    MemAcct a;
    for (size_t i = 0; i < n; i++) { if (deltas[i] >= 0) a.add((uint64_t)deltas[i]); else a.sub((uint64_t)(-deltas[i])); }
    if (peak) *peak = a.peak.load();
    if (current) *current = a.cur.load();
}

struct DevPool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> free_blocks;     // (device, bytes) -> block
    struct Out { int dev; std::shared_ptr<MemAcct> acct; };
    std::map<void *, Out> owner_dev;                               // every block handed out -> the device it lives on, who pays for it
    bool enabled = true;
    DevPool() { const char *v = getenv("SHK_NO_POOL"); enabled = !(v && *v == '1'); }
    static int cur_dev() { int d = 0; (void)hipGetDevice(&d); return d; }
    void *get(size_t &bytes, hipError_t &e) {
        bytes = (bytes + 4095) & ~(size_t)4095;
        // big scratch buffers vary a little from handle to handle: round them up so the cached block fits again
        if (bytes > ((size_t)256 << 20)) bytes = (bytes + ((size_t)256 << 20) - 1) & ~(((size_t)256 << 20) - 1);
        const int dev = cur_dev();
        const std::shared_ptr<MemAcct> &acct = tls_mem_acct;
        if (enabled) {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_blocks.lower_bound(std::make_pair(dev, bytes));
            if (it != free_blocks.end() && it->first.first == dev && it->first.second <= bytes + bytes / 2 + (1u << 20)) {
                void *p = it->second; bytes = it->first.second; free_blocks.erase(it); owner_dev[p] = Out{dev, acct}; e = hipSuccess;
                if (acct) acct->add(bytes);
                return p;
            }
        }
        void *p = nullptr;
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess && enabled) {              // out of memory: drop the cache and retry once
            trim();
            e = hipMalloc(&p, bytes);
        }
        if (e == hipSuccess) { std::lock_guard<std::mutex> lk(mu); owner_dev[p] = Out{dev, acct}; if (acct) acct->add(bytes); }
        return e == hipSuccess ? p : nullptr;
    }
    // a block goes back under the device it was allocated on, whatever device the calling thread has current
    // (an FFI consumer may free a handle from another thread: the HIP current device is per thread)
    void put(void *p, size_t bytes) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu);
        auto it = owner_dev.find(p);
        const int dev = it != owner_dev.end() ? it->second.dev : cur_dev();
        if (it != owner_dev.end()) { if (it->second.acct) it->second.acct->sub(bytes); owner_dev.erase(it); }
        if (!enabled) { (void)hipFree(p); return; }
        free_blocks.emplace(std::make_pair(dev, bytes), p);
    }
    void trim() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &kv : free_blocks) (void)hipFree(kv.second);
        free_blocks.clear();
    }
};
static DevPool &dev_pool() { static DevPool *p = new DevPool(); return *p; }   // never destroyed (HIP teardown order)
void device_pool_trim() { dev_pool().trim(); }
void *device_pool_alloc(size_t &bytes) { hipError_t e; return dev_pool().get(bytes, e); }
void device_pool_release(void *p, size_t bytes) { dev_pool().put(p, bytes); }

// A DevBuf that goes out of scope while an entry point of the C ABI runs is parked until that entry point has drained the
// handle's stream (DeferScope, pipeline.h): an early `return rc` may destroy buffers that kernels, collectives or copies
// still in flight use, and the pool hands a freed block to whichever handle asks next (two handles in flight: batch.py).
// Explicit release() calls — made where the code knows the stream is idle — go to the pool at once.
static thread_local std::vector<std::pair<void *, size_t>> *tls_defer = nullptr;
template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0; size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() {
        if (p && tls_defer) {
            try { tls_defer->push_back(std::make_pair((void *)p, bytes)); p = nullptr; n = 0; bytes = 0; return; }
            catch (...) {}                                 // (no room for the note: give the block back now)
        }
        release();
    }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(bytes, o.bytes); }
    void release() { if (p) { dev_pool().put(p, bytes); p = nullptr; n = 0; bytes = 0; } }
    int alloc(size_t count, std::string &err) {
        release();
        if (count == 0) count = 1;
        size_t b = count * sizeof(T);
        hipError_t e;
        p = (T *)dev_pool().get(b, e);
        if (!p) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return -4; }
        n = count; bytes = b; return 0;
    }
};

// pinned host staging buffer (D2H of contigs at full PCIe rate), cached the same way
struct PinnedBuf {
    char *p = nullptr; size_t bytes = 0;
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::multimap<size_t, void *> &cache() { static auto *c = new std::multimap<size_t, void *>(); return *c; }
    ~PinnedBuf() { if (p) { std::lock_guard<std::mutex> lk(mu()); cache().emplace(bytes, p); } }
    int alloc(size_t b, std::string &err) {
        b = (b + 4095) & ~(size_t)4095; if (!b) b = 4096;
        if (p && bytes >= b) return 0;
        if (p) { std::lock_guard<std::mutex> lk(mu()); cache().emplace(bytes, p); p = nullptr; bytes = 0; }
        {
            std::lock_guard<std::mutex> lk(mu());
            auto it = cache().lower_bound(b);
            if (it != cache().end() && it->first <= 2 * b + (1u << 20)) { p = (char *)it->second; bytes = it->first; cache().erase(it); return 0; }
        }
        hipError_t e = hipHostMalloc((void **)&p, b, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; err = std::string("hipHostMalloc: ") + hipGetErrorString(e); return -4; }
        bytes = b; return 0;
    }
};

DeferScope::DeferScope(IPipeline *pipe) : pipe_(pipe) {
    auto *v = new (std::nothrow) std::vector<std::pair<void *, size_t>>();
    if (v) { try { v->reserve(256); } catch (...) {} }
    list_ = v; prev_ = tls_defer; tls_defer = v;
}
DeferScope::~DeferScope() {
    auto *v = (std::vector<std::pair<void *, size_t>> *)list_;
    tls_defer = (std::vector<std::pair<void *, size_t>> *)prev_;
    if (!v) return;
    if (!v->empty()) {
        if (pipe_) pipe_->drain();                        // nothing in flight reads or writes the parked blocks any more
        else (void)hipDeviceSynchronize();
        for (auto &b : *v) dev_pool().put(b.first, b.second);
    }
    delete v;
}

// streams of freed handles are reused too (create + destroy cost ~0.2 ms per handle); every use of a
// handle's stream ends in a synchronize, so a pooled stream is idle
static std::mutex g_stream_mu;
static std::multimap<int, hipStream_t> &stream_cache() { static auto *c = new std::multimap<int, hipStream_t>(); return *c; }
// (key: device, or device + 1000 for the copy streams of the host-buffer entry points — creating and destroying one per
// handle cost ~1 ms of the 5.7 ms a step from host pinned memory took: profiles/r04)
static hipStream_t stream_pool_get(int dev) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    auto it = stream_cache().find(dev);
    if (it == stream_cache().end()) return nullptr;
    hipStream_t s = it->second; stream_cache().erase(it); return s;
}
static void stream_pool_put(int dev, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        if (stream_cache().size() < 32) { stream_cache().emplace(dev, s); return; }
    }
    (void)hipStreamDestroy(s);
}

static std::mutex &upload_token(int dev) { static std::mutex m[16]; return m[(unsigned)dev % 16u]; }

static inline int grid_for(uint64_t work, int block = 256, int max_blocks = 256 * 16) {
    uint64_t b = (work + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (uint64_t)max_blocks) b = max_blocks;
    return (int)b;
}

// Wait for a stream by polling: the waits of this pipeline are tens of microseconds to a few milliseconds and
// there are eight of them per assembly; a blocking hipStreamSynchronize adds its wake-up latency to every one
// (A/B on one box, 100 steps each: 4.31 against 4.36 ms per step; SHK_BLOCKING_SYNC=1 restores the blocking
// call).  Falls back to it after ~5 ms.  Removing read-backs altogether (deferred overflow checks, pinned
// staging of the small copies) measured no gain and was not kept.
static inline hipError_t stream_wait(hipStream_t s) {
    static const bool blocking = getenv("SHK_BLOCKING_SYNC") != nullptr;
    if (blocking) return hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) return hipStreamSynchronize(s);
    }
}

struct EvTimer {
    hipEvent_t a, b; hipStream_t st; bool ok = false;
    // (an event recorded between two kernels costs the GPU a bubble of ~10 us on the stream — measured, profiles/r04: the
    // stage timers are therefore off unless asked for; `on` = false makes every method a no-op that reports 0)
    explicit EvTimer(hipStream_t s, bool on = true) : st(s) {
        if (!on) return;
        ok = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess;
        if (ok) (void)hipEventRecord(a, st);
    }
    bool on() const { return ok; }
    double stop() {
        if (!ok) return 0.0;
        (void)hipEventRecord(b, st); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        return ms;
    }
    ~EvTimer() { if (ok) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } }
    // end of the timed region without waiting; elapsed() after the stream has been synchronised anyway
    void mark() { if (ok) (void)hipEventRecord(b, st); }
    double elapsed() {
        if (!ok) return 0.0;
        (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        return ms;
    }
    // stop without waiting: the pair of events goes on `pending` and is read when the timings are asked for
    // (a hipEventSynchronize between two phases that have nothing to wait for costs a pipeline bubble)
    struct Pending { std::string name; hipEvent_t a, b; };
    void stop_later(const char *name, std::vector<Pending> &pending) {
        if (!ok) return;
        (void)hipEventRecord(b, st);
        pending.push_back(Pending{name, a, b});
        ok = false;                                      // the events now belong to the list
    }
    static void resolve(std::vector<Pending> &pending, StageTimes &times) {
        for (auto &p : pending) {
            (void)hipEventSynchronize(p.b);
            float ms = 0; (void)hipEventElapsedTime(&ms, p.a, p.b);
            times.add(p.name, ms);
            (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
        }
        pending.clear();
    }
};

static inline uint64_t env_u64(const char *name, uint64_t dflt) {
    const char *v = getenv(name);
    return (v && *v) ? strtoull(v, nullptr, 10) : dflt;
}
// SHK_DEBUG_* (timing experiments: results are wrong) are read by `make ABLATE=1` builds only
static inline uint32_t env_dbg(const char *name) {
#if SHK_ABLATE
    return (uint32_t)env_u64(name, 0);
#else
    (void)name; return 0u;
#endif
}

// Two fills in one launch (a step of the pipeline made sixteen hipMemsetAsync calls, 5-6 us of the GPU each and most of them a
// few dozen bytes): regions of whole 32-bit words.
__global__ __launch_bounds__(256) void k_fill2(uint32_t *a, unsigned long long na, uint32_t va, uint32_t *b, unsigned long long nb, uint32_t vb) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += stride) {
        if (i < na) a[i] = va; else b[i - na] = vb;
    }
}
// out[i] = i * stride (the places of the partitions' deduplicated records when one batch sits in its slices: a launch instead of
// a copy from pageable host memory, which the runtime only starts once the stream has drained — the host then fell behind
// pass 1 and the launches after it reached the GPU one by one, 36 us of idle time behind k_partition)
__global__ __launch_bounds__(256) void k_stride_fill(unsigned long long *out, uint32_t n, unsigned long long stride) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (unsigned long long)i * stride;
}
static inline hipError_t fill2_async(void *a, size_t bytes_a, uint32_t va, void *b, size_t bytes_b, uint32_t vb, hipStream_t st) {
    const unsigned long long na = bytes_a / 4, nb = bytes_b / 4;
    const unsigned long long blocks = std::min<unsigned long long>((na + nb + 255) / 256, 2048ull);
    if (!blocks) return hipSuccess;
    hipLaunchKernelGGL(k_fill2, dim3((uint32_t)blocks), dim3(256), 0, st, (uint32_t *)a, na, va, (uint32_t *)b, nb, vb);
    return hipGetLastError();
}

// ---- the contig text goes to the host by a KERNEL that writes pinned host memory slab by slab and raises a flag per slab
// there (TextArrival, pipeline.h).  hipMemcpyAsync in pieces cost a blit kernel and ~11 us of gap per piece on the GPU box
// (10 pieces: 0.2 ms for 5 MB that cross PCIe in 0.09) and an event query per look; here the host polls plain memory.
static constexpr uint32_t ARRIVAL_SLAB = 64u << 10;
__global__ __launch_bounds__(256) void k_text_to_host(const char *__restrict__ src, char *__restrict__ dst_host, unsigned long long bytes,
                                                      uint32_t n_slabs, uint32_t *__restrict__ flags_host) {
    for (uint32_t s = blockIdx.x; s < n_slabs; s += gridDim.x) {                      // (roughly front to back)
        const unsigned long long o = (unsigned long long)s * ARRIVAL_SLAB;
        const uint32_t m = (uint32_t)(bytes - o < ARRIVAL_SLAB ? bytes - o : ARRIVAL_SLAB);
        const uint4 *a = reinterpret_cast<const uint4 *>(src + o);
        uint4 *d = reinterpret_cast<uint4 *>(dst_host + o);
        const uint32_t nv = m >> 4;
#pragma unroll 4
        for (uint32_t i = threadIdx.x; i < nv; i += 256) d[i] = a[i];
        if (threadIdx.x < (m & 15u)) dst_host[o + (nv << 4) + threadIdx.x] = src[o + (nv << 4) + threadIdx.x];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&flags_host[s], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// first / last E bytes of every emitted contig, side by side, written straight to pinned host memory with a flag behind them
// (the writer's order, links and strand checks need nothing else of a contig before its text has arrived):
// ends[c][0][E] = text[off .. off + m), ends[c][1][E] = text[off + len - m .. off + len), m = min(E, len).  ONE workgroup.
__global__ __launch_bounds__(1024) void k_ends_to_host(const char *__restrict__ text, const unsigned long long *__restrict__ src /* {off, len} per contig */,
                                                       uint32_t nc, uint32_t E, char *__restrict__ ends_host, uint32_t *__restrict__ flag_host) {
    const unsigned long long total = (unsigned long long)nc * 2ull * E;
    for (unsigned long long t = threadIdx.x; t < total; t += blockDim.x) {
        const uint32_t c = (uint32_t)(t / (2ull * E)), r = (uint32_t)(t % (2ull * E)), side = r / E, j = r % E;
        const unsigned long long off = src[2 * c], len = src[2 * c + 1];
        const unsigned long long m = len < E ? len : (unsigned long long)E;
        char ch = 0;
        if (j < m) ch = side ? text[off + len - m + j] : text[off + j];
        ends_host[t] = ch;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_host, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the same copy when the host does not know the size yet (k_plan_emit's plan): plan[0] == 1 or nothing is copied, plan[1] = bytes
__global__ __launch_bounds__(256) void k_text_to_host_planned(const char *__restrict__ src, char *__restrict__ dst_host,
                                                              const unsigned long long *__restrict__ plan) {
    if (plan[0] != 1ull) return;
    const unsigned long long bytes = plan[1];
    const unsigned long long nv = bytes >> 4;
    const uint4 *a = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst_host);
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < nv; i += (unsigned long long)gridDim.x * 256u) d[i] = a[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15ull)) dst_host[(nv << 4) + threadIdx.x] = src[(nv << 4) + threadIdx.x];
}
class SlabArrival : public TextArrival {
public:
    void set(char *dst, size_t bytes, volatile uint32_t *flags, uint32_t n_slabs, hipStream_t st) { base_ = dst; total_ = bytes; flags_ = flags; n_slabs_ = n_slabs; st_ = st; failed_.store(false); }
    const char *base() const override { return base_; }
    size_t total() const override { return total_; }
    static bool spin_until(volatile uint32_t *f, std::atomic<bool> &failed) {
        if (__atomic_load_n(f, __ATOMIC_ACQUIRE)) return true;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t it = 0;; it++) {
            if (__atomic_load_n(f, __ATOMIC_ACQUIRE)) return true;
            if (failed.load(std::memory_order_relaxed)) return false;
            if ((it & 0xFFFu) == 0xFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) { failed.store(true); return false; }   // (a kernel that died: nobody waits for ever)
            __builtin_ia32_pause();
        }
    }
    void wait_range(size_t begin, size_t end) override {
        if (end > total_) end = total_;
        if (begin >= end) return;
        for (size_t s = begin / ARRIVAL_SLAB; s <= (end - 1) / ARRIVAL_SLAB; s++) if (!spin_until(flags_ + s, failed_)) return;
    }
    int finish(std::string &err) override {
        const hipError_t e = hipStreamSynchronize(st_);                    // (the kernel itself: its source buffer may go afterwards)
        if (e != hipSuccess || failed_.load()) { err = std::string("download of the contigs failed") + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string()); return -5; }
        return 0;
    }
    std::atomic<bool> failed_{false};
private:
    char *base_ = nullptr; size_t total_ = 0; volatile uint32_t *flags_ = nullptr; uint32_t n_slabs_ = 0; hipStream_t st_ = nullptr;
};

// m-mers in the minimiser window of a k-mer (pass 1 of the counting step walks them in one block of 8 or 16, or in two
// blocks of 9), and the minimiser length of the counting partitions and of the graph partitions that follows from it.
// k = 31 takes the window of 18 (m = 14): runs of 9.7 k-mers instead of 8.3, a seventh fewer records for pass 1 to store.
// Measured on the bench isolate (profiles/r04_final/window_sweep.txt): w = 20 (m = 12) makes pass 1 faster still, but the
// partitions get uneven (fewer distinct minimisers per partition: std / mean 0.136 against 0.108) and the graph tables pay
// 0.11 ms for it; two-word keys (k = 51, reads cut short by quality masking) lose with 18 — most of their records are the
// partial runs at segment ends, and longer runs make those longer to expand.  SHK_PART_WIN = 16 / 18 / 20 forces a window
// for k >= 31 (results never depend on it).
static inline int part_win(int k) {
    static const int forced = [] { const char *e = getenv("SHK_PART_WIN"); return e ? atoi(e) : 0; }();
    if (k < 23) return 8;
    if (k < 31) return 16;
    if (forced == 16 || forced == 18 || forced == 20) return forced;
    return k <= 32 ? 18 : 16;
}
static inline int part_m(int k) { return k - part_win(k) + 1; }

template <int W> class Pipeline : public IPipeline {
public:
    explicit Pipeline(int k) : k_(k) {}
    ~Pipeline() override {
        drain();                                            // (the member buffers go back to the pool idle)
        EvTimer::resolve(pending_timers_, times_);
        if (copy_stream_) { (void)hipStreamSynchronize(copy_stream_); stream_pool_put(stream_dev_ + 1000, copy_stream_); }
        if (stream_) stream_pool_put(stream_dev_, stream_);
    }
    int init(std::string &err) {
        HIPCHK(hipGetDevice(&stream_dev_));
        { int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, stream_dev_); n_cus_ = cus > 0 ? cus : 256; }
        stream_ = stream_pool_get(stream_dev_);
        if (!stream_) HIPCHK(hipStreamCreate(&stream_));
        HIPCHK(ctl_.alloc(CTL_WORDS, err) ? hipErrorOutOfMemory : hipSuccess);
        if (mbox_.alloc(MBOX_BYTES, err)) return -4;
        return 0;
    }
    StageTimes &times() override { EvTimer::resolve(pending_timers_, times_); return times_; }
    void drain() override {
        if (copy_stream_) (void)hipStreamSynchronize(copy_stream_);
        if (!stream_) return;
        std::string e;
        if (wd_comm_) (void)comm_stream_wait(wd_comm_, stream_, e);
        else (void)hipStreamSynchronize(stream_);
    }
    void *stream() override { return (void *)stream_; }
    int device() const override { return stream_dev_; }
    uint64_t total_instances() const override { return total_instances_; }
    uint64_t n_distinct() const override { return n_distinct_; }
    uint64_t n_solid() const override { return n_solid_; }

    // ---- counting --------------------------------------------------------------------------
    int alloc_table(uint64_t slots, std::string &err) {
        for (int j = 0; j < W; j++) if (int rc = tkeys_[j].alloc(slots, err)) return rc;
        if (int rc = tcnt_.alloc(slots, err)) return rc;
        if (W > 1) if (int rc = tstate_.alloc(slots, err)) return rc;
        tslots_ = slots;
        if (W == 1) HIPCHK(hipMemsetAsync(tkeys_[0].p, 0xFF, slots * 8, stream_));
        else HIPCHK(hipMemsetAsync(tstate_.p, 0, slots * 4, stream_));
        HIPCHK(hipMemsetAsync(tcnt_.p, 0, slots * 4, stream_));
        return 0;
    }
    CountTable<W> table_view() {
        CountTable<W> t;
        for (int j = 0; j < W; j++) t.keys.w[j] = tkeys_[j].p;
        t.cnt = tcnt_.p; t.state = tstate_.p; t.mask = tslots_ - 1;
        return t;
    }

    int count_batch_global(const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg,
                           uint64_t n_bases, std::string &err) {
        if (n_seg >= 0xFFFFFFFFull || n_bases >= 0xFFFFFFFFull) { err = "batch too large (>= 2^32 bases)"; return -1; }
        if (n_seg == 0) return 0;
        // single batch per table in this version; size from the instance upper bound
        const uint64_t inst_ub = n_bases - n_seg * (uint64_t)(k_ - 1);
        if (tslots_ == 0) {
            uint64_t want = inst_ub / 4 + 1024;
            uint64_t slots = 1ull << 16;
            while (slots < want) slots <<= 1;
            if (int rc = alloc_table(slots, err)) return rc;
            pending_.clear();
        }
        pending_.push_back({d_bases, d_seg_off, n_seg});
        for (;;) {
            HIPCHK(hipMemsetAsync(ctl_.p, 0, CTL_WORDS * sizeof(unsigned long long), stream_));
            EvTimer t(stream_);
            hipLaunchKernelGGL(k_count_segments<W>, dim3(grid_for(n_seg)), dim3(256), 0, stream_, d_bases,
                               d_seg_off, (uint32_t)n_seg, k_, table_view(), (uint32_t *)(ctl_.p + 1),
                               ctl_.p + 0);
            HIPCHK(hipGetLastError());
            double ms = t.stop();
            unsigned long long h[2];
            HIPCHK(hipMemcpyAsync(h, ctl_.p, sizeof h, hipMemcpyDeviceToHost, stream_));
            WAIT_STREAM();
            if ((uint32_t)h[1] == 0) {
                times_.add("count_kernel", ms);
                total_instances_ += h[0];
                return 0;
            }
            // table too small: grow 4x and recount every batch seen so far
            if (pending_.size() > 1) { err = "count table overflow across batches"; return -6; }
            times_.add("count_retry", ms);
            if (int rc = alloc_table(tslots_ * 4, err)) return rc;
        }
    }

    // ---- partitioned counting (count_part.h) ---------------------------------------------------
    template <int WBLK, int NBLK>
    void launch_partition_t(const uint32_t *d_bases, const uint32_t *d_seg_off, uint32_t n_seg) {
        hipLaunchKernelGGL((k_partition<W, WBLK, NBLK>), dim3(pp_.G), dim3(PART_THREADS), 0, stream_, d_bases, d_seg_off,
                           n_seg, pp_, recs_.p, fill_.p, (uint32_t *)(ctl_.p + 8));
    }
    void launch_partition(int win, const uint32_t *d_bases, const uint32_t *d_seg_off, uint32_t n_seg) {
        if (win == 20) launch_partition_t<10, 2>(d_bases, d_seg_off, n_seg);
        else if (win == 18) launch_partition_t<9, 2>(d_bases, d_seg_off, n_seg);
        else if (win == 16) launch_partition_t<16, 1>(d_bases, d_seg_off, n_seg);
        else launch_partition_t<8, 1>(d_bases, d_seg_off, n_seg);
    }

    int count_batch(const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg, uint64_t n_bases,
                    std::string &err) override {
        return count_batch_impl(d_bases, d_seg_off, nullptr, nullptr, n_seg, n_bases, err);
    }
    // the packed reads are still in HOST memory: they are uploaded in a few pieces on a second stream, and pass 1 of
    // piece i runs while piece i+1 travels (the pieces append to the same slices: k_partition continues its cursors)
    int count_batch_host(uint32_t *d_bases, uint32_t *d_seg_off, const uint32_t *h_bases, const uint32_t *h_seg_off,
                         uint64_t n_seg, uint64_t n_bases, std::string &err) override {
        return count_batch_impl(d_bases, d_seg_off, h_bases, h_seg_off, n_seg, n_bases, err);
    }
    int count_batch_pieces(const DevPiece *pieces, size_t n_pieces, std::string &err) override {
        std::vector<DevPiece> list;
        uint64_t n_seg = 0, n_bases = 0;
        for (size_t i = 0; i < n_pieces; i++)
            if (pieces[i].n_seg) { list.push_back(pieces[i]); n_seg += pieces[i].n_seg; n_bases += pieces[i].n_bases; }
        if (list.empty()) return 0;
        if (list.size() == 1) return count_batch_impl(list[0].d_bases, list[0].d_seg_off, nullptr, nullptr, n_seg, n_bases, err);
        if (global_mode_) {                                        // (the cross-check path takes batches one by one)
            for (const DevPiece &pc : list)
                if (int rc = count_batch_impl(pc.d_bases, pc.d_seg_off, nullptr, nullptr, pc.n_seg, pc.n_bases, err)) return rc;
            return 0;
        }
        piece_list_ = std::move(list);
        struct Clear { std::vector<DevPiece> &v; ~Clear() { v.clear(); } } clear{piece_list_};
        return count_batch_impl(piece_list_[0].d_bases, piece_list_[0].d_seg_off, nullptr, nullptr, n_seg, n_bases, err);
    }
    // pass 1 over the batch: one launch, or one per piece (count_batch_pieces) appending to the same slices
    void launch_partition_all(int wblk, const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg) {
        if (piece_list_.empty()) { launch_partition(wblk, d_bases, d_seg_off, (uint32_t)n_seg); return; }
        for (const DevPiece &pc : piece_list_) launch_partition(wblk, pc.d_bases, pc.d_seg_off, (uint32_t)pc.n_seg);
    }
    int count_batch_impl(const uint32_t *d_bases, const uint32_t *d_seg_off, const uint32_t *h_bases, const uint32_t *h_seg_off,
                         uint64_t n_seg, uint64_t n_bases, std::string &err) {
        if (h_bases && (global_mode_ || n_seg == 0)) {            // (nothing to overlap: plain upload first)
            HIPCHK(hipMemcpyAsync((void *)d_bases, h_bases, ((n_bases + 15) / 16 + 1) * 4, hipMemcpyHostToDevice, stream_));
            HIPCHK(hipMemcpyAsync((void *)d_seg_off, h_seg_off, (n_seg + 1) * 4, hipMemcpyHostToDevice, stream_));
            h_bases = nullptr;
        }
        if (global_mode_) return count_batch_global(d_bases, d_seg_off, n_seg, n_bases, err);
        if (n_seg >= 0xFFFFFFFFull || n_bases >= 0xFFFFFFFFull) { err = "batch too large (>= 2^32 bases)"; return -1; }
        if (n_seg == 0) return 0;
        // several batches per handle (chunked / streamed / > 2^32-base inputs): the previous batch's records
        // are packed densely into a buffer of their own; pass 2 then reads one run per batch and partition
        if (have_parts_) if (int rc = pack_current_batch(err)) return rc;
        constexpr int RW = 2 * W;
        const int wblk = part_win(k_);                    // m-mers per window
        const uint64_t inst_ub = n_bases - n_seg * (uint64_t)(k_ - 1);
        const int cus = n_cus_;
        const uint64_t n_super = (n_seg + PART_THREADS - 1) / PART_THREADS;
        pp_.k = k_; pp_.m = k_ - wblk + 1; pp_.dbg_nostore = env_dbg("SHK_DEBUG_NOSTORE");
        pp_.dbg_clk = nullptr; pp_.dbg_flush_at = env_dbg("SHK_DEBUG_P1FLUSH");
#if SHK_ABLATE
        if (env_dbg("SHK_DEBUG_P1CLK")) {                   // (timing experiment: the buffer is leaked on purpose, the report goes to stderr)
            static unsigned long long *clk = nullptr;
            if (!clk) (void)hipMalloc((void **)&clk, 64);
            else {
                unsigned long long h[8] = {0};
                (void)hipDeviceSynchronize(); (void)hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
                if (h[4]) fprintf(stderr, "[p1clk] per wave: pre %.0f walk(incl. flush) %.0f flush %.0f tail %.0f cycles (%llu waves)\n", (double)h[0] / h[4], (double)h[1] / h[4], (double)h[2] / h[4], (double)h[3] / h[4], h[4]);
            }
            (void)hipMemset(clk, 0, 64);
            pp_.dbg_clk = clk;
        }
#endif
        pp_.max_n = std::min<uint32_t>(32u * RW - 3u - (uint32_t)(k_ - 1), 63u);
        if (uint64_t mn = env_u64("SHK_PART_MAXN", 0)) pp_.max_n = std::min<uint32_t>(pp_.max_n, (uint32_t)mn);
        pp_.G = (uint32_t)std::min<uint64_t>((uint64_t)std::min(cus, 256), n_super);
        if (uint64_t fg = env_u64("SHK_PART_G", 0)) pp_.G = (uint32_t)std::min<uint64_t>(pp_.G, fg);   // (tests: few workgroups, many tiles per wave)
        uint32_t P = 64;
        // instances per partition: sized so that the distinct k-mers of a 100x isolate load the LDS k-mer table to ~40 %
        const uint64_t per_part = env_u64("SHK_PART_INST", W == 1 ? 100000 : 40000);
        while (P < (uint32_t)PART_MAX_P && (uint64_t)P * per_part < inst_ub) P <<= 1;
        if (uint64_t fp = env_u64("SHK_PART_P", 0)) P = (uint32_t)fp;
        if (forced_P_) P = forced_P_;
        if (!batches_.empty()) P = pp_.P;                 // every batch uses the first batch's partitioning
        pp_.P = P;
        // records per slice: mean run length is ~(WBLK+1)/2 k-mers (shorter if max_n caps it); 2x slack
        const uint64_t per_rec = pp_.max_n >= 16 ? 4 : 2;
        uint64_t cap = inst_ub / (per_rec * P * pp_.G) + 32 + env_u64("SHK_SLICE_PAD", 0);
        if (cap_override_) { cap = cap_override_; cap_override_ = 0; }      // (pass 1 repeated with the room the first run asked for)
        for (int attempt = 0; attempt < 2; attempt++) {
            // The 256 producer workgroups append to slices [p][0..G) at the same time; when consecutive slices lie
            // (almost) a multiple of 2 KB apart their writes keep meeting in the same memory channels
            // (measured on the bench workload: slices of 2032 / 2064 / 4096 bytes 0.90 ms, of 2112 ... 3056 bytes
            // 0.80 ms for k_partition).  Slice sizes stay >= 256 bytes away from a multiple of 2 KB.
            if (env_u64("SHK_SLICE_NOSKEW", 0) == 0)
                while ((cap * RW * 8) % 2048 < 256 || (cap * RW * 8) % 2048 > 1792) cap++;
            if (cap * pp_.G > 0xFFFFFFF0ull) { err = "a partition would hold more than 2^32 records"; return -1; }
            pp_.slice_cap = (uint32_t)cap;
            const uint64_t n_slices = (uint64_t)P * pp_.G;
            if (int rc = recs_.alloc(n_slices * cap * RW, err)) return rc;
            if (int rc = fill_.alloc(n_slices, err)) return rc;
            HIPCHK(fill2_async(fill_.p, n_slices * 4, 0u, ctl_.p, CTL_WORDS * sizeof(unsigned long long), 0u, stream_));
            EvTimer t(stream_);
            if (h_bases && attempt == 0) {
                // ---- upload and pass 1, piece by piece
                uint64_t C = env_u64("SHK_H2D_PIECES", 4);
                if (C < 1) C = 1;
                while (C > 1 && n_seg / C < (uint64_t)PART_THREADS * pp_.G) C--;          // every piece fills the chip
                if (!copy_stream_) copy_stream_ = stream_pool_get(stream_dev_ + 1000);
                if (!copy_stream_) HIPCHK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
                const uint64_t n_words = (n_bases + 15) / 16 + 1;
                std::vector<hipEvent_t> evs((size_t)C, nullptr);
                struct EvGuard { std::vector<hipEvent_t> &v; ~EvGuard() { for (auto e : v) if (e) (void)hipEventDestroy(e); } } evg{evs};
                // on every way out, errors included, the copy stream is idle: the caller's host buffers and the device
                // blocks (which go back to a pool without stream-ordering bookkeeping) are no longer read or written
                // ONE upload at a time per device: two handles in flight that upload together halve each other's PCIe rate while the
                // GPU waits for both (measured: both at 42 GB/s for 3.2 ms, kernels idle meanwhile); in turn, one uploads at the
                // full rate under the kernels of the other.  The token is held until this upload has drained.
                struct CopyDrain { hipStream_t s; std::unique_lock<std::mutex> token; ~CopyDrain() { (void)hipStreamSynchronize(s); } }
                    copy_drain{copy_stream_, std::unique_lock<std::mutex>(upload_token(stream_dev_))};
                for (uint64_t c = 0; c < C; c++) {
                    const uint64_t s0 = n_seg * c / C, s1 = n_seg * (c + 1) / C;
                    const uint64_t w0 = h_seg_off[s0] >> 4, w1 = std::min<uint64_t>(n_words, (((uint64_t)h_seg_off[s1] + 15) >> 4) + 1);
                    HIPCHK(hipMemcpyAsync((void *)(d_seg_off + s0), h_seg_off + s0, (s1 - s0 + 1) * 4, hipMemcpyHostToDevice, copy_stream_));
                    HIPCHK(hipMemcpyAsync((void *)(d_bases + w0), h_bases + w0, (w1 - w0) * 4, hipMemcpyHostToDevice, copy_stream_));
                    HIPCHK(hipEventCreateWithFlags(&evs[c], hipEventDisableTiming));
                    HIPCHK(hipEventRecord(evs[c], copy_stream_));
                    HIPCHK(hipStreamWaitEvent(stream_, evs[c], 0));
                    launch_partition(wblk, d_bases, d_seg_off + s0, (uint32_t)(s1 - s0));
                }
                HIPCHK(hipGetLastError());
                times_.add("h2d_pieces_x1", (double)C);
                if (defer_p1_ && batches_.empty()) {            // (as below: pass 2 follows without a host round trip)
                    HIPCHK(hipMemcpyAsync(mbox64() + MB_P1, ctl_.p + 8, 16, hipMemcpyDeviceToHost, stream_));
                    t.stop_later("partition_with_upload", pending_timers_);
                    p1_ = P1Pending{true, d_bases, d_seg_off, n_seg, n_bases, cap};
                    return finish_partition(n_slices, err);
                }
                t.mark();
                unsigned long long h[2];
                HIPCHK(hipMemcpyAsync(h, ctl_.p + 8, sizeof h, hipMemcpyDeviceToHost, stream_));
                WAIT_STREAM();
                const uint32_t *fl = (const uint32_t *)&h[0];
                if (fl[1]) { err = "a read segment exceeds 32768 bases (split it on the host)"; return -1; }
                if ((uint32_t)h[1] <= cap) {
                    times_.add("partition_with_upload", t.elapsed());
                    if (int rc = finish_partition(n_slices, err)) return rc;
                    return 0;
                }
                times_.add("partition_retry", t.elapsed());
                cap = (uint64_t)(uint32_t)h[1] + 8;               // (the reads are on the device now: the retry is one launch)
                continue;
            }
            launch_partition_all(wblk, d_bases, d_seg_off, n_seg);
            HIPCHK(hipGetLastError());
            if (defer_p1_ && attempt == 0 && batches_.empty() && piece_list_.empty()) {
                // ONE batch whose reads outlive histogram() (the packed entry points): pass 2 is launched behind pass 1 without a
                // host round trip; the flags land in pinned memory and are looked at after pass 2's own read-back.  A slice
                // that overflowed stored nothing beyond its room (wave_flush) and its run is clamped (k_make_runs): pass 2 then
                // ran on a part of the records and both passes are repeated with the exact room (histogram()).
                HIPCHK(hipMemcpyAsync(mbox64() + MB_P1, ctl_.p + 8, 16, hipMemcpyDeviceToHost, stream_));
                t.stop_later("partition_kernel", pending_timers_);
                p1_ = P1Pending{true, d_bases, d_seg_off, n_seg, n_bases, cap};
                return finish_partition(n_slices, err);
            }
            t.mark();
            unsigned long long h[2];
            HIPCHK(hipMemcpyAsync(h, ctl_.p + 8, sizeof h, hipMemcpyDeviceToHost, stream_));
            WAIT_STREAM();          // (one host round trip: flags and the timer together)
            const double ms = t.elapsed();
            const uint32_t *fl = (const uint32_t *)&h[0];
            if (fl[1]) { err = "a read segment exceeds 32768 bases (split it on the host)"; return -1; }
            const uint32_t max_fill = (uint32_t)h[1];
            if (max_fill <= cap) {
                times_.add("partition_kernel", ms);
                if (int rc = finish_partition(n_slices, err)) return rc;
                return 0;
            }
            times_.add("partition_retry", ms);
            cap = (uint64_t)max_fill + 8;               // exact from the counting run
        }
        err = "partition slices overflowed twice";
        return -6;
    }

    // pass 1 is done: run table of the local layout, one run per (partition, producer workgroup)
    int finish_partition(uint64_t n_slices, std::string &err) {
        constexpr int RW = 2 * W;
        have_parts_ = true;
        if (int rc = run_off_.alloc(n_slices, err)) return rc;
        if (int rc = run_cnt_.alloc(n_slices, err)) return rc;
        hipLaunchKernelGGL(k_make_runs, dim3(grid_for(n_slices)), dim3(256), 0, stream_, fill_.p, pp_, recs_.p,
                           (uint32_t)RW, run_off_.p, run_cnt_.p);
        HIPCHK(hipGetLastError());
        run_view_.run_addr16 = run_off_.p; run_view_.run_cnt = run_cnt_.p;
        run_view_.S = pp_.G; run_view_.k = k_; n_count_parts_ = pp_.P; run_view_.dbg = env_dbg("SHK_DEBUG_P2");
        return 0;
    }

    // ---- batches ---------------------------------------------------------------------------------
    struct BatchRecs { DevBuf<uint64_t> dense; std::vector<unsigned long long> part_off; /* [P+1], records */ };

    // the batch that still sits in its [p][g] slices -> a dense buffer (partition-major), slices released
    int pack_current_batch(std::string &err) {
        constexpr int RW = 2 * W;
        DevBuf<unsigned long long> tot, base;
        if (int rc = tot.alloc(pp_.P, err)) return rc;
        if (int rc = base.alloc(pp_.P, err)) return rc;
        hipLaunchKernelGGL(k_part_totals, dim3(pp_.P), dim3(256), 0, stream_, fill_.p, pp_, tot.p);
        HIPCHK(hipGetLastError());
        std::vector<unsigned long long> h(pp_.P);
        HIPCHK(hipMemcpyAsync(h.data(), tot.p, (size_t)pp_.P * 8, hipMemcpyDeviceToHost, stream_));
        WAIT_STREAM();
        std::unique_ptr<BatchRecs> b(new BatchRecs());
        b->part_off.assign(pp_.P + 1, 0);
        for (uint32_t p = 0; p < pp_.P; p++) b->part_off[p + 1] = b->part_off[p] + h[p];
        if (int rc = b->dense.alloc(b->part_off[pp_.P] * RW + 2, err)) return rc;
        HIPCHK(hipMemcpyAsync(base.p, b->part_off.data(), (size_t)pp_.P * 8, hipMemcpyHostToDevice, stream_));
        EvTimer t(stream_, stage_timers_);
        hipLaunchKernelGGL((k_pack_partition<2 * W>), dim3(pp_.P), dim3(256), 0, stream_, recs_.p, fill_.p, pp_, base.p, b->dense.p);
        HIPCHK(hipGetLastError());
        if (t.on()) times_.add("batch_pack_kernel", t.stop());
        WAIT_STREAM();
        batches_.push_back(std::move(b));
        recs_.release(); fill_.release(); run_off_.release(); run_cnt_.release(); have_parts_ = false;
        if (batches_.size() >= env_u64("SHK_MERGE_BATCHES_AT", 192)) return merge_batches(err);
        return 0;
    }

    // Any number of batches per handle (chunked mode hands one on every chunk_size reads, a stream one per
    // chunk): a run table holds 256 runs per partition, so when that many batches have piled up they are
    // copied into one (every record moves once per ~200 batches).
    int merge_batches(std::string &err) {
        constexpr int RW = 2 * W;
        if (batches_.size() < 2) return 0;
        if (int rc = make_batch_run_view(err)) return rc;
        std::unique_ptr<BatchRecs> m(new BatchRecs());
        m->part_off.assign(pp_.P + 1, 0);
        for (uint32_t p = 0; p < pp_.P; p++) {
            unsigned long long tot = 0;
            for (auto &b : batches_) tot += b->part_off[p + 1] - b->part_off[p];
            m->part_off[p + 1] = m->part_off[p] + tot;
        }
        if (int rc = m->dense.alloc(m->part_off[pp_.P] * RW + 2, err)) return rc;
        DevBuf<unsigned long long> base;
        if (int rc = base.alloc(pp_.P, err)) return rc;
        HIPCHK(hipMemcpyAsync(base.p, m->part_off.data(), (size_t)pp_.P * 8, hipMemcpyHostToDevice, stream_));
        EvTimer t(stream_);
        hipLaunchKernelGGL((k_merge_runs<RW>), dim3(pp_.P), dim3(256), 0, stream_, run_view_, base.p, m->dense.p);
        HIPCHK(hipGetLastError());
        if (t.on()) times_.add("batch_merge_kernel", t.stop());
        WAIT_STREAM();
        batches_.clear();
        batches_.push_back(std::move(m));
        run_off_.release(); run_cnt_.release(); have_parts_ = false;
        return 0;
    }

    // run tables over the packed batches: partition p, run b = batch b's records of p
    int make_batch_run_view(std::string &err) {
        const uint32_t nb = (uint32_t)batches_.size();
        if (nb > 256) { err = "more than 256 batches per handle"; return -1; }
        const uint64_t n_runs = (uint64_t)pp_.P * nb;
        std::vector<unsigned long long> addr16(n_runs); std::vector<uint32_t> cnt(n_runs);
        for (uint32_t p = 0; p < pp_.P; p++) {
            unsigned long long r = 0;                     // a partition's record index is 32 bits wide in pass 2
            for (uint32_t b = 0; b < nb; b++) r += batches_[b]->part_off[p + 1] - batches_[b]->part_off[p];
            if (r > 0xFFFFFFF0ull) { err = "a partition holds more than 2^32 records"; return -1; }
        }
        for (uint32_t p = 0; p < pp_.P; p++)
            for (uint32_t b = 0; b < nb; b++) {
                const BatchRecs &B = *batches_[b];
                const unsigned long long c = B.part_off[p + 1] - B.part_off[p];
                if (c > 0xFFFFFFFFull) { err = "partition too large in one batch"; return -1; }
                addr16[(uint64_t)p * nb + b] = ((unsigned long long)(uintptr_t)B.dense.p >> 4) + B.part_off[p] * (unsigned long long)W;
                cnt[(uint64_t)p * nb + b] = (uint32_t)c;
            }
        if (int rc = run_off_.alloc(n_runs, err)) return rc;
        if (int rc = run_cnt_.alloc(n_runs, err)) return rc;
        HIPCHK(hipMemcpyAsync(run_off_.p, addr16.data(), n_runs * 8, hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemcpyAsync(run_cnt_.p, cnt.data(), n_runs * 4, hipMemcpyHostToDevice, stream_));
        WAIT_STREAM();
        run_view_.run_addr16 = run_off_.p; run_view_.run_cnt = run_cnt_.p;
        run_view_.S = nb; run_view_.k = k_; n_count_parts_ = pp_.P; run_view_.dbg = 0;
        have_parts_ = true;
        return 0;
    }
    void set_bloom(bool on) override { bloom_ = on; }
    void set_verbose(bool on) override { if (on) { stage_timers_ = true; keep_stages_ = true; } }
    void single_batch_resident(bool on) override { defer_p1_ = on && !global_mode_; }
    void expect_more_batches() override { if (!forced_P_ && batches_.empty() && !have_parts_) forced_P_ = (uint32_t)PART_MAX_P; }

    // pass 2 into (keys, cnt) with the given emit threshold; sizes the output by retrying.
    // Partitions whose distinct k-mers do not fit the LDS table are listed by the first launch and
    // then repartitioned at k-mer level (k_ovf_scatter -> k_count_buckets); should a bucket region
    // overflow (extreme skew), that partition falls back to in-kernel residue-class re-runs.
    int run_count_partitions(const RunView &rv, uint32_t n_parts, uint32_t threshold, DevBuf<uint64_t> (&keys)[W],
                             DevBuf<uint32_t> &cnt, uint64_t &n_rows, uint64_t hist_out[500], uint64_t &inst_out,
                             uint64_t cap_hint, double &ms_out, std::string &err) {
        if (n_parts == 0) { n_rows = 0; inst_out = 0; memset(hist_out, 0, 500 * 8); ms_out = 0; return 0; }
        constexpr uint32_t S = CountShared<W>::S;
        // (weighted records — sharded counting, deduplicated by their sources — stay in k_count_partitions: the bucket path
        // counts one per entry; a partition that does not fit is re-run by residue classes there)
        const bool repartition = env_u64("SHK_NO_REPARTITION", 0) == 0 && rv.weights == nullptr;
        DevBuf<unsigned long long> dh;
        DevBuf<OvfRec> d_ovf; DevBuf<OvfItem> d_items; DevBuf<uint32_t> d_fill, d_list, d_maxfill, d_new; DevBuf<uint64_t> d_kmers; DevBuf<BucketRef> d_blist; DevBuf<unsigned long long> d_sumfill;
        const bool bloom = bloom_ && !bloom_off_once_;          // (do_bloom: Bloom pre-filter in the k-mer-level repartition)
        unsigned long long bloom_new = 0, bloom_kmer_bytes = 0, bloom_kmer_bytes_exact = 0;
        if (int rc = dh.alloc(500, err)) return rc;
        if (repartition) if (int rc = d_ovf.alloc(n_parts, err)) return rc;
        // Two-kernel pass 2 (default for local reads): k_dedupe_partitions writes every partition's DISTINCT records with
        // their multiplicities (sorted by length), k_count_weighted expands them — each needs half of a CU's LDS, so two
        // workgroups per CU hide each other's latencies (the fused k_count_partitions owns the whole LDS: 0.87 ms on the
        // bench workload).  SHK_COUNT_SPLIT=0 keeps the fused kernel; partitions that do not fit the k-mer table go to the
        // k-mer-level repartition from their RAW records either way.
        // (three- and four-word keys: the two kernels do not fit 64 registers and run one workgroup per CU like the fused one — no gain)
        // (Bloom mode: reads with errors by design — the fused kernel's in-launch sample hands the partitions over sooner)
        const bool split = W <= 2 && repartition && !bloom && rv.weights == nullptr && split_ready_ && split_base_.size() == n_parts && split_total_ > 0 &&
                           env_u64("SHK_COUNT_SPLIT", 1) != 0;
        struct DropSplit { Pipeline<W> *me; bool on; ~DropSplit() { if (on) me->shard_drop_dedup(); } } drop_split{this, split};
        uint64_t cap = cap_hint;
        // (exact modes: the second attempt knows the row count.  Bloom mode is not repeatable to the row — which
        // k-mers a false positive lifts over the threshold depends on the order of arrival — so its retries get slack)
        for (int attempt = 0; attempt < (bloom ? 4 : 2); attempt++) {
            bloom_new = 0; bloom_kmer_bytes = 0; bloom_kmer_bytes_exact = 0;
            for (int j = 0; j < W; j++) if (int rc = keys[j].alloc(cap, err)) return rc;
            if (int rc = cnt.alloc(cap, err)) return rc;
            HIPCHK(fill2_async(dh.p, 500 * 8, 0u, ctl_.p, CTL_WORDS * sizeof(unsigned long long), 0u, stream_));
            KeyArr<W> ok; for (int j = 0; j < W; j++) ok.w[j] = keys[j].p;
            const uint32_t probe_parts = (uint32_t)env_u64("SHK_PROBE_PARTS", 512);     // 0 = off
            const uint32_t n_probe = (repartition && probe_parts && n_parts / 4 >= probe_parts) ? probe_parts : 0u;
            EvTimer t(stream_);
            if (split) {
                // (every attempt: the partitions the dedupe hands over are reported in d_ovf / ctl_, which an attempt starts empty)
                if (int rc = dd_base_.alloc(n_parts, err)) return rc;
                if (int rc = dd_n_.alloc(n_parts, err)) return rc;
                if (int rc = dd_recs_.alloc(split_total_ * 2 * W + 2, err)) return rc;
                if (int rc = dd_w_.alloc(split_total_ + 2, err)) return rc;
                if (split_stride_) {
                    hipLaunchKernelGGL(k_stride_fill, dim3((n_parts + 255) / 256), dim3(256), 0, stream_, dd_base_.p, n_parts, split_stride_);
                    HIPCHK(hipGetLastError());
                } else HIPCHK(hipMemcpyAsync(dd_base_.p, split_base_.data(), (size_t)n_parts * 8, hipMemcpyHostToDevice, stream_));
                // (no sample launch in front: the partitions the dedupe tries are the sample, and its verdict is where the counting
                // kernel's own tally starts — count_part.h)
                const uint32_t defer_after = n_probe / 8;
                EvTimer t_a(stream_, stage_timers_);
                hipLaunchKernelGGL(k_dedupe_partitions<W>, dim3(std::min<uint32_t>(n_parts, 2u * (uint32_t)n_cus_)), dim3(COUNT_THREADS), 0, stream_,
                                   rv, 0u, n_parts, dd_base_.p, dd_recs_.p, dd_w_.p, dd_n_.p, (uint32_t *)(ctl_.p + 12),
                                   d_ovf.p, (uint32_t *)(ctl_.p + 3), defer_after, (S / 10) * 9, (uint32_t *)(ctl_.p + 14) + 1);
                HIPCHK(hipGetLastError());
                if (attempt == 0) t_a.stop_later("count_dedupe_kernel", pending_timers_);
                const uint32_t per_cu = 2u;
                // (groups of `merge` partitions share a table; a group is only worth it when the chip still gets >= 2 groups per workgroup)
                // (1, 2 or 4, and the groups' stride a multiple of 256: the members of a group share the low bits of their number)
                // Default ONE partition per table since the window of pass 1 grew (round 4): measured alternating on the bench isolate,
                // pass 2 0.642 against 0.662 ms with two per table, and the assembly behind it 1.46 against 1.55 ms — rows that come
                // out in pure partition groups suit the graph tables and the collapse's tiles (k = 51 masked: 1.03 against 1.05 ms).
                uint32_t merge = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(env_u64("SHK_COUNT_MERGE", 1), 1), 4);
                if (merge == 3) merge = 2;
                while (merge > 1 && (n_parts % merge != 0 || (n_parts / merge) % 256u != 0 || n_parts / merge < 2u * per_cu * (uint32_t)n_cus_)) merge >>= 1;
                const uint32_t n_groups = (n_parts + merge - 1) / merge;
                hipLaunchKernelGGL(k_count_weighted<W>, dim3(std::min<uint32_t>(n_groups, per_cu * (uint32_t)n_cus_)), dim3(COUNT_THREADS), 0, stream_,
                                   dd_recs_.p, dd_w_.p, dd_base_.p, dd_n_.p, 0u, n_parts, merge, rv.k, threshold, dh.p, ok, cnt.p, (unsigned long long)cap,
                                   // (ctl_[14]: the groups handed out and, in its high word, this kernel's tally — it starts at the dedupe's verdict
                                   // and covers reads whose records repeat but whose k-mers do not fit)
                                   ctl_.p + 0, ctl_.p + 1, d_ovf.p, (uint32_t *)(ctl_.p + 3), ctl_.p + 14, 0u, defer_after,
                                   env_dbg("SHK_DEBUG_P2"));
            } else {
                // (persistent workgroups, one per CU: the tables take the whole LDS)
                auto kern = rv.weights ? k_count_partitions<W, true> : k_count_partitions<W, false>;
                hipLaunchKernelGGL(kern, dim3(std::min<uint32_t>(n_parts, (uint32_t)n_cus_)), dim3(COUNT_THREADS), 0, stream_, rv, n_parts, threshold,
                                   dh.p, ok, cnt.p, (unsigned long long)cap, ctl_.p + 0, ctl_.p + 1,
                                   (uint32_t *)(ctl_.p + 2), (const uint32_t *)nullptr, repartition ? d_ovf.p : (OvfRec *)nullptr,
                                   (uint32_t *)(ctl_.p + 3), (uint32_t *)(ctl_.p + 11), n_probe / 2, n_probe / 8);
            }
            HIPCHK(hipGetLastError());
            t.mark();
            unsigned long long h[4];
            HIPCHK(hipMemcpyAsync(h, ctl_.p, sizeof h, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipMemcpyAsync(hist_out, dh.p, 500 * 8, hipMemcpyDeviceToHost, stream_));   // (final unless partitions overflowed)
            WAIT_STREAM();          // one host round trip: counters, histogram and the timer
            ms_out = t.elapsed();      // (sample + dedupe + count)
            const uint32_t n_ovf = (uint32_t)h[3];
            if (env_u64("SHK_VERBOSE_TALLY", 0)) fprintf(stderr, "[shk] pass 2: %u partitions, %u handed over, tally tried %u / over %u, split %d\n", n_parts, n_ovf,
                                                    (unsigned)((h[3] >> 32) & 0xFFFFu), (unsigned)(h[3] >> 48), (int)split);
            // rows written by the bucket path are ordered by key hash, not grouped by minimiser partition (build_graph regroups)
            rows_scattered_ = (uint64_t)n_ovf * 4u > n_parts;
            if (n_ovf) {
                EvTimer t2(stream_);
                std::vector<OvfRec> ov(n_ovf);
                HIPCHK(hipMemcpy(ov.data(), d_ovf.p, (size_t)n_ovf * sizeof(OvfRec), hipMemcpyDeviceToHost));
                std::vector<OvfItem> items;
                items.reserve(n_ovf);
                std::vector<uint32_t> bad;
                uint32_t n_untried = 0;
                // (read once: inside the loop these were 2 x 16384 walks through the environment — with k_ovf_scatter's launch
                // 0.55 ms behind the read-back on configs[2] with its errors left in, profiles/r04_final/errors_left_in_timeline.txt)
                const unsigned long long fill_pct = env_u64("SHK_OVF_FILL_PCT", 110), cap_pct = env_u64("SHK_OVF_CAP_PCT", bloom ? 100 : 150);
                for (uint32_t i = 0; i < n_ovf; i++) {
                    // A partition of >= 2^32 instances may hold a k-mer whose count saturates (SPEC S4).  The bucket path
                    // counts without the saturating add and keeps 32-bit bucket cursors: such a giant is re-run by residue
                    // classes in k_count_partitions instead, whose inserts saturate (R >= 2^26 there).
                    if (ov[i].instances >= 0xFFFFFFF0ull) { bad.push_back(ov[i].p); continue; }
                    items.emplace_back();
                    OvfItem &item = items.back();
                    // buckets sized by INSTANCES (1.1 table sizes each, any F): the distinct/instance estimate of the
                    // aborted round is biased high (repeats show up late), and a bucket that turns out to hold
                    // too many distinct k-mers only costs itself a second pass over its own k-mer list
                    // (about half of an error-rich bucket's k-mers are distinct: a table at ~45 % keeps the linear
                    // probe chains far below the 48-probe limit; at 60 % one bucket in a few hit it and re-ran by residue classes)
                    const unsigned long long per = (unsigned long long)S * fill_pct / 100;
                    const uint32_t F = (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>((ov[i].instances + per - 1) / per, 2ull), OVF_MAX_F);
                    if (ov[i].est_distinct == 0) n_untried++;
                    // (Bloom mode: at least one sighting per distinct k-mer never reaches the buckets — error-rich
                    // partitions are mostly singletons — so the regions start at the instance count, not 1.5 x it)
                    const unsigned long long capb = ov[i].instances * cap_pct / (100ull * F) + 256;   // 50 % slack
                    item.p = ov[i].p; item.F = F; item.cap = (uint32_t)std::min<unsigned long long>(capb, 0xFFFFFFF0ull);
                    item.pad = 0; item.base = 0;
                }
                // scatter + count; an item whose bucket region overflows (a Poisson tail of heavy k-mers in one
                // bucket) is scattered again with twice the room, at most three times, then re-run by residue classes
                size_t n_good_total = 0;
                const int max_passes = (int)env_u64("SHK_OVF_MAX_PASSES", 4);
                for (int pass = 0; pass < max_passes && !items.empty(); pass++) {
                    const uint32_t ni = (uint32_t)items.size();
                    unsigned long long tot = 0;
                    for (auto &it : items) { it.base = tot; tot += (unsigned long long)it.F * it.cap; }
                    if (int rc = d_items.alloc(ni, err)) return rc;
                    if (int rc = d_fill.alloc((size_t)ni * OVF_MAX_F, err)) return rc;
                    if (int rc = d_kmers.alloc(tot * W, err)) return rc;
                    HIPCHK(hipMemcpyAsync(d_items.p, items.data(), (size_t)ni * sizeof(OvfItem), hipMemcpyHostToDevice, stream_));
                    if (bloom) {
                        if (int rc = d_new.alloc(ni, err)) return rc;
                        hipLaunchKernelGGL((k_ovf_scatter<W, true>), dim3(ni), dim3(COUNT_THREADS), 0, stream_, rv, d_items.p, d_kmers.p, d_fill.p, d_new.p);
                    } else {
                        hipLaunchKernelGGL((k_ovf_scatter<W, false>), dim3(ni), dim3(COUNT_THREADS), 0, stream_, rv, d_items.p, d_kmers.p, d_fill.p, (uint32_t *)nullptr);
                    }
                    HIPCHK(hipGetLastError());
                    if (int rc = d_maxfill.alloc(ni, err)) return rc;
                    unsigned long long n_buckets_ub = 0;
                    for (auto &it : items) n_buckets_ub += it.F;
                    if (int rc = d_blist.alloc(n_buckets_ub, err)) return rc;
                    HIPCHK(fill2_async(ctl_.p + 4, sizeof(unsigned long long), 0u, nullptr, 0, 0u, stream_));
                    if (int rc = d_sumfill.alloc(ni, err)) return rc;
                    hipLaunchKernelGGL(k_ovf_check, dim3(grid_for(ni)), dim3(256), 0, stream_, d_items.p, d_fill.p, ni, d_maxfill.p,
                                       d_blist.p, (uint32_t *)(ctl_.p + 4), d_sumfill.p);
                    HIPCHK(hipGetLastError());
                    std::vector<uint32_t> mxf(ni), newc(bloom ? ni : 0);
                    std::vector<unsigned long long> smf(ni);
                    unsigned long long n_list = 0;
                    HIPCHK(hipMemcpyAsync(smf.data(), d_sumfill.p, (size_t)ni * 8, hipMemcpyDeviceToHost, stream_));
                    if (bloom) HIPCHK(hipMemcpyAsync(newc.data(), d_new.p, (size_t)ni * 4, hipMemcpyDeviceToHost, stream_));
                    HIPCHK(hipMemcpyAsync(mxf.data(), d_maxfill.p, (size_t)ni * 4, hipMemcpyDeviceToHost, stream_));
                    HIPCHK(hipMemcpyAsync(&n_list, ctl_.p + 4, sizeof n_list, hipMemcpyDeviceToHost, stream_));
                    WAIT_STREAM();
                    std::vector<OvfItem> again;
                    uint32_t n_good = 0;
                    for (uint32_t i = 0; i < ni; i++) {
                        if (mxf[i] <= items[i].cap) {
                            n_good++;
                            if (bloom) {                               // (statistics + the histogram's singleton bin)
                                bloom_new += newc[i];
                                bloom_kmer_bytes += smf[i] * 8ull * W;       // k-mer instances that still went to HBM
                            }
                        }
                        else if (pass + 1 < max_passes && (unsigned long long)mxf[i] + 256 < 0xFFFFFFF0ull) {
                            OvfItem it = items[i]; it.cap = mxf[i] + 256; again.push_back(it);       // the exact need is known now
                        } else bad.push_back(items[i].p);
                    }
                    if ((uint32_t)n_list) {
                        // persistent workgroups, two per CU (the k-mer table is half the LDS)
                        const uint32_t bgrid = (uint32_t)std::min<unsigned long long>(n_list, 2ull * (unsigned long long)n_cus_);
                        hipLaunchKernelGGL(k_count_buckets<W>, dim3(bgrid), dim3(COUNT_THREADS), 0, stream_,
                                           d_blist.p, (uint32_t)n_list, d_kmers.p, threshold, dh.p, ok, cnt.p, (unsigned long long)cap,
                                           ctl_.p + 0, ctl_.p + 1, (uint32_t *)(ctl_.p + 2), env_dbg("SHK_DEBUG_B"),
                                           bloom ? 1u : 0u, ctl_.p + 9);
                        HIPCHK(hipGetLastError());
                        WAIT_STREAM();      // d_items / d_kmers are reused by the next pass
                    }
                    n_good_total += n_good;
                    items.swap(again);
                }
                if (!bad.empty()) {
                    if (int rc = d_list.alloc(bad.size(), err)) return rc;
                    HIPCHK(hipMemcpyAsync(d_list.p, bad.data(), bad.size() * 4, hipMemcpyHostToDevice, stream_));
                    hipLaunchKernelGGL(k_count_partitions<W>, dim3(std::min<uint32_t>((uint32_t)bad.size(), (uint32_t)n_cus_)), dim3(COUNT_THREADS), 0, stream_, rv, (uint32_t)bad.size(), threshold,
                                       dh.p, ok, cnt.p, (unsigned long long)cap, ctl_.p + 0, ctl_.p + 1,
                                       (uint32_t *)(ctl_.p + 2), (const uint32_t *)d_list.p, (OvfRec *)nullptr, (uint32_t *)nullptr, (uint32_t *)(ctl_.p + 12), 0u, 0u);
                    HIPCHK(hipGetLastError());
                }
                ms_out += t2.stop();
                times_.add("count_repartitioned_x1", (double)n_good_total);
                times_.add("count_deferred_untried_x1", (double)n_untried);
                times_.add("count_residue_rerun_x1", (double)bad.size());
                HIPCHK(hipMemcpyAsync(h, ctl_.p, sizeof h, hipMemcpyDeviceToHost, stream_));
                HIPCHK(hipMemcpyAsync(hist_out, dh.p, 500 * 8, hipMemcpyDeviceToHost, stream_));
                WAIT_STREAM();
                times_.add("count_bucket_splits_x1", (double)(h[2] >> 32));
                if (bloom) {
                    // k-mers the filter took as new: each is one distinct k-mer (less the false positives) and one
                    // instance that never reached a table.  Those that came back later sit in the tables (n_keys);
                    // the rest were seen once: the histogram's first bin (SPEC S4/S5, Bloom mode: statistical)
                    unsigned long long n_keys = 0;
                    HIPCHK(hipMemcpy(&n_keys, ctl_.p + 9, 8, hipMemcpyDeviceToHost));
                    const unsigned long long singles = bloom_new > n_keys ? bloom_new - n_keys : 0ull;
                    hist_out[0] += singles;
                    h[1] += bloom_new;
                    for (uint32_t i = 0; i < n_ovf; i++) bloom_kmer_bytes_exact += ov[i].instances * 8ull * W;
                    times_.add("bloom_new_kmers_x1e-6", (double)bloom_new * 1e-6);
                    times_.add("bloom_singletons_never_stored_x1e-6", (double)singles * 1e-6);
                    times_.add("bloom_kmer_instances_MB_written", (double)bloom_kmer_bytes / 1e6);
                    times_.add("bloom_kmer_instances_MB_without_filter", (double)bloom_kmer_bytes_exact / 1e6);
                }
            }
            if ((uint32_t)h[2]) { err = "partition too large for the LDS table even after 4096-way residue splitting"; return -6; }
            n_rows = h[0]; inst_out = h[1];
            if (n_rows <= cap) {
                return 0;
            }
            cap = bloom ? n_rows + n_rows / 16 + 4096 : n_rows;   // exact (Bloom mode: see above); run again
        }
        err = "row buffer overflowed twice";
        return -6;
    }

    int histogram(uint64_t histo[500], uint32_t emit_threshold, std::string &err) override {
        if (!global_mode_) {
            memset(histo, 0, 500 * 8);
            n_distinct_ = 0; n_emitted_ = 0; emit_threshold_ = emit_threshold;
            if (!batches_.empty()) {
                if (have_parts_ && recs_.p) if (int rc = pack_current_batch(err)) return rc;
                if (int rc = make_batch_run_view(err)) return rc;
            }
            if (have_parts_) {
                const uint64_t inst_ub_rows = total_rows_hint();
                double ms = 0; uint64_t inst = 0;
                // where k_dedupe_partitions may put the distinct records of partition p (two-kernel pass 2): at the place
                // of p's own raw records — the slice region of p, or the sum of the batches' shares
                split_base_.assign(n_count_parts_, 0); split_total_ = 0; split_stride_ = 0;
                if (batches_.empty()) {
                    split_stride_ = (unsigned long long)pp_.G * pp_.slice_cap;
                    for (uint32_t p = 0; p < n_count_parts_; p++) split_base_[p] = (unsigned long long)p * pp_.G * pp_.slice_cap;
                    split_total_ = (unsigned long long)n_count_parts_ * pp_.G * pp_.slice_cap;
                } else {
                    for (uint32_t p = 0; p < n_count_parts_; p++) {
                        split_base_[p] = split_total_;
                        for (auto &b : batches_) split_total_ += b->part_off[p + 1] - b->part_off[p];
                    }
                }
                split_ready_ = true;
                int rc_p2 = run_count_partitions(run_view_, n_count_parts_, emit_threshold, ekeys_, ecnt_, n_emitted_, histo, inst,
                                                 inst_ub_rows, ms, err);
                if (p1_.on) {
                    // pass 1's flags arrived with pass 2's read-back (count_batch_impl): look at them now
                    if (rc_p2) { std::string e2; (void)e2; (void)hipStreamSynchronize(stream_); }
                    p1_.on = false;
                    const unsigned long long *h = mbox64() + MB_P1;
                    const uint32_t *fl = (const uint32_t *)&h[0];
                    if (fl[1]) { split_ready_ = false; err = "a read segment exceeds 32768 bases (split it on the host)"; return -1; }
                    if ((uint32_t)h[1] > p1_.cap) {
                        // a slice overflowed: both passes again, pass 1 with the exact room (the reads are still on the device)
                        times_.add("partition_retry", 1.0);
                        cap_override_ = (uint64_t)(uint32_t)h[1] + 8;
                        have_parts_ = false; split_ready_ = false;
                        const bool d = defer_p1_; defer_p1_ = false;
                        const int rc1 = count_batch_impl(p1_.d_bases, p1_.d_seg_off, nullptr, nullptr, p1_.n_seg, p1_.n_bases, err);
                        defer_p1_ = d;
                        if (rc1) return rc1;
                        return histogram(histo, emit_threshold, err);
                    }
                }
                if (rc_p2) { split_ready_ = false; return rc_p2; }
                split_ready_ = false;
                times_.add("count_kernel", ms);
                total_instances_ = inst;
            }
            for (int i = 0; i < 500; i++) { histo_[i] = histo[i]; n_distinct_ += histo[i]; }
            return 0;
        }
        DevBuf<unsigned long long> dh;
        if (int rc = dh.alloc(500, err)) return rc;
        HIPCHK(hipMemsetAsync(dh.p, 0, 500 * 8, stream_));
        if (tslots_) {
            EvTimer t(stream_, stage_timers_);
            hipLaunchKernelGGL(k_histogram<W>, dim3(grid_for(tslots_)), dim3(256), 0, stream_, table_view(),
                               tslots_, dh.p);
            HIPCHK(hipGetLastError());
            if (t.on()) times_.add("histogram_kernel", t.stop());
        }
        HIPCHK(hipMemcpyAsync(histo, dh.p, 500 * 8, hipMemcpyDeviceToHost, stream_));
        WAIT_STREAM();
        n_distinct_ = 0;
        for (int i = 0; i < 500; i++) { histo_[i] = histo[i]; n_distinct_ += histo[i]; }
        return 0;
    }

    uint64_t total_rows_hint() const {
        // rows = distinct k-mers above the emit threshold; unknown before the pass, retried if short
        uint64_t slots = (uint64_t)pp_.P * pp_.G * pp_.slice_cap;         // records >= rows / max_n
        if (!batches_.empty()) { slots = 0; for (auto &b : batches_) slots += b->part_off[pp_.P]; }
        return std::max<uint64_t>(1u << 16, slots / 8);
    }

    int compact_into(uint32_t threshold, uint64_t expect, DevBuf<uint64_t> (&keys)[W], DevBuf<uint32_t> &cnt,
                     std::string &err) {
        for (int j = 0; j < W; j++) if (int rc = keys[j].alloc(expect, err)) return rc;
        if (int rc = cnt.alloc(expect, err)) return rc;
        if (!tslots_ || !expect) return 0;
        HIPCHK(hipMemsetAsync(ctl_.p, 0, CTL_WORDS * sizeof(unsigned long long), stream_));
        KeyArr<W> ok; for (int j = 0; j < W; j++) ok.w[j] = keys[j].p;
        hipLaunchKernelGGL(k_compact<W>, dim3(grid_for(tslots_)), dim3(256), 0, stream_, table_view(), tslots_,
                           threshold, ok, cnt.p, ctl_.p + 0);
        HIPCHK(hipGetLastError());
        unsigned long long got = 0;
        HIPCHK(hipMemcpyAsync(&got, ctl_.p, 8, hipMemcpyDeviceToHost, stream_));
        WAIT_STREAM();
        if (got != expect) { err = "compaction count mismatch"; return -6; }
        return 0;
    }

    int filter(uint32_t threshold, std::string &err) override {
        uint64_t expect = 0;
        if (threshold >= 500) { err = "threshold out of range"; return -1; }
        for (uint32_t c = 1; c <= 500; c++) if (c > threshold) expect += histo_[c - 1];
        if (expect >= 0x7FFFFFFFull) { err = "too many solid k-mers for 32-bit node ids"; return -1; }
        EvTimer t(stream_, stage_timers_);
        if (global_mode_) {
            if (int rc = compact_into(threshold, expect, skeys_, scnt_, err)) return rc;
        } else if (threshold == emit_threshold_ || n_emitted_ == 0) {
            if (n_emitted_ != expect && threshold == emit_threshold_) { err = "emitted row count disagrees with the histogram"; return -6; }
            for (int j = 0; j < W; j++) skeys_[j].swap(ekeys_[j]);
            scnt_.swap(ecnt_);
            for (int j = 0; j < W; j++) ekeys_[j].release();
            ecnt_.release();
        } else {
            if (threshold < emit_threshold_) { err = "filter threshold below the emit threshold"; return -6; }
            for (int j = 0; j < W; j++) if (int rc = skeys_[j].alloc(expect, err)) return rc;
            if (int rc = scnt_.alloc(expect, err)) return rc;
            HIPCHK(hipMemsetAsync(ctl_.p, 0, CTL_WORDS * sizeof(unsigned long long), stream_));
            KeyArr<W> ik, ok;
            for (int j = 0; j < W; j++) { ik.w[j] = ekeys_[j].p; ok.w[j] = skeys_[j].p; }
            hipLaunchKernelGGL(k_compact_rows<W>, dim3(grid_for(n_emitted_)), dim3(256), 0, stream_, ik, ecnt_.p,
                               n_emitted_, threshold, ok, scnt_.p, ctl_.p + 0);
            HIPCHK(hipGetLastError());
            unsigned long long got = 0;
            HIPCHK(hipMemcpyAsync(&got, ctl_.p, 8, hipMemcpyDeviceToHost, stream_));
            WAIT_STREAM();
            if (got != expect) { err = "row compaction count mismatch"; return -6; }
            for (int j = 0; j < W; j++) ekeys_[j].release();
            ecnt_.release();
        }
        if (t.on()) times_.add("filter_kernel", t.stop());
        n_solid_ = expect;
        graph_ready_ = false;
        return 0;
    }

    int copy_out(DevBuf<uint64_t> (&keys)[W], DevBuf<uint32_t> &cnt, uint64_t n, uint64_t *hk, uint32_t *hc,
                 std::string &err) {
        std::vector<uint64_t> tmp(n ? n : 1);
        for (int j = 0; j < W; j++) {
            if (n) HIPCHK(hipMemcpy(tmp.data(), keys[j].p, n * 8, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < n; i++) hk[i * W + j] = tmp[i];
        }
        if (n) HIPCHK(hipMemcpy(hc, cnt.p, n * 4, hipMemcpyDeviceToHost));
        return 0;
    }

    int get_distinct(uint64_t *keys, uint32_t *counts, uint64_t cap, std::string &err) override {
        if (cap < n_distinct_) { err = "buffer too small"; return -1; }
        DevBuf<uint64_t> dk[W]; DevBuf<uint32_t> dc;
        if (global_mode_) {
            if (int rc = compact_into(0, n_distinct_, dk, dc, err)) return rc;
        } else if (n_distinct_) {
            // stage inspection: run the counting pass again keeping every row
            if (!have_parts_ || (!recs_.p && batches_.empty())) { err = "partition buffers already released"; return -2; }
            uint64_t rows = 0, hist[500], inst = 0; double ms = 0;
            // (Bloom mode never stores its singletons: the inspection runs the exact counter and reports exact counts)
            bloom_off_once_ = true;
            const int rc = run_count_partitions(run_view_, n_count_parts_, 0, dk, dc, rows, hist, inst, n_distinct_, ms, err);
            bloom_off_once_ = false;
            if (rc) return rc;
            if (bloom_) { if (cap < rows) { err = "buffer too small"; return -1; } return copy_out(dk, dc, rows, keys, counts, err); }
            if (rows != n_distinct_) { err = "distinct row count mismatch"; return -6; }
        }
        return copy_out(dk, dc, n_distinct_, keys, counts, err);
    }
    int get_solid(uint64_t *keys, uint32_t *counts, uint64_t cap, std::string &err) override {
        if (cap < n_solid_) { err = "buffer too small"; return -1; }
        return copy_out(skeys_, scnt_, n_solid_, keys, counts, err);
    }

    // ---- shard layer (one process per GPU; DESIGN.md "Multi-GPU") --------------------------------
    int shard_partition(const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg, uint64_t n_bases,
                        uint32_t n_partitions, std::vector<uint64_t> &part_records, std::string &err) override {
        if (global_mode_) { err = "shard layer needs the partitioned counting mode"; return -1; }
        if (n_partitions < 1 || n_partitions > (uint32_t)PART_MAX_P || (n_partitions & (n_partitions - 1))) {
            err = "n_partitions must be a power of two <= 16384"; return -1;
        }
        forced_P_ = n_partitions;
        part_records.assign(n_partitions, 0);
        if (int rc = count_batch(d_bases, d_seg_off, n_seg, n_bases, err)) return rc;
        if (!have_parts_) return 0;                       // no segments on this rank
        DevBuf<unsigned long long> tot;
        if (int rc = tot.alloc(pp_.P, err)) return rc;
        hipLaunchKernelGGL(k_part_totals, dim3(pp_.P), dim3(256), 0, stream_, fill_.p, pp_, tot.p);
        HIPCHK(hipGetLastError());
        std::vector<unsigned long long> h(pp_.P);
        HIPCHK(hipMemcpyAsync(h.data(), tot.p, (size_t)pp_.P * 8, hipMemcpyDeviceToHost, stream_));
        WAIT_STREAM();
        for (uint32_t p = 0; p < pp_.P; p++) part_records[p] = h[p];
        return 0;
    }
    uint32_t rec_words() const override { return 2 * W; }

    int shard_pack(void *d_send, const uint64_t *base_records, uint32_t n_partitions, std::string &err) override {
        if (!have_parts_) return 0;
        if (n_partitions != pp_.P) { err = "partition count mismatch"; return -1; }
        DevBuf<unsigned long long> base;
        if (int rc = base.alloc(pp_.P, err)) return rc;
        HIPCHK(hipMemcpyAsync(base.p, base_records, (size_t)pp_.P * 8, hipMemcpyHostToDevice, stream_));
        EvTimer t(stream_, stage_timers_);
        hipLaunchKernelGGL((k_pack_partition<2 * W>), dim3(pp_.P), dim3(256), 0, stream_, recs_.p, fill_.p, pp_, base.p,
                           (uint64_t *)d_send);
        HIPCHK(hipGetLastError());
        if (t.on()) times_.add("shard_pack_kernel", t.stop());
        WAIT_STREAM();
        // the local slices are no longer needed once packed
        recs_.release(); fill_.release(); run_off_.release(); run_cnt_.release(); have_parts_ = false;
        return 0;
    }

    // Records deduplicated on the sender (k_dedupe_partitions): part_records comes in with the raw record counts of
    // shard_partition and goes out with the numbers of distinct records; the records wait in dd_recs_ / dd_w_ for
    // shard_pack_dedup (which releases the slices) or shard_drop_dedup (the raw records travel: shard_pack).
    int shard_dedupe(std::vector<uint64_t> &part_records, std::string &err) override {
        if (!have_parts_) return 0;
        if (part_records.size() < pp_.P) { err = "partition count mismatch"; return -1; }
        std::vector<unsigned long long> base(pp_.P);
        unsigned long long n_raw = 0;
        for (uint32_t p = 0; p < pp_.P; p++) { base[p] = n_raw; n_raw += part_records[p]; }
        if (int rc = dd_base_.alloc(pp_.P, err)) return rc;
        if (int rc = dd_n_.alloc(pp_.P, err)) return rc;
        if (int rc = dd_recs_.alloc(n_raw * 2 * W + 2, err)) return rc;
        if (int rc = dd_w_.alloc(n_raw + 2, err)) return rc;
        HIPCHK(hipMemcpyAsync(dd_base_.p, base.data(), (size_t)pp_.P * 8, hipMemcpyHostToDevice, stream_));
        HIPCHK(fill2_async(ctl_.p + 12, 8, 0u, nullptr, 0, 0u, stream_));
        EvTimer t(stream_, stage_timers_);
        // (two workgroups per CU where the kernel's registers allow it, as on one GPU: with one, 588 us against 418 for the
        // bench isolate — found in the timeline of the one-rank leg, profiles/r04_final/sharded_one_rank_timeline.txt)
        hipLaunchKernelGGL(k_dedupe_partitions<W>, dim3(std::min<uint32_t>(pp_.P, (W <= 2 ? 2u : 1u) * (uint32_t)n_cus_)), dim3(COUNT_THREADS), 0, stream_,
                           run_view_, 0u, pp_.P, dd_base_.p, dd_recs_.p, dd_w_.p, dd_n_.p, (uint32_t *)(ctl_.p + 12),
                           (OvfRec *)nullptr, (uint32_t *)nullptr, 0u, 0u, (uint32_t *)nullptr);
        HIPCHK(hipGetLastError());
        t.mark();
        std::vector<uint32_t> h(pp_.P);
        HIPCHK(hipMemcpyAsync(h.data(), dd_n_.p, (size_t)pp_.P * 4, hipMemcpyDeviceToHost, stream_));
        WAIT_STREAM();
        times_.add("shard_dedupe_kernel", t.elapsed());
        unsigned long long n_dd = 0;
        for (uint32_t p = 0; p < pp_.P; p++) { part_records[p] = h[p]; n_dd += h[p]; }
        times_.add("shard_dedupe_records_in_x1e-6", (double)n_raw * 1e-6);
        times_.add("shard_dedupe_records_out_x1e-6", (double)n_dd * 1e-6);
        have_dedup_ = true;                              // (the slices stay until the caller has decided what travels)
        return 0;
    }
    void shard_drop_dedup() override { dd_recs_.release(); dd_w_.release(); dd_base_.release(); dd_n_.release(); have_dedup_ = false; }
    int shard_pack_dedup(void *d_send, void *d_send_w, const uint64_t *base_records, uint32_t n_partitions, std::string &err) override {
        if (!have_dedup_) return 0;
        if (n_partitions != pp_.P) { err = "partition count mismatch"; return -1; }
        DevBuf<unsigned long long> base;
        if (int rc = base.alloc(pp_.P, err)) return rc;
        HIPCHK(hipMemcpyAsync(base.p, base_records, (size_t)pp_.P * 8, hipMemcpyHostToDevice, stream_));
        EvTimer t(stream_, stage_timers_);
        hipLaunchKernelGGL((k_pack_dedup<2 * W>), dim3(pp_.P), dim3(256), 0, stream_, dd_recs_.p, dd_w_.p, dd_base_.p, dd_n_.p, base.p,
                           (uint64_t *)d_send, (uint32_t *)d_send_w);
        HIPCHK(hipGetLastError());
        if (t.on()) times_.add("shard_pack_kernel", t.stop());
        WAIT_STREAM();
        shard_drop_dedup();
        recs_.release(); fill_.release(); run_off_.release(); run_cnt_.release(); have_parts_ = false;
        return 0;
    }

    // d_recv: records received from all sources (d_recv_w: their weights when the sources deduplicated them, else null);
    // run tables [n_owned][n_sources] on the host
    int shard_count(const void *d_recv, const void *d_recv_w, const uint64_t *run_off, const uint32_t *run_cnt, uint32_t n_owned,
                    uint32_t n_sources, uint32_t emit_threshold, uint64_t histo[500], std::string &err) override {
        if (n_sources < 1 || n_sources > 256) { err = "1..256 sources"; return -1; }
        const uint64_t n_runs = (uint64_t)n_owned * n_sources;
        if (int rc = run_off_.alloc(n_runs, err)) return rc;
        if (int rc = run_cnt_.alloc(n_runs, err)) return rc;
        std::vector<unsigned long long> addr16(n_runs);
        for (uint64_t i = 0; i < n_runs; i++) addr16[i] = ((unsigned long long)(uintptr_t)d_recv >> 4) + run_off[i] * (unsigned long long)W;   // RW/2 = W
        if (n_runs) {
            HIPCHK(hipMemcpyAsync(run_off_.p, addr16.data(), n_runs * 8, hipMemcpyHostToDevice, stream_));
            HIPCHK(hipMemcpyAsync(run_cnt_.p, run_cnt, n_runs * 4, hipMemcpyHostToDevice, stream_));
        }
        shard_recv_ = d_recv;
        run_view_.run_addr16 = run_off_.p; run_view_.run_cnt = run_cnt_.p;
        run_view_.S = n_sources; run_view_.k = k_; n_count_parts_ = n_owned; run_view_.dbg = 0;
        run_view_.weights = (const uint32_t *)d_recv_w; run_view_.rec_base16 = (unsigned long long)(uintptr_t)d_recv >> 4;
        uint64_t total_recs = 0;
        for (uint32_t j = 0; j < n_owned; j++) {          // a partition's record index is 32 bits wide in pass 2
            uint64_t r = 0;
            for (uint32_t s = 0; s < n_sources; s++) r += run_cnt[(uint64_t)j * n_sources + s];
            if (r > 0xFFFFFFF0ull) { err = "a partition holds more than 2^32 records"; return -1; }
            total_recs += r;
        }
        memset(histo, 0, 500 * 8);
        n_emitted_ = 0; emit_threshold_ = emit_threshold; n_distinct_ = 0;
        double ms = 0; uint64_t inst = 0;
        have_parts_ = n_runs != 0;
        // Room for the rows: raw records are at least as many as the k-mers they bring, DEDUPLICATED ones are not — the bench
        // isolate's 4.88 M distinct records hold 5.0 M distinct k-mers, so a hint of one row per record made every sharded
        // step count twice (0.46 ms each: the second attempt knows the exact number).  Weights travel only when the records
        // repeat a lot (at most half of the raw ones are distinct), i.e. at coverages where rows per record stay near one.
        const uint64_t rows_hint = d_recv_w ? 2 * total_recs + (1u << 16) : std::max<uint64_t>(1u << 16, total_recs);
        if (int rc = run_count_partitions(run_view_, n_owned, emit_threshold, ekeys_, ecnt_, n_emitted_, histo, inst,
                                          rows_hint, ms, err)) return rc;
        times_.add("count_kernel", ms);
        total_instances_ = inst;
        for (int i = 0; i < 500; i++) { histo_[i] = histo[i]; n_distinct_ += histo[i]; }
        return 0;
    }

    // local rows with count > threshold; device pointers stay owned by the pipeline
    int shard_rows(uint32_t threshold, const void **keys_soa, const void **cnt, uint64_t *n, std::string &err) override {
        if (int rc = filter(threshold, err)) return rc;
        for (int j = 0; j < W; j++) keys_soa[j] = skeys_[j].p;
        *cnt = scnt_.p; *n = n_solid_;
        return 0;
    }

    // install the gathered solid set (every rank holds all of it) and the global statistics
    int shard_set_solid(const void *const *keys_soa, const void *cnt, uint64_t n, const uint64_t histo[500],
                        uint64_t total_instances, std::string &err) override {
        if (n >= 0x7FFFFFFFull) { err = "too many solid k-mers for 32-bit node ids"; return -1; }
        DevBuf<uint64_t> nk[W]; DevBuf<uint32_t> nc;
        for (int j = 0; j < W; j++) {
            if (int rc = nk[j].alloc(n, err)) return rc;
            if (n) HIPCHK(hipMemcpyAsync(nk[j].p, keys_soa[j], n * 8, hipMemcpyDeviceToDevice, stream_));
        }
        if (int rc = nc.alloc(n, err)) return rc;
        if (n) HIPCHK(hipMemcpyAsync(nc.p, cnt, n * 4, hipMemcpyDeviceToDevice, stream_));
        WAIT_STREAM();
        for (int j = 0; j < W; j++) skeys_[j].swap(nk[j]);
        scnt_.swap(nc);
        n_solid_ = n; total_instances_ = total_instances; n_distinct_ = 0;
        for (int i = 0; i < 500; i++) { histo_[i] = histo[i]; n_distinct_ += histo[i]; }
        graph_ready_ = false;
        return 0;
    }

    // ---- graph -----------------------------------------------------------------------------
    Graph<W> graph_view() {
        Graph<W> g;
        for (int j = 0; j < W; j++) g.keys.w[j] = skeys_[j].p;
        g.cnt = scnt_.p; g.adj = adj_.p; g.nb = nb_.p; g.k = k_;
        g.gt.e = gt_.p; g.gt.occ = gt_occ_.p; g.gt.scan = gt_scan_.p; g.gt.scan_n = gt_scan_.p ? (uint32_t)n_solid_ : 0u; g.gt.off = gt_off_.p; g.gt.msk = gt_msk_.p; g.gt.gp_mask = gp_ - 1u; g.gt.gm = part_m(k_); g.gt.dbg = env_dbg("SHK_DEBUG_G");
        g.n = (uint32_t)n_solid_;
        if (sh_active_) {                                   // sharded assembly: which rank owns a neighbour candidate
            g.gt.cp_mask = sh_P_ - 1u; g.gt.world = sh_world_; g.gt.rank = sh_rank_;
            g.gt.world_inv = (uint32_t)((0x100000000ull + sh_world_ - 1u) / sh_world_);
            g.xref = XREF;
        }
        return g;
    }

    int build_graph(std::string &err) override {
        const uint64_t n = n_solid_;
        // count table is no longer needed once the solid set exists
        for (int j = 0; j < W; j++) tkeys_[j].release();
        tcnt_.release(); tstate_.release(); tslots_ = 0;
        recs_.release(); fill_.release(); run_off_.release(); run_cnt_.release(); shard_recv_ = nullptr; batches_.clear();
        // graph partitions: 320-640 rows each on average (mini tables of <= 2048 slots fit 16 KB of LDS); the minimiser length is the counting pass's, so rows that
        // arrive grouped by counting partition are grouped by graph partition too
        gp_ = 64;
        // (measured, `profiles/r02_frag/graph_partition_rows.txt`: 610 rows per partition on average beat 305 — fewer workgroups, the
        // same fixed cost each — while 790 and 980 lose to 400 and 490: partitions above 1024 rows need a table beyond the LDS
        // one and work in global memory.  The average ends up in (320, 640].)
        const uint64_t gp_rows_target = env_u64("SHK_GP_ROWS", 640);
        // (sharded assembly: every rank must cut the k-mer space into the same graph partitions — from the global node count)
        const uint64_t n_for_gp = sh_active_ ? n_solid_global_ : n;
        while (gp_ < 131072u && (uint64_t)gp_ * gp_rows_target < n_for_gp) gp_ <<= 1;
        gt_slots_ = 4 * n + 8ull * gp_;               // >= sum of max(8, pow2 >= 2 x rows)
        if (int rc = gt_.alloc(gt_slots_, err)) return rc;
        if (int rc = gt_occ_.alloc(gt_slots_ / 8 + 8, err)) return rc;
        // the nodes' minimiser scans, kept from k_gp_count for k_graph_local (24 B per node; not for graphs beyond 64 M nodes,
        // and dropped when the rows are regrouped in between: k_graph_local then repeats the scan as before)
        gt_scan_.release();
        if (n && n <= (64ull << 20) && env_u64("SHK_KEEP_SCAN", 1)) if (int rc = gt_scan_.alloc(3 * n, err)) return rc;
        if (int rc = gt_off_.alloc(gp_, err)) return rc;
        if (int rc = gt_msk_.alloc(gp_, err)) return rc;
        DevBuf<uint32_t> gp_of, gp_cnt, gp_roff, gp_rows;
        if (int rc = gp_of.alloc(n, err)) return rc;
        if (int rc = gp_cnt.alloc(gp_, err)) return rc;
        if (int rc = gp_roff.alloc(gp_ + 1, err)) return rc;
        if (int rc = gp_rows.alloc(n, err)) return rc;
        DevBuf<unsigned long long> queries;              // 8 slots per row, grouped by partition; only a prefix is touched
        if (int rc = queries.alloc(8 * n + 8, err)) return rc;
        if (int rc = adj_.alloc((n + 8) & ~3ull, err)) return rc;
        if (keep_stages_) if (int rc = adj0_.alloc(n, err)) return rc;
        if (int rc = nb_.alloc(2 * n + 2, err)) return rc;
        if (int rc = alive_.alloc(n, err)) return rc;
        if (int rc = row_starts_.alloc(((n + 63) / 64) * 2 + 2, err)) return rc;
        // (the mini tables are initialised by their builders; no fills for adj_ and alive_: k_graph_local writes the adjacency
        // byte of every row before k_graph_remote ORs into it, k_row_starts sets the alive flags)
        HIPCHK(fill2_async(gp_cnt.p, (size_t)gp_ * 4, 0u, ctl_.p, CTL_WORDS * sizeof(unsigned long long), 0u, stream_));
        // (sharded assembly: a rank that holds NO solid k-mer still owns partitions and is asked about neighbour candidates by
        // the others — its (empty) mini tables must exist: found by the 250-case campaign on 4 ranks, where such a rank answered
        // from tables nobody had built and took a memory fault)
        if (n || sh_active_) {
            Graph<W> g = graph_view();
            EvTimer t(stream_, stage_timers_);
            hipLaunchKernelGGL(k_gp_count<W>, dim3(grid_for(n)), dim3(256), 0, stream_, g.keys, (uint32_t)n, k_, g.gt,
                               gp_of.p, gp_cnt.p);
            hipLaunchKernelGGL(k_gp_scan, dim3(1), dim3(1024), 0, stream_, gp_cnt.p, gp_, gt_off_.p, gt_msk_.p, gp_roff.p,
                               ctl_.p + 2);
            // (gp_cnt is reused as the row-list cursors: k_gp_scan left it zeroed)
            hipLaunchKernelGGL(k_gp_rows, dim3(grid_for(n)), dim3(256), 0, stream_, gp_of.p, (uint32_t)n, gp_roff.p, gp_cnt.p,
                               gp_rows.p);
            HIPCHK(hipGetLastError());
            const uint64_t regroup_env = env_u64("SHK_REGROUP_ROWS", 2);        // 0 never, 1 always, 2 when the rows are scattered
            if (n && (regroup_env == 1 || (regroup_env == 2 && rows_scattered_))) {
                // The rows are not grouped by minimiser partition (they came out of the bucket path in key-hash order): move them
                // into the order of the row lists once.  Every later pass finds a partition's rows side by side again — the key
                // reads of k_graph_local coalesce, the LDS tiles of the collapse hold neighbours — and the lists become the identity.
                DevBuf<uint64_t> nk[W]; DevBuf<uint32_t> nc, ngp;
                for (int j = 0; j < W; j++) if (int rc = nk[j].alloc(n, err)) return rc;
                if (int rc = nc.alloc(n, err)) return rc;
                if (int rc = ngp.alloc(n, err)) return rc;
                KeyArr<W> out_keys; for (int j = 0; j < W; j++) out_keys.w[j] = nk[j].p;
                hipLaunchKernelGGL(k_regroup_rows<W>, dim3(grid_for(n)), dim3(256), 0, stream_, g.keys, scnt_.p, gp_of.p, gp_rows.p, (uint32_t)n,
                                   out_keys, nc.p, ngp.p);
                HIPCHK(hipGetLastError());
                WAIT_STREAM();                          // (the old arrays go back to the pool below)
                for (int j = 0; j < W; j++) skeys_[j].swap(nk[j]);
                scnt_.swap(nc); gp_of.swap(ngp);
                gt_scan_.release();                     // (indexed by the old row numbers)
                g = graph_view();
                times_.add("graph_rows_regrouped_x1", 1.0);
            }
            // where a group of rows with the same low minimiser-hash bits starts: the tile edges of the collapse (collapse.h)
            hipLaunchKernelGGL(k_row_starts, dim3(grid_for(n)), dim3(256), 0, stream_, gp_of.p, (uint32_t)n, std::min<uint32_t>(gp_, 256u) - 1u,
                               row_starts_.p, alive_.p);
            t.stop_later("graph_table_kernel", pending_timers_);
            EvTimer t2(stream_, stage_timers_);
            hipLaunchKernelGGL(k_graph_local<W>, dim3(gp_), dim3(256), 0, stream_, g.keys, k_, g.gt, gp_roff.p, gp_rows.p,
                               adj_.p, nb_.p, queries.p, gp_cnt.p, (uint32_t *)(ctl_.p + 1));
            hipLaunchKernelGGL(k_graph_remote<W>, dim3(gp_), dim3(256), 0, stream_, g.keys, k_, g.gt, gp_roff.p, queries.p,
                               gp_cnt.p, adj_.p, nb_.p);
            HIPCHK(hipGetLastError());
            t2.stop_later("adjacency_kernel", pending_timers_);
            if (sh_active_ && sh_world_ > 1) {
                // the candidates that live on other ranks: staged compactly before `queries` goes (shard_cross_adjacency)
                HIPCHK(fill2_async(ctl_.p + 13, 8, 0u, nullptr, 0, 0u, stream_));
                hipLaunchKernelGGL(k_xq_total, dim3(gp_), dim3(256), 0, stream_, gp_roff.p, queries.p, gp_cnt.p, (unsigned int *)(ctl_.p + 13));
                unsigned int n_x = 0;
                if (int rc = read_ctl(n_x, 13, err)) return rc;
                xq_n_ = n_x;
                if (int rc = xq_dest_.alloc(n_x, err)) return rc;
                if (int rc = xq_pay_.alloc((size_t)n_x * (W + 1), err)) return rc;
                if (int rc = xq_meta_.alloc(n_x, err)) return rc;
                if (n_x) {
                    HIPCHK(fill2_async(ctl_.p + 13, 8, 0u, nullptr, 0, 0u, stream_));
                    hipLaunchKernelGGL(k_xq_stage<W>, dim3(gp_), dim3(256), 0, stream_, g.keys, k_, gp_roff.p, queries.p, gp_cnt.p,
                                       (unsigned int *)(ctl_.p + 13), n_x, xq_dest_.p, xq_pay_.p, xq_meta_.p);
                    HIPCHK(hipGetLastError());
                }
            }
            if (keep_stages_) HIPCHK(hipMemcpyAsync(adj0_.p, adj_.p, n, hipMemcpyDeviceToDevice, stream_));      // (stage inspection only)
            // (the overflow flags of the tables are read with the next counters that come back anyway — the correction's or the
            // collapse's: check_graph_flags(); the sharded assembly reads them here)
            graph_check_pending_ = true;
            if (sh_active_) {
                HIPCHK(hipMemcpyAsync(mbox64() + MB_CTL, ctl_.p, 3 * 8, hipMemcpyDeviceToHost, stream_));
                WAIT_STREAM();
                if (int rc = check_graph_flags(err)) return rc;
            }
            // (the big scratch of this phase goes back to the pool now, not when the entry point ends: see DeferScope)
            if (!sh_active_ || sh_world_ <= 1) { /* the stream is busy: parked until the call ends, or reused below */ }
        }
        graph_ready_ = true;
        return 0;
    }

    // mbox64()[MB_CTL ..] holds a fresh copy of ctl_[0..2]: the graph tables' overflow flags (build_graph)
    int check_graph_flags(std::string &err) {
        if (!graph_check_pending_) return 0;
        graph_check_pending_ = false;
        const unsigned long long *h = mbox64() + MB_CTL;
        if ((uint32_t)h[1] || h[2] > gt_slots_) { err = "graph table overflow"; return -6; }
        return 0;
    }
    int read_ctl(unsigned int &v, int slot, std::string &err) {
        unsigned long long h = 0;
        HIPCHK(hipMemcpyAsync(&h, ctl_.p + slot, 8, hipMemcpyDeviceToHost, stream_));
        WAIT_STREAM();
        v = (unsigned int)h;
        return 0;
    }

    // marked alive nodes -> removed list (count at ctl_[slot]) -> their edges cleared in the neighbours
    int apply_marks(Graph<W> &g, DevBuf<uint8_t> &mark, DevBuf<uint32_t> &removed, int slot, std::string &err) {
        const uint32_t n = (uint32_t)n_solid_;
        hipLaunchKernelGGL(k_collect_marked, dim3(grid_for(n)), dim3(256), 0, stream_, n, mark.p, alive_.p,
                           removed.p, (unsigned int *)(ctl_.p + slot));
        hipLaunchKernelGGL(k_apply_removed<W>, dim3(1024), dim3(256), 0, stream_, g, removed.p,
                           (const unsigned int *)(ctl_.p + slot));
        HIPCHK(hipGetLastError());
        return 0;
    }

    // SPEC S9.  The FIRST round is launched without waiting for its outcome: what it removed (ctl_[16], ctl_[17]) comes back
    // with the first counters of the collapse, whose first two kernels return at once when the round did remove something
    // (rank_chains) — an error-free isolate, where the round finds nothing, pays no host round trip for the correction.
    // Further rounds (reads with errors) run from finish_correction(), one read-back per round as before.
    int correct(bool tips, bool bubbles, std::string &err) override {
        if (!graph_ready_) { err = "graph not built"; return -2; }
        const uint32_t n = (uint32_t)n_solid_;
        tips_removed_ = bubbles_removed_ = 0; rounds_ = 0;
        corr_pending_ = false; corr_tips_ = tips; corr_bubbles_ = bubbles;
        if (n == 0 || (!tips && !bubbles)) return 0;
        if (int rc = corr_.cand.alloc(2ull * n, err)) return rc;
        if (int rc = corr_.removed.alloc(n, err)) return rc;
        if (int rc = corr_.mark.alloc(n, err)) return rc;
        if (!tips) HIPCHK(hipMemsetAsync(corr_.mark.p, 0, n, stream_));       // (with tips: k_tip_candidates clears it in the first round)
        if (tips) {
            // every candidate may turn out to be a tip: sized for all oriented nodes, so that no count has to
            // come back to the host inside a round (the counters live in ctl_: 3 candidates, 4 tips, 5/6 removed)
            if (int rc = corr_.tip_head.alloc(2ull * n, err)) return rc;
            if (int rc = corr_.tiprec.alloc(2ull * n, err)) return rc;
            if (int rc = corr_.kill.alloc(2ull * n, err)) return rc;
            // (tip_head := NIL by k_tip_candidates in the first round)
        }
        EvTimer t(stream_, stage_timers_);
        if (int rc = correction_round(0, err)) return rc;
        t.stop_later("correct_total", pending_timers_);
        corr_pending_ = true;
        return 0;
    }
    struct CorrScratch { DevBuf<uint32_t> cand, tip_head, removed; DevBuf<uint8_t> mark, kill; DevBuf<TipRec> tiprec;
                         void release() { cand.release(); tip_head.release(); removed.release(); mark.release(); kill.release(); tiprec.release(); } };
    // one round of S9 on the stream; removal counts go to ctl_[16], ctl_[17] (round 0) or ctl_[5], ctl_[6]
    int correction_round(int round, std::string &err) {
        const uint32_t n = (uint32_t)n_solid_;
        Graph<W> g = graph_view();
        const dim3 G(1024), B(256);
        const bool tips = corr_tips_, bubbles = corr_bubbles_;
        const int s_tip = round == 0 ? 16 : 5, s_bub = round == 0 ? 17 : 6;
        HIPCHK(fill2_async(ctl_.p + 3, 4 * 8, 0u, nullptr, 0, 0u, stream_));      // (one launch: a small hipMemsetAsync at an odd offset came out as three fill kernels)
        if (tips) {
            hipLaunchKernelGGL(k_tip_candidates<W>, dim3(grid_for(n)), B, 0, stream_, g, alive_.p,
                               corr_.cand.p, (unsigned int *)(ctl_.p + 3), round == 0 ? (uint2 *)corr_.tip_head.p : (uint2 *)nullptr,
                               round == 0 ? corr_.mark.p : (uint8_t *)nullptr);
            hipLaunchKernelGGL(k_tip_walk<W>, G, B, 0, stream_, g, corr_.cand.p, (const unsigned int *)(ctl_.p + 3),
                               corr_.tiprec.p, (unsigned int *)(ctl_.p + 4), corr_.tip_head.p);
            hipLaunchKernelGGL(k_tip_decide<W>, G, B, 0, stream_, g, corr_.tiprec.p, (const unsigned int *)(ctl_.p + 4),
                               corr_.tip_head.p, corr_.kill.p);
            hipLaunchKernelGGL(k_tip_remove<W>, G, B, 0, stream_, g, corr_.tiprec.p, (const unsigned int *)(ctl_.p + 4),
                               corr_.kill.p, corr_.tip_head.p, corr_.mark.p);
            hipLaunchKernelGGL(k_tip_reset_heads, G, B, 0, stream_, corr_.tiprec.p, (const unsigned int *)(ctl_.p + 4),
                               corr_.tip_head.p);
            HIPCHK(hipGetLastError());
            if (int rc = apply_marks(g, corr_.mark, corr_.removed, s_tip, err)) return rc;
        }
        if (bubbles) {
            HIPCHK(fill2_async(ctl_.p + 3, 8, 0u, nullptr, 0, 0u, stream_));
            hipLaunchKernelGGL(k_fork_candidates<W>, dim3(grid_for(n)), B, 0, stream_, g, alive_.p,
                               corr_.cand.p, (unsigned int *)(ctl_.p + 3));
            hipLaunchKernelGGL(k_bubble<W>, G, B, 0, stream_, g, corr_.cand.p, (const unsigned int *)(ctl_.p + 3), corr_.mark.p);
            HIPCHK(hipGetLastError());
            if (int rc = apply_marks(g, corr_.mark, corr_.removed, s_bub, err)) return rc;
        }
        return 0;
    }
    // the first round's outcome is known (n1 tips, n2 bubble nodes removed): run the remaining rounds, if any
    int finish_correction(unsigned int n1, unsigned int n2, std::string &err) {
        corr_pending_ = false;
        tips_removed_ += n1; bubbles_removed_ += n2; rounds_ = 1;
        EvTimer t(stream_, stage_timers_);
        for (int round = 1; round < 32 && n1 + n2 != 0; round++) {
            if (int rc = correction_round(round, err)) return rc;
            HIPCHK(hipMemcpyAsync(mbox64() + MB_MISC, ctl_.p + 5, 16, hipMemcpyDeviceToHost, stream_));
            WAIT_STREAM();
            n1 = (unsigned int)mbox64()[MB_MISC]; n2 = (unsigned int)mbox64()[MB_MISC + 1];
            tips_removed_ += n1; bubbles_removed_ += n2; rounds_++;
        }
        if (rounds_ > 1) if (t.on()) times_.add("correct_total", t.stop());
        corr_.release();                                  // (the stream is idle or holds nothing that uses them)
        return 0;
    }

    int get_adjacency(uint8_t *adj_initial, uint8_t *adj_final, uint8_t *alive, uint64_t cap,
                      std::string &err) override {
        if (!graph_ready_) { err = "graph not built"; return -2; }
        if (cap < n_solid_) { err = "buffer too small"; return -1; }
        if (!n_solid_) return 0;
        if (adj_initial && !keep_stages_) { err = "the initial adjacency is kept for inspection only on a verbose handle (or with SHK_KEEP_STAGES=1)"; return -2; }
        if (adj_initial) HIPCHK(hipMemcpy(adj_initial, adj0_.p, n_solid_, hipMemcpyDeviceToHost));
        if (adj_final) HIPCHK(hipMemcpy(adj_final, adj_.p, n_solid_, hipMemcpyDeviceToHost));
        if (alive) HIPCHK(hipMemcpy(alive, alive_.p, n_solid_, hipMemcpyDeviceToHost));
        return 0;
    }

    // ---- collapse ----------------------------------------------------------------------------
    // Counters in ctl_: 5 = splitters (k_succ_split, then appended to by k_orphan_cycles), 6 = chains reported
    // (k_rank_tails), 7 = splitters that sit on a circular unitig (statistic), 8 = flags of k_orphan_cycles.
    struct ChainState {                // the ranking of the chains of simple links: per node (chain record, position), per chain a HeadRec
        DevBuf<uint32_t> spl, slot_of;
        DevBuf<uint2> winfo, ol;
        DevBuf<SegRec> segs;
        DevBuf<FragRec> frag;          // indexed by node id, written at fragment heads only
        DevBuf<RankRec> Ra, Rb;
        DevBuf<FinRec> fin;
        DevBuf<HeadRec> d_heads;
        DevBuf<RingMin> ringmin;
        std::vector<HeadRec> heads;
        uint32_t seg_cap = 0;
        unsigned int n_heads = 0;      // chains ranked (heads.size() only when they were downloaded)
        bool heads_on_host = true;
    };
    // rings: also find the smallest k-mer of every circular chain and the strand / rotation it is spelled with (single GPU);
    // the sharded assembly settles rings across ranks itself (shard_graph.h)
    // ONE host round trip: everything from the simple links to the ranked chains is launched back to back and the counters
    // come home with the first chain records.  What the host used to fetch in between is handled on the device or by a
    // bound: the splitter list gets room for total/12 + 65536 entries (a 1/64 sample plus the heads: ~2-4 % of the oriented
    // nodes; if a graph needs more the kernels stop and the pass is repeated with room for all), the number of pointer-jumping
    // rounds comes from that room, and the launch may sit behind a first correction round whose outcome is not known yet —
    // then the kernels return at once when that round removed something, the correction is finished and the pass repeated.
    // (plan: an assembly with few chains has its emission planned on the device — k_plan_emit —, its text written and sent to
    // the host behind the ranking, inside the same launch sequence: the host's ONE wait brings the counters, the chain records
    // and the text.  planned = the device did it; otherwise the caller plans as before.)
    struct EmitPlan { EmitRec *d_off; char *d_out; char *h_out; uint32_t max_heads; unsigned long long out_cap; bool planned; unsigned long long out_bytes; uint32_t n_emit; };
    int rank_chains(ChainState &cs, bool rings, std::string &err, uint32_t keep_on_device_from = 0xFFFFFFFFu, EmitPlan *plan = nullptr) {
        const uint32_t n = (uint32_t)n_solid_;
        const uint32_t total = 2 * n;
        Graph<W> g = graph_view();
        if (plan) plan->planned = false;
        const bool stage_log = getenv("SHK_STAGE_LOG") != nullptr;          // (see shard_assemble)
        auto stage = [&](const char *what) {
            if (!stage_log) return;
            const hipError_t e = hipStreamSynchronize(stream_);
            fprintf(stderr, "[rank_chains n=%u] %s done%s\n", n, what, e == hipSuccess ? "" : " (stream error)"); fflush(stderr);
        };
        if (int rc = cs.winfo.alloc(total, err)) return rc;
        if (int rc = cs.frag.alloc(total, err)) return rc;
        if (int rc = cs.spl.alloc(total, err)) return rc;
        if (int rc = cs.ol.alloc(total, err)) return rc;
        const uint32_t split_mask = (1u << (uint32_t)env_u64("SHK_SPLIT_LOG", SPLIT_LOG_DEFAULT)) - 1u;
        const uint32_t tile_rows = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(env_u64("SHK_TILE_ROWS", LF_ROWS), 1), LF_TILE / 2u);
        const int lf_grid = (int)((n + tile_rows - 1) / tile_rows);
        const uint64_t cap_all = std::min<uint64_t>((uint64_t)total + 1024u, 0xFFFFFFF0ull);      // every alive oriented node a splitter or an orphan ring of its own
        uint64_t seg_cap64 = std::min<uint64_t>(cap_all, env_u64("SHK_SEG_CAP", (uint64_t)total / 12u + 65536u));
        constexpr unsigned int HEADS_SPEC = 512;                           // (40 bytes each, into the pinned mailbox)
        static_assert(MB_MISC * 8 + HEADS_SPEC * sizeof(HeadRec) <= MBOX_BYTES, "mailbox");
        std::vector<HeadRec> &heads = cs.heads;
        unsigned long long hc[4] = {0, 0, 0, 0};
        uint32_t seg_cap = 0;
        EvTimer tr(stream_, stage_timers_);
        for (int attempt = 0;; attempt++) {
            if (attempt > 3) { err = "collapse: the chain ranking did not settle"; return -6; }
            seg_cap = (uint32_t)seg_cap64;
            cs.seg_cap = seg_cap;
            if (int rc = cs.segs.alloc(seg_cap, err)) return rc;
            if (int rc = cs.Ra.alloc(seg_cap, err)) return rc;
            if (int rc = cs.Rb.alloc(seg_cap, err)) return rc;
            if (int rc = cs.slot_of.alloc(seg_cap, err)) return rc;
            if (int rc = cs.fin.alloc(seg_cap, err)) return rc;
            if (int rc = cs.d_heads.alloc(seg_cap, err)) return rc;
            if (int rc = cs.ringmin.alloc(seg_cap, err)) return rc;
            const unsigned long long *skip = corr_pending_ ? ctl_.p + 16 : (const unsigned long long *)nullptr;
            // 5 splitters, 6 chains, 7 ring splitters, 8 flags, 9 alive oriented nodes, 10 nodes walked
            HIPCHK(fill2_async(ctl_.p + 5, 6 * 8, 0u, cs.slot_of.p, (size_t)seg_cap * 4, 0xFFFFFFFFu, stream_));
            unsigned int *d_nspl = (unsigned int *)(ctl_.p + 5);
            uint32_t *d_flags = (uint32_t *)(ctl_.p + 8);
            hipLaunchKernelGGL(k_succ_split<W>, dim3((total + 256 * SS_ITEMS - 1) / (256 * SS_ITEMS)), dim3(256), 0, stream_, g,
                               alive_.p, cs.winfo.p, cs.spl.p, cs.ol.p, d_nspl, split_mask, ctl_.p + 9, skip);
            stage("k_succ_split");
            hipLaunchKernelGGL(k_local_frag<W>, dim3(lf_grid), dim3(LF_THREADS), 0, stream_, n, row_starts_.p, tile_rows, alive_.p, cs.winfo.p, cs.ol.p, cs.frag.p, split_mask, skip,
                               d_nspl, seg_cap, d_flags);
            stage("k_local_frag");
            // (grids: the expected 1/64 sample plus a margin; the kernels loop to the device-side count)
            hipLaunchKernelGGL(k_walk_frags<W>, dim3(grid_for((uint64_t)total / 64u + 16384u, 256, 1 << 20)), dim3(256), 0, stream_,
                               cs.spl.p, (const unsigned int *)d_nspl, cs.frag.p, cs.segs.p, split_mask, ctl_.p + 10, total, d_flags);
            stage("k_walk_frags");
            hipLaunchKernelGGL(k_orphan_cycles<W>, dim3(grid_for(total)), dim3(256), 0, stream_, g, alive_.p, cs.winfo.p, cs.ol.p,
                               cs.frag.p, cs.spl.p, cs.segs.p, d_nspl, seg_cap, d_flags, ctl_.p + 9, ctl_.p + 10);
            HIPCHK(hipGetLastError());
            stage("k_orphan_cycles");
            // ---- rank the splitter list on the device: prefix sums by pointer jumping, rings in the same pass (collapse.h)
            const int gr = grid_for((uint64_t)total / 64u + 65536u);
            // (a chain has at most seg_cap splitters; rounds past the point where every pointer has reached its head add nothing)
            int rounds = 1; { uint64_t reach = RANK_HOPS; while (reach < (uint64_t)seg_cap + 1u) { reach *= RANK_HOPS; rounds++; } }
            hipLaunchKernelGGL(k_rank_init, dim3(gr), dim3(256), 0, stream_, cs.segs.p, (const unsigned int *)d_nspl, cs.Ra.p);
            RankRec *Ri = cs.Ra.p, *Ro = cs.Rb.p;
            for (int r = 0; r < rounds; r++) {
                hipLaunchKernelGGL(k_rank_jump, dim3(gr), dim3(256), 0, stream_, (const unsigned int *)d_nspl, Ri, Ro, (uint32_t)r);
                std::swap(Ri, Ro);
            }
            stage("k_rank_jump");
            const unsigned int *d_ncyc = (const unsigned int *)(ctl_.p + 7);
            hipLaunchKernelGGL(k_rank_tails<W>, dim3(gr), dim3(256), 0, stream_, g, cs.segs.p, (const unsigned int *)d_nspl, Ri, cs.d_heads.p, cs.slot_of.p, cs.ringmin.p,
                               (unsigned int *)(ctl_.p + 6), (unsigned int *)(ctl_.p + 7));
            hipLaunchKernelGGL(k_rank_fin, dim3(gr), dim3(256), 0, stream_, cs.segs.p, (const unsigned int *)d_nspl, Ri, cs.slot_of.p, cs.fin.p);
            stage("k_rank_tails + k_rank_fin");
            hipLaunchKernelGGL(k_tile_final, dim3(lf_grid), dim3(LF_THREADS), 0, stream_, n, row_starts_.p, tile_rows, cs.ol.p, cs.frag.p, cs.fin.p, skip, (const uint32_t *)d_flags);
            stage("k_tile_final");
            if (rings) {
                // rings: their smallest k-mer (these three return at once when there is none)
                hipLaunchKernelGGL(k_ring_min1<W>, dim3(1024), dim3(256), 0, stream_, g, alive_.p, cs.ol.p, cs.ringmin.p, d_ncyc);
                hipLaunchKernelGGL(k_ring_min2<W>, dim3(1024), dim3(256), 0, stream_, g, alive_.p, cs.ol.p, cs.ringmin.p, d_ncyc);
                hipLaunchKernelGGL(k_ring_rot<W>, dim3(gr), dim3(256), 0, stream_, g, cs.d_heads.p, (const unsigned int *)(ctl_.p + 6), cs.ringmin.p,
                                   cs.winfo.p, cs.ol.p, d_ncyc, d_flags);
            }
            if (plan) {
                HIPCHK(fill2_async(ctl_.p + 18, 3 * 8, 0u, nullptr, 0, 0u, stream_));
                hipLaunchKernelGGL(k_plan_emit, dim3(1), dim3(1024), 0, stream_, cs.d_heads.p, (const unsigned int *)(ctl_.p + 6), (uint32_t)k_, plan->max_heads,
                                   plan->out_cap, (const uint32_t *)d_flags, plan->d_off, ctl_.p + 18);
                hipLaunchKernelGGL(k_emit<W>, dim3(grid_for(n)), dim3(256), 0, stream_, g, alive_.p, cs.ol.p, plan->d_off, plan->d_out, (const unsigned long long *)(ctl_.p + 18));
                hipLaunchKernelGGL(k_text_to_host_planned, dim3(64), dim3(256), 0, stream_, plan->d_out, plan->h_out, (const unsigned long long *)(ctl_.p + 18));
            }
            HIPCHK(hipGetLastError());
            // (the first chain records travel with the counters: an isolate has a handful of chains, and a second
            // round trip just for them is 30-40 us of idle GPU)
            heads.assign(HEADS_SPEC, HeadRec());
            HIPCHK(hipMemcpyAsync(mbox64() + MB_CTL, ctl_.p, CTL_WORDS * 8, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipMemcpyAsync(mbox64() + MB_MISC, cs.d_heads.p, (size_t)std::min<uint32_t>(HEADS_SPEC, seg_cap) * sizeof(HeadRec), hipMemcpyDeviceToHost, stream_));
            WAIT_STREAM();
            const unsigned long long *hc0 = mbox64() + MB_CTL;
            if (int rc = check_graph_flags(err)) return rc;
            if (corr_pending_) {
                // the first correction round's outcome: if it removed nodes the kernels above returned at once (or did nothing:
                // no splitters) — the remaining rounds run now, then the pass is repeated on the final graph
                const unsigned int r1 = (unsigned int)hc0[16], r2 = (unsigned int)hc0[17];
                if (int rc = finish_correction(r1, r2, err)) return rc;
                if (r1 + r2 != 0) continue;
            }
            memcpy(hc, hc0 + 5, sizeof hc);
            const uint32_t flag = (uint32_t)hc[3];
            if ((flag == 4 || flag == 2) && seg_cap64 < cap_all) {        // more splitters (4) or orphan rings (2) than the room: once more with room for all
                times_.add("collapse_seg_cap_retry_x1", 1.0);
                seg_cap64 = cap_all;
                continue;
            }
            memcpy(heads.data(), mbox64() + MB_MISC, (size_t)std::min<uint32_t>(HEADS_SPEC, seg_cap) * sizeof(HeadRec));
            if (plan && hc0[18] == 1ull) { plan->planned = true; plan->out_bytes = hc0[19]; plan->n_emit = (uint32_t)hc0[20]; }
            break;
        }
        if ((uint32_t)hc[3]) { err = (uint32_t)hc[3] == 2 ? "collapse: too many short circular unitigs" : ((uint32_t)hc[3] == 3 ? "collapse: ring without a smallest k-mer" : "collapse: broken cycle"); return -6; }
        times_.add("collapse_n_splitters_x1e-3", (double)(unsigned int)hc[0] * 1e-3);
        times_.add("collapse_cycle_splitters_x1e-3", (double)(unsigned int)hc[2] * 1e-3);
        const unsigned int n_heads = (unsigned int)hc[1];
        if (tr.on()) times_.add("collapse_rank_device", tr.stop());
        cs.n_heads = n_heads;
        // (sharded assembly, and fragmented assemblies headed for the device writer: the chain records stay on the device —
        // only their number is needed here)
        if (!rings || n_heads >= keep_on_device_from) { heads.resize(std::min<unsigned int>(n_heads, HEADS_SPEC)); cs.heads_on_host = false; return 0; }
        heads.resize(n_heads);
        cs.heads_on_host = true;
        if (n_heads > HEADS_SPEC)
            HIPCHK(hipMemcpy(heads.data() + HEADS_SPEC, cs.d_heads.p + HEADS_SPEC, (size_t)(n_heads - HEADS_SPEC) * sizeof(HeadRec), hipMemcpyDeviceToHost));
        return 0;
    }

    int collapse(std::vector<RawContig> &out, std::string &err, const char **json, size_t *json_len, uint64_t *n_contigs, TextArrival **arrival) override {
        out.clear();
        if (json) *json = nullptr;
        if (arrival) *arrival = nullptr;
        if (!graph_ready_) { err = "graph not built"; return -2; }
        const uint32_t n = (uint32_t)n_solid_;
        if (n == 0) return 0;
        Graph<W> g = graph_view();
        ChainState cs;
        const uint64_t dw_min = env_u64("SHK_DEVICE_WRITER_MIN", 20000);
        // an isolate (a handful of chains): emission planned on the device, text written and sent home inside the ranking's launch
        // sequence — room for 512 chain records and n + 512 (k - 1) bases (what does not fit is planned on the host as before)
        constexpr uint32_t PLAN_HEADS = 512;
        EmitPlan plan{}; EmitPlan *use_plan = nullptr;
        DevBuf<EmitRec> d_plan_off; DevBuf<char> d_plan_out;
        // (off by default — SHK_DEVICE_PLAN=1: measured on the bench isolate, three alternating runs of 30 steps each: 3.30 / 3.22 /
        // 3.20 ms per step with the plan on the host against 3.37 / 3.22 / 3.27 with it on the device — the round trip it saves is
        // paid back by the text crossing PCIe through a kernel instead of the copy engine.  DESIGN.md section 4)
        if (env_u64("SHK_DEVICE_PLAN", 0) != 0 && dw_min > PLAN_HEADS) {
            const unsigned long long cap = (unsigned long long)n + (unsigned long long)PLAN_HEADS * (unsigned long long)(k_ - 1) + 64ull;
            if (int rc = d_plan_off.alloc(PLAN_HEADS, err)) return rc;
            if (int rc = d_plan_out.alloc(cap + 16, err)) return rc;
            if (int rc = hout_.alloc(cap + 16, err)) return rc;
            plan.d_off = d_plan_off.p; plan.d_out = d_plan_out.p; plan.h_out = hout_.p; plan.max_heads = PLAN_HEADS; plan.out_cap = cap;
            use_plan = &plan;
        }
        // (chains come on both strands: 2 x dw_min chain records mean at least dw_min contigs)
        if (int rc = rank_chains(cs, true, err, json ? (uint32_t)std::min<uint64_t>(2 * dw_min, 0xFFFFFFFFull) : 0xFFFFFFFFu, use_plan)) return rc;
        if (use_plan && plan.planned && cs.heads_on_host) {
            // the text is on the host already (hout_): the contigs in chain-record order, as k_plan_emit laid them out
            std::vector<HeadRec> &hd = cs.heads;
            unsigned long long at = 0; uint32_t ne = 0;
            out.reserve(plan.n_emit);
            for (size_t i = 0; i < hd.size(); i++) {
                if (!hd[i].emit) continue;
                RawContig rc; rc.kc = hd[i].kc;
                rc.ext = hout_.p + at; rc.ext_n = hd[i].len + (uint64_t)(k_ - 1);
                at += rc.ext_n; ne++;
                out.push_back(std::move(rc));
            }
            if (at != plan.out_bytes || ne != plan.n_emit) { err = "collapse: the device's emission plan and the chain records disagree"; return -6; }
            times_.add("collapse_planned_on_device_x1", 1.0);
            return 0;
        }
        if (!cs.heads_on_host) {
            // ---- a fragmented assembly: which chains are emitted, where, and the whole get_assembly() text on the device
            const uint32_t nh = cs.n_heads;
            DevBuf<unsigned long long> sz, fl, off, idx; DevBuf<char> tmp; DevBuf<EmitRec> d_off2; DevBuf<WContig> d_c; DevBuf<char> d_out2;
            if (int rc = sz.alloc((size_t)nh + 1, err)) return rc;
            if (int rc = fl.alloc((size_t)nh + 1, err)) return rc;
            hipLaunchKernelGGL(k_w_plan_sizes, dim3(grid_for(nh)), dim3(256), 0, stream_, cs.d_heads.p, nh, (uint32_t)k_, sz.p, fl.p);
            HIPCHK(hipGetLastError());
            if (int rc = scan_excl(sz, off, nh, tmp, err)) return rc;
            if (int rc = scan_excl(fl, idx, nh, tmp, err)) return rc;
            unsigned long long last[4] = {0, 0, 0, 0};
            HIPCHK(hipMemcpyAsync(&last[0], sz.p + (nh - 1), 8, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipMemcpyAsync(&last[1], off.p + (nh - 1), 8, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipMemcpyAsync(&last[2], fl.p + (nh - 1), 8, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipMemcpyAsync(&last[3], idx.p + (nh - 1), 8, hipMemcpyDeviceToHost, stream_));
            WAIT_STREAM();
            const uint64_t out_bytes2 = last[0] + last[1], n_emit = last[2] + last[3];
            if (n_emit >= 0x7FFFFFF0ull) { err = "device writer: too many contigs"; return -1; }
            if (int rc = d_off2.alloc(nh, err)) return rc;
            if (int rc = d_c.alloc(n_emit + 1, err)) return rc;
            if (int rc = d_out2.alloc(out_bytes2 + 16, err)) return rc;
            HIPCHK(fill2_async(ctl_.p + 14, 8, 0u, nullptr, 0, 0u, stream_));
            hipLaunchKernelGGL(k_w_plan_fill, dim3(grid_for(nh)), dim3(256), 0, stream_, cs.d_heads.p, nh, (uint32_t)k_, off.p, idx.p, d_off2.p, d_c.p, (uint32_t *)(ctl_.p + 14));
            EvTimer t3(stream_, stage_timers_);
            hipLaunchKernelGGL(k_emit<W>, dim3(grid_for(n)), dim3(256), 0, stream_, g, alive_.p, cs.ol.p, d_off2.p, d_out2.p);
            HIPCHK(hipGetLastError());
            unsigned int fl9 = 0;
            if (int rc = read_ctl(fl9, 14, err)) return rc;
            if (t3.on()) times_.add("collapse_emit", t3.stop());
            if (fl9) { err = "device writer: a contig beyond 2^32 bases"; return -1; }
            if (n_emit == 0) { if (json) *json = nullptr; return 0; }
            if (int rc = device_write_json(g, cs, d_c, (uint32_t)n_emit, nh, d_out2.p, out_bytes2, json, json_len, err)) return rc;
            if (n_contigs) *n_contigs = n_emit;
            return 0;
        }
        std::vector<HeadRec> &heads = cs.heads;
        DevBuf<EmitRec> d_off; DevBuf<char> d_out;
        // each unitig exists on both strands: keep the canonical one (decided on the device)
        std::vector<EmitRec> head_off(heads.size());
        std::vector<uint32_t> emitted;
        uint64_t out_bytes = 0;
        for (size_t i = 0; i < heads.size(); i++) {
            EmitRec e; e.off = ~0ull; e.rot = heads[i].rot; e.len = (uint32_t)heads[i].len;
            if (heads[i].emit) { e.off = out_bytes; out_bytes += heads[i].len + (uint64_t)(k_ - 1); emitted.push_back((uint32_t)i); }
            head_off[i] = e;
        }
        if (!emitted.empty()) {
            PinnedBuf &hout = hout_;
            if (int rc = hout.alloc(out_bytes, err)) return rc;
            if (int rc = d_off.alloc(head_off.size(), err)) return rc;
            if (int rc = d_out.alloc(out_bytes, err)) return rc;
            HIPCHK(hipMemcpyAsync(d_off.p, head_off.data(), head_off.size() * sizeof(EmitRec), hipMemcpyHostToDevice, stream_));
            EvTimer t3(stream_, stage_timers_);
            hipLaunchKernelGGL(k_emit<W>, dim3(grid_for(n)), dim3(256), 0, stream_, g, alive_.p, cs.ol.p, d_off.p, d_out.p);
            HIPCHK(hipGetLastError());
            t3.stop_later("collapse_emit", pending_timers_);
            auto tcp = std::chrono::steady_clock::now();
            out.reserve(emitted.size());
            for (uint32_t i : emitted) {
                RawContig rc; rc.kc = heads[i].kc;
                rc.ext = hout.p + head_off[i].off; rc.ext_n = heads[i].len + (uint64_t)(k_ - 1);
                out.push_back(std::move(rc));
            }
            // (off by default — SHK_ARRIVAL_MIN = smallest text in bytes that takes this path: measured on the bench isolate, one
            // handle at a time 3.24 against 3.26 ms per step, two handles in flight 2.93 against 2.67: the writer's pool then waits
            // for the GPU while the other handle's writer waits for the pool.  DESIGN.md section 4)
            const uint64_t pipe_min = env_u64("SHK_ARRIVAL_MIN", ~0ull);
            if (arrival && out_bytes >= pipe_min && emitted.size() <= 4096) {
                // ---- megabases of text: the writer starts while the text is still crossing PCIe.  First the ends of every
                // contig (all that order, links and strand checks read), then the text slab by slab, a flag behind each.
                const uint32_t E = (uint32_t)std::max(k_, 32), nc = (uint32_t)emitted.size();
                std::vector<unsigned long long> &src = ends_src_host_;       // (a member: read by an asynchronous copy)
                src.resize(2 * (size_t)nc);
                for (uint32_t c = 0; c < nc; c++) { src[2 * c] = head_off[emitted[c]].off; src[2 * c + 1] = heads[emitted[c]].len + (uint64_t)(k_ - 1); }
                const uint32_t n_slabs = (uint32_t)((out_bytes + ARRIVAL_SLAB - 1) / ARRIVAL_SLAB);
                if (int rc = ends_src_.alloc(2 * (size_t)nc, err)) return rc;
                if (int rc = hends_.alloc((size_t)nc * 2 * E, err)) return rc;
                if (int rc = hflags_.alloc(((size_t)n_slabs + 16) * 4, err)) return rc;
                volatile uint32_t *flags = (volatile uint32_t *)hflags_.p;       // [0] the ends, [16 ..] the slabs
                memset(hflags_.p, 0, ((size_t)n_slabs + 16) * 4);
                HIPCHK(hipMemcpyAsync(ends_src_.p, src.data(), src.size() * 8, hipMemcpyHostToDevice, stream_));
                hipLaunchKernelGGL(k_ends_to_host, dim3(1), dim3(1024), 0, stream_, d_out.p, ends_src_.p, nc, E, hends_.p, (uint32_t *)hflags_.p);
                const uint32_t grid = (uint32_t)std::min<uint64_t>(n_slabs, env_u64("SHK_ARRIVAL_BLOCKS", 64));
                hipLaunchKernelGGL(k_text_to_host, dim3(grid), dim3(256), 0, stream_, d_out.p, hout.p, (unsigned long long)out_bytes, n_slabs, (uint32_t *)hflags_.p + 16);
                HIPCHK(hipGetLastError());
                d_out_keep_.swap(d_out);                                     // (the source of a copy that outlives this function)
                arrival_.set(hout.p, out_bytes, flags + 16, n_slabs, stream_);
                if (!SlabArrival::spin_until(flags, arrival_.failed_)) { (void)hipStreamSynchronize(stream_); err = "collapse: the contigs' ends did not arrive"; return -5; }
                for (uint32_t c = 0; c < nc; c++) {
                    out[c].head = hends_.p + (size_t)c * 2 * E; out[c].tail = out[c].head + E;
                    out[c].ends_n = (uint32_t)std::min<uint64_t>(E, out[c].ext_n);
                }
                *arrival = &arrival_;
                times_.add("collapse_ends_host_clock", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tcp).count());
                return 0;
            }
            HIPCHK(hipMemcpyAsync(hout.p, d_out.p, out_bytes, hipMemcpyDeviceToHost, stream_));
            WAIT_STREAM();
            times_.add("collapse_d2h_contigs_host_clock", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tcp).count());
        }
        return 0;
    }

    // ---- the writer on the device (writer_gpu.h): fragmented assemblies ------------------------------------------
    template <typename T> int scan_excl(DevBuf<T> &in, DevBuf<T> &out, uint32_t n, DevBuf<char> &tmp, std::string &err) {
        if (int rc = out.alloc((size_t)n + 1, err)) return rc;
        if (!n) return 0;
        size_t bytes = 0;
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in.p, out.p, (int)n, stream_));
        if (tmp.n < bytes) if (int rc = tmp.alloc(bytes + 256, err)) return rc;
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in.p, out.p, (int)n, stream_));
        return 0;
    }
    int device_write_json(Graph<W> &g, ChainState &cs, DevBuf<WContig> &d_c, uint32_t nc, uint32_t n_slots, const char *d_text, uint64_t text_bytes,
                          const char **json, size_t *json_len, std::string &err) {
        typedef unsigned long long u64;
        EvTimer tw(stream_, stage_timers_);
        DevBuf<u64> keys, keys2, slen, skc; DevBuf<uint32_t> vals, vals2, rank_of_slot, rank_of_contig; DevBuf<char> tmp;
        if (int rc = keys.alloc(nc, err)) return rc;
        if (int rc = keys2.alloc(nc, err)) return rc;
        if (int rc = vals.alloc(nc, err)) return rc;
        if (int rc = vals2.alloc(nc, err)) return rc;
        if (int rc = rank_of_slot.alloc(n_slots + 1, err)) return rc;
        if (int rc = rank_of_contig.alloc(nc, err)) return rc;
        if (int rc = slen.alloc(nc, err)) return rc;
        if (int rc = skc.alloc(nc, err)) return rc;
        HIPCHK(hipMemsetAsync(rank_of_slot.p, 0, ((size_t)n_slots + 1) * 4, stream_));
        // ---- order
        hipLaunchKernelGGL(k_w_keys, dim3(grid_for(nc)), dim3(256), 0, stream_, d_text, d_c.p, nc, keys.p, vals.p);
        {
            size_t bytes = 0;
            HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys.p, keys2.p, vals.p, vals2.p, (int)nc, 0, 64, stream_));
            if (int rc = tmp.alloc(bytes + 256, err)) return rc;
            HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, keys.p, keys2.p, vals.p, vals2.p, (int)nc, 0, 64, stream_));
        }
        hipLaunchKernelGGL(k_w_ties, dim3(grid_for(nc)), dim3(256), 0, stream_, d_text, d_c.p, nc, keys2.p, vals2.p);
        hipLaunchKernelGGL(k_w_rank, dim3(grid_for(nc)), dim3(256), 0, stream_, d_c.p, nc, vals2.p, rank_of_slot.p, slen.p, skc.p, rank_of_contig.p);
        HIPCHK(hipGetLastError());
        // ---- links
        DevBuf<u64> links, links2; DevBuf<unsigned int> d_cnt;
        const uint32_t link_cap = 8u * nc + 64u;
        if (int rc = links.alloc(link_cap, err)) return rc;
        if (int rc = links2.alloc(link_cap, err)) return rc;
        if (int rc = d_cnt.alloc(4, err)) return rc;
        HIPCHK(hipMemsetAsync(d_cnt.p, 0, 16, stream_));
        hipLaunchKernelGGL(k_w_links<W>, dim3(grid_for(2ull * nc)), dim3(256), 0, stream_, g, cs.d_heads.p, cs.ol.p, d_c.p, nc, vals2.p, rank_of_slot.p,
                           links.p, d_cnt.p, link_cap, d_cnt.p + 2);
        HIPCHK(hipGetLastError());
        unsigned int h_cnt[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(h_cnt, d_cnt.p, 16, hipMemcpyDeviceToHost, stream_));
        WAIT_STREAM();
        if (h_cnt[2]) { err = "device writer: the graph and the chains disagree about a link (" + std::to_string(h_cnt[2]) + ")"; return -6; }
        if (h_cnt[0] > link_cap) { err = "device writer: more links than 8 per contig"; return -6; }
        uint32_t nl = h_cnt[0];
        if (nl) {
            size_t bytes = 0;
            HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, links.p, links2.p, (int)nl, 0, 64, stream_));
            if (tmp.n < bytes) if (int rc = tmp.alloc(bytes + 256, err)) return rc;
            HIPCHK(hipcub::DeviceRadixSort::SortKeys(tmp.p, bytes, links.p, links2.p, (int)nl, 0, 64, stream_));
            bytes = 0;
            HIPCHK(hipcub::DeviceSelect::Unique(nullptr, bytes, links2.p, links.p, d_cnt.p + 1, (int)nl, stream_));
            if (tmp.n < bytes) if (int rc = tmp.alloc(bytes + 256, err)) return rc;
            HIPCHK(hipcub::DeviceSelect::Unique(tmp.p, bytes, links2.p, links.p, d_cnt.p + 1, (int)nl, stream_));
            HIPCHK(hipMemcpyAsync(h_cnt, d_cnt.p, 16, hipMemcpyDeviceToHost, stream_));
            WAIT_STREAM();
            nl = h_cnt[1];
        }
        // ---- sizes and offsets of the records
        const uint32_t ov = (uint32_t)(k_ - 1);
        DevBuf<u64> s_fa, s_dn, s_1s, s_2s, s_dl, s_1l, s_2l, o_fa, o_dn, o_1s, o_2s, o_dl, o_1l, o_2l;
        for (DevBuf<u64> *b : {&s_fa, &s_dn, &s_1s, &s_2s}) if (int rc = b->alloc((size_t)nc + 1, err)) return rc;
        for (DevBuf<u64> *b : {&s_dl, &s_1l, &s_2l}) if (int rc = b->alloc((size_t)nl + 1, err)) return rc;
        hipLaunchKernelGGL(k_w_contig_sizes, dim3(grid_for(nc)), dim3(256), 0, stream_, nc, slen.p, skc.p, s_fa.p, s_dn.p, s_1s.p, s_2s.p);
        if (nl) hipLaunchKernelGGL(k_w_link_sizes, dim3(grid_for(nl)), dim3(256), 0, stream_, nl, links.p, ov, slen.p, s_dl.p, s_1l.p, s_2l.p);
        HIPCHK(hipGetLastError());
        if (int rc = scan_excl(s_fa, o_fa, nc, tmp, err)) return rc;
        if (int rc = scan_excl(s_dn, o_dn, nc, tmp, err)) return rc;
        if (int rc = scan_excl(s_1s, o_1s, nc, tmp, err)) return rc;
        if (int rc = scan_excl(s_2s, o_2s, nc, tmp, err)) return rc;
        if (int rc = scan_excl(s_dl, o_dl, nl, tmp, err)) return rc;
        if (int rc = scan_excl(s_1l, o_1l, nl, tmp, err)) return rc;
        if (int rc = scan_excl(s_2l, o_2l, nl, tmp, err)) return rc;
        // totals: last offset + last size of every section
        u64 last[14] = {0};
        {
            DevBuf<u64> *so[7][2] = {{&s_fa, &o_fa}, {&s_dn, &o_dn}, {&s_dl, &o_dl}, {&s_1s, &o_1s}, {&s_1l, &o_1l}, {&s_2s, &o_2s}, {&s_2l, &o_2l}};
            for (int q = 0; q < 7; q++) {
                const uint32_t cnt = (q == 2 || q == 4 || q == 6) ? nl : nc;
                if (!cnt) continue;
                HIPCHK(hipMemcpyAsync(&last[2 * q], so[q][0]->p + (cnt - 1), 8, hipMemcpyDeviceToHost, stream_));
                HIPCHK(hipMemcpyAsync(&last[2 * q + 1], so[q][1]->p + (cnt - 1), 8, hipMemcpyDeviceToHost, stream_));
            }
            WAIT_STREAM();
        }
        u64 sect[7];
        for (int q = 0; q < 7; q++) sect[q] = last[2 * q] + last[2 * q + 1];
        // the literals between the sections (exactly outputs.cpp's)
        const std::string lit[8] = {"{\"outfasta\":\"",
                                    "\",\"ncontigs\":" + std::to_string(nc) + ",\"outdot\":\"digraph sparrowhawk {\\n",
                                    "", "}\\n\",\"outgfa\":\"H\\tVN:Z:1.0\\n", "", "\",\"outgfav2\":\"H\\tVN:Z:2.0\\n", "", "\"}"};
        u64 at = 0, lit_at[8], base[7];
        for (int q = 0; q < 8; q++) { lit_at[q] = at; at += lit[q].size(); if (q < 7) { base[q] = at; at += sect[q]; } }
        const u64 total = at;
        DevBuf<char> d_js;
        if (int rc = d_js.alloc(total + 8, err)) return rc;
        if (int rc = hjson_.alloc(total + 8, err)) return rc;
        for (int q = 0; q < 8; q++) if (!lit[q].empty()) HIPCHK(hipMemcpyAsync(d_js.p + lit_at[q], lit[q].data(), lit[q].size(), hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemsetAsync(d_js.p + total, 0, 1, stream_));
        WBases B; B.fasta = base[0]; B.dotn = base[1]; B.dotl = base[2]; B.g1s = base[3]; B.g1l = base[4]; B.g2s = base[5]; B.g2l = base[6];
        hipLaunchKernelGGL(k_w_contig_recs, dim3(grid_for(nc)), dim3(256), 0, stream_, nc, slen.p, skc.p, o_fa.p, o_dn.p, o_1s.p, o_2s.p, B, d_js.p);
        if (nl) hipLaunchKernelGGL(k_w_link_recs, dim3(grid_for(nl)), dim3(256), 0, stream_, nl, links.p, ov, slen.p, o_dl.p, o_1l.p, o_2l.p, B, d_js.p);
        hipLaunchKernelGGL(k_w_copy_seqs, dim3(grid_for((text_bytes + 7) / 8)), dim3(256), 0, stream_, d_text, (u64)text_bytes, d_c.p, nc, rank_of_contig.p,
                           o_fa.p, o_1s.p, o_2s.p, skc.p, B, d_js.p);
        HIPCHK(hipGetLastError());
        if (tw.on()) times_.add("device_writer_kernels", tw.stop());
        const double t0 = now_ms_();
        HIPCHK(hipMemcpyAsync(hjson_.p, d_js.p, total + 1, hipMemcpyDeviceToHost, stream_));
        WAIT_STREAM();
        times_.add("device_writer_d2h_host_clock", now_ms_() - t0);
        times_.add("device_writer_json_MB", (double)total / 1e6);
        times_.add("device_writer_links_x1e-3", nl * 1e-3);
        *json = hjson_.p; if (json_len) *json_len = (size_t)total;
        return 0;
    }

    // ---- sharded assembly (shard_graph.h) ---------------------------------------------------------------------
    int shard_keep_local(uint32_t world, uint32_t rank, uint32_t n_count_partitions, const uint64_t *rows_per_rank,
                         const uint64_t histo[500], uint64_t total_instances, std::string &err) override {
        if (world < 1 || world > ROUTE_MAX_WORLD || rank >= world || n_count_partitions < 1 || (n_count_partitions & (n_count_partitions - 1))) {
            err = "shard_keep_local: bad world / rank / partition count"; return -1;
        }
        if (rows_per_rank[rank] != n_solid_) { err = "shard_keep_local: this rank's row count disagrees with the filter"; return -1; }
        if (n_solid_ >= (1ull << 30)) { err = "sharded assembly: more than 2^30 solid k-mers on one rank"; return -1; }
        sh_world_ = world; sh_rank_ = rank; sh_P_ = n_count_partitions;
        sh_gbase_.assign(world + 1, 0);
        for (uint32_t r = 0; r < world; r++) sh_gbase_[r + 1] = sh_gbase_[r] + rows_per_rank[r];
        n_solid_global_ = sh_gbase_[world];
        total_instances_ = total_instances; n_distinct_ = 0;
        for (int i = 0; i < 500; i++) { histo_[i] = histo[i]; n_distinct_ += histo[i]; }
        sh_active_ = true; graph_ready_ = false;
        return 0;
    }
    bool sharded_graph() const override { return sh_active_; }
    uint64_t n_solid_global() const override { return sh_active_ ? n_solid_global_ : n_solid_; }

    // n_items staged records (dest[i] = rank, or NIL: no record; PW words each) -> one pairwise exchange.  r.recv holds
    // what the other ranks sent here (source-major), r.sidx[i] the index item i got in r.send (the order answers come back in)
    struct Routed {
        DevBuf<uint64_t> send, recv; DevBuf<uint32_t> sidx;
        std::vector<uint64_t> send_cnt, recv_cnt; uint64_t n_send = 0, n_recv = 0;
    };
    // (the size exchange happens on the device: the row of counts is gathered by RCCL and read back ONCE, world x (world + 1)
    // words; `extra` rides along — a word of the caller's, e.g. "my local step failed" — and comes back as extra_all[world])
    template <int PW>
    int route_exchange(ShardComm *c, const uint32_t *dest, const uint64_t *pay, uint32_t n_items, Routed &r, std::string &err,
                       uint64_t extra = 0, std::vector<uint64_t> *extra_all = nullptr) {
        const uint32_t world = sh_world_;
        const uint32_t row = world + 1;
        constexpr uint64_t LOCAL_FAIL = 1ull << 63;       // in the word that rides along: "this rank could not stage its records"
        // Everything this rank can allocate before the sizes are known is allocated first (the send buffer from its upper
        // bound): a failure here still travels with the size exchange and every rank leaves together.
        int rc_local = 0;
        if ((rc_local = rt_row_.alloc(ROUTE_MAX_WORLD + 1, err))) return rc_local;     // (a few hundred bytes, also needed to say so)
        if ((rc_local = rt_all_.alloc((size_t)row * world + 1, err))) return rc_local;
        if (!rc_local) rc_local = rt_cur_.alloc(ROUTE_MAX_WORLD, err);
        if (!rc_local) rc_local = r.sidx.alloc(n_items, err);
        if (!rc_local) rc_local = r.send.alloc((size_t)n_items * PW + 1, err);
        if (rc_local) { n_items = 0; extra |= LOCAL_FAIL; }
        HIPCHK(hipMemsetAsync(rt_row_.p, 0, (ROUTE_MAX_WORLD + 1) * 8, stream_));
        if (n_items) {
            hipLaunchKernelGGL(k_route_count, dim3(grid_for(n_items)), dim3(256), 0, stream_, dest, n_items, rt_row_.p);
            HIPCHK(hipGetLastError());
        }
        rt_extra_host_ = extra;                               // (a member: the asynchronous copy reads it after this line)
        HIPCHK(hipMemcpyAsync(rt_row_.p + world, &rt_extra_host_, 8, hipMemcpyHostToDevice, stream_));
        if (int rc = comm_allgather(c, rt_row_.p, rt_all_.p, (size_t)row * 8, stream_, err)) return rc;
        std::vector<uint64_t> all((size_t)row * world);
        HIPCHK(hipMemcpyAsync(all.data(), rt_all_.p, all.size() * 8, hipMemcpyDeviceToHost, stream_));
        WAIT_STREAM();
        bool peer_failed = false;
        for (uint32_t s = 0; s < world; s++) if (all[(size_t)s * row + world] & LOCAL_FAIL) peer_failed = true;
        if (peer_failed) {                                    // every rank sees the same row: all leave here
            sh_agreed_ = true;
            if (rc_local) return rc_local;
            err = "sharded assembly: another rank ran out of device memory while staging an exchange"; return -5;
        }
        r.send_cnt.assign(world, 0); r.recv_cnt.assign(world, 0); r.n_send = 0; r.n_recv = 0;
        std::vector<unsigned long long> &cur = rt_cur_host_;           // (a member: the asynchronous copy below may read it after this call has returned)
        cur.assign(ROUTE_MAX_WORLD, 0);
        for (uint32_t d = 0; d < world; d++) { r.send_cnt[d] = all[(size_t)sh_rank_ * row + d]; cur[d] = r.n_send; r.n_send += r.send_cnt[d]; }
        for (uint32_t s = 0; s < world; s++) { r.recv_cnt[s] = all[(size_t)s * row + sh_rank_]; r.n_recv += r.recv_cnt[s]; }
        if (extra_all) { extra_all->assign(world, 0); for (uint32_t s = 0; s < world; s++) (*extra_all)[s] = all[(size_t)s * row + world] & ~LOCAL_FAIL; }
        // (the receive buffer can only be sized now: a failure here is this rank's alone — shard_assemble aborts the communicator)
        if (int rc = r.recv.alloc(r.n_recv * PW + 1, err)) return rc;
        if (n_items) {
            HIPCHK(hipMemcpyAsync(rt_cur_.p, cur.data(), ROUTE_MAX_WORLD * 8, hipMemcpyHostToDevice, stream_));
            hipLaunchKernelGGL((k_route_pack<PW>), dim3((n_items + ROUTE_CH - 1) / ROUTE_CH), dim3(256), 0, stream_, dest, pay, n_items,
                               rt_cur_.p, r.send.p, r.sidx.p);
            HIPCHK(hipGetLastError());
        }
        std::vector<uint64_t> so(world), sb(world), ro(world), rb(world);
        uint64_t a = 0, b = 0;
        for (uint32_t q = 0; q < world; q++) { so[q] = a; sb[q] = r.send_cnt[q] * PW * 8; a += sb[q]; ro[q] = b; rb[q] = r.recv_cnt[q] * PW * 8; b += rb[q]; }
        if (int rc = comm_alltoallv(c, r.send.p, so.data(), sb.data(), r.recv.p, ro.data(), rb.data(), stream_, err)) return rc;
        // (no wait: what follows is ordered on the same stream, and every buffer involved outlives the call)
        times_.add("shard_graph_exchanged_MB", (double)(r.n_send * PW * 8) / 1e6);
        return 0;
    }
    // the answers to what was received (one u64 per received record, in r.recv's order) travel back: back[i] = answer to r.send[i]
    int reply_exchange(ShardComm *c, const Routed &r, const unsigned long long *ans, DevBuf<unsigned long long> &back, std::string &err) {
        const uint32_t world = sh_world_;
        if (int rc = back.alloc(r.n_send + 1, err)) return rc;
        std::vector<uint64_t> so(world), sb(world), ro(world), rb(world);
        uint64_t a = 0, b = 0;
        for (uint32_t q = 0; q < world; q++) { so[q] = a; sb[q] = r.recv_cnt[q] * 8; a += sb[q]; ro[q] = b; rb[q] = r.send_cnt[q] * 8; b += rb[q]; }
        return comm_alltoallv(c, ans, so.data(), sb.data(), back.p, ro.data(), rb.data(), stream_, err);
    }

    // Collective.  Every way out leaves the ranks in step: a local failure travels with the next size exchange or agreement
    // round and all ranks return an error together (sh_agreed_); a failure nobody else can know of (device memory that runs
    // out after the sizes were agreed, a HIP error, a collective that fails) aborts the communicator at once, so that the
    // peers' watchdogs (comm_stream_wait) end their waits instead of blocking for ever, and shk_comm_free does not block.
    int shard_assemble(ShardComm *c, bool tips, bool bubbles, std::vector<RawContig> &out, std::string &err) override {
        sh_agreed_ = false;
        wd_comm_ = (c && comm_world(c) > 1) ? c : nullptr;
        const uint64_t waits0 = n_host_waits_;
        const int rc = shard_assemble_impl(c, tips, bubbles, out, err);
        times_.add("shard_host_waits_x1", (double)(n_host_waits_ - waits0));     // stream waits + host-side collectives of this call
        if (rc && wd_comm_ && !sh_agreed_) comm_abort_now(c);
        drain();                                           // (through the watchdog: the stream may hold collectives of a call that failed)
        wd_comm_ = nullptr;
        return rc;
    }
    // (the host-side collectives wait for the stream: counted like the stream waits)
    int counted_allreduce_host_u64(ShardComm *c, uint64_t *v, size_t n, void *st, std::string &e) { n_host_waits_++; return comm_allreduce_host_u64(c, v, n, st, e); }
    int counted_allgather_host_u64(ShardComm *c, const uint64_t *in, size_t n, uint64_t *out, void *st, std::string &e) { n_host_waits_++; return comm_allgather_host_u64(c, in, n, out, st, e); }
    int shard_assemble_impl(ShardComm *c, bool tips, bool bubbles, std::vector<RawContig> &out, std::string &err) {
        // SHK_STAGE_LOG=1: after every step the stream is drained and the step's name goes to stderr (which step a fault belongs to)
        const bool stage_log = getenv("SHK_STAGE_LOG") != nullptr;
        auto stage = [&](const char *what) {
            if (!stage_log) return;
            const hipError_t e = hipStreamSynchronize(stream_);
            fprintf(stderr, "[shard_assemble rank %d] %s done%s\n", comm_rank(c), what, e == hipSuccess ? "" : " (stream error)");
            fflush(stderr);
        };
        out.clear();
        if (!sh_active_) { err = "shard_assemble: the handle was not preprocessed by the sharded path"; return -2; }
        const uint32_t world = sh_world_, rank = sh_rank_;
        const uint32_t n = (uint32_t)n_solid_;
        const unsigned long long gbase = sh_gbase_[rank];
        // every local step's result travels with the next collective: all ranks leave together (as in shk_shard_preprocess)
        auto agree = [&](int local_rc, const char *stage) -> int {
            uint64_t f = local_rc ? 1u : 0u;
            std::string e2;
            if (int rc = counted_allreduce_host_u64(c, &f, 1, stream_, e2)) { if (local_rc) return local_rc; err = e2; return rc; }
            if (f) sh_agreed_ = true;                        // (every rank has seen the flag and leaves)
            if (local_rc) return local_rc;
            if (f) { err = std::string("sharded assembly: another rank failed during ") + stage; return -5; }
            return 0;
        };
        const double t_all0 = now_ms_();
        // ---- 1. adjacency: local tables, local candidates, then the candidates of other ranks
        // (every buffer a collective or a kernel of this call touches lives until the call's last wait: no wait is needed for
        // a buffer's sake; the waits that remain are the ones whose RESULT the host needs)
        xq_n_ = 0;
        EvTimer tg(stream_, stage_timers_);
        const int rc_local = build_graph(err);
        if (world == 1 && rc_local) return rc_local;
        Graph<W> g = graph_view();
        DevBuf<unsigned long long> xnb;                  // per cross query: the neighbour's oriented global id (~0: not a solid k-mer)
        Routed rq, hl;
        DevBuf<unsigned long long> xans, xback;
        DevBuf<uint32_t> hdest; DevBuf<uint64_t> hpay;
        if (world > 1) {
            // (the result of the local build travels with the size exchange of the first routed exchange)
            std::vector<uint64_t> flags;
            const int rc_x = rc_local ? rc_local : 0;
            if (int rc = route_exchange<W + 1>(c, xq_dest_.p, xq_pay_.p, rc_x ? 0u : xq_n_, rq, err, rc_x ? 1u : 0u, &flags)) { if (rc_x) return rc_x; return rc; }
            if (rc_x) { sh_agreed_ = true; return rc_x; }
            for (uint64_t f : flags) if (f) { sh_agreed_ = true; err = "sharded assembly: another rank failed during the local graph build"; return -5; }
            if (int rc = xans.alloc(rq.n_recv + 1, err)) return rc;
            if (rq.n_recv) {
                hipLaunchKernelGGL(k_xq_answer<W>, dim3(grid_for(rq.n_recv)), dim3(256), 0, stream_, g.keys, g.gt, rq.recv.p, rq.n_recv, gbase, xans.p);
                HIPCHK(hipGetLastError());
            }
            if (int rc = reply_exchange(c, rq, xans.p, xback, err)) return rc;
            if (int rc = xnb.alloc(xq_n_ + 1, err)) return rc;
            if (xq_n_) {
                hipLaunchKernelGGL(k_xq_apply, dim3(grid_for(xq_n_)), dim3(256), 0, stream_, xq_meta_.p, rq.sidx.p, xback.p, xq_n_, adj_.p, nb_.p, xnb.p);
                HIPCHK(hipGetLastError());
                if (keep_stages_) HIPCHK(hipMemcpyAsync(adj0_.p, adj_.p, n, hipMemcpyDeviceToDevice, stream_));    // (stage inspection: the complete initial adjacency)
            }
        } else if (int rc = xnb.alloc(1, err)) return rc;
        tg.stop_later("shard_graph_adjacency_total", pending_timers_);
        times_.add("shard_graph_cross_queries_x1e-3", xq_n_ * 1e-3);
        stage("1 adjacency");
        // ---- 2. half links: which links across ranks are simple
        EvTimer th(stream_, stage_timers_);
        DevBuf<uint32_t> xpred;                          // per local oriented node: record of hl.recv that names its simple predecessor on another rank
        if (int rc = xpred.alloc(2ull * n + 2, err)) return rc;
        HIPCHK(hipMemsetAsync(xpred.p, 0xFF, (2ull * n + 2) * 4, stream_));
        HIPCHK(fill2_async(ctl_.p + 13, 8, 0u, nullptr, 0, 0u, stream_));
        if (world > 1) {
            if (int rc = hdest.alloc(2ull * n + 1, err)) return rc;
            if (int rc = hpay.alloc(4ull * n + 2, err)) return rc;
            if (n) {
                hipLaunchKernelGGL(k_hl_stage, dim3(grid_for(2ull * n)), dim3(256), 0, stream_, adj_.p, nb_.p, n, xnb.p, xq_meta_.p, gbase, hdest.p, hpay.p);
                HIPCHK(hipGetLastError());
            }
            if (int rc = route_exchange<2>(c, hdest.p, hpay.p, 2u * n, hl, err)) return rc;
            if (hl.n_recv) {
                hipLaunchKernelGGL(k_hl_apply, dim3(grid_for(hl.n_recv)), dim3(256), 0, stream_, hl.recv.p, hl.n_recv, gbase, n, adj_.p, xpred.p, (uint32_t *)(ctl_.p + 13));
                HIPCHK(hipGetLastError());
            }
        } else if (int rc = hl.recv.alloc(2, err)) return rc;
        th.stop_later("shard_graph_half_links", pending_timers_);
        stage("2 half links");
        // ---- 3. the chains of simple links that stay on this rank (the single-GPU contraction)
        ChainState cs;
        int rc_chain = 0;
        if (n) rc_chain = rank_chains(cs, false, err);
        const uint32_t n_lch = n ? (uint32_t)cs.n_heads : 0u;
        stage("3 local chains");
        // ---- 4. stitching: the local chains of all ranks, ranked by every rank
        EvTimer ts(stream_, stage_timers_);
        std::vector<uint64_t> lcnt(world, 0), lbase(world + 1, 0);
        {
            uint64_t mine[2] = {rc_chain ? 0u : n_lch, rc_chain ? 1u : 0u};
            std::vector<uint64_t> all2((size_t)world * 2);
            if (int rc = counted_allgather_host_u64(c, mine, 2, all2.data(), stream_, err)) { if (rc_chain) return rc_chain; return rc; }
            if (rc_chain) { sh_agreed_ = true; return rc_chain; }
            for (uint32_t r = 0; r < world; r++) { if (all2[2 * r + 1]) { sh_agreed_ = true; err = "sharded assembly: another rank failed during the local contraction"; return -5; } lcnt[r] = all2[2 * r]; }
            for (uint32_t r = 0; r < world; r++) lbase[r + 1] = lbase[r] + lcnt[r];
        }
        const uint64_t M = lbase[world];
        if (M >= 0xFFFFFFF0ull) { sh_agreed_ = true; err = "sharded assembly: more than 2^32 local chains"; return -1; }      // (the same sum on every rank)
        DevBuf<SegRec> lsegs, gsegs;
        DevBuf<unsigned long long> d_gbases;
        if (int rc = lsegs.alloc(n_lch + 1, err)) return rc;
        if (int rc = gsegs.alloc(M + 1, err)) return rc;
        if (int rc = d_gbases.alloc(world + 1, err)) return rc;
        HIPCHK(hipMemcpyAsync(d_gbases.p, sh_gbase_.data(), (size_t)(world + 1) * 8, hipMemcpyHostToDevice, stream_));
        DevBuf<uint32_t> ldest; DevBuf<uint64_t> lpay; DevBuf<unsigned long long> lans, lback;
        Routed ls;
        {
            if (int rc = ldest.alloc(n_lch + 1, err)) return rc;
            if (int rc = lpay.alloc(n_lch + 1, err)) return rc;
            if (n_lch) {
                hipLaunchKernelGGL(k_lchain_stage, dim3(grid_for(n_lch)), dim3(256), 0, stream_, cs.d_heads.p, n_lch, xpred.p, hl.recv.p,
                                   d_gbases.p, world, lsegs.p, ldest.p, lpay.p);
                HIPCHK(hipGetLastError());
            }
            if (world > 1) {
                if (int rc = route_exchange<1>(c, ldest.p, lpay.p, n_lch, ls, err)) return rc;
                if (int rc = lans.alloc(ls.n_recv + 1, err)) return rc;
                if (ls.n_recv) {
                    hipLaunchKernelGGL(k_ls_answer, dim3(grid_for(ls.n_recv)), dim3(256), 0, stream_, ls.recv.p, ls.n_recv, gbase, n, cs.ol.p,
                                       (uint32_t)lbase[rank], lans.p, (uint32_t *)(ctl_.p + 13));
                    HIPCHK(hipGetLastError());
                }
                if (int rc = reply_exchange(c, ls, lans.p, lback, err)) return rc;
            } else { if (int rc = lback.alloc(1, err)) return rc; if (int rc = ls.sidx.alloc(n_lch + 1, err)) return rc; }
            if (n_lch) {
                hipLaunchKernelGGL(k_ls_apply, dim3(grid_for(n_lch)), dim3(256), 0, stream_, ldest.p, ls.sidx.p, lback.p, n_lch, (uint32_t)lbase[rank], lsegs.p);
                HIPCHK(hipGetLastError());
            }
            // gather the chain records of all ranks (32 bytes each)
            std::vector<uint64_t> off(world), len(world);
            for (uint32_t r = 0; r < world; r++) { off[r] = lbase[r] * sizeof(SegRec); len[r] = lcnt[r] * sizeof(SegRec); }
            if (int rc = comm_allgatherv(c, lsegs.p, gsegs.p, off.data(), len.data(), stream_, err)) return rc;
        }
        // rank the gathered list: unitigs (rings across ranks in the same pass: collapse.h)
        DevBuf<RankRec> Ra, Rb; DevBuf<uint32_t> slot_of, uid_of_slot; DevBuf<FinRec> fin; DevBuf<UHead> d_uheads; DevBuf<unsigned int> d_M;
        const uint32_t Mu = (uint32_t)M;
        if (int rc = Ra.alloc(M + 1, err)) return rc;
        if (int rc = Rb.alloc(M + 1, err)) return rc;
        if (int rc = slot_of.alloc(M + 1, err)) return rc;
        if (int rc = fin.alloc(M + 1, err)) return rc;
        if (int rc = d_uheads.alloc(M + 1, err)) return rc;
        if (int rc = d_M.alloc(2, err)) return rc;
        unsigned int n_u = 0;
        std::vector<UHead> uheads;
        if (Mu) {
            const unsigned int hM[2] = {Mu, 0u};
            HIPCHK(hipMemcpyAsync(d_M.p, hM, 8, hipMemcpyHostToDevice, stream_));
            HIPCHK(hipMemsetAsync(slot_of.p, 0xFF, (size_t)M * 4, stream_));
            const int gr = grid_for(M);
            int rounds = 1; { uint64_t reach = RANK_HOPS; while (reach < M + 1u) { reach *= RANK_HOPS; rounds++; } }
            hipLaunchKernelGGL(k_rank_init, dim3(gr), dim3(256), 0, stream_, gsegs.p, (const unsigned int *)d_M.p, Ra.p);
            RankRec *Ri = Ra.p, *Ro = Rb.p;
            for (int r = 0; r < rounds; r++) { hipLaunchKernelGGL(k_rank_jump, dim3(gr), dim3(256), 0, stream_, (const unsigned int *)d_M.p, Ri, Ro); std::swap(Ri, Ro); }
            hipLaunchKernelGGL(k_stitch_tails, dim3(gr), dim3(256), 0, stream_, gsegs.p, (const unsigned int *)d_M.p, Ri, d_uheads.p, slot_of.p, d_M.p + 1);
            hipLaunchKernelGGL(k_rank_fin, dim3(gr), dim3(256), 0, stream_, gsegs.p, (const unsigned int *)d_M.p, Ri, slot_of.p, fin.p);
            HIPCHK(hipGetLastError());
            unsigned int hM2[2] = {0, 0};
            unsigned long long h_fl = 0;
            HIPCHK(hipMemcpyAsync(hM2, d_M.p, 8, hipMemcpyDeviceToHost, stream_));
            HIPCHK(hipMemcpyAsync(&h_fl, ctl_.p + 13, 8, hipMemcpyDeviceToHost, stream_));      // (flags of k_hl_apply / k_ls_answer, read with the counts)
            WAIT_STREAM();
            // (these flags are LOCAL: the verdict must be every rank's — a rank that left alone would leave the others in the
            // next collective and itself one collective ahead for the rest of the process)
            int rc_fl = 0;
            if ((uint32_t)h_fl) {
                err = (uint32_t)h_fl == 1 ? "sharded assembly: a record reached a rank that does not own its k-mer (ownership rules disagree)"
                                         : "sharded assembly: the two sides of a link across ranks disagree";
                rc_fl = -6;
            }
            if (world > 1) { if (int rc = agree(rc_fl, "the stitching of the chains")) return rc; }
            else if (rc_fl) return rc_fl;
            n_u = hM2[1];
            uheads.resize(n_u);
            if (n_u) HIPCHK(hipMemcpy(uheads.data(), d_uheads.p, (size_t)n_u * sizeof(UHead), hipMemcpyDeviceToHost));
        }
        if (ts.on()) times_.add("shard_graph_stitch", ts.stop());
        times_.add("shard_graph_local_chains_x1e-3", n_lch * 1e-3);
        times_.add("shard_graph_unitigs_x1", (double)n_u);
        stage("4 stitching");
        // ---- 5. the unitig graph on the host (identical on every rank: the records are put in the order of their first chain)
        const double t_host0 = now_ms_();
        const bool ug_dbg = getenv("SHK_UG_DEBUG") != nullptr;     // stage times of this host section on stderr
        double t_lap = t_host0;
        auto lap = [&](const char *what) { if (ug_dbg) { const double t = now_ms_(); fprintf(stderr, "[shard_assemble host] %-14s %8.1f ms\n", what, t - t_lap); t_lap = t; } };
        // (the first chains are distinct numbers below M: a table instead of a sort — a metagenome has millions of unitigs)
        std::vector<uint32_t> order(n_u);
        {
            std::vector<uint32_t> slot_at((size_t)M + 1, 0xFFFFFFFFu);
            for (uint32_t i = 0; i < n_u; i++) {
                // (the gathered records and their ranking are the same on every rank: so is this verdict)
                if (uheads[i].root >= M || slot_at[uheads[i].root] != 0xFFFFFFFFu) { sh_agreed_ = true; err = "sharded assembly: two unitigs with one first chain"; return -6; }
                slot_at[uheads[i].root] = i;
            }
            uint32_t at = 0;
            for (uint32_t c0 = 0; c0 < M; c0++) if (slot_at[c0] != 0xFFFFFFFFu) order[at++] = slot_at[c0];
        }
        lap("order");
        std::vector<uint32_t> h_uid_of_slot(n_u);
        for (uint32_t u = 0; u < n_u; u++) h_uid_of_slot[order[u]] = u;
        // first / last k-mer of every unitig: asked of the rank that holds that chain
        std::vector<EndReq> reqs;
        reqs.reserve(world == 1 ? 2 * (size_t)n_u : 2 * (size_t)n_u / world + 1024);
        for (uint32_t u = 0; u < n_u; u++) {
            const UHead &h = uheads[order[u]];
            if (h.root >= lbase[rank] && h.root < lbase[rank + 1]) reqs.push_back(EndReq{u, 0u, (uint32_t)(h.root - lbase[rank]), 0u});
            if (h.tail >= lbase[rank] && h.tail < lbase[rank + 1]) reqs.push_back(EndReq{u, 1u, (uint32_t)(h.tail - lbase[rank]), 0u});
        }
        lap("requests");
        DevBuf<EndReq> d_req; DevBuf<uint64_t> d_ends;
        const size_t ends_words = (size_t)n_u * 2 * W;
        if (int rc = d_req.alloc(reqs.size() + 1, err)) return rc;
        if (int rc = d_ends.alloc(ends_words + 1, err)) return rc;
        if (int rc = uid_of_slot.alloc(n_u + 1, err)) return rc;
        std::vector<uint64_t> h_ends(ends_words + 1, 0);
        if (n_u) {
            HIPCHK(hipMemsetAsync(d_ends.p, 0, ends_words * 8, stream_));
            HIPCHK(hipMemcpyAsync(uid_of_slot.p, h_uid_of_slot.data(), (size_t)n_u * 4, hipMemcpyHostToDevice, stream_));
            if (!reqs.empty()) {
                HIPCHK(hipMemcpyAsync(d_req.p, reqs.data(), reqs.size() * sizeof(EndReq), hipMemcpyHostToDevice, stream_));
                hipLaunchKernelGGL(k_fill_ends<W>, dim3(grid_for(reqs.size())), dim3(256), 0, stream_, g, cs.d_heads.p, d_req.p, (uint32_t)reqs.size(), d_ends.p);
                HIPCHK(hipGetLastError());
            }
            if (int rc = comm_allreduce_u64(c, d_ends.p, ends_words, stream_, err)) return rc;     // (every word is written by exactly one rank)
            HIPCHK(hipMemcpyAsync(h_ends.data(), d_ends.p, ends_words * 8, hipMemcpyDeviceToHost, stream_));
            WAIT_STREAM();
        }
        lap("ends");
        std::vector<UnitigRec> recs(n_u);
        host_par_ranges(n_u, [&](size_t a, size_t b) {
            for (size_t u = a; u < b; u++) {
                const UHead &h = uheads[order[u]];
                for (int w = 0; w < W; w++) { recs[u].first[w] = h_ends[(u * 2 + 0) * W + w]; recs[u].last[w] = h_ends[(u * 2 + 1) * W + w]; }
                recs[u].len = h.len; recs[u].kc = h.kc; recs[u].circ = h.circ;
            }
        });
        lap("records");
        UnitigGraphResult res;
        int rc_ug = unitig_assemble(k_, recs, tips, bubbles, res, err);
        lap("unitig graph");
        // (the same code on the same records: it fails on every rank or on none — an agreement round is only paid where the
        // graph is large enough for one host to run out of memory alone)
        if (n_u >= (1u << 20)) { if (int rc = agree(rc_ug ? -6 : 0, "the unitig graph")) return rc; }
        else if (rc_ug) { sh_agreed_ = true; return -6; }
        tips_removed_ = res.tips_removed; bubbles_removed_ = res.bubbles_removed; rounds_ = res.rounds;
        // rings: the smallest k-mer of their records — a pass over the nodes of the rings, on every rank, merged on the host
        if (!res.need_min.empty()) {
            const uint32_t n_rings = (uint32_t)res.need_min.size();
            std::vector<uint32_t> ring_of_uid(n_u, NIL);
            for (uint32_t i = 0; i < n_rings; i++) ring_of_uid[res.need_min[i]] = i;
            DevBuf<uint32_t> d_ring_of_uid, ring_of, vmin; DevBuf<unsigned long long> pmin; DevBuf<uint64_t> rep;
            if (int rc = d_ring_of_uid.alloc(n_u + 1, err)) return rc;
            if (int rc = ring_of.alloc(n_lch + 1, err)) return rc;
            if (int rc = vmin.alloc(n_rings + 1, err)) return rc;
            if (int rc = pmin.alloc(n_rings + 1, err)) return rc;
            if (int rc = rep.alloc((size_t)n_rings * (W + 2) + 1, err)) return rc;
            HIPCHK(hipMemcpyAsync(d_ring_of_uid.p, ring_of_uid.data(), (size_t)n_u * 4, hipMemcpyHostToDevice, stream_));
            HIPCHK(hipMemsetAsync(pmin.p, 0xFF, (size_t)n_rings * 8, stream_));
            HIPCHK(hipMemsetAsync(vmin.p, 0xFF, (size_t)n_rings * 4, stream_));
            std::vector<uint64_t> h_pmin(n_rings, ~0ull), all_pmin((size_t)world * n_rings);
            if (n_lch) {
                hipLaunchKernelGGL(k_ring_of, dim3(grid_for(n_lch)), dim3(256), 0, stream_, fin.p, (uint32_t)lbase[rank], n_lch, uid_of_slot.p, d_ring_of_uid.p, ring_of.p);
                hipLaunchKernelGGL(k_sring_min1<W>, dim3(grid_for(2ull * n)), dim3(256), 0, stream_, g, cs.ol.p, ring_of.p, pmin.p);
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(h_pmin.data(), pmin.p, (size_t)n_rings * 8, hipMemcpyDeviceToHost, stream_));
                WAIT_STREAM();
            }
            if (int rc = counted_allgather_host_u64(c, h_pmin.data(), n_rings, all_pmin.data(), stream_, err)) return rc;
            for (uint32_t i = 0; i < n_rings; i++) for (uint32_t r = 0; r < world; r++) h_pmin[i] = std::min(h_pmin[i], all_pmin[(size_t)r * n_rings + i]);
            std::vector<uint64_t> h_rep((size_t)n_rings * (W + 2), ~0ull), all_rep((size_t)world * n_rings * (W + 2));
            if (n_lch) {
                HIPCHK(hipMemcpyAsync(pmin.p, h_pmin.data(), (size_t)n_rings * 8, hipMemcpyHostToDevice, stream_));
                hipLaunchKernelGGL(k_sring_min2<W>, dim3(grid_for(2ull * n)), dim3(256), 0, stream_, g, cs.ol.p, ring_of.p, pmin.p, vmin.p);
                hipLaunchKernelGGL(k_sring_report<W>, dim3(grid_for(n_rings)), dim3(256), 0, stream_, g, cs.ol.p, fin.p, (uint32_t)lbase[rank], vmin.p, n_rings, rep.p);
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(h_rep.data(), rep.p, h_rep.size() * 8, hipMemcpyDeviceToHost, stream_));
                WAIT_STREAM();
            }
            if (int rc = counted_allgather_host_u64(c, h_rep.data(), h_rep.size(), all_rep.data(), stream_, err)) return rc;
            std::vector<UnitigMinKey> mk(n_u);
            for (uint32_t i = 0; i < n_rings; i++) {
                UnitigMinKey best;
                for (uint32_t r = 0; r < world; r++) {
                    const uint64_t *o = &all_rep[((size_t)r * n_rings + i) * (W + 2)];
                    if (o[W] == ~0ull) continue;
                    UnitigMinKey cand; cand.valid = true; cand.o = (uint32_t)o[W]; cand.pos = o[W + 1];
                    for (int w = 0; w < W; w++) cand.key[w] = o[w];
                    bool less = !best.valid;
                    if (!less) {
                        bool decided = false;
                        for (int w = W - 1; w >= 0 && !decided; w--) if (cand.key[w] != best.key[w]) { less = cand.key[w] < best.key[w]; decided = true; }
                        if (!decided) less = cand.o < best.o;
                    }
                    if (less) best = cand;
                }
                mk[res.need_min[i]] = best;
            }
            int rc_rr = unitig_resolve_rings(k_, recs, mk, res, err);
            if (rc_rr) { sh_agreed_ = true; return -6; }      // (deterministic on identical input: every rank takes the same way out)
        }
        lap("rings");
        // layout: contig text offsets, and for every unitig record where its nodes go
        std::vector<ULayout> lay(n_u + 1);
        for (auto &L : lay) { L.off = ~0ull; L.node_off = 0; L.ring_len = 0; L.rot = 0; }
        uint64_t text_bytes = 0;
        std::vector<uint64_t> c_off(res.contigs.size());
        for (size_t i = 0; i < res.contigs.size(); i++) {
            const UnitigContig &ct = res.contigs[i];
            c_off[i] = text_bytes;
            uint64_t node_off = 0;
            for (uint32_t r : ct.recs) {
                lay[r].off = text_bytes; lay[r].node_off = node_off; lay[r].ring_len = ct.ring ? ct.len_nodes : 0; lay[r].rot = ct.ring ? ct.rot : 0;
                node_off += recs[r].len;
            }
            text_bytes += ct.len_nodes + (uint64_t)(k_ - 1);
        }
        lap("layout");
        times_.add("shard_graph_unitig_host_clock", now_ms_() - t_host0);
        stage("5 unitig graph");
        // ---- 6. emission
        EvTimer te(stream_, stage_timers_);
        const uint64_t text_w32 = ((text_bytes + 15) / 16 + 1) & ~1ull;           // 2-bit words (16 bases each), an even number: all-reduced as u64
        DevBuf<ULayout> d_lay; DevBuf<char> d_text; DevBuf<uint32_t> d_pack;
        if (int rc = d_lay.alloc(n_u + 1, err)) return rc;
        if (int rc = d_text.alloc(text_w32 * 16 + 16, err)) return rc;
        if (int rc = d_pack.alloc(text_w32 + 2, err)) return rc;
        if (int rc = hout_.alloc(text_w32 * 16 + 16, err)) return rc;
        if (text_bytes) {
            HIPCHK(hipMemsetAsync(d_text.p, 0, text_w32 * 16, stream_));
            HIPCHK(hipMemcpyAsync(d_lay.p, lay.data(), (size_t)(n_u + 1) * sizeof(ULayout), hipMemcpyHostToDevice, stream_));
            if (n) {
                hipLaunchKernelGGL(k_shard_emit<W>, dim3(grid_for(n)), dim3(256), 0, stream_, g, cs.ol.p, fin.p, (uint32_t)lbase[rank], uid_of_slot.p, d_lay.p, d_text.p);
                HIPCHK(hipGetLastError());
            }
            if (world > 1) {
                // every base is written by exactly one rank: packed 2 bits per base (A = 0 = "not mine") the ranks' words add up
                hipLaunchKernelGGL(k_text_pack2, dim3(grid_for(text_w32)), dim3(256), 0, stream_, d_text.p, text_bytes, d_pack.p, text_w32);
                HIPCHK(hipGetLastError());
                if (int rc = comm_allreduce_u64(c, d_pack.p, text_w32 / 2, stream_, err)) return rc;
                hipLaunchKernelGGL(k_text_unpack2, dim3(grid_for(text_w32)), dim3(256), 0, stream_, d_pack.p, text_w32, d_text.p);
                HIPCHK(hipGetLastError());
                times_.add("shard_graph_text_allreduce_MB", (double)(text_w32 * 4) / 1e6);
            }
            HIPCHK(hipMemcpyAsync(hout_.p, d_text.p, text_bytes, hipMemcpyDeviceToHost, stream_));
        }
        stage("6 emission kernels");
        WAIT_STREAM();
        if (te.on()) times_.add("shard_graph_emit", te.stop());
        out.reserve(res.contigs.size());
        for (size_t i = 0; i < res.contigs.size(); i++) {
            RawContig rcg; rcg.kc = res.contigs[i].kc;
            rcg.ext = hout_.p + c_off[i]; rcg.ext_n = res.contigs[i].len_nodes + (uint64_t)(k_ - 1);
            out.push_back(std::move(rcg));
        }
        times_.add("shard_assemble_host_clock", now_ms_() - t_all0);
        return 0;
    }
    // a loop over millions of host records (the unitigs of a metagenome) on several threads
    template <typename F> static void host_par_ranges(size_t n, F &&fn) {
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned T = n < 262144 ? 1u : std::min(16u, std::max(1u, hw));
        if (T == 1) { fn((size_t)0, n); return; }
        std::vector<std::thread> ts;
        for (unsigned t = 1; t < T; t++) ts.emplace_back([&fn, n, t, T] { fn(n * t / T, n * (t + 1) / T); });
        fn((size_t)0, n / T);
        for (auto &t : ts) t.join();
    }
    static double now_ms_() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

private:
    int k_;
    hipStream_t stream_ = nullptr; int stream_dev_ = 0;
    hipStream_t copy_stream_ = nullptr;              // uploads that overlap pass 1 (count_batch_host)
    ShardComm *wd_comm_ = nullptr;                   // set while a collective call of several ranks runs: host waits go through its watchdog
    bool sh_agreed_ = false;                         // shard_assemble: the error it is about to return is known to every rank
    StageTimes times_;
    std::vector<EvTimer::Pending> pending_timers_;
    int n_cus_ = 256;
    DevBuf<unsigned long long> ctl_;
    PinnedBuf mbox_;                                 // pinned host words the small device-to-host copies land in (truly asynchronous)
    unsigned long long *mbox64() { return (unsigned long long *)mbox_.p; }
    static constexpr size_t MBOX_BYTES = 32768;
    static constexpr int MB_P1 = 0 /* 2 words: pass 1's flags */, MB_CTL = 8 /* CTL_WORDS words: a copy of ctl_ */, MB_MISC = 64;
    struct P1Pending { bool on; const uint32_t *d_bases, *d_seg_off; uint64_t n_seg, n_bases, cap; };
    P1Pending p1_{false, nullptr, nullptr, 0, 0, 0};
    uint64_t n_host_waits_ = 0;                      // host waits on the stream so far (WAIT_STREAM, host-side collectives)
    bool stage_timers_ = env_u64("SHK_STAGE_TIMERS", 0) != 0;    // per-stage HIP-event timers (the two counting passes are always timed: the roofline needs them)
    bool keep_stages_ = env_u64("SHK_KEEP_STAGES", 0) != 0;      // keep what only the stage inspection reads (the initial adjacency bytes)
    bool graph_check_pending_ = false, corr_pending_ = false, corr_tips_ = false, corr_bubbles_ = false;
    CorrScratch corr_;
    bool defer_p1_ = false;                          // set by the packed entry points: the batch is the only one and its reads outlive histogram()
    uint64_t cap_override_ = 0;
    // count table
    DevBuf<uint64_t> tkeys_[W]; DevBuf<uint32_t> tcnt_, tstate_; uint64_t tslots_ = 0;
    struct Batch { const uint32_t *bases, *seg_off; uint64_t n_seg; };
    std::vector<Batch> pending_;
    std::vector<DevPiece> piece_list_;               // count_batch_pieces: the pieces of the batch being partitioned
    uint64_t total_instances_ = 0, n_distinct_ = 0, n_solid_ = 0;
    uint64_t histo_[500] = {0};
    // partitioned counting
    bool global_mode_ = env_u64("SHK_COUNT_MODE_GLOBAL", 0) != 0;
    bool bloom_ = false, bloom_off_once_ = false;
    bool have_parts_ = false;
    bool have_dedup_ = false;                        // shard_dedupe ran: dd_* hold the distinct records of every partition
    bool split_ready_ = false;                       // histogram() prepared split_base_ / split_total_ for the two-kernel pass 2
    std::vector<unsigned long long> split_base_; unsigned long long split_total_ = 0, split_stride_ = 0;      // (split_stride_ != 0: split_base_[p] = p * split_stride_)
    DevBuf<uint64_t> dd_recs_; DevBuf<uint32_t> dd_w_, dd_n_; DevBuf<unsigned long long> dd_base_;
    PartParams pp_{};
    DevBuf<uint64_t> recs_; DevBuf<uint32_t> fill_;
    DevBuf<unsigned long long> run_off_; DevBuf<uint32_t> run_cnt_;
    RunView run_view_{}; uint32_t n_count_parts_ = 0; uint32_t forced_P_ = 0;
    std::vector<std::unique_ptr<BatchRecs>> batches_;
    const void *shard_recv_ = nullptr;
    DevBuf<uint64_t> ekeys_[W]; DevBuf<uint32_t> ecnt_;
    uint64_t n_emitted_ = 0; uint32_t emit_threshold_ = 0;
    bool rows_scattered_ = false;                    // most rows came out of the k-mer-level repartition (run_count_partitions)
    // solid set / graph
    DevBuf<uint64_t> skeys_[W]; DevBuf<uint32_t> scnt_;
    DevBuf<uint64_t> gt_; DevBuf<uint8_t> gt_occ_; DevBuf<uint2> gt_scan_; uint64_t gt_slots_ = 0; uint32_t gp_ = 64;
    DevBuf<unsigned long long> gt_off_; DevBuf<uint32_t> gt_msk_;
    DevBuf<uint8_t> adj_, adj0_, alive_;
    DevBuf<uint32_t> row_starts_;                    // one bit per solid row: a group of rows of one minimiser partition starts here (k_row_starts)
    PinnedBuf hout_;                  // contigs as downloaded; RawContig::ext points into it
    PinnedBuf hends_, hflags_;        // their first / last bases (RawContig::head / tail) and the arrival flags when the text arrives slab by slab
    SlabArrival arrival_;
    DevBuf<char> d_out_keep_; DevBuf<unsigned long long> ends_src_;
    std::vector<unsigned long long> ends_src_host_;
    PinnedBuf hjson_;                 // the JSON of a fragmented assembly, written on the device (device_write_json)
    DevBuf<uint32_t> nb_;
    bool graph_ready_ = false;
    uint64_t tips_removed_ = 0, bubbles_removed_ = 0; int rounds_ = 0;
    // sharded assembly (shard_graph.h)
    bool sh_active_ = false;
    uint32_t sh_world_ = 1, sh_rank_ = 0, sh_P_ = 1;
    std::vector<uint64_t> sh_gbase_;                 // [world + 1] first global node id of every rank
    uint64_t n_solid_global_ = 0;
    uint32_t xq_n_ = 0;                              // cross-rank neighbour queries of this rank (build_graph)
    DevBuf<uint32_t> xq_dest_; DevBuf<uint64_t> xq_pay_; DevBuf<unsigned long long> xq_meta_;
    DevBuf<unsigned long long> rt_row_, rt_all_, rt_cur_;   // the router's small buffers (kept: no wait for their sake)
    std::vector<unsigned long long> rt_cur_host_;
    uint64_t rt_extra_host_ = 0;
};

int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
int set_device(int dev) { int prev = 0; (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); return prev; }
int device_download(void *host, const void *dptr, size_t bytes, std::string &err) {
    HIPCHK(hipMemcpy(host, dptr, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

IPipeline *make_pipeline(int k, std::string &err) {
    if (device_count() <= 0) { err = "no HIP device available (libshk_hip has no CPU fallback)"; return nullptr; }
    const int W = (2 * k + 63) / 64;
    int rc = -1;
    IPipeline *p = nullptr;
    if (W == 1) { auto *q = new Pipeline<1>(k); rc = q->init(err); p = q; }
    else if (W == 2) { auto *q = new Pipeline<2>(k); rc = q->init(err); p = q; }
    else if (W == 3) { auto *q = new Pipeline<3>(k); rc = q->init(err); p = q; }
    else if (W == 4) { auto *q = new Pipeline<4>(k); rc = q->init(err); p = q; }
    else if (W == 5) { auto *q = new Pipeline<5>(k); rc = q->init(err); p = q; }
    else if (W == 6) { auto *q = new Pipeline<6>(k); rc = q->init(err); p = q; }
    else if (W == 7) { auto *q = new Pipeline<7>(k); rc = q->init(err); p = q; }
    else if (W == 8) { auto *q = new Pipeline<8>(k); rc = q->init(err); p = q; }
    else { err = "k too large for the compiled key widths"; return nullptr; }
    if (rc != 0) { delete p; return nullptr; }
    return p;
}

int device_upload(const void *host, size_t bytes, void **dptr, std::string &err) {
    *dptr = nullptr;
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 4));
    if (bytes) {
        hipError_t e = hipMemcpy(*dptr, host, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(*dptr); *dptr = nullptr; err = hipGetErrorString(e); return -5; }
    }
    return 0;
}
void device_free(void *dptr) { if (dptr) (void)hipFree(dptr); }
int device_stream_sync(void *stream, std::string &err) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return 0; }
int device_copy_h2d_async(void *dst, const void *src, size_t bytes, void *stream, std::string &err) {
    if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}

// ---- host-side self-test helpers (same arithmetic as the kernels) ---------------------------
template <int W> static int host_canon_t(const char *seq, uint32_t k, uint64_t *out, int *orient) {
    Kmer<W> f = km_zero<W>();
    for (uint32_t i = 0; i < k; i++) {
        uint32_t b;
        switch (seq[i]) { case 'A': case 'a': b = 0; break; case 'C': case 'c': b = 1; break;
                          case 'G': case 'g': b = 2; break; case 'T': case 't': b = 3; break; default: return -1; }
        km_push_back<W>(f, b, (int)k);
    }
    int o; Kmer<W> c = km_canonical<W>(f, (int)k, o);
    for (int j = 0; j < W; j++) out[j] = c.w[j];
    if (orient) *orient = o;
    return 0;
}
// ---- measurement helper: what a pure streaming read reaches on this box (SURVEY.md 8d asks for the
// achievable HBM peak beside the nominal 8 TB/s).  16 bytes per lane and load, 8 loads in flight.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_read(const u32x4_t *__restrict__ src, uint64_t n16, uint32_t *__restrict__ sink) {
    u32x4_t acc = {0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= src[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[0] = 1;      // never with the fill pattern: keeps the loads alive
}
int stream_read_gbs(size_t bytes, int iters, double *gbs, std::string &err) {
    if (!gbs || bytes < (1u << 20) || iters < 1) { err = "bad arguments"; return -1; }
    void *buf = nullptr; uint32_t *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) {
        (void)hipGetLastError(); if (buf) (void)hipFree(buf);
        err = "out of device memory"; return -4;
    }
    hipStream_t st = nullptr;
    (void)hipMemsetAsync(buf, 0x5A, bytes, st); (void)hipMemsetAsync(sink, 0, 4, st);
    const uint64_t n16 = bytes / 16;
    const dim3 grid(256 * 16);
    hipLaunchKernelGGL(k_stream_read, grid, dim3(256), 0, st, (const u32x4_t *)buf, n16, sink);   // warm-up
    double best = 0;
    for (int it = 0; it < iters; it++) {
        EvTimer t(st);
        hipLaunchKernelGGL(k_stream_read, grid, dim3(256), 0, st, (const u32x4_t *)buf, n16, sink);
        const double ms = t.stop();
        if (ms > 0) best = std::max(best, (double)(n16 * 16) / (ms * 1e-3) / 1e9);
    }
    const hipError_t e = hipGetLastError();
    (void)hipFree(buf); (void)hipFree(sink);
    if (e != hipSuccess) { err = hipGetErrorString(e); return -5; }
    *gbs = best;
    return 0;
}

int host_canonical(const char *seq, uint32_t k, uint64_t *out, int *orient) {
    const int W = (2 * k + 63) / 64;
    if (W == 1) return host_canon_t<1>(seq, k, out, orient);
    if (W == 2) return host_canon_t<2>(seq, k, out, orient);
    if (W == 3) return host_canon_t<3>(seq, k, out, orient);
    if (W == 4) return host_canon_t<4>(seq, k, out, orient);
    if (W == 5) return host_canon_t<5>(seq, k, out, orient);
    if (W == 6) return host_canon_t<6>(seq, k, out, orient);
    if (W == 7) return host_canon_t<7>(seq, k, out, orient);
    if (W == 8) return host_canon_t<8>(seq, k, out, orient);
    return -1;
}
uint64_t host_nthash(const char *seq, uint32_t k) {
    // roll across the first k bases exactly as the kernel does
    NtState nt{0, 0};
    for (uint32_t i = 0; i < k; i++) {
        uint32_t b = seq[i] == 'A' ? 0 : seq[i] == 'C' ? 1 : seq[i] == 'G' ? 2 : 3;
        nt_init_step(nt, b, i);
    }
    return nt_canonical(nt);
}

}  // namespace shk
