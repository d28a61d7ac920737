// shard_comm.hip — RCCL collectives of the shard layer (see shard_comm.h) and the host-side exchange plan.
//
// Traffic shape (DESIGN.md "Multi-GPU"): ONE pairwise exchange of super-k-mer records (grouped
// ncclSend/ncclRecv: xGMI is point-to-point, 7 links per GPU, and a pairwise exchange is the pattern that
// keeps all of them busy at once), a 4 KB all-reduce of the histogram and an all-gather of the solid rows.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <stdlib.h>
#include <string.h>

#include "shard_comm.h"
#include "pipeline.h"

namespace shk {

static_assert(SHARD_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {

// the RCCL entry points used here, bound at run time
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                       // optional (a stand-in library may lack it)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;   // optional
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
    std::string why;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // by soname first: a copy that is already mapped (torch ships one) is shared, not doubled.
        // SHK_RCCL_LIBRARY names another library with the same entry points (a site's own RCCL build; the tests'
        // shared-memory stand-in, tests/mock_rccl, which lets several ranks share the one GPU of a test box)
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        const char *over = getenv("SHK_RCCL_LIBRARY");
        // (dlerror() hands its message out once and clears it: one call per failure)
        if (over && *over) {
            h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!h) { const char *e = dlerror(); r.why = std::string("SHK_RCCL_LIBRARY: ") + (e ? e : "cannot be loaded"); return; }
        }
        std::string last;
        for (const char *n : names) {
            if (h) break;
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) { const char *e = dlerror(); last = e ? e : "cannot be loaded"; }
        }
        if (!h) { r.why = "librccl.so.1 not found: " + last; return; }
        auto sym = [&](const char *n) -> void * { void *p = dlsym(h, n); if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.ok = r.why.empty();
        r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
        r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
    });
    return r;
}

}  // namespace

// One RCCL call carries at most this many bytes (default 256 MiB; SHK_COMM_PIECE_BYTES lets the tests cut small exchanges
// into many pieces).  A multiple of 8.
static uint64_t piece_bytes() {
    static const uint64_t v = [] {
        const char *e = getenv("SHK_COMM_PIECE_BYTES");
        unsigned long long x = e && *e ? strtoull(e, nullptr, 10) : 0ull;
        if (x < 8) x = 256ull << 20;
        return (uint64_t)(x & ~7ull);
    }();
    return v;
}

struct ShardComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    // staging of the small host-side collectives, allocated with the communicator: a rank that runs out of device
    // memory later can still take part in them (the error agreement of shk_shard_preprocess rides on them)
    void *stage = nullptr; size_t stage_bytes = 0;
    bool broken = false;                   // a collective failed: the communicator is aborted, not destroyed
};

// (a failed collective leaves the communicator in an undefined state: it is marked and later aborted, so that
// peers blocked in the same collective fail fast instead of waiting for this rank for ever)
#define RCCLCHK(call)                                                                         \
    do {                                                                                      \
        const ncclResult_t _r = (call);                                                       \
        if (_r != ncclSuccess) {                                                              \
            err = std::string(#call) + ": " + (R.GetErrorString ? R.GetErrorString(_r) : "?"); \
            if (c) c->broken = true;                                                          \
            return -5;                                                                        \
        }                                                                                     \
    } while (0)

int comm_unique_id(uint8_t id[SHARD_UNIQUE_ID_BYTES], std::string &err) {
    Rccl &R = rccl();
    ShardComm *c = nullptr;
    if (!R.ok) { err = R.why; return -5; }
    ncclUniqueId u;
    RCCLCHK(R.GetUniqueId(&u));
    memcpy(id, u.internal, SHARD_UNIQUE_ID_BYTES);
    return 0;
}

ShardComm *comm_create(const uint8_t id[SHARD_UNIQUE_ID_BYTES], int rank, int world, std::string &err) {
    Rccl &R = rccl();
    if (!R.ok) { err = R.why; return nullptr; }
    if (world < 1 || rank < 0 || rank >= world) { err = "comm: bad rank / world"; return nullptr; }
    ShardComm *c = new (std::nothrow) ShardComm();
    if (!c) { err = "out of host memory"; return nullptr; }
    c->rank = rank; c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) { err = "no HIP device"; delete c; return nullptr; }
    ncclUniqueId u;
    memcpy(u.internal, id, SHARD_UNIQUE_ID_BYTES);
    const ncclResult_t r = R.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        err = std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(r) : "?");
        delete c; return nullptr;
    }
    // room for the largest small collective: world x (2 PART_MAX_P + 2) counts (raw and deduplicated records per partition), plus this rank's contribution
    c->stage_bytes = ((size_t)world + 1) * (2 * 16384 + 8) * 8;
    if (hipMalloc(&c->stage, c->stage_bytes) != hipSuccess) {
        (void)hipGetLastError();
        err = "out of device memory (communicator staging)";
        (void)R.CommDestroy(c->comm); delete c; return nullptr;
    }
    return c;
}

void comm_destroy(ShardComm *c) {
    if (!c) return;
    Rccl &R = rccl();
    if (c->comm && R.ok) {
        if (c->broken && R.CommAbort) (void)R.CommAbort(c->comm);
        else (void)R.CommDestroy(c->comm);
    }
    if (c->stage) (void)hipFree(c->stage);
    delete c;
}
void comm_mark_broken(ShardComm *c) { if (c) c->broken = true; }
bool comm_broken(const ShardComm *c) { return c && c->broken; }
void comm_abort_now(ShardComm *c) {
    if (!c) return;
    c->broken = true;
    Rccl &R = rccl();
    if (c->comm && R.ok && R.CommAbort) { (void)R.CommAbort(c->comm); c->comm = nullptr; }      // (comm_destroy then has nothing left to tear down)
}
static double comm_timeout_s() {
    static const double v = [] {
        const char *e = getenv("SHK_COMM_TIMEOUT_S");
        if (!e || !*e) return 300.0;
        const double x = strtod(e, nullptr);
        return x < 0 ? 300.0 : x;
    }();
    return v;
}
int comm_stream_wait(ShardComm *c, void *stream, std::string &err) {
    hipStream_t st = (hipStream_t)stream;
    if (!c || c->world <= 1) {
        // (nothing can keep a one-rank collective waiting: poll for a few milliseconds — the waits are short — then block)
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e;
        while ((e = hipStreamQuery(st)) == hipErrorNotReady)
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) { e = hipStreamSynchronize(st); break; }
        if (e != hipSuccess) { err = std::string("stream: ") + hipGetErrorString(e); return -5; }
        return 0;
    }
    if (!c->comm) { err = "the communicator was aborted"; return -5; }
    const auto t0 = std::chrono::steady_clock::now();
    auto last_check = t0;
    const double limit = comm_timeout_s();
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) { err = std::string("stream: ") + hipGetErrorString(e); comm_abort_now(c); return -5; }
        const auto now = std::chrono::steady_clock::now();
        if (now - last_check > std::chrono::milliseconds(5)) {
            last_check = now;
            const std::string ae = comm_async_error(c);
            if (!ae.empty()) { err = ae; comm_abort_now(c); (void)hipStreamSynchronize(st); return -5; }
            if (limit > 0 && std::chrono::duration<double>(now - t0).count() > limit) {
                err = "a collective did not complete within SHK_COMM_TIMEOUT_S (a peer left the call or died): communicator aborted";
                comm_abort_now(c); (void)hipStreamSynchronize(st); return -5;
            }
            std::this_thread::yield();
        }
    }
}
// an error RCCL found asynchronously (a peer that died, a link that failed): "" when there is none or the
// library cannot tell
std::string comm_async_error(ShardComm *c) {
    Rccl &R = rccl();
    if (!c || !R.ok || !R.CommGetAsyncError) return "";
    ncclResult_t ar = ncclSuccess;
    if (R.CommGetAsyncError(c->comm, &ar) != ncclSuccess || ar == ncclSuccess || ar == ncclInProgress) return "";
    c->broken = true;
    return std::string("RCCL asynchronous error: ") + (R.GetErrorString ? R.GetErrorString(ar) : "?");
}
int comm_rank(const ShardComm *c) { return c ? c->rank : 0; }
int comm_world(const ShardComm *c) { return c ? c->world : 1; }
int comm_device(const ShardComm *c) { return c ? c->device : 0; }

int comm_allreduce_u64(ShardComm *c, void *d_buf, size_t n, void *stream, std::string &err) {
    Rccl &R = rccl();
    if (!c || !R.ok) { err = "comm: not initialised"; return -1; }
    if (!c->comm) { err = "comm: the communicator was aborted"; return -5; }
    if (!n) return 0;
    // (pieces of <= 256 MiB per call, as everywhere in this file: see comm_alltoallv)
    const size_t PIECE_N = (size_t)piece_bytes() / 8;
    for (size_t o = 0; o < n; o += PIECE_N) {
        uint64_t *at = (uint64_t *)d_buf + o;
        RCCLCHK(R.AllReduce(at, at, std::min(PIECE_N, n - o), ncclUint64, ncclSum, c->comm, (hipStream_t)stream));
    }
    return 0;
}

int comm_allgather(ShardComm *c, const void *d_send, void *d_recv, size_t bytes, void *stream, std::string &err) {
    Rccl &R = rccl();
    if (!c || !R.ok) { err = "comm: not initialised"; return -1; }
    if (!c->comm) { err = "comm: the communicator was aborted"; return -5; }
    if (!bytes) return 0;
    if (bytes % 8) { err = "comm_allgather: bytes must be a multiple of 8"; return -1; }
    const size_t PIECE = (size_t)piece_bytes();
    if (bytes <= PIECE) {
        RCCLCHK(R.AllGather(d_send, d_recv, bytes / 8, ncclUint64, c->comm, (hipStream_t)stream));
        return 0;
    }
    // a large contribution travels as grouped broadcasts of <= 256 MiB (an all-gather cannot be cut without changing its layout)
    RCCLCHK(R.GroupStart());
    for (int s = 0; s < c->world; s++)
        for (size_t o = 0; o < bytes; o += PIECE) {
            char *dst = (char *)d_recv + (size_t)s * bytes + o;
            RCCLCHK(R.Broadcast(s == c->rank ? (const void *)((const char *)d_send + o) : (const void *)dst, dst, std::min(PIECE, bytes - o) / 8,
                                ncclUint64, s, c->comm, (hipStream_t)stream));
        }
    RCCLCHK(R.GroupEnd());
    return 0;
}

int comm_alltoallv(ShardComm *c, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes,
                   void *d_recv, const uint64_t *recv_off, const uint64_t *recv_bytes, void *stream, std::string &err, uint32_t elem) {
    Rccl &R = rccl();
    if (!c || !R.ok) { err = "comm: not initialised"; return -1; }
    if (!c->comm) { err = "comm: the communicator was aborted"; return -5; }
    if (elem != 8 && elem != 4) { err = "comm_alltoallv: elements of 4 or 8 bytes"; return -1; }
    for (int p = 0; p < c->world; p++)
        if ((send_off[p] | send_bytes[p] | recv_off[p] | recv_bytes[p]) % elem) { err = "comm_alltoallv: offsets and sizes must be multiples of the element size"; return -1; }
    const ncclDataType_t ty = elem == 8 ? ncclUint64 : ncclUint32;
    // The part a rank keeps for itself never touches the fabric: a device copy on the same stream.  (It also has to be one: a
    // grouped ncclSend / ncclRecv of a rank to ITSELF delivered only part of a 1.2 GB block on the GPU box — the rest of
    // the receive buffer kept what the pool block held before; found with the configs[4] share through a one-rank
    // communicator, round 3.)  Blocks for other ranks go out in pieces of <= 256 MiB: several sends to one peer inside a
    // group are matched in order.
    const int me = c->rank;
    if (send_bytes[me] != recv_bytes[me]) { err = "comm_alltoallv: a rank's block for itself must have one size"; return -1; }
    if (send_bytes[me] &&
        hipMemcpyAsync((char *)d_recv + recv_off[me], (const char *)d_send + send_off[me], send_bytes[me], hipMemcpyDeviceToDevice,
                       (hipStream_t)stream) != hipSuccess) {
        err = std::string("comm_alltoallv: ") + hipGetErrorString(hipGetLastError()); return -5;
    }
    if (c->world == 1) return 0;
    const uint64_t PIECE = piece_bytes();
    RCCLCHK(R.GroupStart());
    for (int p = 0; p < c->world; p++) {
        if (p == me) continue;
        for (uint64_t o = 0; o < send_bytes[p]; o += PIECE)
            RCCLCHK(R.Send((const char *)d_send + send_off[p] + o, std::min<uint64_t>(PIECE, send_bytes[p] - o) / elem, ty, p, c->comm, (hipStream_t)stream));
        for (uint64_t o = 0; o < recv_bytes[p]; o += PIECE)
            RCCLCHK(R.Recv((char *)d_recv + recv_off[p] + o, std::min<uint64_t>(PIECE, recv_bytes[p] - o) / elem, ty, p, c->comm, (hipStream_t)stream));
    }
    RCCLCHK(R.GroupEnd());
    return 0;
}

int comm_allgatherv(ShardComm *c, const void *d_send, void *d_recv, const uint64_t *off, const uint64_t *bytes,
                    void *stream, std::string &err) {
    Rccl &R = rccl();
    if (!c || !R.ok) { err = "comm: not initialised"; return -1; }
    if (!c->comm) { err = "comm: the communicator was aborted"; return -5; }
    for (int s = 0; s < c->world; s++)
        if ((off[s] | bytes[s]) % 4) { err = "comm_allgatherv: offsets and sizes must be multiples of 4"; return -1; }
    RCCLCHK(R.GroupStart());
    const uint64_t PIECE = piece_bytes();
    for (int s = 0; s < c->world; s++)
        for (uint64_t o = 0; o < bytes[s]; o += PIECE) {
            char *dst = (char *)d_recv + off[s] + o;
            RCCLCHK(R.Broadcast(s == c->rank ? (const void *)((const char *)d_send + o) : (const void *)dst, dst,
                                std::min<uint64_t>(PIECE, bytes[s] - o) / 4, ncclUint32, s, c->comm, (hipStream_t)stream));
        }
    RCCLCHK(R.GroupEnd());
    return 0;
}

// ---- small host-side collectives ---------------------------------------------------------------------
namespace {
struct PoolBlock {
    void *p = nullptr; size_t bytes = 0;
    explicit PoolBlock(size_t b, bool unused = false) : bytes(b ? b : 8) { if (!unused) p = device_pool_alloc(bytes); }
    ~PoolBlock() { if (p) device_pool_release(p, bytes); }
};
}  // namespace

int comm_allreduce_host_u64(ShardComm *c, uint64_t *host_inout, size_t n, void *stream, std::string &err) {
    if (!n) return 0;
    if (!c) { err = "comm: not initialised"; return -1; }
    PoolBlock b(n * 8 <= c->stage_bytes ? 0 : n * 8, n * 8 <= c->stage_bytes);
    void *buf = n * 8 <= c->stage_bytes ? c->stage : b.p;
    if (!buf) { err = "device allocation failed (all-reduce staging)"; return -4; }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(buf, host_inout, n * 8, hipMemcpyHostToDevice, st) != hipSuccess) { err = "hipMemcpyAsync failed"; return -5; }
    if (int rc = comm_allreduce_u64(c, buf, n, stream, err)) { (void)hipStreamSynchronize(st); return rc; }
    if (hipMemcpyAsync(host_inout, buf, n * 8, hipMemcpyDeviceToHost, st) != hipSuccess) {
        err = std::string("all-reduce: ") + hipGetErrorString(hipGetLastError()); c->broken = true; (void)hipStreamSynchronize(st); return -5;
    }
    return comm_stream_wait(c, stream, err);
}

int comm_allgather_host_u64(ShardComm *c, const uint64_t *host_in, size_t n, uint64_t *host_out, void *stream, std::string &err) {
    if (!n) return 0;
    if (!c) { err = "comm: not initialised"; return -1; }
    const size_t world = (size_t)comm_world(c);
    const bool fits = n * 8 * (world + 1) <= c->stage_bytes;
    PoolBlock blk(fits ? 0 : n * 8 * (world + 1), fits);
    char *base = fits ? (char *)c->stage : (char *)blk.p;
    if (!base) { err = "device allocation failed (all-gather staging)"; return -4; }
    void *in = base, *out = base + n * 8;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(in, host_in, n * 8, hipMemcpyHostToDevice, st) != hipSuccess) { err = "hipMemcpyAsync failed"; return -5; }
    if (int rc = comm_allgather(c, in, out, n * 8, stream, err)) { (void)hipStreamSynchronize(st); return rc; }
    if (hipMemcpyAsync(host_out, out, n * 8 * world, hipMemcpyDeviceToHost, st) != hipSuccess) {
        err = std::string("all-gather: ") + hipGetErrorString(hipGetLastError()); c->broken = true; (void)hipStreamSynchronize(st); return -5;
    }
    return comm_stream_wait(c, stream, err);
}

// ---- host-side plan -------------------------------------------------------------------------------
uint32_t choose_partitions(uint64_t total_instances_ub, uint32_t world, uint64_t per_part) {
    uint32_t P = 64;
    while (P < 16384u && (uint64_t)P * per_part < total_instances_ub) P <<= 1;
    while (P < world) P <<= 1;
    return P;
}

int plan_exchange(const uint64_t *pr, uint32_t world, uint32_t P, uint32_t rank, ExchangePlan &out, std::string &err) {
    if (!pr || world < 1 || rank >= world || P < world) { err = "plan_exchange: bad arguments"; return -1; }
    const uint64_t *mine = pr + (uint64_t)rank * P;
    out.base.assign(P, 0); out.send_counts.assign(world, 0); out.recv_counts.assign(world, 0);
    uint64_t off = 0;
    for (uint32_t d = 0; d < world; d++)                       // destination-major, partitions ascending
        for (uint32_t p = d; p < P; p += world) { out.base[p] = off; off += mine[p]; out.send_counts[d] += mine[p]; }
    out.owned.clear();
    for (uint32_t p = rank; p < P; p += world) out.owned.push_back(p);
    const uint32_t n_owned = (uint32_t)out.owned.size();
    std::vector<uint64_t> recv_base(world, 0);
    for (uint32_t s = 0; s < world; s++) {
        uint64_t t = 0;
        for (uint32_t j = 0; j < n_owned; j++) t += pr[(uint64_t)s * P + out.owned[j]];
        out.recv_counts[s] = t;
    }
    for (uint32_t s = 1; s < world; s++) recv_base[s] = recv_base[s - 1] + out.recv_counts[s - 1];
    out.run_off.assign((uint64_t)n_owned * world, 0); out.run_cnt.assign((uint64_t)n_owned * world, 0);
    for (uint32_t s = 0; s < world; s++) {
        uint64_t within = 0;                                   // source s sends its partitions for this rank in ascending order
        for (uint32_t j = 0; j < n_owned; j++) {
            const uint64_t c = pr[(uint64_t)s * P + out.owned[j]];
            if (c > 0xFFFFFFFFull) { err = "plan_exchange: more than 2^32 records of one partition on one rank"; return -1; }
            out.run_off[(uint64_t)j * world + s] = recv_base[s] + within;
            out.run_cnt[(uint64_t)j * world + s] = (uint32_t)c;
            within += c;
        }
    }
    return 0;
}

}  // namespace shk
