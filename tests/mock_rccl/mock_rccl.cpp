// mock_rccl.cpp — a stand-in for librccl.so.1 that moves device memory between the PROCESSES OF ONE HOST through
// POSIX shared memory.  TEST INFRASTRUCTURE ONLY (tests/test_dist.py): RCCL refuses two ranks on one GPU, and the
// test box has one GPU, so the multi-rank logic of libshk_hip.so's shard layer (csrc/shard_comm.hip: offsets of the
// pairwise exchange, grouped send/recv, all-reduce, gather by broadcasts) could otherwise first run on the driver's
// 8-GPU node.  libshk_hip.so loads this library instead of RCCL when SHK_RCCL_LIBRARY names it.
//
// Round 4: rewritten as a POINT-TO-POINT transport with NCCL's (lack of) guarantees and nothing more — the first version
// ran every group between two global barriers, which real RCCL does not have, and that hid two bugs of round 3:
//   * one byte FIFO per ordered pair of ranks (a ring in shared memory): messages between a pair arrive in issue order,
//     NOTHING orders the traffic of different pairs;
//   * sends are EAGER: a send completes as soon as its bytes fit the ring, whether or not the peer has posted the receive
//     (a rank may be several collectives ahead of a slow peer); a full ring makes the sender wait for the receiver;
//   * no barrier anywhere after communicator creation: a rank whose group is empty does nothing at all;
//   * collectives are built from the same sends and receives (all-reduce / all-gather: everybody to everybody; broadcast:
//     root to everybody), every message tagged with its kind and size — a rank that issues a different sequence of
//     collectives than its peers is told so (ncclInvalidUsage) instead of exchanging garbage;
//   * the operations of a group make progress together, in no fixed order, in pieces.
// MOCK_RCCL_JITTER=1 makes it HOSTILE (tools/fuzz_sharded.sh): random pauses before and between pieces, a random visiting
// order of the pending operations in every round, random piece sizes — per-peer delivery delays, completion out of order
// across peers, sends long done when the receive is posted.  MOCK_RCCL_SEED seeds it (default: from the rank).
// Everything is synchronous with respect to the stream (the stream is drained before memory is touched, the device is
// drained at the end of a group): what the stand-in cannot imitate is RCCL's asynchrony on the device.
// MOCK_RCCL_TIMEOUT_S (default 120): a group that makes no progress for that long fails (a peer left or died).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <string>
#include <vector>

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t RING_BYTES = (size_t)64 << 20;        // per ordered pair (tmpfs pages exist only once touched)
constexpr size_t PIECE_MAX = (size_t)8 << 20;

struct Ring { std::atomic<uint64_t> head, tail; char pad[48]; };     // bytes written / consumed (monotonic); data follows in the pair's slot
struct Control {
    std::atomic<int> arrived;
    std::atomic<int> aborted;                               // a rank called ncclCommAbort: peers fail instead of waiting for it
    Ring ring[MAX_RANKS][MAX_RANKS];                        // [src][dst]
};
struct MsgHeader { uint64_t bytes; uint32_t kind, magic; };  // kind: 0 p2p, 1 all-reduce, 2 all-gather, 3 broadcast

struct Comm {
    int rank = 0, n = 1;
    std::string name;
    Control *ctl = nullptr;
    char *data = nullptr;                                   // n x n ring slots
    size_t map_bytes = 0;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    bool jitter = false;
    char *slot(int s, int d) const { return data + ((size_t)s * n + d) * RING_BYTES; }
    uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
};

// what the caller asked for (queued inside a group)
struct Op { int kind; const void *send; void *recv; size_t bytes; int peer; };     // kind as MsgHeader, 4 = recv (p2p)

// one direction of one message, in flight
struct Xfer {
    bool is_send; int peer; uint32_t kind; size_t bytes;
    const char *dsrc = nullptr;     // send: device source (or host source when `hsrc`)
    const char *hsrc = nullptr;
    char *ddst = nullptr;           // recv: device destination (nullptr: into `hdst`)
    char *hdst = nullptr;
    size_t deliver = 0;             // recv: bytes that really reach ddst (MOCK_RCCL_TRUNCATE_BYTES)
    size_t done = 0; bool header_done = false;
    bool finished() const { return header_done && done == bytes; }
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm *g_comm = nullptr;
thread_local hipStream_t g_stream = nullptr;

size_t dsize(ncclDataType_t t) {
    switch (t) { case ncclUint8: case ncclInt8: return 1; case ncclUint32: case ncclInt32: case ncclFloat32: return 4;
                 case ncclUint64: case ncclInt64: case ncclFloat64: return 8; default: return 0; }
}
double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
void nap(long ns) { struct timespec ts = {0, ns}; nanosleep(&ts, nullptr); }

// ring I/O with wrap-around; `dev`: the other side of the copy is device memory
bool ring_put(Comm *c, int dst, const char *src, size_t n, bool dev) {
    Ring &r = c->ctl->ring[c->rank][dst];
    char *base = c->slot(c->rank, dst);
    const uint64_t h = r.head.load(std::memory_order_relaxed);
    size_t at = (size_t)(h % RING_BYTES), first = std::min(n, RING_BYTES - at);
    if (dev) {
        if (hipMemcpy(base + at, src, first, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (n > first && hipMemcpy(base, src + first, n - first, hipMemcpyDeviceToHost) != hipSuccess) return false;
    } else { memcpy(base + at, src, first); if (n > first) memcpy(base, src + first, n - first); }
    r.head.store(h + n, std::memory_order_release);
    return true;
}
bool ring_get(Comm *c, int src, char *dst, size_t n, size_t keep, bool dev) {       // consumes n bytes, delivers the first `keep` of them
    Ring &r = c->ctl->ring[src][c->rank];
    const char *base = c->slot(src, c->rank);
    const uint64_t t = r.tail.load(std::memory_order_relaxed);
    size_t at = (size_t)(t % RING_BYTES), first = std::min(n, RING_BYTES - at);
    const size_t k1 = std::min(keep, first), k2 = keep > first ? keep - first : 0;
    if (dev) {
        if (k1 && hipMemcpy(dst, base + at, k1, hipMemcpyHostToDevice) != hipSuccess) return false;
        if (k2 && hipMemcpy(dst + first, base, k2, hipMemcpyHostToDevice) != hipSuccess) return false;
    } else { if (k1) memcpy(dst, base + at, k1); if (k2) memcpy(dst + first, base, k2); }
    r.tail.store(t + n, std::memory_order_release);
    return true;
}

// a random part of the n bytes that could move now, a multiple of 16 (device addresses stay 16-byte aligned: copies of odd
// sizes from odd device addresses are not what the library ever asks a transport for — the first jittered campaign ended in a
// GPU memory fault at a page boundary that a run with the stages serialised did not show, and odd pieces were the one thing
// only this stand-in did)
size_t jitter_piece(Comm *c, size_t n) {
    size_t m = 1 + (size_t)(c->next() % n);
    m &= ~(size_t)15;
    return m ? m : std::min<size_t>(n, 16);
}

// one step of one transfer; returns 1 progress, 0 none, <0 error (ncclResult as negative)
int progress(Comm *c, Xfer &x) {
    if (x.is_send) {
        Ring &r = c->ctl->ring[c->rank][x.peer];
        size_t space = RING_BYTES - (size_t)(r.head.load(std::memory_order_relaxed) - r.tail.load(std::memory_order_acquire));
        if (!x.header_done) {
            if (space < sizeof(MsgHeader)) return 0;
            MsgHeader h{(uint64_t)x.bytes, x.kind, 0x534B4D4Bu};
            if (!ring_put(c, x.peer, (const char *)&h, sizeof h, false)) return -(int)ncclUnhandledCudaError;
            x.header_done = true;
            return 1;
        }
        if (x.done == x.bytes) return 0;
        size_t n = std::min({space, x.bytes - x.done, PIECE_MAX});
        if (c->jitter && n > 64) n = jitter_piece(c, n);
        if (!n) return 0;
        const bool dev = x.hsrc == nullptr;
        if (!ring_put(c, x.peer, (dev ? x.dsrc : x.hsrc) + x.done, n, dev)) return -(int)ncclUnhandledCudaError;
        x.done += n;
        return 1;
    }
    Ring &r = c->ctl->ring[x.peer][c->rank];
    size_t avail = (size_t)(r.head.load(std::memory_order_acquire) - r.tail.load(std::memory_order_relaxed));
    if (!x.header_done) {
        if (avail < sizeof(MsgHeader)) return 0;
        MsgHeader h;
        if (!ring_get(c, x.peer, (char *)&h, sizeof h, sizeof h, false)) return -(int)ncclUnhandledCudaError;
        if (h.magic != 0x534B4D4Bu || h.kind != x.kind || h.bytes != (uint64_t)x.bytes) {
            fprintf(stderr, "[mock rccl rank %d] message from rank %d: kind %u, %llu bytes — this rank expects kind %u, %zu bytes (the ranks' collectives differ)\n",
                    c->rank, x.peer, h.kind, (unsigned long long)h.bytes, x.kind, x.bytes);
            return -(int)ncclInvalidUsage;
        }
        x.header_done = true;
        return 1;
    }
    if (x.done == x.bytes) return 0;
    size_t n = std::min({avail, x.bytes - x.done, PIECE_MAX});
    if (c->jitter && n > 64) n = jitter_piece(c, n);
    if (!n) return 0;
    const size_t keep = x.done >= x.deliver ? 0 : std::min(n, x.deliver - x.done);
    const bool dev = x.ddst != nullptr;
    if (!ring_get(c, x.peer, (dev ? x.ddst : x.hdst) + x.done, n, keep, dev)) return -(int)ncclUnhandledCudaError;
    x.done += n;
    return 1;
}

ncclResult_t run_group(Comm *c, hipStream_t st, std::vector<Op> &ops) {
    if (ops.empty()) return ncclSuccess;                                      // (no barrier: a rank with nothing to do does nothing)
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    size_t trunc = 0;
    // MOCK_RCCL_TRUNCATE_BYTES=n: a received p2p block delivers only its first n bytes (the rest of the buffer keeps what it
    // held) — the fault a real library showed with blocks above 1 GiB; the tests use it to see the loss REPORTED
    if (const char *tb = getenv("MOCK_RCCL_TRUNCATE_BYTES")) trunc = (size_t)strtoull(tb, nullptr, 10);
    std::vector<Xfer> xs;
    struct Reduce { size_t op; std::vector<std::vector<uint64_t>> part; std::vector<uint64_t> mine; };
    std::vector<Reduce> reds;
    reds.reserve(ops.size());                                                // (pointers into it are kept below)
    for (size_t i = 0; i < ops.size(); i++) {
        const Op &o = ops[i];
        if (o.kind == 0) { Xfer x{true, o.peer, 0, o.bytes}; x.dsrc = (const char *)o.send; xs.push_back(x); }
        else if (o.kind == 4) { Xfer x{false, o.peer, 0, o.bytes}; x.ddst = (char *)o.recv; x.deliver = (trunc && o.bytes > trunc) ? trunc : o.bytes; xs.push_back(x); }
        else if (o.kind == 1) {                                               // all-reduce: my vector to everybody, everybody's to me
            reds.push_back(Reduce{i, {}, {}});
            Reduce &r = reds.back();
            r.mine.resize(o.bytes / 8); r.part.assign(c->n, {});
            if (o.bytes && hipMemcpy(r.mine.data(), o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            for (int p = 0; p < c->n; p++) {
                if (p == c->rank) continue;
                r.part[p].resize(o.bytes / 8);
                Xfer s{true, p, 1, o.bytes}; s.hsrc = (const char *)r.mine.data(); xs.push_back(s);
                Xfer g{false, p, 1, o.bytes}; g.hdst = (char *)r.part[p].data(); g.deliver = o.bytes; xs.push_back(g);
            }
        } else if (o.kind == 2) {                                             // all-gather
            for (int p = 0; p < c->n; p++) {
                if (p == c->rank) continue;
                Xfer s{true, p, 2, o.bytes}; s.dsrc = (const char *)o.send; xs.push_back(s);
                Xfer g{false, p, 2, o.bytes}; g.ddst = (char *)o.recv + (size_t)p * o.bytes; g.deliver = o.bytes; xs.push_back(g);
            }
        } else if (o.kind == 3) {                                             // broadcast from o.peer
            if (o.peer == c->rank) { for (int p = 0; p < c->n; p++) if (p != c->rank) { Xfer s{true, p, 3, o.bytes}; s.dsrc = (const char *)o.send; xs.push_back(s); } }
            else { Xfer g{false, o.peer, 3, o.bytes}; g.ddst = (char *)o.recv; g.deliver = o.bytes; xs.push_back(g); }
        }
    }
    // Per pair and direction the transfers must run in issue order (one byte FIFO); across pairs and directions anything goes.
    // `front[q]`: the first unfinished transfer of queue q = (direction, peer).
    const int NQ = 2 * MAX_RANKS;
    std::vector<std::vector<size_t>> queue(NQ);
    for (size_t i = 0; i < xs.size(); i++) queue[(xs[i].is_send ? 0 : MAX_RANKS) + xs[i].peer].push_back(i);
    std::vector<size_t> front(NQ, 0);
    static const double limit = [] { const char *e = getenv("MOCK_RCCL_TIMEOUT_S"); return e && *e ? strtod(e, nullptr) : 120.0; }();
    double last_progress = now_s();
    std::vector<int> order(NQ);
    for (int q = 0; q < NQ; q++) order[q] = q;
    if (c->jitter) nap((long)(c->next() % 300000));                           // up to 0.3 ms before this rank even starts
    for (;;) {
        bool all_done = true, moved = false;
        if (c->jitter) for (int q = NQ - 1; q > 0; q--) std::swap(order[q], order[(size_t)(c->next() % (uint64_t)(q + 1))]);
        for (int qi = 0; qi < NQ; qi++) {
            const int q = order[qi];
            if (front[q] >= queue[q].size()) continue;
            all_done = false;
            if (c->jitter && (c->next() & 3u) == 0) continue;                 // this peer's turn is skipped: delivery delays per peer
            Xfer &x = xs[queue[q][front[q]]];
            const int r = progress(c, x);
            if (r < 0) return (ncclResult_t)(-r);
            if (r > 0) moved = true;
            if (x.finished()) front[q]++;
            if (c->jitter && (c->next() & 15u) == 0) nap((long)(c->next() % 100000));
        }
        if (all_done) break;
        if (moved) last_progress = now_s();
        else {
            if (c->ctl->aborted.load()) return ncclSystemError;
            if (limit > 0 && now_s() - last_progress > limit) {
                fprintf(stderr, "[mock rccl rank %d] no progress for %.0f s: a peer left the collective\n", c->rank, limit);
                return ncclSystemError;
            }
            nap(2000);
        }
    }
    for (Reduce &r : reds) {
        const Op &o = ops[r.op];
        for (int p = 0; p < c->n; p++) if (p != c->rank) for (size_t i = 0; i < r.mine.size(); i++) r.mine[i] += r.part[p][i];
        if (o.bytes && hipMemcpy(o.recv, r.mine.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    for (const Op &o : ops)                                                   // all-gather / broadcast: the rank's own part
        if (o.kind == 2) { if (o.bytes && hipMemcpy((char *)o.recv + (size_t)c->rank * o.bytes, o.send, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError; }
        else if (o.kind == 3 && o.peer == c->rank && o.recv != o.send) { if (o.bytes && hipMemcpy(o.recv, o.send, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError; }
    // hipMemcpy from PAGEABLE host memory (the shared-memory rings) may return once the bytes are staged, before the DMA to
    // the device has finished, and the library's stream is a non-blocking one — nothing orders its next kernel behind that DMA.
    // A real collective is stream-ordered; this stand-in waits for the device here so that it is too.
    if (hipDeviceSynchronize() != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t submit(Comm *c, hipStream_t st, const Op &o) {
    if (!c) return ncclInvalidArgument;
    if (g_depth > 0) {
        if (g_comm && g_comm != c) return ncclInvalidUsage;
        g_comm = c; g_stream = st; g_ops.push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> one{o};
    return run_group(c, st, one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id->internal, 0, sizeof id->internal);
    struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/shk_mock_rccl_%d_%ld_%ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank; c->n = nranks; c->name = id.internal;
    const size_t ctl_bytes = (sizeof(Control) + 4095) & ~(size_t)4095;
    c->map_bytes = ctl_bytes + (size_t)nranks * nranks * RING_BYTES;
    int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->ctl = (Control *)p;                                   // (a fresh shm object is zero-filled: counters start at 0)
    c->data = (char *)p + ctl_bytes;
    const char *j = getenv("MOCK_RCCL_JITTER");
    c->jitter = j && *j && *j != '0';
    const char *sd = getenv("MOCK_RCCL_SEED");
    c->rng = (sd && *sd ? strtoull(sd, nullptr, 10) : 0x1234567ull) * 0x9E3779B97F4A7C15ull + (uint64_t)(rank + 1) * 0xD1B54A32D192ED03ull;
    if (!c->rng) c->rng = 1;
    *out = (ncclComm_t)c;
    // communicator creation is collective in NCCL too: everybody has mapped the object before anybody unlinks it
    c->ctl->arrived.fetch_add(1);
    const double t0 = now_s();
    while (c->ctl->arrived.load() < nranks) { if (now_s() - t0 > 120) return ncclSystemError; nap(20000); }
    if (rank == 0) shm_unlink(c->name.c_str());              // the mappings keep it alive
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    munmap((void *)c->ctl, c->map_bytes);
    delete c;
    return ncclSuccess;
}
// a rank that gives up: its peers' pending groups fail instead of waiting for ever (what ncclCommAbort + the peers' own
// error handling amount to on a real node)
ncclResult_t ncclCommAbort(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    c->ctl->aborted.store(1);
    return ncclCommDestroy(comm);
}
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *err) {
    Comm *c = (Comm *)comm;
    if (err) *err = (c && c->ctl->aborted.load()) ? ncclSystemError : ncclSuccess;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) { case ncclSuccess: return "no error"; case ncclInvalidUsage: return "mock rccl: invalid usage (mismatched collectives)";
                 case ncclUnhandledCudaError: return "mock rccl: HIP error"; case ncclSystemError: return "mock rccl: system error (a peer left, or aborted)";
                 case ncclInvalidArgument: return "mock rccl: invalid argument"; default: return "mock rccl: internal error"; }
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    if (g_comm) rc = run_group(g_comm, g_stream, g_ops);     // (an empty group: nothing happens, as with RCCL)
    g_ops.clear(); g_comm = nullptr; g_stream = nullptr;
    return rc;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    return submit((Comm *)comm, st, Op{0, buf, nullptr, count * dsize(t), peer});
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    return submit((Comm *)comm, st, Op{4, nullptr, buf, count * dsize(t), peer});
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    if (t != ncclUint64 || op != ncclSum) return ncclInvalidArgument;
    return submit((Comm *)comm, st, Op{1, send, recv, count * 8, -1});
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st) {
    return submit((Comm *)comm, st, Op{2, send, recv, count * dsize(t), -1});
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t st) {
    return submit((Comm *)comm, st, Op{3, send, recv, count * dsize(t), root});
}

}  // extern "C"
