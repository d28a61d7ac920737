// mock_rccl.cpp — a stand-in for librccl.so.1 that moves device memory between the PROCESSES OF ONE HOST through
// POSIX shared memory.  TEST INFRASTRUCTURE ONLY (tests/test_dist.py): RCCL refuses two ranks on one GPU, and the
// test box has one GPU, so the multi-rank logic of libshk_hip.so's shard layer (csrc/shard_comm.hip: offsets of the
// pairwise exchange, grouped send/recv, all-reduce, gather by broadcasts) could otherwise first run on the driver's
// 8-GPU node.  libshk_hip.so loads this library instead of RCCL when SHK_RCCL_LIBRARY names it.
//
// Semantics kept from NCCL: every rank issues the same sequence of collectives; send/recv between a pair match in
// issue order; operations inside ncclGroupStart/ncclGroupEnd are executed together at the end (so a rank may post
// its sends to all peers before any receive).  Everything is synchronous with respect to the stream (the stream is
// drained before memory is touched).  Protocol per group: every rank copies what it contributes into its outbox in
// shared memory with a directory of (kind, peer, bytes) entries, barrier, every rank pulls what it is owed, barrier.
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
#include <string>
#include <vector>

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t OUTBOX_BYTES = (size_t)1 << 30;       // per rank (tmpfs pages exist only once touched)
constexpr int MAX_ENTRIES = 4096;

struct Entry { int kind; int peer; size_t off, bytes; };          // kind: 0 send, 1 all-reduce, 2 all-gather, 3 broadcast(root)
struct Control {
    std::atomic<int> arrived; std::atomic<int> generation;
    int n_entries[MAX_RANKS];
    Entry entries[MAX_RANKS][MAX_ENTRIES];
};
struct Op { int kind; const void *send; void *recv; size_t bytes; int peer; size_t elem; };

struct Comm {
    int rank = 0, n = 1;
    std::string name;
    Control *ctl = nullptr;
    char *data = nullptr;                                 // n outboxes
    size_t map_bytes = 0;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm *g_comm = nullptr;
thread_local hipStream_t g_stream = nullptr;

size_t dsize(ncclDataType_t t) {
    switch (t) { case ncclUint8: case ncclInt8: return 1; case ncclUint32: case ncclInt32: case ncclFloat32: return 4;
                 case ncclUint64: case ncclInt64: case ncclFloat64: return 8; default: return 0; }
}

void barrier(Comm *c) {
    const int gen = c->ctl->generation.load();
    if (c->ctl->arrived.fetch_add(1) + 1 == c->n) { c->ctl->arrived.store(0); c->ctl->generation.fetch_add(1); }
    else {
        struct timespec ts = {0, 20000};
        while (c->ctl->generation.load() == gen) nanosleep(&ts, nullptr);
    }
}

ncclResult_t run_group(Comm *c, hipStream_t st, std::vector<Op> &ops) {
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    char *mine = c->data + (size_t)c->rank * OUTBOX_BYTES;
    size_t at = 0; int ne = 0;
    for (const Op &o : ops) {
        const bool contributes = o.kind == 0 || o.kind == 1 || o.kind == 2 || (o.kind == 3 && o.peer == c->rank);
        if (!contributes) continue;
        if (at + o.bytes > OUTBOX_BYTES || ne >= MAX_ENTRIES) return ncclInternalError;
        if (o.bytes && hipMemcpy(mine + at, o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        c->ctl->entries[c->rank][ne++] = Entry{o.kind, o.peer, at, o.bytes};
        at += (o.bytes + 63) & ~(size_t)63;
    }
    c->ctl->n_entries[c->rank] = ne;
    barrier(c);
    // per source: how many of its entries of each kind have been consumed by me
    std::vector<int> next_send(c->n, 0), next_coll(c->n, 0);
    auto find = [&](int src, int kind, int want_peer, std::vector<int> &cursor) -> const Entry * {
        for (int i = cursor[src]; i < c->ctl->n_entries[src]; i++) {
            const Entry &e = c->ctl->entries[src][i];
            if (e.kind == kind && (kind != 0 || e.peer == want_peer)) { cursor[src] = i + 1; return &e; }
        }
        return nullptr;
    };
    ncclResult_t rc = ncclSuccess;
    std::vector<uint64_t> acc;
    for (const Op &o : ops) {
        if (o.kind == 4) {                                                   // recv from o.peer
            const Entry *e = find(o.peer, 0, c->rank, next_send);
            if (!e || e->bytes != o.bytes) { rc = ncclInvalidUsage; break; }
            // MOCK_RCCL_TRUNCATE_BYTES=n: a received block delivers only its first n bytes (the rest of the buffer keeps what it
            // held) — the fault a real library showed with blocks above 1 GiB; the tests use it to see the loss REPORTED
            size_t deliver = o.bytes;
            if (const char *tb = getenv("MOCK_RCCL_TRUNCATE_BYTES")) { const size_t lim = (size_t)strtoull(tb, nullptr, 10); if (lim && deliver > lim) deliver = lim; }
            if (deliver && hipMemcpy(o.recv, c->data + (size_t)o.peer * OUTBOX_BYTES + e->off, deliver, hipMemcpyHostToDevice) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
        } else if (o.kind == 1) {                                            // all-reduce (uint64 sum)
            const size_t n = o.bytes / 8;
            acc.assign(n, 0);
            for (int s = 0; s < c->n; s++) {
                const Entry *e = find(s, 1, -1, next_coll);
                if (!e || e->bytes != o.bytes) { rc = ncclInvalidUsage; break; }
                const uint64_t *p = (const uint64_t *)(c->data + (size_t)s * OUTBOX_BYTES + e->off);
                for (size_t i = 0; i < n; i++) acc[i] += p[i];
            }
            if (rc != ncclSuccess) break;
            if (o.bytes && hipMemcpy(o.recv, acc.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
        } else if (o.kind == 2) {                                            // all-gather
            for (int s = 0; s < c->n; s++) {
                const Entry *e = find(s, 2, -1, next_coll);
                if (!e || e->bytes != o.bytes) { rc = ncclInvalidUsage; break; }
                if (o.bytes && hipMemcpy((char *)o.recv + (size_t)s * o.bytes, c->data + (size_t)s * OUTBOX_BYTES + e->off, o.bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
            }
            if (rc != ncclSuccess) break;
        } else if (o.kind == 3) {                                            // broadcast from o.peer
            const Entry *e = find(o.peer, 3, -1, next_coll);
            if (!e || e->bytes != o.bytes) { rc = ncclInvalidUsage; break; }
            if (o.bytes && hipMemcpy(o.recv, c->data + (size_t)o.peer * OUTBOX_BYTES + e->off, o.bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
        }
    }
    // hipMemcpy from PAGEABLE host memory (the shared-memory outboxes) may return once the bytes are staged, before the DMA to
    // the device has finished, and the library's stream is a non-blocking one — nothing orders its next kernel behind that DMA.
    // A real collective is stream-ordered; this stand-in waits for the device here so that it is too.
    if (hipDeviceSynchronize() != hipSuccess && rc == ncclSuccess) rc = ncclUnhandledCudaError;
    barrier(c);                                                              // (outboxes may be overwritten from here on)
    return rc;
}

// (the communicator and stream of the last operation: a group in which THIS rank has nothing to send or receive — a rank
// without nodes in an exchange of the sharded assembly — must still take part in the group's two barriers, or the other
// ranks wait for it and it runs one group ahead of them from then on.  A real RCCL has no such barrier: a rank without
// operations simply does nothing.  Found by the 250-case campaign on 4 ranks, round 3.)
Comm *g_last_comm = nullptr; hipStream_t g_last_stream = nullptr;
ncclResult_t submit(Comm *c, hipStream_t st, const Op &o) {
    g_last_comm = c; g_last_stream = st;
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
    c->map_bytes = ((sizeof(Control) + 4095) & ~(size_t)4095) + (size_t)nranks * OUTBOX_BYTES;
    int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->ctl = (Control *)p;                                   // (a fresh shm object is zero-filled: counters start at 0)
    c->data = (char *)p + ((sizeof(Control) + 4095) & ~(size_t)4095);
    *out = (ncclComm_t)c;
    barrier(c);                                              // everybody has mapped it
    if (rank == 0) shm_unlink(c->name.c_str());              // the mappings keep it alive
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    if (g_last_comm == c) { g_last_comm = nullptr; g_last_stream = nullptr; }
    munmap((void *)c->ctl, c->map_bytes);
    delete c;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) { case ncclSuccess: return "no error"; case ncclInvalidUsage: return "mock rccl: invalid usage (mismatched collectives)";
                 case ncclUnhandledCudaError: return "mock rccl: HIP error"; case ncclSystemError: return "mock rccl: system error";
                 case ncclInvalidArgument: return "mock rccl: invalid argument"; default: return "mock rccl: internal error"; }
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    if (g_comm) rc = run_group(g_comm, g_stream, g_ops);
    else if (g_last_comm && g_last_comm->n > 1) rc = run_group(g_last_comm, g_last_stream, g_ops);      // an empty group: the barriers alone
    g_ops.clear(); g_comm = nullptr; g_stream = nullptr;
    return rc;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    return submit((Comm *)comm, st, Op{0, buf, nullptr, count * dsize(t), peer, dsize(t)});
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    return submit((Comm *)comm, st, Op{4, nullptr, buf, count * dsize(t), peer, dsize(t)});
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    if (t != ncclUint64 || op != ncclSum) return ncclInvalidArgument;
    return submit((Comm *)comm, st, Op{1, send, recv, count * 8, -1, 8});
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st) {
    return submit((Comm *)comm, st, Op{2, send, recv, count * dsize(t), -1, dsize(t)});
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t st) {
    return submit((Comm *)comm, st, Op{3, send, recv, count * dsize(t), root, dsize(t)});
}

}  // extern "C"
