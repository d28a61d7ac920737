// shard_comm.h — the collectives of the shard layer (DESIGN.md "Multi-GPU"), RCCL over xGMI, behind a plain
// C++ interface (no RCCL types leave shard_comm.hip).  One communicator per process and GPU; the caller of the
// C ABI hands in the ncclUniqueId bytes (include/shk.h: shk_comm_*).  The reference has no collectives at all
// (SURVEY.md §2): this replaces the rayon read-parallel driver the north_star names (row a15).
//
// librccl.so.1 is opened on first use (dlopen), so the single-GPU library has no link-time dependency on it
// and a process that already holds a copy (torch's) shares that one.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

namespace shk {

struct ShardComm;                                     // opaque: ncclComm_t + rank/world + device

static constexpr size_t SHARD_UNIQUE_ID_BYTES = 128;  // == NCCL_UNIQUE_ID_BYTES (checked in shard_comm.hip)

int comm_unique_id(uint8_t id[SHARD_UNIQUE_ID_BYTES], std::string &err);            // ncclGetUniqueId
ShardComm *comm_create(const uint8_t id[SHARD_UNIQUE_ID_BYTES], int rank, int world, std::string &err);   // ncclCommInitRank, current device
void comm_destroy(ShardComm *c);
int comm_rank(const ShardComm *c);
int comm_world(const ShardComm *c);
int comm_device(const ShardComm *c);
// a rank that cannot go on (a collective failed locally): comm_destroy will abort the communicator (ncclCommAbort)
// instead of destroying it, so peers blocked in a collective fail fast
void comm_mark_broken(ShardComm *c);
std::string comm_async_error(ShardComm *c);           // ncclCommGetAsyncError as text; "" when there is none
// A rank that leaves a collective call ALONE (a local failure nobody else knows of: out of device memory after the sizes
// were agreed, a HIP error) aborts the communicator at once: its own queued collectives are cancelled, shk_comm_free will not
// block, and peers see the failure through RCCL's asynchronous error or, at the latest, through the watchdog below.
void comm_abort_now(ShardComm *c);
bool comm_broken(const ShardComm *c);
// Host wait of the shard layer: polls `stream`, looks at RCCL's asynchronous error every few milliseconds, and gives up
// after SHK_COMM_TIMEOUT_S seconds (default 300; 0 = wait for ever) — a peer that died or left cannot hold this rank in a
// collective for ever.  On error / timeout the communicator is aborted (the stream's collectives are cancelled, so the
// buffers they use may be freed) and -5 is returned.  With c == nullptr or a one-rank communicator: a plain wait.
int comm_stream_wait(ShardComm *c, void *stream, std::string &err);

// All operate on device memory and are enqueued on `stream` (a hipStream_t); none of them waits for the stream.
// sum of n uint64 over the ranks, in place
int comm_allreduce_u64(ShardComm *c, void *d_buf, size_t n, void *stream, std::string &err);
// every rank contributes `bytes` bytes; d_recv holds world * bytes, rank-major
int comm_allgather(ShardComm *c, const void *d_send, void *d_recv, size_t bytes, void *stream, std::string &err);
// pairwise exchange: this rank sends send_bytes[d] from d_send + send_off[d] to rank d and receives recv_bytes[s]
// from rank s at d_recv + recv_off[s] — ONE grouped ncclSend/ncclRecv set, every xGMI link of the GPU busy at once
// (all byte counts and offsets are multiples of 8)
int comm_alltoallv(ShardComm *c, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes,
                   void *d_recv, const uint64_t *recv_off, const uint64_t *recv_bytes, void *stream, std::string &err, uint32_t elem = 8 /* 8: u64 elements, 4: u32 */);
// rank s contributes bytes[s] bytes that end up at d_recv + off[s] on every rank (d_send: this rank's part)
int comm_allgatherv(ShardComm *c, const void *d_send, void *d_recv, const uint64_t *off, const uint64_t *bytes,
                    void *stream, std::string &err);

// small host-side collectives (staged through a device buffer of the pool; they wait for `stream`)
int comm_allreduce_host_u64(ShardComm *c, uint64_t *host_inout, size_t n, void *stream, std::string &err);
int comm_allgather_host_u64(ShardComm *c, const uint64_t *host_in, size_t n, uint64_t *host_out /* [world][n] */, void *stream,
                            std::string &err);

// ---- pure host logic of the record exchange (tested on CPU against sparrowhawk_amd/dist.py: plan_exchange) ----
// part_records_all: [world][P] records rank s holds for partition p; partition p belongs to rank p % world.
struct ExchangePlan {
    std::vector<uint32_t> owned;          // partitions this rank counts, ascending
    std::vector<uint64_t> base;           // [P] record offset of partition p in the send buffer (destination-major)
    std::vector<uint64_t> send_counts;    // [world] records sent to each destination
    std::vector<uint64_t> recv_counts;    // [world] records received from each source
    std::vector<uint64_t> run_off;        // [n_owned][world] record offset of source s's run of owned partition j in the receive buffer
    std::vector<uint32_t> run_cnt;        // [n_owned][world]
};
int plan_exchange(const uint64_t *part_records_all, uint32_t world, uint32_t P, uint32_t rank, ExchangePlan &out,
                  std::string &err);
// power of two in [64, 16384], >= world, ~per_part k-mer instances per partition
uint32_t choose_partitions(uint64_t total_instances_ub, uint32_t world, uint64_t per_part);

}  // namespace shk
