// pipeline.h — host-visible interface of the device pipeline (implemented in pipeline.hip).
// One object per handle; templated on the key width W inside, type-erased here.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <memory>

namespace shk {

struct ShardComm;             // shard_comm.h

// The text of the contigs may still be on its way from the device when the writer starts (pipeline.hip: collapse): it
// arrives in slabs, roughly front to back, and the writer copies a range as soon as its slabs are there — the download of a
// 5 Mbp contig and its three copies into the JSON (FASTA, GFA1, GFA2) run side by side.
class TextArrival {
public:
    virtual ~TextArrival() {}
    virtual const char *base() const = 0;             // first byte of the arriving text (pinned host memory)
    virtual size_t total() const = 0;
    virtual void wait_range(size_t begin, size_t end) = 0;     // returns when [begin, end) has arrived (any thread)
    void wait_all() { wait_range(0, total()); }
    virtual int finish(std::string &err) = 0;         // waits for all of it; non-zero if the download failed
};

struct RawContig {            // one unitig as spelled by the device, arbitrary strand
    // The sequence either lives in the pipeline's pinned download buffer (ext: valid until the next
    // collapse() or the pipeline's destruction — a 5 Mbp contig is not copied again) or in `own`.
    const char *ext = nullptr; size_t ext_n = 0;
    std::string own;
    uint64_t kc = 0;          // sum of k-mer counts over its nodes
    // with a TextArrival: copies of the first and the last min(size, ends_n) bases, there before the text itself
    const char *head = nullptr, *tail = nullptr; uint32_t ends_n = 0;
    const char *data() const { return ext ? ext : own.data(); }
    size_t size() const { return ext ? ext_n : own.size(); }
};

struct StageTimes {           // milliseconds (HIP events on the pipeline's stream / host clock)
    std::map<std::string, double> ms;
    void add(const std::string &k, double v) { ms[k] += v; }
};

// one piece of a batch of packed segments, already in device memory (offsets relative to its own d_bases)
struct DevPiece { const uint32_t *d_bases, *d_seg_off; uint64_t n_seg, n_bases; };

class IPipeline {
public:
    virtual ~IPipeline() {}
    // counting (SPEC S4): may be called once per batch of packed segments
    virtual int count_batch(const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg,
                            uint64_t n_bases, std::string &err) = 0;
    // the same batch with the packed reads still in host memory (h_*): uploaded into d_* piece by piece, overlapped with pass 1
    virtual int count_batch_host(uint32_t *d_bases, uint32_t *d_seg_off, const uint32_t *h_bases, const uint32_t *h_seg_off,
                                 uint64_t n_seg, uint64_t n_bases, std::string &err) = 0;
    // ONE batch that lies in several pieces (a FASTQ text parsed piece by piece while the next piece was uploaded): pass 1
    // runs over the pieces one after the other into the same slices
    virtual int count_batch_pieces(const DevPiece *pieces, size_t n_pieces, std::string &err) = 0;
    // more batches will follow the first one (chunked / streamed input of unknown size): partition for the worst case
    virtual void expect_more_batches() = 0;
    // do_bloom (docs/src/assembly.md:18): partitions that go through the k-mer-level repartition pass a Bloom
    // pre-filter first, so their singletons are never stored; counts may then be one too high, never too low
    virtual void set_bloom(bool on) = 0;
    // verbose handle (AssemblyHelper.new's second argument): per-stage HIP-event timers and the stage data the inspection
    // calls serve (initial adjacency) are kept; a quiet handle times the two counting passes only
    virtual void set_verbose(bool on) = 0;
    // the next count_batch / count_batch_host hands over the ONLY batch of this handle and its packed reads stay where they are
    // until histogram() has returned: pass 2 may then be launched behind pass 1 without a host round trip in between
    virtual void single_batch_resident(bool on) = 0;
    // spectrum histogram (SPEC S5).  Rows with count <= emit_threshold will never be asked for
    // by filter(): the counting pass may drop them as soon as they are histogrammed.
    virtual int histogram(uint64_t histo[500], uint32_t emit_threshold, std::string &err) = 0;
    // keep k-mers with count > threshold (SPEC S7); returns number kept via n_solid()
    virtual int filter(uint32_t threshold, std::string &err) = 0;
    virtual uint64_t total_instances() const = 0;
    virtual uint64_t n_distinct() const = 0;
    virtual uint64_t n_solid() const = 0;
    virtual int get_distinct(uint64_t *keys, uint32_t *counts, uint64_t cap, std::string &err) = 0;
    virtual int get_solid(uint64_t *keys, uint32_t *counts, uint64_t cap, std::string &err) = 0;
    // assembly (SPEC S8-S10)
    virtual int build_graph(std::string &err) = 0;
    virtual int correct(bool tips, bool bubbles, std::string &err) = 0;
    // json (optional): a FRAGMENTED assembly (>= SHK_DEVICE_WRITER_MIN contigs, default 20 000) is turned into the
    // get_assembly() JSON on the device (writer_gpu.h: order, links, FASTA / DOT / GFA1 / GFA2 text) and `out` stays empty;
    // *json then points at the NUL-terminated text in pinned host memory owned by the pipeline, *n_contigs says how many
    // arrival (optional): when the caller can start on text that is still arriving, *arrival is set (owned by the pipeline,
    // valid until the next collapse) and the contigs carry their first / last bases; otherwise the text is complete on return
    virtual int collapse(std::vector<RawContig> &out, std::string &err, const char **json = nullptr, size_t *json_len = nullptr,
                         uint64_t *n_contigs = nullptr, TextArrival **arrival = nullptr) = 0;
    virtual int get_adjacency(uint8_t *adj_initial, uint8_t *adj_final, uint8_t *alive,
                              uint64_t cap, std::string &err) = 0;
    // shard layer (one process per GPU): partition -> pack -> [all-to-all] -> count -> rows ->
    // [all-gather] -> set_solid -> build_graph/correct/collapse as usual
    virtual int shard_partition(const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg, uint64_t n_bases,
                                uint32_t n_partitions, std::vector<uint64_t> &part_records, std::string &err) = 0;
    virtual uint32_t rec_words() const = 0;
    virtual int shard_pack(void *d_send, const uint64_t *base_records, uint32_t n_partitions, std::string &err) = 0;
    // records deduplicated by their SOURCE before the exchange (count_part.h: k_dedupe_partitions): part_records in = raw
    // counts of shard_partition, out = distinct records per partition; shard_pack_dedup then fills the send buffers
    // (records and their u32 weights, same element offsets); shard_drop_dedup forgets them (the raw records travel instead: shard_pack)
    virtual int shard_dedupe(std::vector<uint64_t> &part_records, std::string &err) = 0;
    virtual int shard_pack_dedup(void *d_send, void *d_send_w, const uint64_t *base_records, uint32_t n_partitions, std::string &err) = 0;
    virtual void shard_drop_dedup() = 0;
    virtual int shard_count(const void *d_recv, const void *d_recv_w /* nullable: weights */, const uint64_t *run_off, const uint32_t *run_cnt,
                            uint32_t n_owned, uint32_t n_sources, uint32_t emit_threshold, uint64_t histo[500], std::string &err) = 0;
    virtual int shard_rows(uint32_t threshold, const void **keys_soa, const void **cnt, uint64_t *n, std::string &err) = 0;
    virtual int shard_set_solid(const void *const *keys_soa, const void *cnt, uint64_t n, const uint64_t histo[500],
                                uint64_t total_instances, std::string &err) = 0;
    // sharded ASSEMBLY (shard_graph.h): instead of shard_set_solid — which installs the gathered solid set on every rank —
    // the rank keeps the rows of its own counting partitions (after shard_rows) and assembles with the graph spread over
    // the ranks.  rows_per_rank: [world] solid rows of every rank; n_count_partitions: the P of the counting pass (partition
    // p belongs to rank p % world).  shard_assemble is collective over `comm`.
    virtual int shard_keep_local(uint32_t world, uint32_t rank, uint32_t n_count_partitions, const uint64_t *rows_per_rank,
                                 const uint64_t histo[500], uint64_t total_instances, std::string &err) = 0;
    virtual bool sharded_graph() const = 0;
    virtual uint64_t n_solid_global() const = 0;
    virtual int shard_assemble(ShardComm *comm, bool tips, bool bubbles, std::vector<RawContig> &out, std::string &err) = 0;
    virtual StageTimes &times() = 0;
    // waits until nothing is in flight on the pipeline's streams (through the shard layer's watchdog while a collective call
    // of several ranks runs); never fails — an error stays on the stream for the next call to find
    virtual void drain() = 0;
    virtual void *stream() = 0;
    virtual int device() const = 0;                  // the HIP device this pipeline's stream and buffers live on
};

// Device buffers that go out of scope on the calling thread while this object lives are kept until it dies; then the
// pipeline's stream is drained and they go back to the pool (see DevBuf in pipeline.hip).  One per entry point of the C ABI.
class DeferScope {
public:
    explicit DeferScope(IPipeline *pipe);
    ~DeferScope();
    DeferScope(const DeferScope &) = delete;
    DeferScope &operator=(const DeferScope &) = delete;
private:
    IPipeline *pipe_; void *list_; void *prev_;
};

// returns nullptr (and err) if no device / bad k
IPipeline *make_pipeline(int k, std::string &err);
int device_count();
int current_device();          // the calling thread's current HIP device
int set_device(int dev);       // makes `dev` current for the calling thread, returns the previous one

// host-side helpers implemented with the same kmer.h arithmetic as the kernels
int host_canonical(const char *seq, uint32_t k, uint64_t *out_words, int *orient);
uint64_t host_nthash(const char *seq, uint32_t k);

// upload helper used by the host-buffer entry points
int device_upload(const void *host, size_t bytes, void **dptr, std::string &err);
void device_free(void *dptr);
void device_pool_trim();
// the process-wide device block cache (pipeline.hip): bytes is rounded up to the block actually handed out
void *device_pool_alloc(size_t &bytes);
// best rate (GB/s) of `iters` pure streaming reads of a `bytes` buffer on the current device
int stream_read_gbs(size_t bytes, int iters, double *gbs, std::string &err);
void device_pool_release(void *p, size_t bytes);
// peak device memory per handle (Assembler.ts:69-71,137 reports peak wasm memory with every assembly): an accounting context
// is made per handle, carried by the thread that serves it (mem_acct_set returns the previous one), and every block the pool
// hands out meanwhile is charged to it until it goes back
std::shared_ptr<void> mem_acct_new();
std::shared_ptr<void> mem_acct_set(std::shared_ptr<void> a);
uint64_t mem_acct_peak(const std::shared_ptr<void> &a);
uint64_t mem_acct_current(const std::shared_ptr<void> &a);
// the counter alone, on the host (tests): applies signed byte deltas, returns the high-water mark and what is left
void mem_acct_replay(const int64_t *deltas, size_t n, uint64_t *peak, uint64_t *current);
int device_stream_sync(void *stream, std::string &err);      // waits for a hipStream_t
int device_copy_h2d_async(void *dst, const void *src, size_t bytes, void *stream, std::string &err);
int device_download(void *host, const void *dptr, size_t bytes, std::string &err);      // blocking device -> host copy

}  // namespace shk
