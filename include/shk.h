/*
 * shk.h — C ABI of libshk_hip.so: the MI355X (gfx950) implementation of the sparrowhawk-asm
 * assembly path (k-mer count -> filter -> de Bruijn graph -> correct -> collapse -> contigs).
 *
 * This is the drop-in boundary: exactly what a Rust `AssemblyHelper` (the wasm-bindgen struct
 * the reference's worker drives, /root/reference/www/src/workers/Assembler.ts:15-39) binds over
 * FFI.  See INTEGRATION.md for the Rust-side `extern "C"` block.  Plain pointers and sizes only;
 * no C++ or torch types cross this boundary.
 *
 * Threading: a handle is not thread-safe; one handle drives one HIP device (the current device
 * at shk_new time).  Every function returns SHK_OK (0) or a negative error code and never
 * aborts; the message is available from shk_last_error().  Strings returned by the library are
 * owned by the handle and stay valid until the next call on it or shk_free().
 *
 * There is no CPU fallback: without a usable HIP device shk_new() fails with SHK_E_DEVICE.
 */
#ifndef SHK_H
#define SHK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHK_OK 0
#define SHK_E_PARAM  (-1)  /* even k, k out of range, bad argument                         */
#define SHK_E_STATE  (-2)  /* call out of order (Assembler.ts:92-111,121-139 call sequence) */
#define SHK_E_PARSE  (-3)  /* malformed FASTQ / gzip                                        */
#define SHK_E_OOM    (-4)  /* host or device memory                                         */
#define SHK_E_DEVICE (-5)  /* no HIP device / HIP runtime error                             */
#define SHK_E_INTERNAL (-6)

#define SHK_HISTO_BINS 500 /* KmerHistogram.vue:45 */
#define SHK_K_MIN 15
#define SHK_K_MAX 255      /* compiled key widths: 1..8 64-bit words (SPEC S3; docs/src/assembly.md:13 "up until 255", the UI offers 21..89) */

typedef struct shk_handle shk_handle;

/* Progress hook — replaces the crate's post_state(&str) -> postMessage({assemblyState})
 * (/root/reference/AGENTS.md:236-250; strings: AssemblyPage.vue:458-609).  Fires on the
 * calling thread. */
typedef void (*shk_progress_cb)(const char *state, void *user);

/* AssemblyHelper::new(k, verbose, min_count, min_qual, chunk_size, do_bloom, do_fit,
 *                     no_bubble_collapse, no_dead_end_removal)        Assembler.ts:15-29,94-99
 * Same nine parameters, same order.  Returns NULL on failure; shk_new_error() tells why.
 * verbose: results never depend on it.  A verbose handle also records per-stage HIP-event timers (shk_get_timings) and keeps
 * the initial adjacency bytes for shk_get_adjacency; a quiet one times the two counting passes only — an event between two
 * kernels costs the GPU ~10 us — and drops what only the inspection reads (SHK_STAGE_TIMERS=1 / SHK_KEEP_STAGES=1 turn
 * either on for quiet handles). */
shk_handle *shk_new(uint32_t k, int verbose, uint32_t min_count, uint32_t min_qual,
                    uint64_t chunk_size, int do_bloom, int do_fit,
                    int no_bubble_collapse, int no_dead_end_removal);
int shk_new_error(void);                 /* error code of the last failed shk_new on this thread */
const char *shk_new_error_message(void);

void shk_free(shk_handle *h);            /* Assembler.ts:141-143 (resetAll drops the handle) */
const char *shk_last_error(shk_handle *h);
void shk_set_progress_cb(shk_handle *h, shk_progress_cb cb, void *user);

/* AssemblyHelper::preprocess(file1, file2|null)                        Assembler.ts:35,100
 * fq1/fq2: whole FASTQ files in memory, plain or gzip (fastx_wasm.rs:9,53-70); fq2 may be NULL.
 * The buffers are not retained. */
int shk_preprocess(shk_handle *h, const uint8_t *fq1, size_t n1, const uint8_t *fq2, size_t n2);

/* Streaming form of preprocess for inputs that do not fit host memory at once (modelled on the
 * reference's chunked bridge, rust/deacon-bridge/src/lib.rs:112,158).  Each chunk must hold
 * whole FASTQ records (plain text); shk_finish_reads() completes preprocessing. */
int shk_push_reads(shk_handle *h, const uint8_t *fastq_chunk, size_t n);
int shk_finish_reads(shk_handle *h);

/* Device-resident form of preprocess: reads already parsed, quality-masked, cut into valid
 * segments (SPEC S2) and 2-bit packed in HBM (layout: DESIGN.md "Data layout").
 *   d_bases   : uint32 words, base i of the stream in bits [2*(i%16), 2*(i%16)+1] of word i/16;
 *               at least ceil(n_bases/16)+1 words allocated
 *   d_seg_off : uint32[n_seg+1], base offset of each segment in the stream (ascending,
 *               d_seg_off[n_seg] == n_bases); every segment is >= k bases and of any length (a segment of more
 *               than 2048 k-mers is walked in pieces by the counting pass; before round 4 segments beyond 163840
 *               bases were refused)
 * Both are HIP device pointers on the handle's device; they are read, not retained.
 * n_reads is only used for progress/statistics. */
int shk_preprocess_packed_device(shk_handle *h, const void *d_bases, const void *d_seg_off,
                                 uint64_t n_seg, uint64_t n_bases, uint64_t n_reads);

/* The same reads, packed, in HOST memory (pinned memory makes the upload run at the link's rate): uploaded on the
 * handle's stream, then counted as shk_preprocess_packed_device does.  This is where SURVEY.md 8(d) starts the
 * clock of the throughput metric ("packed reads resident in host pinned memory"). */
int shk_preprocess_packed_host(shk_handle *h, const uint32_t *bases, const uint32_t *seg_off,
                               uint64_t n_seg, uint64_t n_bases, uint64_t n_reads);

/* AssemblyHelper::get_preprocessing_info() -> String                   Assembler.ts:36,110
 * JSON {"nkmers":int,"histo":[500 ints],"used_min_count":int}          Assembler.ts:1-5 */
const char *shk_get_preprocessing_info(shk_handle *h);

/* AssemblyHelper::assemble()                                           Assembler.ts:37,124 */
int shk_assemble(shk_handle *h);

/* AssemblyHelper::get_assembly() -> String                             Assembler.ts:38,127
 * JSON {"outfasta":str,"ncontigs":int,"outdot":str,"outgfa":str,"outgfav2":str}  :7-13 */
const char *shk_get_assembly(shk_handle *h);

/* ---- shard layer: one process per GPU (DESIGN.md "Multi-GPU").  It replaces the crate's rayon
 * read-parallel driver (north_star; not in the reference tree).  The
 * k-mer space is split by minimiser-hash partition: partition p belongs to rank p % world.
 * Sequence on every rank (collectives done by the caller, e.g. sparrowhawk_amd/dist.py):
 *   shk_shard_partition  reads -> super-k-mer records per partition; part_records[p] = records held
 *   shk_shard_pack       records copied densely to d_send at base_records[p] (caller's order)
 *        -- ONE all-to-all of d_send blocks (RCCL) --
 *   shk_shard_count      counts the owned partitions from d_recv; run tables [n_owned][n_sources]
 *                        give record offset/count of each source's run; local histogram out
 *        -- all-reduce of the 500-bin histogram and the instance count --
 *   shk_shard_rows       fit + filter on the global histogram; local solid rows (device pointers,
 *                        W key arrays + one count array, valid until shk_shard_set_solid)
 *        -- all-gather of the solid rows --
 *   shk_shard_set_solid  installs the whole solid set; the handle is then "preprocessed" and
 *                        shk_assemble() runs as usual (identically on every rank)
 * n_partitions: a power of two <= 16384, identical on all ranks. */
int shk_shard_partition(shk_handle *h, const void *d_bases, const void *d_seg_off, uint64_t n_seg,
                        uint64_t n_bases, uint64_t n_reads, uint32_t n_partitions,
                        uint64_t *part_records /* [n_partitions] out */);
uint32_t shk_shard_record_bytes(shk_handle *h);
int shk_shard_pack(shk_handle *h, void *d_send, const uint64_t *base_records, uint32_t n_partitions);
int shk_shard_count(shk_handle *h, const void *d_recv, const uint64_t *run_off, const uint32_t *run_cnt,
                    uint32_t n_owned, uint32_t n_sources, uint64_t *histo500_local,
                    uint64_t *n_instances_local);
int shk_shard_rows(shk_handle *h, const uint64_t *histo500_global, const void **d_keys /* [W] out */,
                   const void **d_cnt, uint64_t *n_rows, uint32_t *used_min_count);
int shk_shard_set_solid(shk_handle *h, const void *const *d_keys /* [W] */, const void *d_cnt,
                        uint64_t n_rows, uint64_t n_instances_global);

/* ---- the same sequence with the collectives INSIDE the library: RCCL over xGMI (librccl.so.1, opened on
 * first use).  One communicator per process and GPU.  The caller only distributes the 128-byte id that rank 0
 * obtains from shk_comm_unique_id() (ncclGetUniqueId) by whatever channel it has (MPI, a TCP store, a file),
 * then every rank — with its GPU current — calls shk_comm_init (ncclCommInitRank).  shk_shard_preprocess is
 * collective over the communicator: every rank calls it with its own share of the reads (n_seg may be 0) and
 * the same n_partitions (0 = chosen from the global instance count).  It runs
 *   pass 1 -> all-gather of the per-partition record counts (the size exchange) -> pack -> ONE pairwise
 *   exchange of the records (grouped ncclSend/ncclRecv) -> pass 2 over the owned partitions -> all-reduce
 *   of histogram + instance count (501 x u64) -> fit / filter -> all-gather of the solid rows -> install,
 * leaving every rank's handle "preprocessed".  By default the GRAPH STAYS SHARDED: every rank keeps the solid k-mers of
 * its own partitions, and shk_assemble() on these handles is COLLECTIVE over the same communicator (call it on every
 * rank; keep the communicator until it has returned): adjacency with one pairwise exchange of neighbour queries and one
 * of answers, non-branching paths contracted per rank and stitched across ranks, tips / bubbles on the graph of unitigs,
 * every rank writing the bases of its own k-mers into the contigs (csrc/shard_graph.h).  Every rank ends with the same
 * get_assembly() text; the stage-inspection getters then describe the rank's own rows.  With SHK_SHARD_GRAPH=0 in the
 * environment the solid set is gathered instead (all-gather of the solid rows) and shk_assemble() runs locally, on the
 * whole graph, on every rank.  A Rust host binds these five functions and never writes a collective itself. */
#define SHK_UNIQUE_ID_BYTES 128
typedef struct shk_comm shk_comm;
int shk_comm_unique_id(uint8_t id[SHK_UNIQUE_ID_BYTES]);
shk_comm *shk_comm_init(const uint8_t id[SHK_UNIQUE_ID_BYTES], int rank, int world);   /* NULL on failure */
const char *shk_comm_error(void);        /* message of the last failed shk_comm_* call on this thread */
int shk_comm_rank(const shk_comm *c);
int shk_comm_world(const shk_comm *c);
void shk_comm_free(shk_comm *c);
int shk_shard_preprocess(shk_handle *h, shk_comm *c, const void *d_bases, const void *d_seg_off, uint64_t n_seg,
                         uint64_t n_bases, uint64_t n_reads, uint32_t n_partitions);
/* host-only: the exchange plan shk_shard_preprocess derives from the gathered record counts (exposed for the
 * CPU tests).  part_records_all: [world][n_partitions].  Outputs (caller-allocated): base [n_partitions],
 * send_counts / recv_counts [world], run_off / run_cnt [n_owned][world], n_owned = partitions p with
 * p % world == rank. */
int shk_plan_exchange(const uint64_t *part_records_all, uint32_t world, uint32_t n_partitions, uint32_t rank,
                      uint64_t *base, uint64_t *send_counts, uint64_t *recv_counts, uint64_t *run_off,
                      uint32_t *run_cnt);
uint32_t shk_choose_partitions(uint64_t total_instances_ub, uint32_t world, uint32_t key_words);

/* ---- host-side packer (the parser the preprocess entry points use), exposed so a caller can
 * stage packed reads in HBM itself (bench.py, the multi-GPU shard layer). */
typedef struct shk_packed {
    uint32_t *bases;       /* ceil(n_bases/16)+1 words */
    uint32_t *seg_off;     /* n_seg+1 */
    uint64_t n_seg, n_bases, n_reads, n_input_bases;
} shk_packed;
/* returns SHK_OK or SHK_E_PARSE / SHK_E_OOM; *err (optional) receives a static message */
int shk_pack_fastq(const uint8_t *fq, size_t n, uint32_t k, uint32_t min_qual, shk_packed *out,
                   const char **err);
void shk_packed_free(shk_packed *p);

/* ---- stage inspection (used by the parity tests; SURVEY.md §7 step 1 stages a-f).
 * All copy device state to caller-provided host arrays.  Key layout: W = ceil(2k/64) uint64
 * words per k-mer, w[0] least significant (SPEC S3).  Order of rows is unspecified. */
uint32_t shk_key_words(shk_handle *h);
uint64_t shk_total_instances(shk_handle *h);            /* valid k-mer windows counted (S4)   */
uint64_t shk_n_distinct(shk_handle *h);
int shk_get_distinct(shk_handle *h, uint64_t *keys, uint32_t *counts, uint64_t cap);
uint64_t shk_n_solid(shk_handle *h);
int shk_get_solid(shk_handle *h, uint64_t *keys, uint32_t *counts, uint64_t cap);
int shk_get_histo(shk_handle *h, uint64_t *histo500);
uint32_t shk_used_min_count(shk_handle *h);
/* after shk_assemble: per solid node (same row order as shk_get_solid) */
int shk_get_adjacency(shk_handle *h, uint8_t *adj_initial, uint8_t *adj_final, uint8_t *alive,
                      uint64_t cap);
/* timing of the last preprocess/assemble, milliseconds, by stage name; returns a JSON object.  Two entries are not times:
 * "peak_device_bytes" = the most device memory the handle held at once, "device_bytes_now" = what it holds at the call
 * (the reference reports the peak wasm memory with every assembly: Assembler.ts:69-71,137). */
const char *shk_get_timings(shk_handle *h);
/* the same figure alone: high-water mark of the device bytes charged to this handle (its pool blocks, the packed reads
 * uploaded for it, the exchange buffers of the shard layer) */
uint64_t shk_peak_device_bytes(shk_handle *h);
/* host-only (tests): the counter behind it — applies signed byte deltas in order, reports high-water mark and remainder */
void shk_host_mem_counter(const int64_t *deltas, size_t n, uint64_t *peak, uint64_t *current);

/* ---- host-only self tests of the device/host shared k-mer arithmetic (no GPU needed) */
int shk_host_canonical(const char *seq, uint32_t k, uint64_t *out_words /*W*/, int *orient);
uint64_t shk_host_nthash(const char *seq, uint32_t k);   /* canonical ntHash of seq[0..k) */
int shk_host_fit(const uint64_t *histo500, uint32_t *used_min_count); /* 1 ok, 0 fit failed */
/* the output writer alone (SPEC S10-S11: canonical strand, order, links, FASTA / DOT / GFA1 / GFA2, JSON) on
 * unitigs handed in as text, any strand, any order: seqs = the spellings back to back, offsets[n+1] their
 * bounds, kc[n] their k-mer count sums.  Returns a malloc'd NUL-terminated JSON (shk_host_free) or NULL. */
char *shk_host_assembly_json(const char *seqs, const uint64_t *offsets, const uint64_t *kc, uint64_t n_contigs, uint32_t k);
/* the same on text that arrives while the writer works, piece_bytes at a time every delay_us microseconds (the device path
 * downloads the contigs in pieces and the writer copies what is there: csrc/pipeline.h TextArrival) — same JSON */
char *shk_host_assembly_json_arriving(const char *seqs, const uint64_t *offsets, const uint64_t *kc, uint64_t n_contigs, uint32_t k,
                                      uint64_t piece_bytes, uint32_t delay_us);
void shk_host_free(void *p);
/* host-only: the gzip reader of shk_preprocess alone (fastx_wasm.rs:53-70: gz sniff, multi-member) — BGZF blocks in
 * parallel, a large plain member by the multi-threaded two-pass inflater (csrc/inflate_mt.cpp), the rest by zlib.  *out is
 * malloc'd (shk_host_free); plain input is copied through.  *mt_members (optional): members the multi-threaded inflater
 * has handled in this process so far; *reader_seconds (optional): the time the reader itself took (without the copy into
 * *out).  SHK_GUNZIP_THREADS (environment): its thread count, 0 or 1 = zlib only. */
int shk_host_gunzip(const uint8_t *gz, size_t n, uint8_t **out, size_t *out_n, uint64_t *mt_members, double *reader_seconds);
/* the DEVICE inflater alone (csrc/inflate_gpu.hip; needs a GPU): shk_preprocess hands a plain gzip member of >= 4 MiB
 * (SHK_GUNZIP_DEVICE_MIN) to it first — the compressed bytes are what crosses PCIe — and reads on the host whatever it does
 * not take.  0: *out (malloc'd, shk_host_free) holds the member's bytes, equal to zlib's; 1: not taken, *why (optional) says
 * why; < 0: error.  *ms_total (optional): upload + kernels + checks. */
int shk_device_gunzip(const uint8_t *gz, size_t n, uint8_t **out, size_t *out_n, const char **why, double *ms_total);
/* SPEC S9 (tips, bubbles) and S10 (chains of simple links, the circular cut) on UNITIG records instead of k-mers — what the
 * sharded assembly runs on every rank's host once the k-mer-level contraction is done on the GPUs (csrc/unitig_graph.h:
 * one record per strand of a unitig; first / last: [n_recs][W] words of its first / last k-mer as spelled; min_*: the
 * smallest oriented node of a record and its position, read for the records of rings only).  Exposed for the CPU tests.
 * Returns a malloc'd text (shk_host_free): "removed <tip nodes> <bubble nodes>", then one line per contig
 * "<ring> <rot> <nodes> <kc> : <record> <record> ...", or NULL on inconsistent input. */
char *shk_host_unitig_assemble(uint32_t k, uint64_t n_recs, const uint64_t *first, const uint64_t *last, const uint64_t *len,
                               const uint64_t *kc, const uint8_t *circ, const uint64_t *min_key, const uint8_t *min_o,
                               const uint64_t *min_pos, int tips, int bubbles);

/* Device buffers of freed handles are cached process-wide for the next handle (a handle lives
 * for one preprocess+assemble); this returns the cache to the driver.  SHK_NO_POOL=1 disables it. */
void shk_release_cached_memory(void);

/* Measurement helper (bench.py): best rate in GB/s of `iters` pure streaming reads of a
 * `bytes` device buffer on the current HIP device — the achievable HBM read peak of this box, reported
 * beside the nominal 8 TB/s.  No counterpart in the reference. */
int shk_measure_stream_read(size_t bytes, int iters, double *gbs);

const char *shk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SHK_H */
