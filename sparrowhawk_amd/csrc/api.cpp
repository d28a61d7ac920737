// api.cpp — the C ABI of libshk_hip.so (include/shk.h): the stateful AssemblyHelper the
// reference's worker drives (www/src/workers/Assembler.ts:15-39,73-143), its call-order state
// machine, the progress strings (AssemblyPage.vue:458-609) and the JSON getters.
#include "../../include/shk.h"

#include <atomic>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "fastq.h"
#include "fastq_gpu.h"
#include "inflate_gpu.h"
#include "inflate_mt.h"
#include "outputs.h"
#include "pipeline.h"
#include "shard_comm.h"
#include "unitig_graph.h"

namespace shk {
bool spectrum_fit(const uint64_t *histo500, uint32_t *out);
}

using namespace shk;

namespace {

thread_local int g_new_err = 0;
thread_local std::string g_new_msg;

enum class St { Fresh, Streaming, Sharding, Preprocessed, Assembled, Failed };

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct shk_handle {
    uint32_t k = 31, min_count = 5, min_qual = 20;
    uint64_t chunk_size = 0;
    bool verbose = false, do_bloom = false, do_fit = false, no_bubble = false, no_deadend = false;
    St st = St::Fresh;
    IPipeline *pipe = nullptr;
    shk_progress_cb cb = nullptr;
    void *cb_user = nullptr;
    std::string err, first_err, pre_json, timings_json;
    ByteVec asm_json;                  // NUL-terminated
    const char *asm_json_dev = nullptr; // a fragmented assembly's JSON, written on the device: pinned memory owned by the pipeline
    uint64_t histo[SHK_HISTO_BINS] = {0};
    uint32_t used_min_count = 0;
    bool fit_ok = false;
    PackedReads stream_reads;          // shk_push_reads accumulator
    uint64_t n_reads = 0;
    uint64_t batches_started = 0;      // batches handed to the pipeline (a failure after the first one poisons the handle)
    ShardComm *shard_comm = nullptr;   // sharded assembly: the communicator shk_shard_preprocess ran on (shk_assemble is collective over it)
    AssemblyText text;
    std::shared_ptr<void> mem = mem_acct_new();   // device bytes this handle holds / held at most (pipeline.h: mem_acct_*)

    const char *mode() const { return do_bloom ? "bloom" : (chunk_size > 0 ? "chunked" : "bulk"); }
    void post(const std::string &s) { if (cb) cb(s.c_str(), cb_user); }
    void post_mode(const char *suffix) { post(std::string("preprocess:") + mode() + ":" + suffix); }
    // `loop:start` / `loop:end` exist for bulk and bloom only: the reference UI defines no such state for the
    // chunked mode (AssemblyPage.vue:548-579 has :start, :fitting, :filtering and :loop:<n>[:<pct>])
    void post_loop_edge(const char *suffix) { if (do_bloom || chunk_size == 0) post_mode(suffix); }
    uint64_t progress_every() const { return (!do_bloom && chunk_size > 0) ? chunk_size : 100000; }
};

static int fail(shk_handle *h, int code, const std::string &msg) { h->err = msg; return code; }

// Every entry point that touches the device runs with the HANDLE's device current (the HIP current device is
// per thread and new threads start on device 0: an FFI consumer may call from any thread) and restores the
// caller's device afterwards; no exception crosses the C ABI (shk.h: "never aborts").
struct DevGuard {
    int prev;
    explicit DevGuard(int dev) : prev(set_device(dev)) {}
    ~DevGuard() { (void)set_device(prev); }
    DevGuard(const DevGuard &) = delete;
    DevGuard &operator=(const DevGuard &) = delete;
};
// the calling thread allocates (and frees) device blocks on behalf of this handle while the guard lives
struct MemGuard {
    std::shared_ptr<void> prev;
    explicit MemGuard(const std::shared_ptr<void> &a) : prev(mem_acct_set(a)) {}
    ~MemGuard() { (void)mem_acct_set(prev); }
    MemGuard(const MemGuard &) = delete;
    MemGuard &operator=(const MemGuard &) = delete;
};
enum class Poison { Never, AfterFirstBatch, Always };
template <typename F> static int guarded(shk_handle *h, Poison poison, F &&body) {
    if (!h) return SHK_E_PARAM;
    if (h->st == St::Failed) return fail(h, SHK_E_STATE, "the handle failed earlier (" + h->first_err + "): free it and start again");
    DevGuard g(h->pipe ? h->pipe->device() : current_device());
    MemGuard mg(h->mem);
    // device buffers that go out of scope inside the call (early returns included) are parked until the handle's stream
    // has drained, then go back to the pool: nothing in flight can be handed to another handle (pipeline.h)
    DeferScope park(h->pipe);
    int rc;
    try { rc = body(); }
    catch (const std::bad_alloc &) { rc = fail(h, SHK_E_OOM, "out of host memory"); }
    catch (const std::exception &e) { rc = fail(h, SHK_E_INTERNAL, std::string("unexpected exception: ") + e.what()); }
    catch (...) { rc = fail(h, SHK_E_INTERNAL, "unexpected exception"); }
    // a failed call leaves counted batches / a half-built graph behind: a second attempt on the same handle
    // would double-count them, so the handle only accepts shk_free / shk_last_error from here on
    if (rc != SHK_OK && rc != SHK_E_STATE &&
        (poison == Poison::Always || (poison == Poison::AfterFirstBatch && (h->batches_started > 0 || h->st == St::Sharding))))
        { h->st = St::Failed; h->first_err = h->err; }
    return rc;
}

extern "C" {

// the communicators that exist: a handle keeps the one its sharded preprocess ran on, and shk_assemble must find out that it
// has been freed meanwhile (SHK_E_STATE, not a use after free)
static std::mutex g_live_mu;
static std::set<ShardComm *> &live_comms() { static auto *s = new std::set<ShardComm *>(); return *s; }
static bool comm_is_live(ShardComm *c) { std::lock_guard<std::mutex> lk(g_live_mu); return live_comms().count(c) != 0; }

void shk_release_cached_memory(void) { device_pool_trim(); big_trim(); }
int shk_measure_stream_read(size_t bytes, int iters, double *gbs) {
    std::string err;
    const int rc = stream_read_gbs(bytes, iters, gbs, err);
    return rc == 0 ? SHK_OK : rc == -1 ? SHK_E_PARAM : rc == -4 ? SHK_E_OOM : SHK_E_DEVICE;
}

const char *shk_version(void) { return "sparrowhawk_amd 0.1 (gfx950)"; }
int shk_new_error(void) { return g_new_err; }
const char *shk_new_error_message(void) { return g_new_msg.c_str(); }

shk_handle *shk_new(uint32_t k, int verbose, uint32_t min_count, uint32_t min_qual, uint64_t chunk_size,
                    int do_bloom, int do_fit, int no_bubble_collapse, int no_dead_end_removal) {
    g_new_err = 0; g_new_msg.clear();
    if ((k & 1u) == 0 || k < SHK_K_MIN || k > SHK_K_MAX) {
        g_new_err = SHK_E_PARAM; g_new_msg = "k must be odd and within [15, 255]"; return nullptr;
    }
    if (min_qual > 93) { g_new_err = SHK_E_PARAM; g_new_msg = "min_qual out of range"; return nullptr; }
    if (min_count >= SHK_HISTO_BINS) { g_new_err = SHK_E_PARAM; g_new_msg = "min_count out of range"; return nullptr; }
    if (do_bloom && min_count < 3) {     // AssemblyPage.vue:430-432,628-632
        g_new_err = SHK_E_PARAM; g_new_msg = "Bloom mode requires min_count >= 3"; return nullptr;
    }
    shk_handle *h = new (std::nothrow) shk_handle();
    if (!h) { g_new_err = SHK_E_OOM; g_new_msg = "out of host memory"; return nullptr; }
    h->k = k; h->verbose = verbose != 0; h->min_count = min_count; h->min_qual = min_qual;
    h->chunk_size = chunk_size; h->do_bloom = do_bloom != 0; h->do_fit = do_fit != 0;
    h->no_bubble = no_bubble_collapse != 0; h->no_deadend = no_dead_end_removal != 0;
    std::string err;
    {
        MemGuard mg(h->mem);
        h->pipe = make_pipeline((int)k, err);
    }
    if (!h->pipe) { g_new_err = SHK_E_DEVICE; g_new_msg = err; delete h; return nullptr; }
    h->pipe->set_bloom(h->do_bloom);
    h->pipe->set_verbose(h->verbose);
    return h;
}

void shk_free(shk_handle *h) {
    if (!h) return;
    DevGuard g(h->pipe ? h->pipe->device() : current_device());
    delete h->pipe;
    delete h;
}

const char *shk_last_error(shk_handle *h) { return h ? h->err.c_str() : "null handle"; }
void shk_set_progress_cb(shk_handle *h, shk_progress_cb cb, void *user) { if (h) { h->cb = cb; h->cb_user = user; } }

// common tail of every preprocess entry point: packed segments are in HBM
// one batch of packed segments in HBM -> pass 1 (several batches per handle are allowed)
static int count_one_batch(shk_handle *h, const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg,
                           uint64_t n_bases) {
    std::string err;
    const double t0 = now_ms();
    h->batches_started++;
    int rc = h->pipe->count_batch(d_bases, d_seg_off, n_seg, n_bases, err);
    h->pipe->times().add("preprocess_device_total_host_clock", now_ms() - t0);
    if (rc) return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err);
    return SHK_OK;
}

// common tail of every preprocess entry point: all batches are in -> histogram, fit, filter
static int finish_counting(shk_handle *h) {
    std::string err;
    const double t0 = now_ms();
    h->post_loop_edge("loop:end");
    if (!h->do_bloom && h->chunk_size == 0) h->post("preprocess:bulk:sorting");
    // the fit never returns less than 1 and falls back to min_count (SPEC S6)
    const uint32_t emit_thr = h->do_fit ? (h->min_count < 1u ? h->min_count : 1u) : h->min_count;
    int rc = h->pipe->histogram(h->histo, emit_thr, err);
    if (rc) return fail(h, SHK_E_DEVICE, err);
    h->used_min_count = h->min_count; h->fit_ok = false;
    if (h->do_fit) {
        h->post_mode("fitting");
        uint32_t v = 0;
        if (spectrum_fit(h->histo, &v)) { h->used_min_count = v; h->fit_ok = true; }
    }
    h->post_mode("filtering");
    rc = h->pipe->filter(h->used_min_count, err);
    if (rc) return fail(h, rc == -4 ? SHK_E_OOM : SHK_E_DEVICE, err);
    h->post("preprocess:saving");
    h->pre_json = preprocessing_json(h->pipe->n_solid(), h->histo, h->used_min_count);
    h->pipe->times().add("preprocess_device_total_host_clock", now_ms() - t0);
    h->st = St::Preprocessed;
    h->post("preprocess:end");
    return SHK_OK;
}

static int run_counting(shk_handle *h, const uint32_t *d_bases, const uint32_t *d_seg_off, uint64_t n_seg,
                        uint64_t n_bases) {
    // (one batch, and the caller's reads stay in place until this function returns: the two counting passes run back to back)
    h->pipe->single_batch_resident(true);
    int rc = count_one_batch(h, d_bases, d_seg_off, n_seg, n_bases);
    if (!rc) rc = finish_counting(h);
    h->pipe->single_batch_resident(false);
    return rc;
}

// bases per batch of the host-parsed paths (a batch is limited to 2^32 packed bases by its 32-bit offsets)
static uint64_t batch_bases() {
    const char *v = getenv("SHK_BATCH_BASES");
    const uint64_t b = (v && *v) ? strtoull(v, nullptr, 10) : (1ull << 31);
    return b < 1024 ? 1024 : (b > (3ull << 30) ? (3ull << 30) : b);
}

// hand the packed stream on as one batch (upload + pass 1) and empty it; read counters are kept
static int flush_host_batch(shk_handle *h, PackedReads &pr) {
    if (pr.n_seg() == 0) { pr.reset_stream(); return SHK_OK; }
    pr.finish();
    std::string err;
    void *d_bases = nullptr, *d_off = nullptr;
    const double t0 = now_ms();
    int rc = device_upload(pr.bases.data(), pr.bases.size() * 4, &d_bases, err);
    if (!rc) rc = device_upload(pr.seg_off.data(), pr.seg_off.size() * 4, &d_off, err);
    if (rc) { device_free(d_bases); device_free(d_off); return fail(h, SHK_E_OOM, err); }
    h->pipe->times().add("h2d_upload_host_clock", now_ms() - t0);
    rc = count_one_batch(h, (const uint32_t *)d_bases, (const uint32_t *)d_off, pr.n_seg(), pr.n_bases);
    device_free(d_bases); device_free(d_off);
    pr.reset_stream();
    return rc;
}

// chunked mode hands a batch on every chunk_size reads (docs/src/assembly.md:17: "reads per batch")
static uint64_t flush_every_reads(const shk_handle *h) { return (!h->do_bloom && h->chunk_size > 0) ? h->chunk_size : 0; }

// start of the first FASTQ record at or after `from` (a line starting with '@' whose line after next starts
// with '+': a quality line may start with '@' too, but then the line after next is a sequence); n = none,
// SIZE_MAX = the text does not look like 4-line FASTQ here
static size_t next_record_start(const uint8_t *t, size_t n, size_t from) {
    size_t p = from;
    if (p >= n) return n;
    if (p > 0 && t[p - 1] != '\n') {
        const void *nl = memchr(t + p, '\n', n - p);
        if (!nl) return n;
        p = (size_t)((const uint8_t *)nl - t) + 1;
    }
    for (int tries = 0; tries < 8 && p < n; tries++) {
        const void *e0 = memchr(t + p, '\n', n - p);
        if (!e0) return n;
        const size_t b = (size_t)((const uint8_t *)e0 - t) + 1;
        if (b >= n) return n;
        const void *e1 = memchr(t + b, '\n', n - b);
        if (!e1) return n;
        const size_t c = (size_t)((const uint8_t *)e1 - t) + 1;
        if (t[p] == '@' && c < n && t[c] == '+') return p;
        p = b;
    }
    return SIZE_MAX;
}

// A text of more than one batch: pieces of ~2 * batch_bases() bytes, cut at record boundaries, go through
// the device parser one after the other, each counted as its own batch (pass 1) before the next is parsed;
// a helper thread uploads piece i+1 while piece i is parsed and counted (H2D is two thirds of the entry point).
// handled = false (and nothing counted) when the very first piece is not regular 4-line FASTQ: the caller
// then runs the host parser over everything.  A later piece that is not regular is parsed on the host from
// there to the end of its file, with the record numbers and progress of the whole file.
// one_batch: the text fits one batch.  It is still cut into a few pieces so that piece i+1 travels while piece i is
// parsed, but the packed pieces are kept and counted together as ONE batch at the end (pass 1 runs over the pieces
// into the same slices: the partitioning of a single batch, no batch packing, no merge).
static int preprocess_device_pieces(shk_handle *h, const uint8_t *t1, size_t l1, const uint8_t *t2, size_t l2,
                                    size_t n1, size_t total, bool &handled, bool one_batch = false) {
    handled = false;
    std::string err;
    size_t piece_bytes = (size_t)(2 * batch_bases());
    if (one_batch) {
        const char *pv = getenv("SHK_FASTQ_PIECES");
        const size_t C = (pv && *pv) ? (size_t)std::max<long long>(1, atoll(pv)) : 4;
        // the last piece is parsed with nothing left to upload: it is the small one (a tenth of the text)
        const size_t tot = l1 + (t2 ? l2 : 0);
        piece_bytes = std::max<size_t>(C >= 3 ? tot / 10 * 9 / (C - 1) : tot / C, 1024);
    }
    std::vector<GpuPacked> kept;                          // one_batch: the parsed pieces, counted together
    auto free_kept = [&]() { for (auto &g : kept) gpu_packed_free(g); kept.clear(); };
    auto count_kept = [&]() -> int {
        if (kept.empty()) return SHK_OK;
        std::vector<DevPiece> pcs;
        for (auto &g : kept) pcs.push_back(DevPiece{g.d_bases, g.d_seg_off, g.n_seg, g.n_bases});
        std::string e2;
        const double tc = now_ms();
        h->batches_started++;
        const int rc = h->pipe->count_batch_pieces(pcs.data(), pcs.size(), e2);
        h->pipe->times().add("preprocess_device_total_host_clock", now_ms() - tc);
        free_kept();
        if (rc) return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), e2);
        return SHK_OK;
    };
    struct Piece { int file; size_t off, end; bool host_rest; };
    std::vector<Piece> pieces;
    for (int f = 0; f < 2; f++) {
        const uint8_t *t = f ? t2 : t1;
        const size_t len = f ? l2 : l1;
        if (!t) continue;
        size_t off = 0;
        while (off < len) {
            size_t end = len;
            if (len - off > piece_bytes + piece_bytes / 8) {
                end = next_record_start(t, len, off + piece_bytes);
                if (end == SIZE_MAX || end <= off) { pieces.push_back(Piece{f, off, len, true}); break; }   // no record boundary: host from here
            }
            pieces.push_back(Piece{f, off, end, false});
            off = end;
        }
    }
    const int device = h->pipe->device();
    uint64_t reads_done = 0, file_reads = 0;
    bool counted_any = false;
    const double t0 = now_ms();
    GpuText cur, nxt;
    bool have_cur = false;
    struct Joining { std::thread t; ~Joining() { if (t.joinable()) t.join(); } } upl;   // joined on every way out, exceptions included
    std::thread &uploader = upl.t;
    int up_rc = 0; std::string up_err;
    auto join_upload = [&]() { if (uploader.joinable()) uploader.join(); };
    auto drop_all = [&]() { join_upload(); gpu_text_free(cur); gpu_text_free(nxt); free_kept(); };
    // the rest of a file through the host parser (a piece that is not regular 4-line FASTQ, or no boundary found)
    auto host_rest = [&](const Piece &pc) -> int {
        const uint8_t *t = pc.file ? t2 : t1;
        const size_t len = pc.file ? l2 : l1, done_before = pc.file ? n1 : 0;
        PackedReads pr;
        pr.n_reads = reads_done;
        auto prog = [&](uint64_t reads, uint64_t bytes, uint64_t) {
            const uint64_t pct = total ? (100 * (done_before + pc.off + bytes)) / total : 100;
            h->post_mode(("loop:" + std::to_string(reads) + ":" + std::to_string(pct)).c_str());
        };
        int flush_rc = SHK_OK;
        auto flush = [&](PackedReads &p) -> int { flush_rc = flush_host_batch(h, p); return flush_rc ? -7 : 0; };
        if (int rck = count_kept()) return rck;           // (one_batch: what the device parsed so far is a batch of its own now)
        if (!counted_any) h->pipe->expect_more_batches();
        int rc2 = pack_fastq(t + pc.off, len - pc.off, h->k, h->min_qual, pr, err, h->progress_every(), prog, 0, batch_bases(), flush, file_reads);
        if (rc2 == -7) return flush_rc;
        if (rc2) return fail(h, rc2 == -3 ? SHK_E_PARSE : (rc2 == -4 ? SHK_E_OOM : SHK_E_PARAM), err);
        if (int rc3 = flush_host_batch(h, pr)) return rc3;
        reads_done = pr.n_reads;
        counted_any = true;
        return SHK_OK;
    };
    for (size_t i = 0; i < pieces.size(); i++) {
        const Piece pc = pieces[i];
        if (i == 0 || pieces[i - 1].file != pc.file) file_reads = 0;
        if (pc.host_rest) {
            if (!counted_any && i == 0) { drop_all(); return SHK_OK; }          // handled stays false: the caller's host path
            if (int rc = host_rest(pc)) { drop_all(); return rc; }
            continue;
        }
        const uint8_t *t = pc.file ? t2 : t1;
        if (!have_cur) {
            if (int rc = gpu_upload_text(t + pc.off, pc.end - pc.off, device, cur, err)) return fail(h, rc == -4 ? SHK_E_OOM : SHK_E_DEVICE, err);
            have_cur = true;
        }
        // the next piece travels while this one is parsed and counted
        const bool prefetch = i + 1 < pieces.size() && !pieces[i + 1].host_rest;
        if (prefetch) {
            const Piece nx = pieces[i + 1];
            const uint8_t *tn = nx.file ? t2 : t1;
            uploader = std::thread([&, nx, tn]() {
                MemGuard mg(h->mem);                       // (the text block this thread allocates belongs to the handle)
                try { up_rc = gpu_upload_text(tn + nx.off, nx.end - nx.off, device, nxt, up_err); }
                catch (...) { up_rc = -4; up_err = "out of host memory (uploader)"; }
            });
        }
        GpuPacked gp;
        const double tp0 = now_ms();
        int rc = gpu_pack_fastq(nullptr, 0, nullptr, 0, h->k, h->min_qual, h->progress_every(), h->pipe->stream(), gp, err, reads_done, &cur);
        h->pipe->times().add("fastq_piece_parse_host_clock", now_ms() - tp0);
        if (rc < 0) { gpu_packed_free(gp); drop_all(); return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err); }
        if (rc == 1) {
            gpu_packed_free(gp);
            join_upload(); gpu_text_free(cur); gpu_text_free(nxt); have_cur = false;
            if (!counted_any) { free_kept(); return SHK_OK; }             // handled stays false
            Piece rest = pc; rest.end = pc.file ? l2 : l1; rest.host_rest = true;
            if (int rc2 = host_rest(rest)) return rc2;
            while (i + 1 < pieces.size() && pieces[i + 1].file == pc.file) i++;      // the rest of this file is done
            continue;
        }
        if (!counted_any && !one_batch) h->pipe->expect_more_batches();
        h->pipe->times().add("fastq_h2d_text", gp.h2d_ms);
        h->pipe->times().add("fastq_device_kernels", gp.kernels_ms);
        h->pipe->times().add("fastq_device_pieces_x1", 1.0);
        const uint64_t every = h->progress_every();
        const size_t done_before = pc.file ? n1 : 0;
        for (size_t j = 0; j < gp.progress_bytes.size(); j++) {
            const uint64_t bytes = gp.progress_bytes[j] & ~(1ull << 63);
            const uint64_t pct = total ? (100 * (done_before + pc.off + bytes)) / total : 100;
            h->post_mode(("loop:" + std::to_string(every * (gp.first_mark + j + 1)) + ":" + std::to_string(pct)).c_str());
        }
        int rc2 = SHK_OK;
        reads_done += gp.n_reads; file_reads += gp.n_reads;
        if (one_batch) { kept.push_back(gp); gp = GpuPacked(); }       // (counted with the other pieces at the end)
        else {
            rc2 = gp.n_seg ? count_one_batch(h, gp.d_bases, gp.d_seg_off, gp.n_seg, gp.n_bases) : SHK_OK;
            gpu_packed_free(gp);
        }
        if (rc2) { drop_all(); return rc2; }
        counted_any = true;
        // hand over to the uploaded next piece
        const double tj0 = now_ms();
        join_upload();
        h->pipe->times().add("fastq_piece_wait_for_upload_host_clock", now_ms() - tj0);
        gpu_text_free(cur); have_cur = false;
        if (prefetch) {
            if (up_rc) { gpu_text_free(nxt); return fail(h, up_rc == -4 ? SHK_E_OOM : SHK_E_DEVICE, up_err); }
            cur = nxt; nxt = GpuText(); have_cur = true;
        }
    }
    join_upload(); gpu_text_free(cur); gpu_text_free(nxt);
    h->pipe->times().add("fastq_device_parse_pack_host_clock", now_ms() - t0);
    if (int rck = count_kept()) return rck;
    handled = true;
    h->n_reads = reads_done;
    return finish_counting(h);
}

// Both files (or the one) are plain gzip members the device inflater takes: inflate -> device parser -> one batch.
// handled = false (nothing counted, nothing posted beyond what the host path posts again) when any file is not taken or
// turns out not to be regular 4-line FASTQ: the caller's host reader then starts over.
static int preprocess_device_gzip(shk_handle *h, const uint8_t *fq1, size_t n1, const uint8_t *fq2, size_t n2, size_t total, bool &handled) {
    handled = false;
    const uint8_t *gz[2] = {fq1, fq2}; const size_t gn[2] = {n1, fq2 ? n2 : 0};
    const int nf = fq2 ? 2 : 1;
    for (int f = 0; f < nf; f++) if (gn[f] < 18 || gz[f][0] != 0x1F || gz[f][1] != 0x8B) return SHK_OK;
    std::string err;
    const double t0 = now_ms();
    GpuText text[2];
    GpuPacked packed[2];
    auto drop = [&]() { for (int f = 0; f < 2; f++) { gpu_text_free(text[f]); gpu_packed_free(packed[f]); } };
    double ms_h2d = 0, ms_search = 0, ms_decode = 0, ms_resolve = 0;
    for (int f = 0; f < nf; f++) {
        GpuInflateStats st;
        const int rc = gpu_inflate_member(gz[f], gn[f], h->pipe->device(), h->pipe->stream(), text[f], err, &st);
        if (rc == 1) { drop(); h->pipe->times().add("gunzip_device_not_taken_x1", 1.0); return SHK_OK; }
        if (rc) { drop(); return fail(h, rc == -4 ? SHK_E_OOM : SHK_E_DEVICE, err); }
        ms_h2d += st.h2d_ms; ms_search += st.search_ms; ms_decode += st.decode_ms; ms_resolve += st.resolve_ms;
    }
    const double t1 = now_ms();
    uint64_t text_total = 0;
    for (int f = 0; f < nf; f++) text_total += text[f].e;
    if (text_total / 2 > batch_bases()) { drop(); return SHK_OK; }             // (several batches: the host reader's piece-wise path)
    uint64_t reads_done = 0, text_before = 0;
    for (int f = 0; f < nf; f++) {
        const int rc = gpu_pack_fastq(nullptr, 0, nullptr, 0, h->k, h->min_qual, h->progress_every(), h->pipe->stream(), packed[f], err, reads_done, &text[f]);
        if (rc < 0) { drop(); return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err); }
        if (rc == 1) { drop(); h->pipe->times().add("gunzip_device_not_taken_x1", 1.0); return SHK_OK; }      // not regular FASTQ: the host parser owns the messages
        // progress as the text path posts it: every `every` reads, with the share of the (inflated) text consumed so far
        const uint64_t every = h->progress_every();
        for (size_t j = 0; j < packed[f].progress_bytes.size(); j++) {
            const uint64_t bytes = packed[f].progress_bytes[j] & ~(1ull << 63);
            const uint64_t pct = text_total ? (100 * (text_before + bytes)) / text_total : 100;
            h->post_mode(("loop:" + std::to_string(every * (packed[f].first_mark + j + 1)) + ":" + std::to_string(pct)).c_str());
        }
        reads_done += packed[f].n_reads;
        text_before += text[f].e;
        h->pipe->times().add("fastq_device_kernels", packed[f].kernels_ms);
        gpu_text_free(text[f]);
    }
    (void)total;
    h->pipe->times().add("gunzip_device_host_clock", t1 - t0);
    h->pipe->times().add("gunzip_device_h2d", ms_h2d);
    h->pipe->times().add("gunzip_device_search", ms_search);
    h->pipe->times().add("gunzip_device_decode", ms_decode);
    h->pipe->times().add("gunzip_device_windows_resolve_crc", ms_resolve);
    h->pipe->times().add("gunzip_device_members_x1", (double)nf);
    h->pipe->times().add("fastq_device_parse_pack_host_clock", now_ms() - t1);
    h->n_reads = reads_done;
    std::vector<DevPiece> pcs;
    for (int f = 0; f < nf; f++) if (packed[f].n_seg) pcs.push_back(DevPiece{packed[f].d_bases, packed[f].d_seg_off, packed[f].n_seg, packed[f].n_bases});
    const double tc = now_ms();
    h->batches_started++;
    const int rc = h->pipe->count_batch_pieces(pcs.data(), pcs.size(), err);
    h->pipe->times().add("preprocess_device_total_host_clock", now_ms() - tc);
    if (rc) { drop(); return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err); }
    handled = true;
    const int rf = finish_counting(h);
    drop();
    return rf;
}

static int preprocess_impl(shk_handle *h, const uint8_t *fq1, size_t n1, const uint8_t *fq2, size_t n2) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Fresh) return fail(h, SHK_E_STATE, "preprocess: handle already used (Assembler.ts:92: one preprocess per handle)");
    if (!fq1) return fail(h, SHK_E_PARAM, "preprocess: file1 is required");
    h->post("preprocess:start");
    h->post_mode("start");
    h->post_loop_edge("loop:start");
    std::string err;
    const double t0 = now_ms();
    const size_t total = n1 + (fq2 ? n2 : 0);
    // ---- device-side parsing (fastq_gpu.hip): regular 4-line FASTQ is parsed, masked, segmented and
    // packed by streaming kernels.  Irregular input and every malformed record go to the host parser
    // below, which owns the error messages.
    // gzip (plain members: one thread per file; BGZF: block-parallel) is inflated once, for either parser
    // ---- .fastq.gz (the reference's real input: fastx_wasm.rs:53-70): a plain gzip member of some size is inflated ON THE
    // DEVICE — the compressed bytes are what crosses PCIe — and its text goes straight to the device parser; whatever the
    // device inflater does not take (several members, BGZF, binary data, a damaged stream) is read on the host below
    {
        const char *gd = getenv("SHK_GUNZIP_DEVICE");
        const char *fh = getenv("SHK_HOST_PARSER");
        if (!(gd && *gd == '0') && !(fh && *fh == '1')) {
            bool handled = false;
            const int rc = preprocess_device_gzip(h, fq1, n1, fq2, n2, total, handled);
            if (rc || handled) return rc;
        }
    }
    ByteVec st1, st2;
    const uint8_t *t1 = nullptr, *t2 = nullptr; size_t l1 = 0, l2 = 0;
    {
        const uint64_t mt0 = inflate_mt_members();
        int rc = maybe_inflate_pair(fq1, n1, fq2, n2, st1, st2, t1, l1, t2, l2, err);
        if (rc) return fail(h, rc == -3 ? SHK_E_PARSE : SHK_E_OOM, err);
        h->pipe->times().add("gunzip_host_clock", now_ms() - t0);
        h->pipe->times().add("gunzip_mt_members_x1", (double)(inflate_mt_members() - mt0));     // members the multi-threaded inflater took
    }
    const size_t text_total = l1 + (fq2 ? l2 : 0);
    const char *force_host = getenv("SHK_HOST_PARSER");
    // a single-batch text of some size: cut into pieces, piece i+1 uploaded while piece i is parsed (H2D is two thirds
    // of this entry point), all counted as one batch
    {
        const char *mv = getenv("SHK_FASTQ_PIPELINE_MIN");
        const size_t pipe_min = (mv && *mv) ? (size_t)strtoull(mv, nullptr, 10) : ((size_t)64 << 20);
        if (!(force_host && *force_host == '1') && text_total / 2 <= batch_bases() && text_total >= pipe_min) {
            bool handled = false;
            int rc = preprocess_device_pieces(h, t1, l1, fq2 ? t2 : nullptr, l2, n1, total, handled, true);
            if (rc || handled) return rc;
            // (not regular 4-line FASTQ from the first piece on: the host parser below; the single-shot device path
            // would find the same)
            force_host = "1";
        }
    }
    if (!(force_host && *force_host == '1') && text_total / 2 <= batch_bases()) {
        const double t1c = now_ms();
        GpuPacked gp;
        int rc = gpu_pack_fastq(t1, l1, fq2 ? t2 : nullptr, l2, h->k, h->min_qual, h->progress_every(), h->pipe->stream(), gp, err);
        if (rc < 0) { gpu_packed_free(gp); return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err); }
        if (rc == 0) {
            h->pipe->times().add("fastq_device_parse_pack_host_clock", now_ms() - t1c);
            h->pipe->times().add("fastq_h2d_text", gp.h2d_ms);
            h->pipe->times().add("fastq_device_kernels", gp.kernels_ms);
            const uint64_t every = h->progress_every();
            for (size_t j = 0; j < gp.progress_bytes.size(); j++) {
                const bool second = (gp.progress_bytes[j] >> 63) != 0;
                const uint64_t bytes = gp.progress_bytes[j] & ~(1ull << 63);
                const uint64_t pct = total ? (100 * ((second ? n1 : 0) + bytes)) / total : 100;
                h->post_mode(("loop:" + std::to_string(every * (j + 1)) + ":" + std::to_string(pct)).c_str());
            }
            h->n_reads = gp.n_reads;
            rc = run_counting(h, gp.d_bases, gp.d_seg_off, gp.n_seg, gp.n_bases);
            gpu_packed_free(gp);
            return rc;
        }
        gpu_packed_free(gp);                            // rc == 1: not regular -> host parser
    }
    // ---- texts of several batches: pieces cut at record boundaries, each through the device parser
    if (!(force_host && *force_host == '1') && text_total / 2 > batch_bases()) {
        bool handled = false;
        int rc = preprocess_device_pieces(h, t1, l1, fq2 ? t2 : nullptr, l2, n1, total, handled);
        if (rc || handled) return rc;
    }
    // ---- host parser (irregular framing, malformed records, inputs of several batches)
    PackedReads pr;
    size_t done_before = 0;
    auto prog = [&](uint64_t reads, uint64_t bytes, uint64_t) {
        const uint64_t pct = total ? (100 * (done_before + bytes)) / total : 100;
        h->post_mode(("loop:" + std::to_string(reads) + ":" + std::to_string(pct)).c_str());
    };
    int flush_rc = SHK_OK;
    auto flush = [&](PackedReads &p) -> int { flush_rc = flush_host_batch(h, p); return flush_rc ? -7 : 0; };
    if (flush_every_reads(h) || text_total / 2 > batch_bases()) h->pipe->expect_more_batches();
    int rc = pack_fastq(t1, l1, h->k, h->min_qual, pr, err, h->progress_every(), prog, flush_every_reads(h), batch_bases(), flush);
    if (!rc && fq2) {
        done_before = n1;
        rc = pack_fastq(t2, l2, h->k, h->min_qual, pr, err, h->progress_every(), prog, flush_every_reads(h), batch_bases(), flush);
    }
    if (rc == -7) return flush_rc;                       // the batch hand-over failed: its error is set
    if (rc) return fail(h, rc == -3 ? SHK_E_PARSE : (rc == -4 ? SHK_E_OOM : SHK_E_PARAM), err);
    h->n_reads = pr.n_reads;
    h->pipe->times().add("fastq_parse_pack_host_clock", now_ms() - t0);
    if (int rc2 = flush_host_batch(h, pr)) return rc2;
    return finish_counting(h);
}

static int push_reads_impl(shk_handle *h, const uint8_t *chunk, size_t n) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Fresh && h->st != St::Streaming) return fail(h, SHK_E_STATE, "push_reads: handle already preprocessed");
    if (h->st == St::Fresh) {
        h->post("preprocess:start"); h->post_mode("start"); h->post_loop_edge("loop:start");
        h->st = St::Streaming;
    }
    std::string err;
    auto prog = [&](uint64_t reads, uint64_t, uint64_t) { h->post_mode(("loop:" + std::to_string(reads)).c_str()); };
    int flush_rc = SHK_OK;
    auto flush = [&](PackedReads &p) -> int { flush_rc = flush_host_batch(h, p); return flush_rc ? -7 : 0; };
    h->pipe->expect_more_batches();                      // the total is unknown while chunks keep coming
    // a large chunk (whole records, like every chunk) is parsed on the device and counted as a batch of its
    // own; small chunks — and any chunk the device parser finds irregular — are packed on the host below
    {
        const char *force_host = getenv("SHK_HOST_PARSER");
        const char *mv = getenv("SHK_STREAM_DEVICE_MIN");
        const size_t dev_min = (mv && *mv) ? (size_t)strtoull(mv, nullptr, 10) : ((size_t)8 << 20);
        if (!(force_host && *force_host == '1') && n >= dev_min) {
            ByteVec st;
            const uint8_t *t = nullptr; size_t l = 0;
            int rc = maybe_inflate(chunk, n, st, t, l, err);
            if (rc) return fail(h, rc == -3 ? SHK_E_PARSE : SHK_E_OOM, err);
            if (l / 2 <= batch_bases()) {
                GpuPacked gp;
                rc = gpu_pack_fastq(t, l, nullptr, 0, h->k, h->min_qual, h->progress_every(), h->pipe->stream(), gp, err, h->stream_reads.n_reads);
                if (rc < 0) { gpu_packed_free(gp); return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err); }
                if (rc == 0) {
                    const uint64_t every = h->progress_every();
                    for (size_t j = 0; j < gp.progress_bytes.size(); j++) prog(every * (gp.first_mark + j + 1), 0, 0);
                    h->pipe->times().add("fastq_device_chunks_x1", 1.0);
                    const int rc2 = gp.n_seg ? count_one_batch(h, gp.d_bases, gp.d_seg_off, gp.n_seg, gp.n_bases) : SHK_OK;
                    h->stream_reads.n_reads += gp.n_reads; h->stream_reads.n_input_bases += gp.n_input_bases;
                    gpu_packed_free(gp);
                    return rc2;
                }
                gpu_packed_free(gp);                    // not regular 4-line FASTQ: the host parser decides
            }
        }
    }
    int rc = pack_fastq(chunk, n, h->k, h->min_qual, h->stream_reads, err, h->progress_every(), prog,
                        flush_every_reads(h), batch_bases(), flush);
    if (rc == -7) return flush_rc;
    if (rc) return fail(h, rc == -3 ? SHK_E_PARSE : (rc == -4 ? SHK_E_OOM : SHK_E_PARAM), err);
    return SHK_OK;
}

static int finish_reads_impl(shk_handle *h) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Streaming) return fail(h, SHK_E_STATE, "finish_reads: no reads pushed");
    h->n_reads = h->stream_reads.n_reads;
    int rc = flush_host_batch(h, h->stream_reads);
    h->stream_reads.clear();
    if (rc) return rc;
    return finish_counting(h);
}

static int preprocess_packed_device_impl(shk_handle *h, const void *d_bases, const void *d_seg_off, uint64_t n_seg,
                                 uint64_t n_bases, uint64_t n_reads) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Fresh) return fail(h, SHK_E_STATE, "preprocess: handle already used");
    if (!d_bases || !d_seg_off) return fail(h, SHK_E_PARAM, "null device pointer");
    h->post("preprocess:start"); h->post_mode("start"); h->post_loop_edge("loop:start");
    h->n_reads = n_reads;
    h->post_mode(("loop:" + std::to_string(n_reads) + ":100").c_str());
    return run_counting(h, (const uint32_t *)d_bases, (const uint32_t *)d_seg_off, n_seg, n_bases);
}

static int preprocess_packed_host_impl(shk_handle *h, const uint32_t *bases, const uint32_t *seg_off, uint64_t n_seg,
                                       uint64_t n_bases, uint64_t n_reads) {
    if (h->st != St::Fresh) return fail(h, SHK_E_STATE, "preprocess: handle already used");
    if (!bases || !seg_off) return fail(h, SHK_E_PARAM, "null host pointer");
    if (n_bases >= 0xFFFFFFFFull || n_seg >= 0xFFFFFFFFull) return fail(h, SHK_E_PARAM, "batch too large (>= 2^32 bases)");
    std::string err;
    struct Block { void *p = nullptr; size_t bytes = 0; ~Block() { if (p) device_pool_release(p, bytes); } } db, ds;
    db.bytes = (size_t)((n_bases + 15) / 16 + 1) * 4; ds.bytes = (size_t)(n_seg + 1) * 4;
    const size_t want_b = db.bytes, want_s = ds.bytes;
    db.p = device_pool_alloc(db.bytes); ds.p = device_pool_alloc(ds.bytes);
    if (!db.p || !ds.p) return fail(h, SHK_E_OOM, "preprocess: device memory for the packed reads");
    const double t0 = now_ms();
    void *st = h->pipe->stream();
    h->post("preprocess:start"); h->post_mode("start"); h->post_loop_edge("loop:start");
    h->n_reads = n_reads;
    h->post_mode(("loop:" + std::to_string(n_reads) + ":100").c_str());
    // upload and pass 1 overlap piece by piece (Pipeline::count_batch_host), then histogram / fit / filter as usual
    h->batches_started++;
    h->pipe->single_batch_resident(true);
    int rc = h->pipe->count_batch_host((uint32_t *)db.p, (uint32_t *)ds.p, bases, seg_off, n_seg, n_bases, err);
    if (rc) rc = fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err);
    else rc = finish_counting(h);
    h->pipe->single_batch_resident(false);
    { std::string e2; (void)device_stream_sync(st, e2); }      // the blocks go back to the pool idle, also after a failure
    h->pipe->times().add("h2d_packed_reads_MB", (double)(want_b + want_s) / 1e6);
    h->pipe->times().add("preprocess_from_host_total_host_clock", now_ms() - t0);
    return rc;
}

// ---- shard layer: the single-GPU preprocess cut at its two exchange points ----------------------
static uint32_t emit_threshold_of(const shk_handle *h) {
    return h->do_fit ? (h->min_count < 1u ? h->min_count : 1u) : h->min_count;
}

static int shard_partition_impl(shk_handle *h, const void *d_bases, const void *d_seg_off, uint64_t n_seg, uint64_t n_bases,
                        uint64_t n_reads, uint32_t n_partitions, uint64_t *part_records) {
    if (!h || !part_records) return SHK_E_PARAM;
    if (h->st != St::Fresh) return fail(h, SHK_E_STATE, "shard_partition: handle already used");
    h->post("preprocess:start"); h->post_mode("start"); h->post_loop_edge("loop:start");
    h->n_reads = n_reads;
    std::vector<uint64_t> pr;
    std::string err;
    int rc = h->pipe->shard_partition((const uint32_t *)d_bases, (const uint32_t *)d_seg_off, n_seg, n_bases, n_partitions, pr, err);
    if (rc) return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err);
    memcpy(part_records, pr.data(), (size_t)n_partitions * 8);
    h->post_mode(("loop:" + std::to_string(n_reads) + ":100").c_str());
    h->post_loop_edge("loop:end");
    h->st = St::Sharding;
    return SHK_OK;
}

uint32_t shk_shard_record_bytes(shk_handle *h) { return h && h->pipe ? h->pipe->rec_words() * 8u : 0u; }

static int shard_pack_impl(shk_handle *h, void *d_send, const uint64_t *base_records, uint32_t n_partitions) {
    if (!h || !base_records) return SHK_E_PARAM;
    if (h->st != St::Sharding) return fail(h, SHK_E_STATE, "shard_pack: call shard_partition first");
    std::string err;
    int rc = h->pipe->shard_pack(d_send, base_records, n_partitions, err);
    return rc ? fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err) : SHK_OK;
}

static int shard_count_impl(shk_handle *h, const void *d_recv, const uint64_t *run_off, const uint32_t *run_cnt,
                    uint32_t n_owned, uint32_t n_sources, uint64_t *histo500_local, uint64_t *n_instances_local,
                    const void *d_recv_w = nullptr /* weights of the records (deduplicated by their sources) */) {
    if (!h || !histo500_local) return SHK_E_PARAM;
    if (h->st != St::Sharding) return fail(h, SHK_E_STATE, "shard_count: call shard_partition first");
    std::string err;
    if (!h->do_bloom && h->chunk_size == 0) h->post("preprocess:bulk:sorting");
    int rc = h->pipe->shard_count(d_recv, d_recv_w, run_off, run_cnt, n_owned, n_sources, emit_threshold_of(h), histo500_local, err);
    if (rc) return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err);
    if (n_instances_local) *n_instances_local = h->pipe->total_instances();
    return SHK_OK;
}

static int shard_rows_impl(shk_handle *h, const uint64_t *histo500_global, const void **d_keys, const void **d_cnt,
                   uint64_t *n_rows, uint32_t *used_min_count) {
    if (!h || !histo500_global || !d_keys || !d_cnt || !n_rows) return SHK_E_PARAM;
    if (h->st != St::Sharding) return fail(h, SHK_E_STATE, "shard_rows: call shard_count first");
    memcpy(h->histo, histo500_global, sizeof h->histo);
    h->used_min_count = h->min_count; h->fit_ok = false;
    if (h->do_fit) {
        h->post_mode("fitting");
        uint32_t v = 0;
        if (spectrum_fit(h->histo, &v)) { h->used_min_count = v; h->fit_ok = true; }
    }
    h->post_mode("filtering");
    std::string err;
    int rc = h->pipe->shard_rows(h->used_min_count, d_keys, d_cnt, n_rows, err);
    if (rc) return fail(h, rc == -4 ? SHK_E_OOM : SHK_E_DEVICE, err);
    if (used_min_count) *used_min_count = h->used_min_count;
    return SHK_OK;
}

static int shard_set_solid_impl(shk_handle *h, const void *const *d_keys, const void *d_cnt, uint64_t n_rows,
                        uint64_t n_instances_global) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Sharding) return fail(h, SHK_E_STATE, "shard_set_solid: call shard_rows first");
    std::string err;
    int rc = h->pipe->shard_set_solid(d_keys, d_cnt, n_rows, h->histo, n_instances_global, err);
    if (rc) return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err);
    h->post("preprocess:saving");
    h->pre_json = preprocessing_json(h->pipe->n_solid(), h->histo, h->used_min_count);
    h->st = St::Preprocessed;
    h->post("preprocess:end");
    return SHK_OK;
}

const char *shk_get_preprocessing_info(shk_handle *h) {
    if (!h) return nullptr;
    if (h->st != St::Preprocessed && h->st != St::Assembled) { h->err = "get_preprocessing_info before preprocess"; return nullptr; }
    return h->pre_json.c_str();
}

static int assemble_impl(shk_handle *h) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Preprocessed) return fail(h, SHK_E_STATE, "assemble: preprocess first (and only once)");
    std::string err;
    const double t0 = now_ms();
    h->post("assembly:start");
    if (h->pipe->sharded_graph()) {
        // the graph is spread over the ranks of the communicator shk_shard_preprocess ran on: collective (csrc/shard_graph.h).
        // The three phases run interleaved across ranks; the states are posted in the reference's order.
        if (!h->shard_comm || !comm_is_live(h->shard_comm)) return fail(h, SHK_E_STATE, "assemble: the communicator of the sharded preprocess is gone (shk_comm_free before shk_assemble)");
        h->post("assembly:create_graph");
        std::vector<RawContig> contigs;
        if (h->pipe->n_solid_global() >= (1u << 20)) writer_prewarm(3000);
        int rc = h->pipe->shard_assemble(h->shard_comm, !h->no_deadend, !h->no_bubble, contigs, err);
        // (a failure the other ranks cannot know of has aborted the communicator already: Pipeline::shard_assemble)
        if (rc) return fail(h, rc == -4 ? SHK_E_OOM : SHK_E_DEVICE, err);
        h->post("assembly:correct_graph");
        h->post("assembly:collapse_graph");
        h->pipe->times().add("assemble_device_total_host_clock", now_ms() - t0);
        h->post("assembly:saving");
        const double t1 = now_ms();
        build_assembly_text(contigs, h->k, h->text);
        h->asm_json.swap(h->text.json);
        h->pipe->times().add("outputs_host_clock", now_ms() - t1);
        for (auto &kv : h->text.stage_ms) h->pipe->times().add(kv.first, kv.second);
        h->st = St::Assembled;
        h->post("assembly:end");
        return SHK_OK;
    }
    h->post("assembly:create_graph");
    int rc = h->pipe->build_graph(err);
    if (rc) return fail(h, rc == -4 ? SHK_E_OOM : SHK_E_DEVICE, err);
    h->post("assembly:correct_graph");
    rc = h->pipe->correct(!h->no_deadend, !h->no_bubble, err);
    if (rc) return fail(h, rc == -4 ? SHK_E_OOM : SHK_E_DEVICE, err);
    h->post("assembly:collapse_graph");
    if (h->pipe->n_solid() >= (1u << 20)) writer_prewarm(3000);     // megabases of output in about a millisecond
    std::vector<RawContig> contigs;
    const char *dev_json = nullptr; size_t dev_json_len = 0; uint64_t dev_nc = 0;
    TextArrival *arrival = nullptr;                    // megabases of contigs: their text is still crossing PCIe when the writer starts
    rc = h->pipe->collapse(contigs, err, &dev_json, &dev_json_len, &dev_nc, &arrival);
    if (rc) return fail(h, rc == -4 ? SHK_E_OOM : SHK_E_DEVICE, err);
    h->pipe->times().add("assemble_device_total_host_clock", now_ms() - t0);
    h->post("assembly:saving");
    if (dev_json) {                                    // a fragmented assembly: its text was written on the device (csrc/writer_gpu.h)
        h->asm_json_dev = dev_json;
        h->pipe->times().add("outputs_host_clock", 0.0);
        h->st = St::Assembled;
        h->post("assembly:end");
        return SHK_OK;
    }
    const double t1 = now_ms();
    struct ArrivalDone { TextArrival *a; ~ArrivalDone() { if (a) { std::string e; (void)a->finish(e); } } } arrival_done{arrival};   // (also when the writer throws)
    build_assembly_text(contigs, h->k, h->text, arrival);
    if (arrival) { arrival_done.a = nullptr; if (int rc2 = arrival->finish(err)) return fail(h, rc2 == -4 ? SHK_E_OOM : SHK_E_DEVICE, err); }
    h->asm_json.swap(h->text.json);
    h->pipe->times().add("outputs_host_clock", now_ms() - t1);
    if (arrival) h->pipe->times().add("outputs_with_arrival_x1", 1.0);
    for (auto &kv : h->text.stage_ms) h->pipe->times().add(kv.first, kv.second);
    h->st = St::Assembled;
    h->post("assembly:end");
    return SHK_OK;
}

const char *shk_get_assembly(shk_handle *h) {
    if (!h) return nullptr;
    if (h->st != St::Assembled) { h->err = "get_assembly before assemble"; return nullptr; }
    if (h->asm_json_dev) return h->asm_json_dev;
    return h->asm_json.empty() ? "" : (const char *)h->asm_json.data();
}

// ---- packer ------------------------------------------------------------------------------
int shk_pack_fastq(const uint8_t *fq, size_t n, uint32_t k, uint32_t min_qual, shk_packed *out, const char **errp) {
    static thread_local std::string msg;
    if (!out) return SHK_E_PARAM;
    PackedReads pr;
    int rc = pack_fastq(fq, n, k, min_qual, pr, msg);
    if (rc) { if (errp) *errp = msg.c_str(); return rc == -3 ? SHK_E_PARSE : (rc == -4 ? SHK_E_OOM : SHK_E_PARAM); }
    pr.finish();
    out->n_seg = pr.n_seg(); out->n_bases = pr.n_bases; out->n_reads = pr.n_reads; out->n_input_bases = pr.n_input_bases;
    out->bases = (uint32_t *)malloc(pr.bases.size() * 4);
    out->seg_off = (uint32_t *)malloc(pr.seg_off.size() * 4);
    if (!out->bases || !out->seg_off) { free(out->bases); free(out->seg_off); return SHK_E_OOM; }
    memcpy(out->bases, pr.bases.data(), pr.bases.size() * 4);
    memcpy(out->seg_off, pr.seg_off.data(), pr.seg_off.size() * 4);
    return SHK_OK;
}
void shk_packed_free(shk_packed *p) { if (p) { free(p->bases); free(p->seg_off); p->bases = nullptr; p->seg_off = nullptr; } }

// ---- stage inspection -----------------------------------------------------------------------
uint32_t shk_key_words(shk_handle *h) { return h ? (2 * h->k + 63) / 64 : 0; }
uint64_t shk_total_instances(shk_handle *h) { return h && h->pipe ? h->pipe->total_instances() : 0; }
uint64_t shk_n_distinct(shk_handle *h) { return h && h->pipe ? h->pipe->n_distinct() : 0; }
uint64_t shk_n_solid(shk_handle *h) { return h && h->pipe ? h->pipe->n_solid() : 0; }
uint32_t shk_used_min_count(shk_handle *h) { return h ? h->used_min_count : 0; }

static int get_distinct_impl(shk_handle *h, uint64_t *keys, uint32_t *counts, uint64_t cap) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Preprocessed) return fail(h, SHK_E_STATE, "get_distinct: only between preprocess and assemble");
    std::string err; int rc = h->pipe->get_distinct(keys, counts, cap, err);
    return rc ? fail(h, rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE, err) : SHK_OK;
}
static int get_solid_impl(shk_handle *h, uint64_t *keys, uint32_t *counts, uint64_t cap) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Preprocessed && h->st != St::Assembled) return fail(h, SHK_E_STATE, "get_solid before preprocess");
    std::string err; int rc = h->pipe->get_solid(keys, counts, cap, err);
    return rc ? fail(h, rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE, err) : SHK_OK;
}
int shk_get_histo(shk_handle *h, uint64_t *histo500) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Preprocessed && h->st != St::Assembled) return fail(h, SHK_E_STATE, "get_histo before preprocess");
    memcpy(histo500, h->histo, sizeof h->histo);
    return SHK_OK;
}
static int get_adjacency_impl(shk_handle *h, uint8_t *adj_initial, uint8_t *adj_final, uint8_t *alive, uint64_t cap) {
    if (!h) return SHK_E_PARAM;
    if (h->st != St::Assembled) return fail(h, SHK_E_STATE, "get_adjacency before assemble");
    std::string err; int rc = h->pipe->get_adjacency(adj_initial, adj_final, alive, cap, err);
    return rc ? fail(h, rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE, err) : SHK_OK;
}
const char *shk_get_timings(shk_handle *h) {
    if (!h || !h->pipe) return "{}";
    DevGuard g(h->pipe->device());
    std::string j = "{";
    bool first = true;
    for (auto &kv : h->pipe->times().ms) {
        if (!first) j += ",";
        first = false;
        char buf[64]; snprintf(buf, sizeof buf, "%.6f", kv.second);
        j += "\"" + kv.first + "\":" + buf;
    }
    {   // not a time: the most device memory this handle held at once, in bytes (Assembler.ts:69-71,137 reports peak memory)
        char buf[96]; snprintf(buf, sizeof buf, "%s\"peak_device_bytes\":%llu,\"device_bytes_now\":%llu", first ? "" : ",",
                               (unsigned long long)mem_acct_peak(h->mem), (unsigned long long)mem_acct_current(h->mem));
        j += buf;
    }
    j += "}";
    h->timings_json.swap(j);
    return h->timings_json.c_str();
}

uint64_t shk_peak_device_bytes(shk_handle *h) { return h ? mem_acct_peak(h->mem) : 0; }
void shk_host_mem_counter(const int64_t *deltas, size_t n, uint64_t *peak, uint64_t *current) { mem_acct_replay(deltas, n, peak, current); }

// ---- collectives inside the library (shard_comm.hip: RCCL) ---------------------------------------------
struct shk_comm { ShardComm *c = nullptr; };
static thread_local std::string g_comm_err;

static int shard_preprocess_impl(shk_handle *h, shk_comm *cm, const void *d_bases, const void *d_seg_off, uint64_t n_seg,
                                 uint64_t n_bases, uint64_t n_reads, uint32_t n_partitions) {
    if (!cm || !cm->c) return fail(h, SHK_E_PARAM, "shard_preprocess: null communicator");
    if (h->st != St::Fresh) return fail(h, SHK_E_STATE, "shard_preprocess: handle already used");
    if (comm_device(cm->c) != h->pipe->device()) return fail(h, SHK_E_PARAM, "shard_preprocess: communicator and handle live on different devices");
    ShardComm *c = cm->c;
    const uint32_t world = (uint32_t)comm_world(c), rank = (uint32_t)comm_rank(c);
    const uint32_t W = (2 * h->k + 63) / 64;
    void *st = h->pipe->stream();
    std::string err;
    auto cfail = [&](int rc) { return fail(h, rc == -4 ? SHK_E_OOM : (rc == -1 ? SHK_E_PARAM : SHK_E_DEVICE), err); };
    const double t0 = now_ms();
    // SHK_STAGE_LOG=1: after every step the stream is drained and the step's name goes to stderr (which step a fault belongs to)
    const bool stage_log = getenv("SHK_STAGE_LOG") != nullptr;
    auto stage = [&](const char *what) {
        if (!stage_log) return;
        std::string e2; const int r2 = device_stream_sync(st, e2);
        fprintf(stderr, "[shard_preprocess rank %u] %s done%s\n", rank, what, r2 ? " (stream error)" : ""); fflush(stderr);
    };
    // A failure on ONE rank (device memory, a slice that overflows, a partition beyond 2^32 records: all depend on that
    // rank's share of the reads) must not leave the others blocked in the next collective: every local step's result
    // travels with the next small collective — as an extra element of one that exists anyway, or as a one-word
    // all-reduce in front of the two big exchanges — and all ranks leave together.  A collective that fails itself
    // marks the communicator broken (shk_comm_free then aborts it: peers fail fast).
    auto peer_failed = [&](const char *stage) {
        return fail(h, SHK_E_DEVICE, std::string("shard_preprocess: another rank failed during ") + stage + " (this rank's state is intact up to there; free the handle)");
    };
    // SHK_FAULT_INJECT=<step> (pass1 | pack | count | rows | keep | alloc) makes that local step of THIS process fail: the tests
    // set it on one rank to see every rank leave with an error instead of hanging.  It never changes a result.
    const char *inject = getenv("SHK_FAULT_INJECT");
    auto injected = [&](const char *step) -> int {
        return (inject && !strcmp(inject, step)) ? fail(h, SHK_E_INTERNAL, std::string("injected fault (SHK_FAULT_INJECT=") + step + ")") : SHK_OK;
    };
    auto agree = [&](int local_rc, const char *stage) -> int {
        uint64_t f = local_rc ? 1u : 0u;
        std::string e2;
        if (int rc = comm_allreduce_host_u64(c, &f, 1, st, e2)) { if (local_rc) return local_rc; err = e2; return cfail(rc); }
        if (local_rc) return local_rc;
        return f ? peer_failed(stage) : SHK_OK;
    };
    // ---- the partition count must be the same everywhere: from the global instance count
    if (n_partitions == 0) {
        uint64_t inst = n_bases > n_seg * (uint64_t)(h->k - 1) ? n_bases - n_seg * (uint64_t)(h->k - 1) : 0;
        if (int rc = comm_allreduce_host_u64(c, &inst, 1, st, err)) return cfail(rc);
        n_partitions = choose_partitions(inst, world, W == 1 ? 100000 : 40000);
    }
    if (n_partitions < world) return fail(h, SHK_E_PARAM, "shard_preprocess: fewer partitions than ranks");
    const uint32_t P = n_partitions;
    // ---- pass 1 on this rank's reads, then the records are deduplicated HERE, before they cross the fabric (at 100x a
    // super-k-mer record recurs ~50 times; count_part.h: k_dedupe_partitions): distinct records + u32 weights travel.
    // SHK_SHARD_DEDUPE: 0 = never, 1 = always, default = when the distinct records of all ranks are at most half of the raw
    // ones (error-rich reads do not deduplicate, and their partitions need the k-mer-level repartition, which counts
    // unweighted records).  Bloom mode counts raw records too.  The decision is taken from the gathered rows: same everywhere.
    std::vector<uint64_t> part(2 * (size_t)P + 2, 0);      // [0,P) records to send, [P,2P) raw records, [2P] = this rank failed, [2P+1] = 0 raw / 1 auto / 2 always
    int rc_p1 = shard_partition_impl(h, d_bases, d_seg_off, n_seg, n_bases, n_reads, P, part.data());
    if (!rc_p1) rc_p1 = injected("pass1");
    if (!rc_p1) {
        memcpy(&part[P], &part[0], (size_t)P * 8);
        const char *dd = getenv("SHK_SHARD_DEDUPE");
        const uint64_t mode = (dd && *dd == '0') || h->do_bloom ? 0u : ((dd && *dd == '1') ? 2u : 1u);
        if (mode) {
            std::vector<uint64_t> pr(part.begin(), part.begin() + P);
            std::string e2;
            if (int r2 = h->pipe->shard_dedupe(pr, e2)) rc_p1 = fail(h, r2 == -4 ? SHK_E_OOM : SHK_E_DEVICE, e2);
            else memcpy(&part[0], pr.data(), (size_t)P * 8);
        }
        part[2 * (size_t)P + 1] = mode;
    }
    part[2 * (size_t)P] = rc_p1 ? 1u : 0u;
    stage("pass 1 + dedupe");
    // ---- the size exchange: every rank learns what every rank holds per partition
    const size_t ROW = 2 * (size_t)P + 2;
    std::vector<uint64_t> all_raw((size_t)world * ROW), all((size_t)world * P), raw_counts((size_t)world * P);
    if (int rc = comm_allgather_host_u64(c, part.data(), ROW, all_raw.data(), st, err)) { if (rc_p1) return rc_p1; return cfail(rc); }
    if (rc_p1) return rc_p1;
    bool weighted = true, forced = false;
    uint64_t sum_dd = 0, sum_raw = 0;
    for (uint32_t r = 0; r < world; r++) {
        const uint64_t *row = &all_raw[(size_t)r * ROW];
        if (row[2 * (size_t)P]) return peer_failed("pass 1");
        if (row[2 * (size_t)P + 1] == 0) weighted = false;
        if (row[2 * (size_t)P + 1] == 2) forced = true;
        for (uint32_t p = 0; p < P; p++) { sum_dd += row[p]; sum_raw += row[P + p]; }
        memcpy(&raw_counts[(size_t)r * P], row + P, (size_t)P * 8);
    }
    if (weighted && !forced && sum_dd * 2 > sum_raw) weighted = false;
    for (uint32_t r = 0; r < world; r++) memcpy(&all[(size_t)r * P], &all_raw[(size_t)r * ROW + (weighted ? 0 : P)], (size_t)P * 8);
    if (!weighted) h->pipe->shard_drop_dedup();
    h->pipe->times().add("shard_records_deduplicated_x1", weighted ? 1.0 : 0.0);
    ExchangePlan plan;
    const uint64_t rec_bytes = (uint64_t)h->pipe->rec_words() * 8u;
    uint64_t n_send = 0, n_recv = 0;
    // ---- pack (destination-major) and exchange
    struct Block { void *p = nullptr; size_t bytes = 0; ~Block() { if (p) device_pool_release(p, bytes); } } send, recv, send_w, recv_w, gk[8], gc;
    // (declared after the blocks, so it runs before they go back to the pool: on every way out — errors included —
    // the stream is drained first; the pool has no stream-ordering bookkeeping)
    struct DrainOnExit { ShardComm *c; void *st; ~DrainOnExit() { std::string e; (void)comm_stream_wait(c, st, e); } } drain{c, st};
    {
        int rc_pack = SHK_OK;
        if (int rc = plan_exchange(all.data(), world, P, rank, plan, err)) rc_pack = cfail(rc);
        if (!rc_pack && weighted) {
            // (a partition's record index — and a record's multiplicity — are 32 bits wide in pass 2: the limit is on the RAW records)
            for (uint32_t p = rank; p < P && !rc_pack; p += world) {
                uint64_t t = 0;
                for (uint32_t r = 0; r < world; r++) t += raw_counts[(size_t)r * P + p];
                if (t > 0xFFFFFFF0ull) rc_pack = fail(h, SHK_E_PARAM, "a partition holds more than 2^32 records");
            }
        }
        if (!rc_pack) {
            for (uint32_t r = 0; r < world; r++) { n_send += plan.send_counts[r]; n_recv += plan.recv_counts[r]; }
            send.bytes = (size_t)(n_send * rec_bytes + 64); send.p = device_pool_alloc(send.bytes);
            recv.bytes = (size_t)(n_recv * rec_bytes + 64); recv.p = device_pool_alloc(recv.bytes);
            if (weighted) {
                send_w.bytes = (size_t)(n_send * 4 + 64); send_w.p = device_pool_alloc(send_w.bytes);
                recv_w.bytes = (size_t)(n_recv * 4 + 64); recv_w.p = device_pool_alloc(recv_w.bytes);
            }
            if (!send.p || !recv.p || (weighted && (!send_w.p || !recv_w.p))) rc_pack = fail(h, SHK_E_OOM, "shard_preprocess: device memory for the record exchange");
        }
        if (!rc_pack && weighted) {
            std::string e2;
            if (int r2 = h->pipe->shard_pack_dedup(send.p, send_w.p, plan.base.data(), P, e2)) rc_pack = fail(h, r2 == -4 ? SHK_E_OOM : (r2 == -1 ? SHK_E_PARAM : SHK_E_DEVICE), e2);
        } else if (!rc_pack) rc_pack = shard_pack_impl(h, send.p, plan.base.data(), P);
        if (!rc_pack) rc_pack = injected("pack");
        stage("pack");
        if (int rc = agree(rc_pack, "the packing of the records")) return rc;
    }
    {
        std::vector<uint64_t> so(world), sb(world), ro(world), rb(world);
        uint64_t a = 0, b = 0;
        for (uint32_t r = 0; r < world; r++) {
            so[r] = a; sb[r] = plan.send_counts[r] * rec_bytes; a += sb[r];
            ro[r] = b; rb[r] = plan.recv_counts[r] * rec_bytes; b += rb[r];
        }
        const double tx = now_ms();
        if (int rc = comm_alltoallv(c, send.p, so.data(), sb.data(), recv.p, ro.data(), rb.data(), st, err)) return cfail(rc);
        if (weighted) {
            // the weights: same element offsets as the records, 4 bytes each
            for (uint32_t r = 0; r < world; r++) { so[r] = so[r] / rec_bytes * 4; sb[r] = plan.send_counts[r] * 4; ro[r] = ro[r] / rec_bytes * 4; rb[r] = plan.recv_counts[r] * 4; }
            if (int rc = comm_alltoallv(c, send_w.p, so.data(), sb.data(), recv_w.p, ro.data(), rb.data(), st, err, 4)) return cfail(rc);
        }
        if (int rc = comm_stream_wait(c, st, err)) { comm_mark_broken(c); return cfail(rc); }
        h->pipe->times().add("shard_exchange_host_clock", now_ms() - tx);
        h->pipe->times().add("shard_exchange_sent_MB", (double)(n_send * (rec_bytes + (weighted ? 4 : 0))) / 1e6);
    }
    stage("exchange");
    if (send_w.p) { device_pool_release(send_w.p, send_w.bytes); send_w.p = nullptr; }
    device_pool_release(send.p, send.bytes); send.p = nullptr;
    // ---- pass 2 over the owned partitions, then the global histogram ([501] = ranks that failed)
    // ([500] k-mer instances counted, [501] ranks that failed, [502] k-mer instances this rank's READS hold: after the
    // all-reduce [500] must equal [502] — a record exchange that lost or duplicated data cannot pass unnoticed)
    uint64_t red[SHK_HISTO_BINS + 3] = {0};
    const uint64_t inst_in_reads = n_bases > n_seg * (uint64_t)(h->k - 1) ? n_bases - n_seg * (uint64_t)(h->k - 1) : 0;
    int rc_cnt = shard_count_impl(h, recv.p, plan.run_off.data(), plan.run_cnt.data(), (uint32_t)plan.owned.size(), world, red,
                                  &red[SHK_HISTO_BINS], weighted ? recv_w.p : nullptr);
    stage("pass 2");
    if (!rc_cnt) rc_cnt = injected("count");
    if (rc_cnt) memset(red, 0, sizeof red);
    red[SHK_HISTO_BINS + 1] = rc_cnt ? 1u : 0u;
    red[SHK_HISTO_BINS + 2] = inst_in_reads;
    if (int rc = comm_allreduce_host_u64(c, red, SHK_HISTO_BINS + 3, st, err)) { if (rc_cnt) return rc_cnt; return cfail(rc); }
    if (rc_cnt) return rc_cnt;
    if (red[SHK_HISTO_BINS + 1]) return peer_failed("pass 2");
    if (red[SHK_HISTO_BINS] != red[SHK_HISTO_BINS + 2])
        return fail(h, SHK_E_INTERNAL, "shard_preprocess: the ranks counted " + std::to_string(red[SHK_HISTO_BINS]) + " k-mer instances, their reads hold " +
                                       std::to_string(red[SHK_HISTO_BINS + 2]) + ": the record exchange lost or duplicated data");
    // ---- fit / filter (identical on every rank), local solid rows
    const void *keys[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; const void *cnt = nullptr;
    uint64_t n_local = 0; uint32_t used = 0;
    int rc_rows = shard_rows_impl(h, red, keys, &cnt, &n_local, &used);
    stage("filter");
    if (stage_log && !rc_rows && n_local && n_local < (1u << 24)) {          // (debug aid: are the rank's solid k-mers distinct?)
        std::vector<std::vector<uint64_t>> kw(W, std::vector<uint64_t>(n_local));
        bool ok = true;
        for (uint32_t j = 0; j < W && ok; j++) { void *dp = const_cast<void *>(keys[j]); std::string e3; ok = device_download(kw[j].data(), dp, n_local * 8, e3) == 0; }
        if (ok) {
            std::vector<uint32_t> idx(n_local);
            for (uint32_t i = 0; i < n_local; i++) idx[i] = i;
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { for (int j = (int)W - 1; j >= 0; j--) if (kw[j][a] != kw[j][b]) return kw[j][a] < kw[j][b]; return false; });
            uint64_t dup = 0;
            for (uint32_t i = 1; i < n_local; i++) {
                bool eq = true; for (uint32_t j = 0; j < W; j++) eq = eq && kw[j][idx[i]] == kw[j][idx[i - 1]];
                if (eq && dup < 4) fprintf(stderr, "[shard_preprocess rank %u]   duplicate k-mer %016llx at rows %u and %u\n", rank, (unsigned long long)kw[0][idx[i]], idx[i - 1], idx[i]);
                dup += eq;
            }
            fprintf(stderr, "[shard_preprocess rank %u] %llu local solid k-mers, %llu duplicates\n", rank, (unsigned long long)n_local, (unsigned long long)dup); fflush(stderr);
        }
    }
    if (!rc_rows) rc_rows = injected("rows");
    // ---- all-gather of the solid rows
    std::vector<uint64_t> counts2((size_t)world * 2), counts(world);
    const uint64_t mine2[2] = {rc_rows ? 0u : n_local, rc_rows ? 1u : 0u};
    if (int rc = comm_allgather_host_u64(c, mine2, 2, counts2.data(), st, err)) { if (rc_rows) return rc_rows; return cfail(rc); }
    if (rc_rows) return rc_rows;
    for (uint32_t r = 0; r < world; r++) { if (counts2[2 * r + 1]) return peer_failed("the filter"); counts[r] = counts2[2 * r]; }
    // ---- the graph stays sharded (default): every rank keeps its own rows and shk_assemble() runs collectively over this
    // communicator (csrc/shard_graph.h).  SHK_SHARD_GRAPH=0: round 2's path — gather the solid set, assemble on every rank.
    {
        const char *sg = getenv("SHK_SHARD_GRAPH");
        if (!(sg && *sg == '0')) {
            int rc_keep = SHK_OK;
            { std::string e2; const int r2 = h->pipe->shard_keep_local(world, rank, P, counts.data(), red, red[SHK_HISTO_BINS], e2);
              if (r2) rc_keep = fail(h, r2 == -1 ? SHK_E_PARAM : SHK_E_DEVICE, e2); }
            if (!rc_keep) rc_keep = injected("keep");
            if (int rc = agree(rc_keep, "the hand-over to the sharded assembly")) return rc;
            h->shard_comm = c;
            h->post("preprocess:saving");
            h->pre_json = preprocessing_json(h->pipe->n_solid_global(), h->histo, h->used_min_count);
            h->st = St::Preprocessed;
            h->post("preprocess:end");
            h->pipe->times().add("shard_preprocess_host_clock", now_ms() - t0);
            return SHK_OK;
        }
    }
    uint64_t n_total = 0;
    std::vector<uint64_t> off8(world), len8(world), off4(world), len4(world);
    for (uint32_t r = 0; r < world; r++) { off8[r] = n_total * 8; len8[r] = counts[r] * 8; off4[r] = n_total * 4; len4[r] = counts[r] * 4; n_total += counts[r]; }
    {
        int rc_alloc = SHK_OK;
        for (uint32_t j = 0; j < W && !rc_alloc; j++) {
            gk[j].bytes = (size_t)(n_total * 8 + 64); gk[j].p = device_pool_alloc(gk[j].bytes);
            if (!gk[j].p) rc_alloc = fail(h, SHK_E_OOM, "shard_preprocess: device memory for the solid set");
        }
        if (!rc_alloc) {
            gc.bytes = (size_t)(n_total * 4 + 64); gc.p = device_pool_alloc(gc.bytes);
            if (!gc.p) rc_alloc = fail(h, SHK_E_OOM, "shard_preprocess: device memory for the solid set");
        }
        if (!rc_alloc) rc_alloc = injected("alloc");
        if (int rc = agree(rc_alloc, "the allocation of the solid set")) return rc;
    }
    for (uint32_t j = 0; j < W; j++)
        if (int rc = comm_allgatherv(c, keys[j], gk[j].p, off8.data(), len8.data(), st, err)) return cfail(rc);
    if (int rc = comm_allgatherv(c, cnt, gc.p, off4.data(), len4.data(), st, err)) return cfail(rc);
    if (int rc = comm_stream_wait(c, st, err)) { comm_mark_broken(c); return cfail(rc); }
    const void *kp[8] = {gk[0].p, gk[1].p, gk[2].p, gk[3].p, gk[4].p, gk[5].p, gk[6].p, gk[7].p};
    if (int rc = shard_set_solid_impl(h, kp, gc.p, n_total, red[SHK_HISTO_BINS])) return rc;
    h->pipe->times().add("shard_preprocess_host_clock", now_ms() - t0);
    return SHK_OK;
}

int shk_comm_unique_id(uint8_t id[SHK_UNIQUE_ID_BYTES]) {
    try { return comm_unique_id(id, g_comm_err) == 0 ? SHK_OK : SHK_E_DEVICE; }
    catch (...) { g_comm_err = "unexpected exception"; return SHK_E_INTERNAL; }
}
shk_comm *shk_comm_init(const uint8_t id[SHK_UNIQUE_ID_BYTES], int rank, int world) {
    try {
        if (!id) { g_comm_err = "null id"; return nullptr; }
        ShardComm *c = comm_create(id, rank, world, g_comm_err);
        if (!c) return nullptr;
        shk_comm *w = new (std::nothrow) shk_comm();
        if (!w) { comm_destroy(c); g_comm_err = "out of host memory"; return nullptr; }
        w->c = c;
        { std::lock_guard<std::mutex> lk(g_live_mu); live_comms().insert(c); }
        return w;
    } catch (...) { g_comm_err = "unexpected exception"; return nullptr; }
}
const char *shk_comm_error(void) { return g_comm_err.c_str(); }
int shk_comm_rank(const shk_comm *c) { return c ? comm_rank(c->c) : 0; }
int shk_comm_world(const shk_comm *c) { return c ? comm_world(c->c) : 0; }
void shk_comm_free(shk_comm *c) {
    if (!c) return;
    DevGuard g(comm_device(c->c));
    { std::lock_guard<std::mutex> lk(g_live_mu); live_comms().erase(c->c); }
    comm_destroy(c->c);
    delete c;
}
int shk_shard_preprocess(shk_handle *h, shk_comm *c, const void *d_bases, const void *d_seg_off, uint64_t n_seg,
                         uint64_t n_bases, uint64_t n_reads, uint32_t n_partitions) {
    return guarded(h, Poison::AfterFirstBatch, [&] {
        const int rc = shard_preprocess_impl(h, c, d_bases, d_seg_off, n_seg, n_bases, n_reads, n_partitions);
        // a collective that failed on this rank alone: abort now, so that the peers' waits end (shard_comm.h: comm_stream_wait)
        if (rc != SHK_OK && c && c->c && comm_broken(c->c)) comm_abort_now(c->c);
        return rc;
    });
}
int shk_plan_exchange(const uint64_t *part_records_all, uint32_t world, uint32_t n_partitions, uint32_t rank,
                      uint64_t *base, uint64_t *send_counts, uint64_t *recv_counts, uint64_t *run_off, uint32_t *run_cnt) {
    try {
        ExchangePlan pl; std::string err;
        if (!base || !send_counts || !recv_counts || !run_off || !run_cnt) return SHK_E_PARAM;
        if (plan_exchange(part_records_all, world, n_partitions, rank, pl, err)) return SHK_E_PARAM;
        memcpy(base, pl.base.data(), pl.base.size() * 8);
        memcpy(send_counts, pl.send_counts.data(), (size_t)world * 8);
        memcpy(recv_counts, pl.recv_counts.data(), (size_t)world * 8);
        memcpy(run_off, pl.run_off.data(), pl.run_off.size() * 8);
        memcpy(run_cnt, pl.run_cnt.data(), pl.run_cnt.size() * 4);
        return SHK_OK;
    } catch (...) { return SHK_E_OOM; }
}
uint32_t shk_choose_partitions(uint64_t total_instances_ub, uint32_t world, uint32_t key_words) {
    return choose_partitions(total_instances_ub, world ? world : 1, key_words <= 1 ? 100000 : 40000);
}

// ---- the guarded entry points (device guard, no exception across the ABI, failed-handle state) ----
int shk_preprocess(shk_handle *h, const uint8_t *fq1, size_t n1, const uint8_t *fq2, size_t n2) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return preprocess_impl(h, fq1, n1, fq2, n2); });
}
int shk_push_reads(shk_handle *h, const uint8_t *chunk, size_t n) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return push_reads_impl(h, chunk, n); });
}
int shk_finish_reads(shk_handle *h) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return finish_reads_impl(h); });
}
int shk_preprocess_packed_device(shk_handle *h, const void *d_bases, const void *d_seg_off, uint64_t n_seg,
                                 uint64_t n_bases, uint64_t n_reads) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return preprocess_packed_device_impl(h, d_bases, d_seg_off, n_seg, n_bases, n_reads); });
}
int shk_preprocess_packed_host(shk_handle *h, const uint32_t *bases, const uint32_t *seg_off, uint64_t n_seg, uint64_t n_bases,
                               uint64_t n_reads) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return preprocess_packed_host_impl(h, bases, seg_off, n_seg, n_bases, n_reads); });
}
int shk_shard_partition(shk_handle *h, const void *d_bases, const void *d_seg_off, uint64_t n_seg, uint64_t n_bases,
                        uint64_t n_reads, uint32_t n_partitions, uint64_t *part_records) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return shard_partition_impl(h, d_bases, d_seg_off, n_seg, n_bases, n_reads, n_partitions, part_records); });
}
int shk_shard_pack(shk_handle *h, void *d_send, const uint64_t *base_records, uint32_t n_partitions) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return shard_pack_impl(h, d_send, base_records, n_partitions); });
}
int shk_shard_count(shk_handle *h, const void *d_recv, const uint64_t *run_off, const uint32_t *run_cnt,
                    uint32_t n_owned, uint32_t n_sources, uint64_t *histo500_local, uint64_t *n_instances_local) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return shard_count_impl(h, d_recv, run_off, run_cnt, n_owned, n_sources, histo500_local, n_instances_local); });
}
int shk_shard_rows(shk_handle *h, const uint64_t *histo500_global, const void **d_keys, const void **d_cnt,
                   uint64_t *n_rows, uint32_t *used_min_count) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return shard_rows_impl(h, histo500_global, d_keys, d_cnt, n_rows, used_min_count); });
}
int shk_shard_set_solid(shk_handle *h, const void *const *d_keys, const void *d_cnt, uint64_t n_rows,
                        uint64_t n_instances_global) {
    return guarded(h, Poison::AfterFirstBatch, [&] { return shard_set_solid_impl(h, d_keys, d_cnt, n_rows, n_instances_global); });
}
int shk_assemble(shk_handle *h) {
    return guarded(h, Poison::Always, [&] { return assemble_impl(h); });
}
int shk_get_distinct(shk_handle *h, uint64_t *keys, uint32_t *counts, uint64_t cap) {
    return guarded(h, Poison::Never, [&] { return get_distinct_impl(h, keys, counts, cap); });
}
int shk_get_solid(shk_handle *h, uint64_t *keys, uint32_t *counts, uint64_t cap) {
    return guarded(h, Poison::Never, [&] { return get_solid_impl(h, keys, counts, cap); });
}
int shk_get_adjacency(shk_handle *h, uint8_t *adj_initial, uint8_t *adj_final, uint8_t *alive, uint64_t cap) {
    return guarded(h, Poison::Never, [&] { return get_adjacency_impl(h, adj_initial, adj_final, alive, cap); });
}

// ---- host-only self tests --------------------------------------------------------------------
int shk_host_canonical(const char *seq, uint32_t k, uint64_t *out_words, int *orient) {
    if (!seq || !out_words || (k & 1u) == 0 || k > SHK_K_MAX) return SHK_E_PARAM;
    return host_canonical(seq, k, out_words, orient) == 0 ? SHK_OK : SHK_E_PARAM;
}
uint64_t shk_host_nthash(const char *seq, uint32_t k) { return host_nthash(seq, k); }
int shk_host_fit(const uint64_t *histo500, uint32_t *used) { return spectrum_fit(histo500, used) ? 1 : 0; }
char *shk_host_assembly_json(const char *seqs, const uint64_t *offsets, const uint64_t *kc, uint64_t n_contigs, uint32_t k) {
    try {
        if ((!seqs && n_contigs) || !offsets || (!kc && n_contigs)) return nullptr;
        std::vector<RawContig> contigs((size_t)n_contigs);
        for (uint64_t i = 0; i < n_contigs; i++) {
            if (offsets[i + 1] < offsets[i] + k) return nullptr;            // a unitig spells at least one k-mer
            contigs[i].ext = seqs + offsets[i]; contigs[i].ext_n = (size_t)(offsets[i + 1] - offsets[i]); contigs[i].kc = kc[i];
        }
        AssemblyText text;
        build_assembly_text(contigs, k, text);
        char *out = (char *)malloc(text.json.size());        // (the JSON carries its terminator)
        if (!out) return nullptr;
        memcpy(out, text.json.data(), text.json.size());
        return out;
    } catch (...) { return nullptr; }
}
// the same writer on text that ARRIVES while it works (the device path hands it text that is still crossing PCIe): a thread
// releases the bytes piece by piece into a buffer that starts out as garbage, the contigs carry their ends as the device
// path's do — the JSON must be the one shk_host_assembly_json gives (tests; no GPU)
char *shk_host_assembly_json_arriving(const char *seqs, const uint64_t *offsets, const uint64_t *kc, uint64_t n_contigs, uint32_t k,
                                      uint64_t piece_bytes, uint32_t delay_us) {
    try {
        if ((!seqs && n_contigs) || !offsets || (!kc && n_contigs) || !piece_bytes) return nullptr;
        const size_t total = (size_t)offsets[n_contigs];
        struct Fake : TextArrival {
            std::vector<char> buf; std::atomic<size_t> ready{0};
            const char *base() const override { return buf.data(); }
            size_t total() const override { return buf.size(); }
            void wait_range(size_t, size_t end) override { if (end > buf.size()) end = buf.size(); while (ready.load(std::memory_order_acquire) < end) std::this_thread::yield(); }
            int finish(std::string &) override { wait_all(); return 0; }
        } fake;
        fake.buf.assign(total, '#');                       // (what has not arrived is not sequence)
        const uint32_t E = std::max<uint32_t>(k, 32);
        std::vector<char> ends((size_t)n_contigs * 2 * E, 0);
        std::vector<RawContig> contigs((size_t)n_contigs);
        for (uint64_t i = 0; i < n_contigs; i++) {
            if (offsets[i + 1] < offsets[i] + k) return nullptr;
            const size_t len = (size_t)(offsets[i + 1] - offsets[i]), m = std::min<size_t>(E, len);
            contigs[i].ext = fake.buf.data() + offsets[i]; contigs[i].ext_n = len; contigs[i].kc = kc[i];
            memcpy(&ends[(size_t)i * 2 * E], seqs + offsets[i], m);
            memcpy(&ends[(size_t)i * 2 * E + E], seqs + offsets[i] + len - m, m);
            contigs[i].head = &ends[(size_t)i * 2 * E]; contigs[i].tail = contigs[i].head + E; contigs[i].ends_n = (uint32_t)m;
        }
        std::thread feeder([&] {
            for (size_t o = 0; o < total; o += (size_t)piece_bytes) {
                if (delay_us) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
                const size_t m = std::min<size_t>((size_t)piece_bytes, total - o);
                memcpy(fake.buf.data() + o, seqs + o, m);
                fake.ready.store(o + m, std::memory_order_release);
            }
        });
        struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join{feeder};
        AssemblyText text;
        build_assembly_text(contigs, k, text, total ? &fake : nullptr);
        char *out = (char *)malloc(text.json.size());
        if (!out) return nullptr;
        memcpy(out, text.json.data(), text.json.size());
        return out;
    } catch (...) { return nullptr; }
}
void shk_host_free(void *p) { free(p); }
// the device inflater alone (csrc/inflate_gpu.hip): 0 = *out (malloc'd, shk_host_free) holds the member's bytes; 1 = the
// member was not taken (*why says why: the product then reads it on the host); < 0 = error
int shk_device_gunzip(const uint8_t *gz, size_t n, uint8_t **out, size_t *out_n, const char **why, double *ms_total) {
    try {
        if (!gz || !out || !out_n) return SHK_E_PARAM;
        *out = nullptr; *out_n = 0;
        static thread_local std::string msg;
        std::string err;
        GpuText text; GpuInflateStats st;
        const int rc = gpu_inflate_member(gz, n, current_device(), nullptr, text, err, &st, true);
        if (why) { msg = rc == 1 ? st.why_not : err; *why = msg.c_str(); }
        if (ms_total) *ms_total = st.total_ms;
        if (rc == 1) return 1;
        if (rc) return rc == -4 ? SHK_E_OOM : SHK_E_DEVICE;
        const size_t bytes = text.e;
        uint8_t *o = (uint8_t *)malloc(bytes ? bytes : 1);
        if (!o) { gpu_text_free(text); return SHK_E_OOM; }
        const int rd = device_download(o, text.d, bytes, err);
        gpu_text_free(text);
        if (rd) { free(o); if (why) { msg = err; *why = msg.c_str(); } return SHK_E_DEVICE; }
        *out = o; *out_n = bytes;
        return SHK_OK;
    } catch (...) { return SHK_E_OOM; }
}
int shk_host_gunzip(const uint8_t *gz, size_t n, uint8_t **out, size_t *out_n, uint64_t *mt_members, double *reader_seconds) {
    try {
        if (!gz || !out || !out_n) return SHK_E_PARAM;
        ByteVec st; const uint8_t *p = nullptr; size_t pn = 0; std::string err;
        const double t0 = now_ms();
        const int rc = maybe_inflate(gz, n, st, p, pn, err);
        if (reader_seconds) *reader_seconds = (now_ms() - t0) * 1e-3;
        if (rc) return rc == -3 ? SHK_E_PARSE : SHK_E_OOM;
        *out = (uint8_t *)malloc(pn ? pn : 1);
        if (!*out) return SHK_E_OOM;
        if (pn) memcpy(*out, p, pn);
        *out_n = pn;
        if (mt_members) *mt_members = inflate_mt_members();
        return SHK_OK;
    } catch (...) { return SHK_E_OOM; }
}
char *shk_host_unitig_assemble(uint32_t k, uint64_t n_recs, const uint64_t *first, const uint64_t *last, const uint64_t *len,
                               const uint64_t *kc, const uint8_t *circ, const uint64_t *min_key, const uint8_t *min_o,
                               const uint64_t *min_pos, int tips, int bubbles) {
    try {
        if ((k & 1u) == 0 || k < SHK_K_MIN || k > SHK_K_MAX || (n_recs && (!first || !last || !len || !kc || !circ))) return nullptr;
        const uint32_t W = (2 * k + 63) / 64;
        std::vector<UnitigRec> recs((size_t)n_recs);
        for (uint64_t r = 0; r < n_recs; r++) {
            for (uint32_t j = 0; j < W; j++) { recs[r].first[j] = first[r * W + j]; recs[r].last[j] = last[r * W + j]; }
            recs[r].len = len[r]; recs[r].kc = kc[r]; recs[r].circ = circ[r];
        }
        UnitigGraphResult res; std::string err;
        auto fail_text = [](const std::string &e) -> char * { const std::string t = "error: " + e; char *o = (char *)malloc(t.size() + 1); if (o) memcpy(o, t.c_str(), t.size() + 1); return o; };
        if (unitig_assemble((int)k, recs, tips != 0, bubbles != 0, res, err)) return fail_text(err);
        std::vector<UnitigMinKey> mk((size_t)n_recs);
        for (uint32_t r : res.need_min) {
            if (!min_key || !min_o || !min_pos) return nullptr;
            for (uint32_t j = 0; j < W; j++) mk[r].key[j] = min_key[(uint64_t)r * W + j];
            mk[r].o = min_o[r]; mk[r].pos = min_pos[r]; mk[r].valid = true;
        }
        if (unitig_resolve_rings((int)k, recs, mk, res, err)) return fail_text(err);
        std::string text = "removed " + std::to_string(res.tips_removed) + " " + std::to_string(res.bubbles_removed) + "\n";
        for (const UnitigContig &c : res.contigs) {
            text += std::to_string(c.ring ? 1 : 0) + " " + std::to_string(c.rot) + " " + std::to_string(c.len_nodes) + " " + std::to_string(c.kc) + " :";
            for (uint32_t r : c.recs) text += " " + std::to_string(r);
            text += "\n";
        }
        char *out = (char *)malloc(text.size() + 1);
        if (!out) return nullptr;
        memcpy(out, text.c_str(), text.size() + 1);
        return out;
    } catch (...) { return nullptr; }
}

}  // extern "C"
