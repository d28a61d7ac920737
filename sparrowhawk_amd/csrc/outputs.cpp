// outputs.cpp — see outputs.h.  The four texts are produced directly in their JSON-escaped form
// (the only consumer is get_assembly(), Assembler.ts:127): sequences are copied in bulk, only
// the short separators need escaping, so a 5 Mbp contig costs three memcpys, not a per-byte scan.
#include "outputs.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <tuple>

namespace shk {

static inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }

static void revcomp_range(const char *s, size_t n, char *r, size_t a, size_t b) {       // r[a..b) of the reverse complement
    for (size_t i = a; i < b; i++) r[i] = comp(s[n - 1 - i]);
}
// (revcomp of a whole contig: defined below the worker pool — big ones are done in pieces)

// is revcomp(s) < s ?  decided at the first differing position, without materialising it
static bool revcomp_is_smaller(const char *s, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const char a = comp(s[n - 1 - i]), b = s[i];
        if (a != b) return a < b;
    }
    return false;
}

void json_escape_into(std::string &dst, const std::string &s) {
    dst.push_back('"');
    for (char c : s) {
        switch (c) {
            case '\n': dst += "\\n"; break;
            case '\t': dst += "\\t"; break;
            case '"': dst += "\\\""; break;
            case '\\': dst += "\\\\"; break;
            default: dst.push_back(c);
        }
    }
    dst.push_back('"');
}

std::string preprocessing_json(uint64_t nkmers, const uint64_t *h, uint32_t used) {
    std::string j = "{\"nkmers\":" + std::to_string(nkmers) + ",\"histo\":[";
    for (int i = 0; i < 500; i++) { if (i) j.push_back(','); j += std::to_string(h[i]); }
    j += "],\"used_min_count\":" + std::to_string(used) + "}";
    return j;
}

namespace {
// The writers emit JSON-escaped text (NL/TAB/QUOTE are the only specials they produce) into a sink: one
// sink measures, the other writes through a pointer — every section's size is known before a byte is
// written, so the sections go straight to their places in one buffer, the big ones in parallel.
static inline size_t n_digits(uint64_t v) { size_t n = 1; while (v >= 10) { v /= 10; n++; } return n; }
struct SizeSink {
    size_t n = 0;
    void seq(const char *, size_t m) { n += m; }
    void nl() { n += 2; }
    void tab() { n += 2; }
    void quote() { n += 2; }
    void raw(const char *p) { n += strlen(p); }
    void raw(const std::string &p) { n += p.size(); }
    void raw(const char *, size_t m) { n += m; }
    void num(uint64_t v) { n += n_digits(v); }
};
struct PtrSink {
    char *p;
    // a sequence of at least `big` bytes is not copied here: its place is skipped and the copy is queued, so that
    // it can be done in pieces by all threads
    size_t big = (size_t)-1;
    std::vector<std::pair<char *, std::pair<const char *, size_t>>> *deferred = nullptr;
    std::mutex *mu = nullptr;
    TextArrival *arr = nullptr;        // the sequences may still be arriving from the device: wait for the bytes about to be copied
    void seq(const char *q, size_t m) {
        if (m >= big && deferred) { std::lock_guard<std::mutex> lk(*mu); deferred->push_back({p, {q, m}}); }
        else {
            if (arr && q >= arr->base() && q < arr->base() + arr->total()) { const size_t o = (size_t)(q - arr->base()); arr->wait_range(o, o + m); }
            memcpy(p, q, m);
        }
        p += m;
    }
    void nl() { *p++ = '\\'; *p++ = 'n'; }
    void tab() { *p++ = '\\'; *p++ = 't'; }
    void quote() { *p++ = '\\'; *p++ = '"'; }
    void raw(const char *q) { const size_t m = strlen(q); memcpy(p, q, m); p += m; }
    void raw(const std::string &q) { memcpy(p, q.data(), q.size()); p += q.size(); }
    void raw(const char *q, size_t m) { memcpy(p, q, m); p += m; }
    void num(uint64_t v) { const size_t m = n_digits(v); for (size_t i = m; i-- > 0;) { p[i] = (char)('0' + v % 10); v /= 10; } p += m; }
};
}  // namespace

// ---- a small persistent worker pool -------------------------------------------------------------------
// The writer's work is megabytes of memcpy (a 5 Mbp contig goes into FASTA, GFA1 and GFA2) or millions of small
// records (a fragmented metagenome: 4 M contigs); a handle lives for one assembly, so threads are kept across
// handles (spawning three per assembly cost ~0.1 ms of the 0.35 ms the writer took on the bench workload).
namespace {
class WorkPool {
public:
    static WorkPool &get() { static WorkPool *p = new WorkPool(16u, "SHK_WRITER_THREADS"); return *p; }     // never destroyed (process teardown order)
    // a fragmented assembly (10^4 .. 10^7 contigs: a metagenome) is seconds of hashing, sorting and text on 16 threads; its
    // jobs are coarse enough for every core of the host (created on first use)
    static WorkPool &big() { static WorkPool *p = new WorkPool(64u, "SHK_WRITER_THREADS_BIG"); return *p; }
    unsigned size() const { return (unsigned)n_workers_ + 1u; }                   // workers + the caller
    // runs fn(task) for task in [0, n_tasks) on the pool and the calling thread; returns when all are done
    template <typename F> void run(size_t n_tasks, F &&fn) {
        if (n_tasks == 0) return;
        // (a job started from inside a task of another job runs inline: the pool is not re-entrant)
        if (n_tasks == 1 || n_workers_ == 0 || in_task()) { for (size_t i = 0; i < n_tasks; i++) fn(i); return; }
        std::lock_guard<std::mutex> job_lock(job_mu_);                           // one job at a time (handles on several threads)
        std::function<void(size_t)> f = [&fn](size_t i) { fn(i); };
        uint64_t gen;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &f; n_tasks_ = n_tasks; next_ = 0; pending_ = n_tasks; failed_ = false; gen = ++gen_;
            gen_atomic_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
        work(gen);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;                                                           // (under mu_: no worker is inside this job any more)
        if (failed_) throw std::bad_alloc();
    }
private:
    WorkPool(unsigned cap, const char *env_name) {
        unsigned hc = std::thread::hardware_concurrency();
        if (hc == 0) hc = 4;
        unsigned n = std::min(hc, cap);
        if (const char *v = getenv(env_name)) { const long t = strtol(v, nullptr, 10); if (t >= 1 && t <= 256) n = (unsigned)t; }
        n_workers_ = n - 1;
        for (unsigned i = 1; i < n; i++) std::thread([this] { loop(); }).detach();
    }
    // Tasks are claimed under the mutex together with the job's generation: a worker that wakes up late (or is still
    // leaving the previous job) can never take a task index of one job and run it against another.  Tasks are coarse
    // (tens to hundreds per job), so the lock is not a bottleneck.
    void work(uint64_t gen) {
        for (;;) {
            size_t i; const std::function<void(size_t)> *f;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (gen_ != gen || next_ >= n_tasks_ || !fn_) return;
                i = next_++; f = fn_;
            }
            bool threw = false;
            in_task() = true;
            try { (*f)(i); } catch (...) { threw = true; }        // (out of memory in a task: reported by run(), never a terminate)
            in_task() = false;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (threw) failed_ = true;
                if (--pending_ == 0) done_cv_.notify_all();
            }
        }
    }
    void loop() {
        uint64_t seen = 0, seen_warm = 0;
        auto until = std::chrono::steady_clock::now();               // end of the current warm window
        for (;;) {
            if (std::chrono::steady_clock::now() >= until) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen || warm_gen_ != seen_warm; });
                if (warm_gen_ != seen_warm) { seen_warm = warm_gen_; until = std::chrono::steady_clock::now() + std::chrono::microseconds(warm_us_.load()); }
                if (gen_ == seen) continue;                          // woken to stay warm: spin below on the next turn
                seen = gen_;
            } else {
                // a job is about to come, or another one of the same assembly (WorkPool::prewarm): stay awake for it
                // instead of paying the wake-up latency of a sleeping thread (tens to hundreds of microseconds)
                while (gen_atomic_.load(std::memory_order_acquire) == seen && warm_atomic_.load(std::memory_order_acquire) == seen_warm &&
                       std::chrono::steady_clock::now() < until) std::this_thread::yield();
                std::lock_guard<std::mutex> lk(mu_);
                if (warm_gen_ != seen_warm) { seen_warm = warm_gen_; until = std::chrono::steady_clock::now() + std::chrono::microseconds(warm_us_.load()); }
                if (gen_ == seen) continue;
                seen = gen_;
            }
            work(seen);
        }
    }
    static bool &in_task() { static thread_local bool f = false; return f; }
    unsigned n_workers_ = 0;
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_tasks_ = 0, pending_ = 0, next_ = 0;
    bool failed_ = false;
    uint64_t gen_ = 0, warm_gen_ = 0;
    std::atomic<uint64_t> gen_atomic_{0}, warm_atomic_{0};
    std::atomic<long> warm_us_{0};
public:
    // tells the workers that a job will arrive within about `us` microseconds: they wake up now and spin until then
    void prewarm(long us) {
        if (n_workers_ == 0) return;
        { std::lock_guard<std::mutex> lk(mu_); warm_us_.store(us); warm_gen_++; warm_atomic_.store(warm_gen_, std::memory_order_release); }
        cv_.notify_all();
    }
};

std::string revcomp(const char *s, size_t n) {
    std::string r(n, 'A');
    if (n < ((size_t)1 << 20)) { revcomp_range(s, n, &r[0], 0, n); return r; }
    const size_t pieces = WorkPool::get().size() * 2;
    WorkPool::get().run(pieces, [&](size_t t) { revcomp_range(s, n, &r[0], n * t / pieces, n * (t + 1) / pieces); });
    return r;
}

// parallel loop over [0, n) in `pieces` contiguous ranges
template <typename F> void par_ranges(size_t n, size_t min_per_piece, F &&fn, WorkPool *use = nullptr) {
    WorkPool &pool = use ? *use : WorkPool::get();
    size_t pieces = std::min<size_t>(pool.size() * 4, n / std::max<size_t>(min_per_piece, 1));
    if (pieces < 2) { fn(0, n); return; }
    pool.run(pieces, [&](size_t t) { fn(n * t / pieces, n * (t + 1) / pieces); });
}
}  // namespace

void writer_prewarm(long microseconds) { WorkPool::get().prewarm(microseconds); }

void build_assembly_text(std::vector<RawContig> &contigs, uint32_t k, AssemblyText &out, TextArrival *arrival) {
    // Text that is still arriving (pipeline.h: TextArrival): everything in front of the copies reads the contigs' ENDS only
    // (RawContig::head / tail: the first / last max(k, 32) bases); whatever needs more — a strand or order decision that is
    // still open after those bases, a reverse complement — waits for the whole text first (rare: ties).
    auto all_here = [&] { if (arrival) { arrival->wait_all(); arrival = nullptr; for (auto &c : contigs) { c.head = c.tail = nullptr; c.ends_n = 0; } } };
    auto first_bases = [](const RawContig &c) { return c.head ? c.head : c.data(); };                       // >= min(size, max(k, 32)) of them
    auto last_k = [k](const RawContig &c) { return c.tail ? c.tail + (c.ends_n - k) : c.data() + c.size() - k; };
    if (arrival) for (auto &c : contigs) if (!c.head || c.ends_n < std::min<uint64_t>(c.size(), std::max<uint32_t>(k, 32))) { all_here(); break; }
    const char *pm = getenv("SHK_WRITER_PAR_MIN");          // (tests force the parallel paths on small outputs)
    const size_t par_min = (pm && *pm) ? (size_t)strtoull(pm, nullptr, 10) : ((size_t)1 << 20);
    const bool many = contigs.size() >= (par_min >= ((size_t)1 << 20) ? (size_t)20000 : (size_t)2);
    const size_t grain = many && par_min < ((size_t)1 << 20) ? 1 : 2048;
    // (many = a fragmented assembly; the forced-parallel test mode keeps the small pool)
    WorkPool &wp = (many && contigs.size() >= 200000) ? WorkPool::big() : WorkPool::get();
    out.stage_ms.clear();
    auto clock_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = clock_ms();
    auto lap = [&](const char *name) { const double t = clock_ms(); out.stage_ms.emplace_back(name, t - t_last); t_last = t; };
    // SPEC S10: each unitig is emitted as min(seq, revcomp(seq))
    if (arrival) {
        // from the ends alone: position i of revcomp(s) is comp(s[n-1-i]) — decided within the first ends_n positions, or open
        bool open = false;
        for (auto &c : contigs) {
            const size_t n = c.size(), m = c.ends_n;
            int verdict = 0;                              // -1 revcomp smaller, +1 not, 0 open
            for (size_t i = 0; i < m && !verdict; i++) {
                const char a = comp(c.tail[m - 1 - i]), b = c.head[i];
                if (a != b) verdict = a < b ? -1 : 1;
            }
            if (verdict == 0 && m >= n) verdict = 1;      // (the whole contig was compared: it is its own reverse complement)
            if (verdict <= 0) { open = true; break; }      // a reverse complement has to be built, or the ends do not decide
        }
        if (open) all_here();
    }
    auto canon = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            RawContig &c = contigs[i];
            if (revcomp_is_smaller(c.data(), c.size())) { c.own = revcomp(c.data(), c.size()); c.ext = nullptr; c.ext_n = 0; }
        }
    };
    if (arrival) {}                                        // (every contig is the smaller strand already: decided above)
    else
    { if (many) par_ranges(contigs.size(), grain, canon, &wp); else canon(0, contigs.size()); }
    lap("outputs_canonical_strand");
    // SPEC S11: order by (length desc, sequence asc).  An index is sorted, not the records; with many contigs the
    // ranges are sorted in parallel and merged pairwise.
    const size_t nc = contigs.size();
    // (sorted through 16-byte keys — length and the first 32 bases, 2-bit packed — so that the comparisons stay in one
    // array; the sequences themselves are only touched on a tie)
    struct SortKey { uint64_t prefix; uint32_t len, idx; };
    std::vector<SortKey> order(nc);
    auto make_keys = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            const RawContig &c = contigs[i];
            const char *p = first_bases(c); const size_t n = c.size(), m = std::min<size_t>(n, 32);
            uint64_t pf = 0;
            for (size_t j = 0; j < m; j++) { const char ch = p[j]; pf = (pf << 2) | (uint64_t)(ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3); }
            pf <<= 2 * (32 - m);
            order[i] = SortKey{pf, (uint32_t)std::min<size_t>(n, 0xFFFFFFFFu), (uint32_t)i};
        }
    };
    if (many) par_ranges(nc, grain, make_keys, &wp); else make_keys(0, nc);
    auto before = [&](const SortKey &x, const SortKey &y) {
        if (x.len != y.len) return x.len > y.len;
        if (x.prefix != y.prefix) return x.prefix < y.prefix;
        const RawContig &a = contigs[x.idx], &b = contigs[y.idx];
        if (a.size() != b.size()) return a.size() > b.size();           // (lengths beyond 2^32 - 1)
        if (arrival) arrival->wait_all();                               // (a tie on length and 32 bases: the texts decide)
        const int c = memcmp(a.data(), b.data(), a.size());
        return c != 0 ? c < 0 : x.idx < y.idx;                          // (equal spellings cannot occur; keeps the order total)
    };
    if (many && nc >= 4) {
        WorkPool &pool = wp;
        size_t pieces = 1; while (pieces * 2 <= pool.size() * 2 && nc / (pieces * 2) >= grain) pieces *= 2;
        std::vector<size_t> cut(pieces + 1);
        for (size_t t = 0; t <= pieces; t++) cut[t] = nc * t / pieces;
        pool.run(pieces, [&](size_t t) { std::sort(order.begin() + cut[t], order.begin() + cut[t + 1], before); });
        std::vector<SortKey> tmp(nc);
        for (size_t width = 1; width < pieces; width *= 2) {
            const size_t n_merges = pieces / (2 * width);
            pool.run(n_merges, [&](size_t j) {
                const size_t a = cut[2 * width * j], m = cut[2 * width * j + width], b = cut[2 * width * (j + 1)];
                std::merge(order.begin() + a, order.begin() + m, order.begin() + m, order.begin() + b, tmp.begin() + a, before);
            });
            order.swap(tmp);
        }
    } else std::sort(order.begin(), order.end(), before);
    {
        std::vector<RawContig> sorted(nc);
        auto place = [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) sorted[i] = std::move(contigs[order[i].idx]); };
        if (many) par_ranges(nc, grain, place, &wp); else place(0, nc);
        contigs.swap(sorted);
    }
    std::vector<SortKey>().swap(order);
    out.ncontigs = nc;
    lap("outputs_order");

    // links: first k-mer of every (contig, orientation) -> id; '+' entries win over '-'.  K-mers are handled
    // 2-bit packed (8 words hold k <= 255): a fragmented assembly has 10^4..10^7 contigs and this map is
    // the whole cost of the writer then.
    constexpr int KW = 8;
    struct Key {
        uint64_t w[KW];
        bool operator==(const Key &o) const { for (int i = 0; i < KW; i++) if (w[i] != o.w[i]) return false; return true; }
    };
    const int kw_used = (int)((2 * k + 63) / 64);             // words a k-mer of this k occupies: the rest stay zero
    struct KeyHash {
        int n;
        size_t operator()(const Key &x) const {
            uint64_t h = 0x9e3779b97f4a7c15ull;
            for (int i = 0; i < n; i++) { h ^= x.w[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
            return (size_t)h;
        }
    };
    auto code = [](char c) -> uint64_t { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; };
    auto shl2 = [&](Key &x, uint64_t b) {                      // x = x * 4 + b
        for (int i = kw_used - 1; i > 0; i--) x.w[i] = (x.w[i] << 2) | (x.w[i - 1] >> 62);
        x.w[0] = (x.w[0] << 2) | b;
    };
    auto mask_k = [&](Key &x) {                                // keep the low 2k bits
        const uint32_t used = 2 * k;
        for (uint32_t i = 0; i < (uint32_t)KW; i++) {
            if (64 * i >= used) x.w[i] = 0;
            else if (used - 64 * i < 64) x.w[i] &= (1ull << (used - 64 * i)) - 1ull;
        }
    };
    auto pack = [&](const char *p) {                           // first base in the top 2 bits of the 2k-bit value
        Key x{};
        for (uint32_t i = 0; i < k; i++) { const uint32_t pos = 2 * (k - 1 - i); x.w[pos >> 6] |= code(p[i]) << (pos & 63); }
        return x;
    };
    auto pack_rc = [&](const char *p) {                        // reverse complement of p[0..k)
        Key x{};
        for (uint32_t i = 0; i < k; i++) { const uint32_t pos = 2 * i; x.w[pos >> 6] |= (3 - code(p[i])) << (pos & 63); }
        return x;
    };
    // Flat open-addressing table, one 64-bit word per slot: fingerprint of the key (23 bits) | orientation | contig
    // (40 bits).  The key of an entry is recomputed from its contig, but only when the fingerprints agree, so a probe
    // is one cache miss; many threads fill the table with compare-and-swap.  Of several entries with one key the
    // SMALLEST word survives — same key, same fingerprint: '+' before '-', then the smaller contig id — which is
    // what "all '+' entries first, then all '-', the first one in wins" gave the serial writer, and is order-free.
    size_t cap = 16; while (cap < 4 * nc + 4) cap <<= 1;
    std::unique_ptr<std::atomic<uint64_t>[]> hv(new std::atomic<uint64_t>[cap]);
    par_ranges(cap, 1 << 16, [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) hv[i].store(~0ull, std::memory_order_relaxed); }, &wp);
    const KeyHash hasher{kw_used};
    auto dec_c = [](uint64_t v) { return (uint32_t)(v & ((1ull << 40) - 1)); };
    auto dec_o = [](uint64_t v) { return (uint32_t)((v >> 40) & 1); };
    auto key_of_enc = [&](uint64_t v) -> Key {
        const RawContig &c = contigs[dec_c(v)];
        return dec_o(v) ? pack_rc(last_k(c)) : pack(first_bases(c));
    };
    auto fp_of = [](size_t h) -> uint64_t { uint64_t f = (uint64_t)(h >> 41); return f == 0x7FFFFF ? 0x7FFFFE : f; };   // (all ones is "empty")
    auto put2 = [&](const Key &x, uint64_t contig, uint64_t o) {
        const size_t h = hasher(x);
        const uint64_t v = (fp_of(h) << 41) | (o << 40) | contig;
        size_t s = h & (cap - 1);
        for (;;) {
            uint64_t cur = hv[s].load(std::memory_order_acquire);
            if (cur == ~0ull && hv[s].compare_exchange_strong(cur, v, std::memory_order_acq_rel)) return;
            if ((cur >> 41) == (v >> 41) && key_of_enc(cur) == x) {
                while (v < cur && !hv[s].compare_exchange_weak(cur, v, std::memory_order_acq_rel)) {}
                return;
            }
            s = (s + 1) & (cap - 1);
        }
    };
    auto get2 = [&](const Key &x) -> uint64_t {
        const size_t h = hasher(x);
        const uint64_t f = fp_of(h);
        size_t s = h & (cap - 1);
        for (;;) {
            const uint64_t cur = hv[s].load(std::memory_order_relaxed);
            if (cur == ~0ull) return ~0ull;
            if ((cur >> 41) == f && key_of_enc(cur) == x) return cur;
            s = (s + 1) & (cap - 1);
        }
    };
    auto fill = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            put2(pack(first_bases(contigs[i])), i, 0);
            put2(pack_rc(last_k(contigs[i])), i, 1);           // first k-mer of the '-' orientation
        }
    };
    if (many) par_ranges(nc, grain, fill, &wp); else fill(0, nc);
    lap("outputs_link_table");
    typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t> Link;
    auto find_links = [&](size_t a, size_t b, std::vector<Link> &dst) {
        for (size_t i = a; i < b; i++) for (uint32_t o = 0; o < 2; o++) {
            // last k-mer of the orientation: of '+' the contig's last k-mer, of '-' the reverse complement of its first
            Key cand = o ? pack_rc(first_bases(contigs[i])) : pack(last_k(contigs[i]));
            shl2(cand, 0); mask_k(cand);                    // drop the first base, append A
            for (uint64_t bb = 0; bb < 4; bb++) {
                cand.w[0] = (cand.w[0] & ~3ull) | bb;
                const uint64_t hit = get2(cand);
                if (hit == ~0ull) continue;
                const uint32_t cj = dec_c(hit), oj = dec_o(hit);
                Link L((uint32_t)i + 1, o, cj + 1, oj), M(cj + 1, !oj, (uint32_t)i + 1, !o);
                dst.push_back(M < L ? M : L);
            }
        }
    };
    std::vector<Link> links;
    if (many) {
        WorkPool &pool = wp;
        const size_t pieces = std::max<size_t>(1, std::min<size_t>(pool.size() * 4, nc / grain));
        std::vector<std::vector<Link>> part(pieces);
        pool.run(pieces, [&](size_t t) { find_links(nc * t / pieces, nc * (t + 1) / pieces, part[t]); });
        size_t tot = 0; for (auto &v : part) tot += v.size();
        links.reserve(tot);
        for (auto &v : part) links.insert(links.end(), v.begin(), v.end());
    } else find_links(0, nc, links);
    lap("outputs_link_find");
    std::sort(links.begin(), links.end());
    links.erase(std::unique(links.begin(), links.end()), links.end());
    lap("outputs_link_sort");

    // Straight into the JSON text in key order (a 5 Mbp contig is copied three times — FASTA, GFA1, GFA2 —
    // and never staged in per-format strings).  Every record's size is known before a byte is written, so the
    // records go to their places in one buffer from many threads: the big sequences in pieces, the many small
    // records in ranges.
    size_t seq_bytes = 0;
    for (auto &c : contigs) seq_bytes += c.size();
    const std::string ov = std::to_string(k - 1);
    // the per-contig and per-link records of each section, as functions of a sink
    auto fasta_rec = [&](auto &w, size_t i) {
        w.raw(">contig_"); w.num(i + 1); w.raw(" len="); w.num(contigs[i].size()); w.raw(" kc="); w.num(contigs[i].kc); w.nl();
        w.seq(contigs[i].data(), contigs[i].size()); w.nl();
    };
    auto dot_node = [&](auto &w, size_t i) {
        w.raw("  "); w.quote(); w.num(i + 1); w.quote(); w.raw(" [label="); w.quote(); w.num(i + 1); w.raw(" len=");
        w.num(contigs[i].size()); w.raw(" kc="); w.num(contigs[i].kc); w.quote(); w.raw("];"); w.nl();
    };
    auto dot_link = [&](auto &w, size_t j) {
        const Link &L = links[j];
        const uint32_t a = std::get<0>(L), ao = std::get<1>(L), b = std::get<2>(L), bo = std::get<3>(L);
        w.raw("  "); w.quote(); w.num(a); w.quote(); w.raw(" -> "); w.quote(); w.num(b); w.quote();
        w.raw(" [label="); w.quote(); w.raw(ao ? "-" : "+"); w.raw(bo ? "-" : "+"); w.quote(); w.raw("];"); w.nl();
    };
    auto gfa1_seg = [&](auto &w, size_t i) {
        w.raw("S"); w.tab(); w.num(i + 1); w.tab(); w.seq(contigs[i].data(), contigs[i].size()); w.tab(); w.raw("LN:i:"); w.num(contigs[i].size());
        w.tab(); w.raw("KC:i:"); w.num(contigs[i].kc); w.nl();
    };
    auto gfa1_link = [&](auto &w, size_t j) {
        const Link &L = links[j];
        const uint32_t a = std::get<0>(L), ao = std::get<1>(L), b = std::get<2>(L), bo = std::get<3>(L);
        w.raw("L"); w.tab(); w.num(a); w.tab(); w.raw(ao ? "-" : "+"); w.tab(); w.num(b); w.tab();
        w.raw(bo ? "-" : "+"); w.tab(); w.raw(ov); w.raw("M"); w.nl();
    };
    auto gfa2_seg = [&](auto &w, size_t i) {
        w.raw("S"); w.tab(); w.num(i + 1); w.tab(); w.num(contigs[i].size()); w.tab(); w.seq(contigs[i].data(), contigs[i].size()); w.tab();
        w.raw("KC:i:"); w.num(contigs[i].kc); w.nl();
    };
    auto gfa2_link = [&](auto &w, size_t j) {
        const Link &L = links[j];
        const uint32_t a = std::get<0>(L), ao = std::get<1>(L), b = std::get<2>(L), bo = std::get<3>(L);
        const uint64_t la = contigs[a - 1].size(), lb = contigs[b - 1].size();
        w.raw("E"); w.tab(); w.raw("*"); w.tab(); w.num(a); w.raw(ao ? "-" : "+"); w.tab(); w.num(b);
        w.raw(bo ? "-" : "+"); w.tab();
        if (!ao) { w.num(la - (k - 1)); w.tab(); w.num(la); w.raw("$"); w.tab(); }
        else { w.raw("0"); w.tab(); w.raw(ov); if ((uint64_t)(k - 1) == la) w.raw("$"); w.tab(); }
        if (!bo) { w.raw("0"); w.tab(); w.raw(ov); if ((uint64_t)(k - 1) == lb) w.raw("$"); w.tab(); }
        else { w.num(lb - (k - 1)); w.tab(); w.num(lb); w.raw("$"); w.tab(); }
        w.raw(ov); w.raw("M"); w.nl();
    };
    // A "part" = a run of records of one kind, preceded by a literal: {literal, kind, count}.  The JSON is the
    // concatenation of the parts; sizes are measured per record (prefix sums), then every range of records and
    // every big sequence copy becomes a task.
    enum Kind { NONE, FASTA, DOTN, DOTL, G1S, G1L, G2S, G2L };
    struct Part { std::string lit; Kind kind; size_t count; };
    std::vector<Part> parts;
    parts.push_back({"{\"outfasta\":\"", FASTA, nc});
    parts.push_back({"\",\"ncontigs\":" + std::to_string(nc) + ",\"outdot\":\"digraph sparrowhawk {\\n", DOTN, nc});
    parts.push_back({"", DOTL, links.size()});
    parts.push_back({"}\\n\",\"outgfa\":\"H\\tVN:Z:1.0\\n", G1S, nc});
    parts.push_back({"", G1L, links.size()});
    parts.push_back({"\",\"outgfav2\":\"H\\tVN:Z:2.0\\n", G2S, nc});
    parts.push_back({"", G2L, links.size()});
    parts.push_back({"\"}", NONE, 0});
    auto emit_rec = [&](auto &w, Kind kd, size_t i) {
        switch (kd) {
            case FASTA: fasta_rec(w, i); break; case DOTN: dot_node(w, i); break; case DOTL: dot_link(w, i); break;
            case G1S: gfa1_seg(w, i); break; case G1L: gfa1_link(w, i); break; case G2S: gfa2_seg(w, i); break;
            case G2L: gfa2_link(w, i); break; default: break;
        }
    };
    // every part is cut into ranges of records; a range is measured (SizeSink) and later written (PtrSink) by one
    // task, so only the ranges' offsets are kept — no per-record tables for millions of contigs
    struct Range { size_t part, a, b, bytes, off; };
    std::vector<Range> ranges;
    std::vector<size_t> part_off(parts.size() + 1, 0);
    for (size_t p = 0; p < parts.size(); p++) {
        const size_t cnt = parts[p].count;
        if (!cnt) continue;
        const size_t per = many ? std::max<size_t>(grain, cnt / (wp.size() * 4) + 1) : cnt;
        for (size_t a = 0; a < cnt; a += per) ranges.push_back(Range{p, a, std::min(cnt, a + per), 0, 0});
    }
    auto measure = [&](size_t r) { SizeSink z; for (size_t i = ranges[r].a; i < ranges[r].b; i++) emit_rec(z, parts[ranges[r].part].kind, i); ranges[r].bytes = z.n; };
    if (many && ranges.size() > 1) wp.run(ranges.size(), measure); else for (size_t r = 0; r < ranges.size(); r++) measure(r);
    {
        size_t at = 0, r = 0;
        for (size_t p = 0; p < parts.size(); p++) {
            part_off[p] = at; at += parts[p].lit.size();
            for (; r < ranges.size() && ranges[r].part == p; r++) { ranges[r].off = at; at += ranges[r].bytes; }
        }
        part_off[parts.size()] = at;
    }
    const size_t total = part_off[parts.size()];
    lap("outputs_measure");
    // (uninitialised storage: every byte is written exactly once below, by the threads that own the ranges — a resize that
    // zero-fills a gigabyte of JSON first cost a third of the writer on a metagenome; large blocks are recycled: bytebuf.h)
    ByteVec js;
    js.resize(total + 1);
    char *base = (char *)js.data();
    base[total] = 0;
    // sequences above `big` bytes are cut out of their record and copied in pieces by all threads
    const size_t big = std::max<size_t>(par_min / 4, 1);
    std::vector<std::pair<char *, std::pair<const char *, size_t>>> deferred;     // (filled by PtrSink::seq)
    std::mutex deferred_mu;
    for (size_t p = 0; p < parts.size(); p++) memcpy(base + part_off[p], parts[p].lit.data(), parts[p].lit.size());
    auto write_range = [&](size_t r) {
        PtrSink w{base + ranges[r].off};
        w.big = seq_bytes >= par_min ? big : (size_t)-1; w.deferred = &deferred; w.mu = &deferred_mu; w.arr = arrival;
        for (size_t i = ranges[r].a; i < ranges[r].b; i++) emit_rec(w, parts[ranges[r].part].kind, i);
    };
    if (ranges.size() > 1 && (many || seq_bytes >= par_min)) wp.run(ranges.size(), write_range);
    else for (size_t r = 0; r < ranges.size(); r++) write_range(r);
    if (!deferred.empty()) {
        // one task = one piece of a source sequence and ALL its destinations (a contig goes into FASTA, GFA1 and GFA2): the
        // piece is read from memory once — text that has just arrived over PCIe sits in no cache — and written three times
        // from the core's cache.  Tasks are taken in the order of the sources, which is the order the text arrives in.
        std::sort(deferred.begin(), deferred.end(), [](const auto &x, const auto &y) { return x.second.first != y.second.first ? x.second.first < y.second.first : x.first < y.first; });
        struct Task { const char *src; size_t n; uint32_t first, count; };         // destinations: dsts[first .. first + count)
        std::vector<char *> dsts;
        std::vector<Task> tasks;
        const size_t piece2 = (size_t)128 << 10;
        for (size_t i = 0; i < deferred.size();) {
            size_t j = i;
            while (j < deferred.size() && deferred[j].second.first == deferred[i].second.first && deferred[j].second.second == deferred[i].second.second) j++;
            const char *src = deferred[i].second.first; const size_t n = deferred[i].second.second;
            for (size_t o = 0; o < n; o += piece2) {
                Task t{src + o, std::min(piece2, n - o), (uint32_t)dsts.size(), (uint32_t)(j - i)};
                for (size_t q = i; q < j; q++) dsts.push_back(deferred[q].first + o);
                tasks.push_back(t);
            }
            i = j;
        }
        TextArrival *arr = arrival;
        wp.run(tasks.size(), [&tasks, &dsts, arr](size_t i) {
            const Task &t = tasks[i];
            if (arr && t.src >= arr->base() && t.src < arr->base() + arr->total()) { const size_t o = (size_t)(t.src - arr->base()); arr->wait_range(o, o + t.n); }
            for (uint32_t q = 0; q < t.count; q++) memcpy(dsts[t.first + q], t.src, t.n);
        });
    }
    if (arrival) arrival->wait_all();
    WorkPool::get().prewarm(0);                         // the writer is done: the workers go back to sleep
    if (&wp != &WorkPool::get()) wp.prewarm(0);
    lap("outputs_write");
    out.json = std::move(js);
    out.fasta.clear(); out.gfa1.clear(); out.gfa2.clear(); out.dot.clear();
}

}  // namespace shk
