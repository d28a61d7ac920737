// fastq.cpp — see fastq.h
#include "fastq.h"
#include "inflate_mt.h"

#include <sys/mman.h>
#include <unordered_map>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <algorithm>
#include <zlib.h>
#include <atomic>
#include <thread>

namespace shk {

// ---- large host blocks (bytebuf.h) ---------------------------------------------------------------------------------
namespace {
struct BigPool {
    std::mutex mu;
    std::vector<std::pair<size_t, void *>> kept;          // (bytes, block)
    std::unordered_map<void *, size_t> size_of;           // every big block alive (handed out or kept) -> its real size
    size_t kept_bytes = 0;
    static constexpr size_t MIN_BIG = (size_t)4 << 20;
    // what may stay cached per process: SHK_HOST_POOL_MAX bytes (default 4 GiB — the text of one .fastq.gz isolate with its
    // marker buffers; a node runs one process per GPU, so eight of these)
    size_t keep_max = [] {
        const char *e = getenv("SHK_HOST_POOL_MAX");
        if (e && *e) return (size_t)strtoull(e, nullptr, 10);
        return (size_t)4 << 30;
    }();
};
BigPool &big_pool() { static BigPool *p = new BigPool(); return *p; }     // never destroyed (process teardown order)
inline size_t big_round(size_t bytes) { return (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1); }
}  // namespace
void *big_alloc(size_t bytes) {
    if (bytes < BigPool::MIN_BIG) return malloc(bytes ? bytes : 1);
    const size_t want = big_round(bytes);
    BigPool &bp = big_pool();
    {
        std::lock_guard<std::mutex> lk(bp.mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < bp.kept.size(); i++)
            if (bp.kept[i].first >= want && bp.kept[i].first <= want * 2 && (best == (size_t)-1 || bp.kept[i].first < bp.kept[best].first)) best = i;
        if (best != (size_t)-1) {
            void *p = bp.kept[best].second;
            bp.kept_bytes -= bp.kept[best].first;
            // (the block keeps its real size in size_of: callers free with THEIR size)
            bp.kept[best] = bp.kept.back(); bp.kept.pop_back();
            return p;
        }
    }
    void *p = aligned_alloc((size_t)2 << 20, want);
    if (!p) return nullptr;
    (void)madvise(p, want, MADV_HUGEPAGE);
    try { std::lock_guard<std::mutex> lk(bp.mu); bp.size_of[p] = want; }
    catch (...) { free(p); return nullptr; }
    return p;
}
void big_free(void *p, size_t bytes) {
    if (!p) return;
    if (bytes < BigPool::MIN_BIG) { free(p); return; }
    BigPool &bp = big_pool();
    {
        std::lock_guard<std::mutex> lk(bp.mu);
        auto it = bp.size_of.find(p);
        const size_t real = it != bp.size_of.end() ? it->second : 0;       // (recorded at allocation: a recycled block may be larger than `bytes`)
        if (real >= BigPool::MIN_BIG && bp.kept_bytes + real <= bp.keep_max && bp.kept.size() < 512) {
            try { bp.kept.emplace_back(real, p); bp.kept_bytes += real; return; } catch (...) {}
        }
        if (it != bp.size_of.end()) bp.size_of.erase(it);
    }
    free(p);
}
void big_trim() {
    BigPool &bp = big_pool();
    std::lock_guard<std::mutex> lk(bp.mu);
    for (auto &kv : bp.kept) { bp.size_of.erase(kv.second); free(kv.second); }
    bp.kept.clear(); bp.kept_bytes = 0;
}

void PackedReads::clear() {
    bases.clear(); seg_off.clear(); n_bases = n_reads = n_input_bases = 0; cur = 0;
}

void PackedReads::reset_stream() {
    bases.clear(); seg_off.clear(); n_bases = 0; cur = 0;
}

void PackedReads::finish() {
    if (seg_off.empty()) seg_off.push_back(0);
    // materialise the partial word and one spare word so kernels may read one word past the end
    std::vector<uint32_t> &b = bases;
    const size_t full = (size_t)(n_bases >> 4);
    b.resize(full);
    b.push_back((n_bases & 15) ? cur : 0u);
    b.push_back(0u);
}

static int inflate_all(const uint8_t *in, size_t n, ByteVec &out, std::string &err) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) { err = "zlib init failed"; return -3; }
    size_t produced = 0, consumed = 0;
    // a large member first goes to the multi-threaded inflater (inflate_mt.cpp: block starts found speculatively, chunks
    // decoded with markers for the unknown window, CRC checked); whatever it declines is zlib's
    {
        const char *tv = getenv("SHK_GUNZIP_THREADS");
        unsigned T = std::thread::hardware_concurrency();
        if (T > 128) T = 128;
        if (tv && *tv) T = (unsigned)std::max(0, atoi(tv));
        size_t used = 0;
        if (T >= 2 && n >= ((size_t)2 << 20) && inflate_member_parallel(in, n, out, 0, used, T) == 0) {
            produced = out.size(); consumed = used;
            if (consumed >= n) { inflateEnd(&zs); return 0; }
        } else out.clear();
    }
    out.resize(produced + (n - consumed) * 4 + 4096);
    for (;;) {
        if (produced == out.size()) out.resize(out.size() * 2);
        size_t in_chunk = n - consumed; if (in_chunk > (1u << 30)) in_chunk = 1u << 30;
        size_t out_chunk = out.size() - produced; if (out_chunk > (1u << 30)) out_chunk = 1u << 30;
        zs.next_in = (Bytef *)(in + consumed); zs.avail_in = (uInt)in_chunk;
        zs.next_out = out.data() + produced; zs.avail_out = (uInt)out_chunk;
        int rc = inflate(&zs, Z_NO_FLUSH);
        consumed += in_chunk - zs.avail_in;
        produced += out_chunk - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (consumed >= n) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); err = "zlib reset failed"; return -3; }   // next member
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { inflateEnd(&zs); err = "gzip stream corrupt"; return -3; }
        if (consumed >= n && zs.avail_out != 0) { inflateEnd(&zs); err = "gzip stream truncated"; return -3; }
    }
    inflateEnd(&zs);
    out.resize(produced);
    return 0;
}

// ---- BGZF (bgzip) input: a gzip file made of independent blocks of <= 64 KiB, each announcing its
// compressed size in a 'BC' extra subfield (SAM spec section 4.1).  The blocks are inflated in parallel.
static bool bgzf_block(const uint8_t *b, size_t n, size_t &bsize) {
    if (n < 18 || b[0] != 0x1F || b[1] != 0x8B || b[2] != 8 || !(b[3] & 4)) return false;
    const size_t xlen = b[10] | ((size_t)b[11] << 8);
    if (12 + xlen > n) return false;
    for (size_t o = 12; o + 4 <= 12 + xlen;) {
        const size_t slen = b[o + 2] | ((size_t)b[o + 3] << 8);
        if (b[o] == 'B' && b[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) {
            bsize = (size_t)(b[o + 4] | ((size_t)b[o + 5] << 8)) + 1;
            return bsize >= 12 + xlen + 8 && bsize <= n;
        }
        o += 4 + slen;
    }
    return false;
}

static int inflate_bgzf(const uint8_t *in, size_t n, ByteVec &out, std::string &err, bool &is_bgzf) {
    struct Blk { size_t in_off, in_len, out_off, out_len, hdr; };
    std::vector<Blk> blocks;
    size_t p = 0, total = 0;
    is_bgzf = false;
    while (p < n) {
        size_t bs = 0;
        if (!bgzf_block(in + p, n - p, bs)) { if (blocks.empty()) return 0; err = "BGZF block chain broken"; is_bgzf = true; return -3; }
        const size_t xlen = in[p + 10] | ((size_t)in[p + 11] << 8);
        const uint8_t *t = in + p + bs - 4;
        const size_t isize = t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
        blocks.push_back({p, bs, total, isize, 12 + xlen});
        total += isize; p += bs;
    }
    is_bgzf = true;
    out.resize(total);
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
    if (nt > blocks.size()) nt = (unsigned)blocks.size();
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        z_stream zs;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= blocks.size() || bad.load()) break;
            const Blk &b = blocks[i];
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; break; }
            zs.next_in = (Bytef *)(in + b.in_off + b.hdr); zs.avail_in = (uInt)(b.in_len - b.hdr - 8);
            zs.next_out = out.data() + b.out_off; zs.avail_out = (uInt)b.out_len;
            const int rc = inflate(&zs, Z_FINISH);
            const bool ok = (rc == Z_STREAM_END) && zs.avail_out == 0;
            inflateEnd(&zs);
            if (!ok) { bad = 1; break; }
        }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; i++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (bad.load()) { err = "gzip stream corrupt"; return -3; }
    return 0;
}

int maybe_inflate(const uint8_t *buf, size_t n, ByteVec &storage, const uint8_t *&p, size_t &pn,
                  std::string &err) {
    p = buf; pn = n;
    if (n >= 2 && buf[0] == 0x1F && buf[1] == 0x8B) {
        bool is_bgzf = false;
        if (int rc = inflate_bgzf(buf, n, storage, err, is_bgzf)) return rc;
        if (!is_bgzf) if (int rc = inflate_all(buf, n, storage, err)) return rc;
        p = storage.data(); pn = storage.size();
    }
    return 0;
}

// both files of a pair at once (a plain gzip member cannot be split, two files can)
int maybe_inflate_pair(const uint8_t *b1, size_t n1, const uint8_t *b2, size_t n2, ByteVec &s1,
                       ByteVec &s2, const uint8_t *&p1, size_t &l1, const uint8_t *&p2, size_t &l2,
                       std::string &err) {
    p2 = nullptr; l2 = 0;
    if (!b2) return maybe_inflate(b1, n1, s1, p1, l1, err);
    int rc2 = 0; std::string err2;
    std::thread t([&]() { rc2 = maybe_inflate(b2, n2, s2, p2, l2, err2); });
    const int rc1 = maybe_inflate(b1, n1, s1, p1, l1, err);
    t.join();
    if (rc1) return rc1;
    if (rc2) { err = err2; return rc2; }
    return 0;
}

namespace {
struct Lut {
    uint8_t code[256];
    Lut() {
        memset(code, 4, sizeof code);
        code[(int)'A'] = code[(int)'a'] = 0; code[(int)'C'] = code[(int)'c'] = 1;
        code[(int)'G'] = code[(int)'g'] = 2; code[(int)'T'] = code[(int)'t'] = 3;
    }
};
const Lut LUT;

inline void append_run(PackedReads &o, const uint8_t *codes, size_t len) {
    uint64_t nb = o.n_bases; uint32_t cur = o.cur;
    for (size_t i = 0; i < len; i++) {
        const uint32_t sh = 2u * (uint32_t)(nb & 15);
        cur |= (uint32_t)codes[i] << sh;
        nb++;
        if ((nb & 15) == 0) { o.bases.push_back(cur); cur = 0; }
    }
    o.n_bases = nb; o.cur = cur;
    o.seg_off.push_back((uint32_t)nb);
}
}  // namespace

int pack_fastq(const uint8_t *buf, size_t n, uint32_t k, uint32_t min_qual, PackedReads &out,
               std::string &err, uint64_t every, const ProgressFn &progress,
               uint64_t flush_reads, uint64_t flush_bases, const FlushFn &flush, uint64_t rec_base) {
    ByteVec inflated;
    if (n >= 2 && buf[0] == 0x1F && buf[1] == 0x8B) {
        const uint8_t *q = nullptr; size_t qn = 0;
        if (int rc = maybe_inflate(buf, n, inflated, q, qn, err)) return rc;
        buf = q; n = qn;
    }
    if (out.seg_off.empty()) out.seg_off.push_back(0);
    std::vector<uint8_t> run;
    size_t p = 0; uint64_t rec = rec_base;
    const int minq = (int)min_qual;
    while (p < n) {
        if (buf[p] == '\n') { p++; continue; }
        if (buf[p] == '\r' && p + 1 < n && buf[p + 1] == '\n') { p += 2; continue; }
        const uint8_t *line[4]; size_t ll[4];
        for (int i = 0; i < 4; i++) {
            if (p >= n) { err = "truncated FASTQ record " + std::to_string(rec); return -3; }
            const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', n - p);
            if (!nl && i < 3) { err = "truncated FASTQ record " + std::to_string(rec); return -3; }
            size_t e = nl ? (size_t)(nl - buf) : n;
            line[i] = buf + p; ll[i] = e - p;
            if (ll[i] > 0 && line[i][ll[i] - 1] == '\r') ll[i]--;
            p = e + 1;
        }
        if (ll[0] == 0 || line[0][0] != '@' || ll[2] == 0 || line[2][0] != '+' || ll[1] != ll[3]) {
            err = "malformed FASTQ record " + std::to_string(rec); return -3;
        }
        // SPEC S2: maximal runs of valid bases; runs shorter than k are dropped
        const uint8_t *s = line[1], *q = line[3];
        const size_t L = ll[1];
        run.clear();
        for (size_t i = 0; i <= L; i++) {
            uint8_t c = 4;
            if (i < L) {
                c = LUT.code[s[i]];
                if (c < 4 && (int)q[i] - 33 < minq) c = 4;
            }
            if (c < 4) { run.push_back(c); continue; }
            if (run.size() >= k) {
                if (out.n_bases + run.size() + run.size() / 256 >= 0xFFFFFFF0ull) { err = "input exceeds 2^32 bases per batch"; return -1; }
                // the kernels take segments of at most 32768 bases: a longer run goes in as pieces that
                // overlap by k-1 bases, so that every k-mer window lies in exactly one piece
                const size_t kPiece = 16384;
                if (run.size() <= kPiece + k - 1) append_run(out, run.data(), run.size());
                else for (size_t p0 = 0; p0 + k - 1 < run.size(); p0 += kPiece)
                    append_run(out, run.data() + p0, std::min(run.size() - p0, kPiece + k - 1));
            }
            run.clear();
        }
        out.n_input_bases += L;
        out.n_reads++; rec++;
        if (every && progress && (out.n_reads % every) == 0) progress(out.n_reads, p > n ? n : p, n);
        if (flush && ((flush_reads && (out.n_reads % flush_reads) == 0) || (flush_bases && out.n_bases >= flush_bases))) {
            if (int rc = flush(out)) { err = "batch hand-over failed"; return rc; }
            if (out.seg_off.empty()) out.seg_off.push_back(0);
        }
    }
    return 0;
}

}  // namespace shk
