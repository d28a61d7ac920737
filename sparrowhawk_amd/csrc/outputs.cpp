// outputs.cpp — see outputs.h.  The four texts are produced directly in their JSON-escaped form
// (the only consumer is get_assembly(), Assembler.ts:127): sequences are copied in bulk, only
// the short separators need escaping, so a 5 Mbp contig costs three memcpys, not a per-byte scan.
#include "outputs.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <thread>
#include <tuple>
#include <unordered_map>

namespace shk {

static inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }

static std::string revcomp(const char *s, size_t n) {
    std::string r(n, 'A');
    for (size_t i = 0; i < n; i++) r[i] = comp(s[n - 1 - i]);
    return r;
}

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

// Large output strings are recycled across handles: a fresh 15 MB std::string costs page faults
// on every assemble (measured 1 -> 4 ms), a recycled one does not.
static std::mutex g_pool_mu;
static std::vector<std::string> g_pool;
std::string take_big_string() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.empty()) return std::string();
    size_t best = 0;
    for (size_t i = 1; i < g_pool.size(); i++) if (g_pool[i].capacity() > g_pool[best].capacity()) best = i;
    std::string s = std::move(g_pool[best]);
    g_pool.erase(g_pool.begin() + best);
    return s;                                           // (with its old size: the writer overwrites it)
}
void give_big_string(std::string &&s) {
    if (s.capacity() < (1u << 20)) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() < 8) g_pool.push_back(std::move(s));
}

namespace {
// The writers emit JSON-escaped text (NL/TAB/QUOTE are the only specials they produce) into a sink: one
// sink measures, the other writes through a pointer — every section's size is known before a byte is
// written, so the sections go straight to their places in one buffer, the big ones in parallel.
static inline size_t n_digits(uint64_t v) { size_t n = 1; while (v >= 10) { v /= 10; n++; } return n; }
struct SizeSink {
    size_t n = 0;
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
    void nl() { *p++ = '\\'; *p++ = 'n'; }
    void tab() { *p++ = '\\'; *p++ = 't'; }
    void quote() { *p++ = '\\'; *p++ = '"'; }
    void raw(const char *q) { const size_t m = strlen(q); memcpy(p, q, m); p += m; }
    void raw(const std::string &q) { memcpy(p, q.data(), q.size()); p += q.size(); }
    void raw(const char *q, size_t m) { memcpy(p, q, m); p += m; }
    void num(uint64_t v) { const size_t m = n_digits(v); for (size_t i = m; i-- > 0;) { p[i] = (char)('0' + v % 10); v /= 10; } p += m; }
};
}  // namespace

void build_assembly_text(std::vector<RawContig> &contigs, uint32_t k, AssemblyText &out) {
    // SPEC S10: each unitig is emitted as min(seq, revcomp(seq))
    for (auto &c : contigs)
        if (revcomp_is_smaller(c.data(), c.size())) { c.own = revcomp(c.data(), c.size()); c.ext = nullptr; c.ext_n = 0; }
    // SPEC S11: order by (length desc, sequence asc)
    std::sort(contigs.begin(), contigs.end(), [](const RawContig &a, const RawContig &b) {
        if (a.size() != b.size()) return a.size() > b.size();
        return memcmp(a.data(), b.data(), a.size()) < 0;
    });
    const size_t nc = contigs.size();
    out.ncontigs = nc;

    // links: first k-mer of every (contig, orientation) -> id; '+' entries win over '-'.  K-mers are handled
    // 2-bit packed (4 words hold k <= 127): a fragmented assembly has 10^4..10^6 contigs and this map is
    // the whole cost of the writer then.
    struct Key {
        uint64_t w[4];
        bool operator==(const Key &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
    };
    struct KeyHash {
        size_t operator()(const Key &x) const {
            uint64_t h = 0x9e3779b97f4a7c15ull;
            for (int i = 0; i < 4; i++) { h ^= x.w[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
            return (size_t)h;
        }
    };
    auto code = [](char c) -> uint64_t { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; };
    auto shl2 = [](Key &x, uint64_t b) {                       // x = x * 4 + b
        x.w[3] = (x.w[3] << 2) | (x.w[2] >> 62); x.w[2] = (x.w[2] << 2) | (x.w[1] >> 62);
        x.w[1] = (x.w[1] << 2) | (x.w[0] >> 62); x.w[0] = (x.w[0] << 2) | b;
    };
    auto mask_k = [&](Key &x) {                                // keep the low 2k bits
        const uint32_t used = 2 * k;
        for (uint32_t i = 0; i < 4; i++) {
            if (64 * i >= used) x.w[i] = 0;
            else if (used - 64 * i < 64) x.w[i] &= (1ull << (used - 64 * i)) - 1ull;
        }
    };
    auto pack = [&](const char *p) {                           // first base in the top 2 bits of the 2k-bit value
        Key x{{0, 0, 0, 0}};
        for (uint32_t i = 0; i < k; i++) { const uint32_t pos = 2 * (k - 1 - i); x.w[pos >> 6] |= code(p[i]) << (pos & 63); }
        return x;
    };
    auto pack_rc = [&](const char *p) {                        // reverse complement of p[0..k)
        Key x{{0, 0, 0, 0}};
        for (uint32_t i = 0; i < k; i++) { const uint32_t pos = 2 * i; x.w[pos >> 6] |= (3 - code(p[i])) << (pos & 63); }
        return x;
    };
    // flat open-addressing table (the first entry of a key wins, like unordered_map::emplace)
    size_t cap = 16; while (cap < 4 * nc + 4) cap <<= 1;
    std::vector<Key> hk(cap); std::vector<uint64_t> hv(cap, ~0ull);
    const KeyHash hasher;
    auto put = [&](const Key &x, uint64_t v) {
        size_t s = hasher(x) & (cap - 1);
        while (hv[s] != ~0ull) { if (hk[s] == x) return; s = (s + 1) & (cap - 1); }
        hk[s] = x; hv[s] = v;
    };
    auto get = [&](const Key &x) -> uint64_t {
        size_t s = hasher(x) & (cap - 1);
        while (hv[s] != ~0ull) { if (hk[s] == x) return hv[s]; s = (s + 1) & (cap - 1); }
        return ~0ull;
    };
    std::vector<Key> tail_plus(nc), tail_minus(nc);
    for (size_t i = 0; i < nc; i++) put(pack(contigs[i].data()), i * 2);
    for (size_t i = 0; i < nc; i++) {
        const char *first = contigs[i].data(), *last = contigs[i].data() + contigs[i].size() - k;
        put(pack_rc(last), i * 2 + 1);                  // first k-mer of the '-' orientation
        tail_plus[i] = pack(last);                      // last k-mer of '+'
        tail_minus[i] = pack_rc(first);                 // last k-mer of '-'
    }
    typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t> Link;
    std::vector<Link> links;
    for (size_t i = 0; i < nc; i++) for (uint32_t o = 0; o < 2; o++) {
        Key cand = o ? tail_minus[i] : tail_plus[i];
        shl2(cand, 0); mask_k(cand);                    // drop the first base, append A
        for (uint64_t b = 0; b < 4; b++) {
            cand.w[0] = (cand.w[0] & ~3ull) | b;
            const uint64_t hit = get(cand);
            if (hit == ~0ull) continue;
            const uint32_t cj = (uint32_t)(hit >> 1), oj = (uint32_t)(hit & 1);
            Link L((uint32_t)i + 1, o, cj + 1, oj), M(cj + 1, !oj, (uint32_t)i + 1, !o);
            links.push_back(M < L ? M : L);
        }
    }
    std::sort(links.begin(), links.end());
    links.erase(std::unique(links.begin(), links.end()), links.end());

    // Straight into the JSON text in key order (a 5 Mbp contig is copied three times — FASTA, GFA1, GFA2 —
    // and never staged in per-format strings).
    size_t seq_bytes = 0;
    for (auto &c : contigs) seq_bytes += c.size();
    std::vector<std::string> ids(nc), lens(nc), kcs(nc);
    for (size_t i = 0; i < nc; i++) {
        ids[i] = std::to_string(i + 1); lens[i] = std::to_string(contigs[i].size()); kcs[i] = std::to_string(contigs[i].kc);
    }
    const std::string ov = std::to_string(k - 1);
    auto sec_fasta = [&](auto &w) {
        w.raw("{\"outfasta\":\"");
        for (size_t i = 0; i < nc; i++) {
            w.raw(">contig_"); w.raw(ids[i]); w.raw(" len="); w.raw(lens[i]); w.raw(" kc="); w.raw(kcs[i]); w.nl();
            w.raw(contigs[i].data(), contigs[i].size()); w.nl();
        }
        w.raw("\",\"ncontigs\":"); w.num(nc);
    };
    auto sec_dot = [&](auto &w) {
        w.raw(",\"outdot\":\"");
        w.raw("digraph sparrowhawk {"); w.nl();
        for (size_t i = 0; i < nc; i++) {
            w.raw("  "); w.quote(); w.raw(ids[i]); w.quote(); w.raw(" [label="); w.quote(); w.raw(ids[i]); w.raw(" len=");
            w.raw(lens[i]); w.raw(" kc="); w.raw(kcs[i]); w.quote(); w.raw("];"); w.nl();
        }
        for (const Link &L : links) {
            const uint32_t a = std::get<0>(L), ao = std::get<1>(L), b = std::get<2>(L), bo = std::get<3>(L);
            w.raw("  "); w.quote(); w.num(a); w.quote(); w.raw(" -> "); w.quote(); w.num(b); w.quote();
            w.raw(" [label="); w.quote(); w.raw(ao ? "-" : "+"); w.raw(bo ? "-" : "+"); w.quote(); w.raw("];"); w.nl();
        }
        w.raw("}"); w.nl();
    };
    auto sec_gfa1 = [&](auto &w) {
        w.raw("\",\"outgfa\":\"");
        w.raw("H"); w.tab(); w.raw("VN:Z:1.0"); w.nl();
        for (size_t i = 0; i < nc; i++) {
            w.raw("S"); w.tab(); w.raw(ids[i]); w.tab(); w.raw(contigs[i].data(), contigs[i].size()); w.tab(); w.raw("LN:i:"); w.raw(lens[i]);
            w.tab(); w.raw("KC:i:"); w.raw(kcs[i]); w.nl();
        }
        for (const Link &L : links) {
            const uint32_t a = std::get<0>(L), ao = std::get<1>(L), b = std::get<2>(L), bo = std::get<3>(L);
            w.raw("L"); w.tab(); w.num(a); w.tab(); w.raw(ao ? "-" : "+"); w.tab(); w.num(b); w.tab();
            w.raw(bo ? "-" : "+"); w.tab(); w.raw(ov); w.raw("M"); w.nl();
        }
    };
    auto sec_gfa2 = [&](auto &w) {
        w.raw("\",\"outgfav2\":\"");
        w.raw("H"); w.tab(); w.raw("VN:Z:2.0"); w.nl();
        for (size_t i = 0; i < nc; i++) {
            w.raw("S"); w.tab(); w.raw(ids[i]); w.tab(); w.raw(lens[i]); w.tab(); w.raw(contigs[i].data(), contigs[i].size()); w.tab();
            w.raw("KC:i:"); w.raw(kcs[i]); w.nl();
        }
        for (const Link &L : links) {
            const uint32_t a = std::get<0>(L), ao = std::get<1>(L), b = std::get<2>(L), bo = std::get<3>(L);
            const uint64_t la = contigs[a - 1].size(), lb = contigs[b - 1].size();
            w.raw("E"); w.tab(); w.raw("*"); w.tab(); w.num(a); w.raw(ao ? "-" : "+"); w.tab(); w.num(b);
            w.raw(bo ? "-" : "+"); w.tab();
            if (!ao) { w.num(la - (k - 1)); w.tab(); w.num(la); w.raw("$"); w.tab(); }
            else { w.raw("0"); w.tab(); w.raw(ov); if ((uint64_t)(k - 1) == la) w.raw("$"); w.tab(); }
            if (!bo) { w.raw("0"); w.tab(); w.raw(ov); if ((uint64_t)(k - 1) == lb) w.raw("$"); w.tab(); }
            else { w.num(lb - (k - 1)); w.tab(); w.num(lb); w.raw("$"); w.tab(); }
            w.raw(ov); w.raw("M"); w.nl();
        }
        w.raw("\"}");
    };
    SizeSink z1, z2, z3, z4;
    sec_fasta(z1); sec_dot(z2); sec_gfa1(z3); sec_gfa2(z4);
    const size_t total = z1.n + z2.n + z3.n + z4.n;
    // (a recycled string keeps its size: growing it is the only time its bytes are filled twice)
    std::string js = take_big_string();
    if (js.size() < total) js.resize(total);
    char *base = &js[0];
    auto run = [&](int which) {
        if (which == 0) { PtrSink w{base}; sec_fasta(w); PtrSink d{base + z1.n}; sec_dot(d); }
        else if (which == 1) { PtrSink w{base + z1.n + z2.n}; sec_gfa1(w); }
        else { PtrSink w{base + z1.n + z2.n + z3.n}; sec_gfa2(w); }
    };
    const char *pm = getenv("SHK_WRITER_PAR_MIN");          // (tests force the threaded path on small outputs)
    const size_t par_min = (pm && *pm) ? (size_t)strtoull(pm, nullptr, 10) : ((size_t)1 << 20);
    if (seq_bytes >= par_min) {                             // three copies of megabytes: one thread each
        std::thread t1(run, 1), t2(run, 2);
        run(0);
        t1.join(); t2.join();
    } else { run(0); run(1); run(2); }
    js.resize(total);
    out.json = std::move(js);
    out.fasta.clear(); out.gfa1.clear(); out.gfa2.clear(); out.dot.clear();
}

}  // namespace shk
