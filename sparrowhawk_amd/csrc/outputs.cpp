// outputs.cpp — see outputs.h.  The four texts are produced directly in their JSON-escaped form
// (the only consumer is get_assembly(), Assembler.ts:127): sequences are copied in bulk, only
// the short separators need escaping, so a 5 Mbp contig costs three memcpys, not a per-byte scan.
#include "outputs.h"

#include <string.h>

#include <algorithm>
#include <mutex>
#include <tuple>
#include <unordered_map>

namespace shk {

static inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }

static std::string revcomp(const std::string &s) {
    std::string r(s.size(), 'A');
    const size_t n = s.size();
    for (size_t i = 0; i < n; i++) r[i] = comp(s[n - 1 - i]);
    return r;
}

// is revcomp(s) < s ?  decided at the first differing position, without materialising it
static bool revcomp_is_smaller(const std::string &s) {
    const size_t n = s.size();
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
    s.clear();
    return s;
}
void give_big_string(std::string &&s) {
    if (s.capacity() < (1u << 20)) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() < 8) g_pool.push_back(std::move(s));
}

namespace {
// appends JSON-escaped text: NL/TAB/QUOTE are the only specials the writers produce
struct Esc {
    std::string s;
    void nl() { s += "\\n"; }
    void tab() { s += "\\t"; }
    void quote() { s += "\\\""; }
    void raw(const char *p) { s += p; }                      // text without specials
    void raw(const std::string &p) { s += p; }
    void num(uint64_t v) { s += std::to_string(v); }
};
}  // namespace

void build_assembly_text(std::vector<RawContig> &contigs, uint32_t k, AssemblyText &out) {
    // SPEC S10: each unitig is emitted as min(seq, revcomp(seq))
    for (auto &c : contigs)
        if (revcomp_is_smaller(c.seq)) c.seq = revcomp(c.seq);
    // SPEC S11: order by (length desc, sequence asc)
    std::sort(contigs.begin(), contigs.end(), [](const RawContig &a, const RawContig &b) {
        if (a.seq.size() != b.seq.size()) return a.seq.size() > b.seq.size();
        return a.seq < b.seq;
    });
    const size_t nc = contigs.size();
    out.ncontigs = nc;

    // links: first k-mer of every (contig, orientation) -> id; '+' entries win over '-'
    std::unordered_map<std::string, uint64_t> head;
    head.reserve(nc * 2 + 1);
    std::vector<std::string> tail_plus(nc), tail_minus(nc);
    for (size_t i = 0; i < nc; i++) head.emplace(contigs[i].seq.substr(0, k), i * 2);
    for (size_t i = 0; i < nc; i++) {
        const std::string &s = contigs[i].seq;
        std::string last = s.substr(s.size() - k, k);
        std::string first = s.substr(0, k);
        head.emplace(revcomp(last), i * 2 + 1);       // first k-mer of the '-' orientation
        tail_plus[i] = std::move(last);               // last k-mer of '+'
        tail_minus[i] = revcomp(first);               // last k-mer of '-'
    }
    typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t> Link;
    std::vector<Link> links;
    const char B[4] = {'A', 'C', 'G', 'T'};
    for (size_t i = 0; i < nc; i++) for (uint32_t o = 0; o < 2; o++) {
        const std::string &t = o ? tail_minus[i] : tail_plus[i];
        std::string cand = t.substr(1) + "A";
        for (int b = 0; b < 4; b++) {
            cand[k - 1] = B[b];
            auto it = head.find(cand);
            if (it == head.end()) continue;
            const uint32_t cj = (uint32_t)(it->second >> 1), oj = (uint32_t)(it->second & 1);
            Link L((uint32_t)i + 1, o, cj + 1, oj), M(cj + 1, !oj, (uint32_t)i + 1, !o);
            links.push_back(M < L ? M : L);
        }
    }
    std::sort(links.begin(), links.end());
    links.erase(std::unique(links.begin(), links.end()), links.end());

    size_t seq_bytes = 0;
    for (auto &c : contigs) seq_bytes += c.seq.size();
    Esc fa, g1, g2, dt;
    fa.s = take_big_string(); g1.s = take_big_string(); g2.s = take_big_string();
    fa.s.reserve(seq_bytes + nc * 64 + 16);
    g1.s.reserve(seq_bytes + nc * 80 + links.size() * 40 + 32);
    g2.s.reserve(seq_bytes + nc * 80 + links.size() * 64 + 32);
    g1.raw("H"); g1.tab(); g1.raw("VN:Z:1.0"); g1.nl();
    g2.raw("H"); g2.tab(); g2.raw("VN:Z:2.0"); g2.nl();
    dt.raw("digraph sparrowhawk {"); dt.nl();
    for (size_t i = 0; i < nc; i++) {
        const std::string id = std::to_string(i + 1), len = std::to_string(contigs[i].seq.size()),
                          kc = std::to_string(contigs[i].kc);
        fa.raw(">contig_"); fa.raw(id); fa.raw(" len="); fa.raw(len); fa.raw(" kc="); fa.raw(kc); fa.nl();
        fa.raw(contigs[i].seq); fa.nl();
        g1.raw("S"); g1.tab(); g1.raw(id); g1.tab(); g1.raw(contigs[i].seq); g1.tab(); g1.raw("LN:i:"); g1.raw(len);
        g1.tab(); g1.raw("KC:i:"); g1.raw(kc); g1.nl();
        g2.raw("S"); g2.tab(); g2.raw(id); g2.tab(); g2.raw(len); g2.tab(); g2.raw(contigs[i].seq); g2.tab();
        g2.raw("KC:i:"); g2.raw(kc); g2.nl();
        dt.raw("  "); dt.quote(); dt.raw(id); dt.quote(); dt.raw(" [label="); dt.quote(); dt.raw(id); dt.raw(" len=");
        dt.raw(len); dt.raw(" kc="); dt.raw(kc); dt.quote(); dt.raw("];"); dt.nl();
    }
    const std::string ov = std::to_string(k - 1);
    for (const Link &L : links) {
        const uint32_t a = std::get<0>(L), ao = std::get<1>(L), b = std::get<2>(L), bo = std::get<3>(L);
        const uint64_t la = contigs[a - 1].seq.size(), lb = contigs[b - 1].seq.size();
        g1.raw("L"); g1.tab(); g1.num(a); g1.tab(); g1.raw(ao ? "-" : "+"); g1.tab(); g1.num(b); g1.tab();
        g1.raw(bo ? "-" : "+"); g1.tab(); g1.raw(ov); g1.raw("M"); g1.nl();
        g2.raw("E"); g2.tab(); g2.raw("*"); g2.tab(); g2.num(a); g2.raw(ao ? "-" : "+"); g2.tab(); g2.num(b);
        g2.raw(bo ? "-" : "+"); g2.tab();
        if (!ao) { g2.num(la - (k - 1)); g2.tab(); g2.num(la); g2.raw("$"); g2.tab(); }
        else { g2.raw("0"); g2.tab(); g2.raw(ov); if ((uint64_t)(k - 1) == la) g2.raw("$"); g2.tab(); }
        if (!bo) { g2.raw("0"); g2.tab(); g2.raw(ov); if ((uint64_t)(k - 1) == lb) g2.raw("$"); g2.tab(); }
        else { g2.num(lb - (k - 1)); g2.tab(); g2.num(lb); g2.raw("$"); g2.tab(); }
        g2.raw(ov); g2.raw("M"); g2.nl();
        dt.raw("  "); dt.quote(); dt.num(a); dt.quote(); dt.raw(" -> "); dt.quote(); dt.num(b); dt.quote();
        dt.raw(" [label="); dt.quote(); dt.raw(ao ? "-" : "+"); dt.raw(bo ? "-" : "+"); dt.quote(); dt.raw("];"); dt.nl();
    }
    dt.raw("}"); dt.nl();

    std::string &js = out.json;
    js = take_big_string();
    js.reserve(fa.s.size() + g1.s.size() + g2.s.size() + dt.s.size() + 128);
    js += "{\"outfasta\":\""; js += fa.s;
    js += "\",\"ncontigs\":" + std::to_string(nc);
    js += ",\"outdot\":\""; js += dt.s;
    js += "\",\"outgfa\":\""; js += g1.s;
    js += "\",\"outgfav2\":\""; js += g2.s;
    js += "\"}";
    give_big_string(std::move(fa.s)); give_big_string(std::move(g1.s)); give_big_string(std::move(g2.s));
    out.fasta.clear(); out.gfa1.clear(); out.gfa2.clear(); out.dot.clear();
}

}  // namespace shk
