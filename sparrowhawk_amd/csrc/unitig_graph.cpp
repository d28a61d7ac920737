// unitig_graph.cpp — SPEC S9 / S10 on unitig records (see unitig_graph.h for why that is exact).
#include "unitig_graph.h"

#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <thread>

#include "kmer.h"

namespace shk {
namespace {

// A metagenome leaves millions of unitig records (an isolate a handful): the passes over ALL records — sorting the ends,
// looking up every record's out-neighbours once, finding the tip / bubble candidates of a round — run on several host threads
// from a size on; the walks themselves start from the few candidates.
unsigned ug_threads(size_t n) {
    if (n < 65536) return 1;
    unsigned t = std::thread::hardware_concurrency();
    return t < 1 ? 1 : (t > 32 ? 32 : t);
}
template <typename F> void par_ranges(size_t n, F &&fn) {
    const unsigned T = ug_threads(n);
    if (T == 1) { fn((size_t)0, n, 0u); return; }
    std::vector<std::thread> ts;
    for (unsigned t = 1; t < T; t++) ts.emplace_back([&fn, n, t, T] { fn(n * t / T, n * (t + 1) / T, t); });
    fn((size_t)0, n / T, 0u);
    for (auto &t : ts) t.join();
}
template <typename It, typename Less> void par_sort(It b, It e, Less less) {
    const size_t n = (size_t)(e - b);
    unsigned C = 1;
    while (C * 2 <= ug_threads(n) && n / (C * 2) >= 32768) C *= 2;
    if (C == 1) { std::sort(b, e, less); return; }
    std::vector<size_t> edge(C + 1);
    for (unsigned c = 0; c <= C; c++) edge[c] = n * c / C;
    { std::vector<std::thread> ts;
      for (unsigned c = 0; c < C; c++) ts.emplace_back([&, c] { std::sort(b + edge[c], b + edge[c + 1], less); });
      for (auto &t : ts) t.join(); }
    for (unsigned w = 1; w < C; w *= 2) {
        std::vector<std::thread> ts;
        for (unsigned j = 0; j < C / (2 * w); j++)
            ts.emplace_back([&, j, w] { std::inplace_merge(b + edge[2 * w * j], b + edge[2 * w * j + w], b + edge[2 * w * j + 2 * w], less); });
        for (auto &t : ts) t.join();
    }
}

template <int W> struct UG {
    const int k;
    const std::vector<UnitigRec> &R;
    const uint32_t n;
    std::vector<Kmer<W>> F, T;                   // first / last k-mer of every record, as spelled
    std::vector<uint32_t> mirror;
    std::vector<uint8_t> alive;
    std::vector<uint32_t> outn;                  // [n][4] the records whose first k-mer overlaps a record's last k-mer (UG_NIL padded), looked up once
    uint64_t T_LEN;                              // T_TIP = T_BUB = 2k nodes

    UG(int k_, const std::vector<UnitigRec> &recs) : k(k_), R(recs), n((uint32_t)recs.size()), T_LEN(2ull * (uint64_t)k_) {}

    static Kmer<W> load(const uint64_t *w) { Kmer<W> x; for (int i = 0; i < W; i++) x.w[i] = w[i]; return x; }
    Kmer<W> prefix(const Kmer<W> &x) const {      // the first k-1 bases as a 2(k-1)-bit integer
        Kmer<W> r;
        for (int i = 0; i < W; i++) r.w[i] = (x.w[i] >> 2) | (i + 1 < W ? x.w[i + 1] << 62 : 0ull);
        return r;
    }
    Kmer<W> suffix(const Kmer<W> &x) const {      // the last k-1 bases
        Kmer<W> r = x;
        const int bits = 2 * (k - 1);
        for (int i = 0; i < W; i++) {
            const int lo = 64 * i;
            if (bits <= lo) r.w[i] = 0;
            else if (bits < lo + 64) r.w[i] &= (1ull << (bits - lo)) - 1ull;
        }
        return r;
    }
    Kmer<W> canon(const Kmer<W> &x, int &o) const { return km_canonical<W>(x, k, o); }

    // Index of the chain starts: an open-addressing table of record numbers, placed by a hash of the first k-1 bases of the
    // record's first k-mer.  One probe sequence answers both questions a record's END asks: "which records start with my last
    // k-1 bases" (its out-neighbours, <= 4) and "which record starts with revcomp(my last k-mer)" (its mirror strand).  Built
    // and read by all threads (one CAS per insert); replaces two sorts of all records and two binary searches per record
    // (4 M records: 2.3 s -> 0.3 s on 8 cores).
    static uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }
    uint64_t hash_of(const Kmer<W> &p) const { uint64_t h = 0x9E3779B97F4A7C15ull; for (int i = 0; i < W; i++) h = mix(h ^ p.w[i]); return h; }

    int init(std::string &err) {
        const bool dbg = getenv("SHK_UG_DEBUG") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!dbg) return;
            const auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "[unitig graph]   init: %-8s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        };
        F.resize(n); T.resize(n); mirror.assign(n, UG_NIL); alive.assign(n, 1);
        par_ranges(n, [&](size_t a, size_t b, unsigned) { for (size_t r = a; r < b; r++) { F[r] = load(R[r].first); T[r] = load(R[r].last); } });
        lap("load");
        size_t cap = 64;
        while (cap < (size_t)n * 2 + 16) cap <<= 1;
        const size_t cmask = cap - 1;
        // a slot = hash tag : record number — a probe only touches a record's k-mers when the tag agrees
        constexpr uint64_t EMPTY = ~0ull;
        std::unique_ptr<std::atomic<uint64_t>[]> slot(new std::atomic<uint64_t>[cap]);
        par_ranges(cap, [&](size_t a, size_t b, unsigned) { for (size_t i = a; i < b; i++) slot[i].store(EMPTY, std::memory_order_relaxed); });
        std::atomic<int> bad{0};
        par_ranges(n, [&](size_t a, size_t b, unsigned) {
            for (size_t r = a; r < b; r++) {
                if (R[r].circ) continue;
                const uint64_t h = hash_of(prefix(F[r])), mine = (h & 0xFFFFFFFF00000000ull) | (uint64_t)r;
                for (size_t i = (size_t)h & cmask;; i = (i + 1) & cmask) {
                    uint64_t e = slot[i].load(std::memory_order_acquire);
                    if (e == EMPTY && slot[i].compare_exchange_strong(e, mine, std::memory_order_acq_rel)) break;
                    // (e holds the slot's entry now) a record with the same start: of two such records the later one meets the earlier
                    if ((e >> 32) == (mine >> 32) && km_eq<W>(F[(uint32_t)e], F[r])) { bad |= 2; break; }
                }
            }
        });
        lap("insert");
        outn.assign((size_t)n * 4, UG_NIL);
        lap("outn");
        par_ranges(n, [&](size_t a, size_t b, unsigned) {
            for (size_t r = a; r < b; r++) {
                if (R[r].circ) continue;
                const Kmer<W> want = km_revcomp<W>(T[r], k);
                const uint64_t hw = hash_of(prefix(want));
                uint32_t m = UG_NIL;
                for (size_t i = (size_t)hw & cmask;; i = (i + 1) & cmask) {
                    const uint64_t e = slot[i].load(std::memory_order_relaxed);
                    if (e == EMPTY) break;
                    if ((e >> 32) == (hw >> 32) && km_eq<W>(F[(uint32_t)e], want)) { m = (uint32_t)e; break; }
                }
                if (m == UG_NIL) { bad |= 1; continue; }
                mirror[r] = m;
                const Kmer<W> sfx = suffix(T[r]);
                const uint64_t hs = hash_of(sfx);
                uint32_t found[4]; int c = 0;
                for (size_t i = (size_t)hs & cmask;; i = (i + 1) & cmask) {
                    const uint64_t e = slot[i].load(std::memory_order_relaxed);
                    if (e == EMPTY) break;
                    if (c < 4 && (e >> 32) == (hs >> 32) && km_eq<W>(prefix(F[(uint32_t)e]), sfx)) found[c++] = (uint32_t)e;
                }
                std::sort(found, found + c);                           // (ascending record numbers, as a sorted index would list them)
                for (int q = 0; q < c; q++) outn[r * 4 + q] = found[q];
            }
        });
        lap("lookup");
        if (bad.load() & 2) { err = "unitig graph: two chains start at the same oriented node"; return -1; }
        if (bad.load() & 1) { err = "unitig graph: a chain without its mirror strand"; return -1; }
        std::atomic<int> unpaired{0};
        par_ranges(n, [&](size_t a, size_t b, unsigned) { for (size_t r = a; r < b; r++) if (!R[r].circ && mirror[mirror[r]] != r) unpaired = 1; });
        if (unpaired.load()) { err = "unitig graph: mirror strands do not pair up"; return -1; }
        return 0;
    }

    // out-neighbours of r's last node: the alive records whose first k-mer overlaps its last k-mer by k-1
    int outs(uint32_t r, uint32_t (&o)[4]) const {
        if (R[r].circ) return 0;
        int c = 0;
        for (int i = 0; i < 4; i++) { const uint32_t s = outn[(size_t)r * 4 + i]; if (s == UG_NIL) break; if (alive[s]) o[c++] = s; }
        return c;
    }
    int outdeg(uint32_t r) const { uint32_t o[4]; return outs(r, o); }
    // in-neighbours of r's first node (as records that END there): the mirrors of the out-neighbours of r's mirror
    int ins(uint32_t r, uint32_t (&o)[4]) const {
        if (R[r].circ) return 0;
        const int c = outs(mirror[r], o);
        for (int i = 0; i < c; i++) o[i] = mirror[o[i]];
        return c;
    }
    int indeg(uint32_t r) const { uint32_t o[4]; return ins(r, o); }

    // the records a round starts from, ascending (found by several threads on a large graph)
    template <typename P> std::vector<uint32_t> candidates(P &&pred) const {
        const unsigned T = ug_threads(n);
        std::vector<std::vector<uint32_t>> part(T);
        par_ranges(n, [&](size_t a, size_t b, unsigned t) { for (size_t r = a; r < b; r++) if (pred((uint32_t)r)) part[t].push_back((uint32_t)r); });
        std::vector<uint32_t> all;
        for (auto &p : part) all.insert(all.end(), p.begin(), p.end());
        return all;
    }

    void kill(const std::vector<uint32_t> &doomed, uint64_t &nodes) {
        for (uint32_t r : doomed) {
            if (!alive[r]) continue;
            alive[r] = 0; alive[mirror[r]] = 0;
            nodes += R[r].len;
        }
    }

    uint64_t tip_round() {
        struct Tip { uint64_t len, sum; std::vector<uint32_t> path; };
        std::map<uint32_t, std::vector<Tip>> attached;               // junction record (its first node) -> tips
        // (a start whose own chain is longer than T, or whose end is a dead end or a fork, cannot be a tip: decided in the
        // parallel pass — in a metagenome nearly every record is an isolated unitig)
        for (uint32_t v : candidates([&](uint32_t r) { return alive[r] && !R[r].circ && R[r].len <= T_LEN && indeg(r) == 0 && outdeg(r) == 1; })) {
            Tip t; t.path.push_back(v); t.len = R[v].len; t.sum = R[v].kc;
            if (t.len > T_LEN) continue;                             // |P| > T inside the first chain: not a tip
            uint32_t cur = v;
            for (;;) {
                uint32_t o[4];
                if (outs(cur, o) != 1) break;                        // not a tip
                const uint32_t nx = o[0];
                if (indeg(nx) >= 2) { attached[nx].push_back(std::move(t)); break; }
                t.path.push_back(nx); t.len += R[nx].len; t.sum += R[nx].kc; cur = nx;
                if (t.len > T_LEN) break;                            // not a tip
            }
        }
        std::vector<uint32_t> doomed;
        for (auto &kv : attached) {
            std::vector<Tip> &tips = kv.second;
            const size_t d = (size_t)indeg(kv.first), t = tips.size();
            size_t best = 0;
            if (t == d) {
                for (size_t q = 1; q < t; q++) {                      // max of (|P|, sum of counts, smaller first canonical k-mer)
                    const Tip &A = tips[q], &B = tips[best];
                    bool better;
                    if (A.len != B.len) better = A.len > B.len;
                    else if (A.sum != B.sum) better = A.sum > B.sum;
                    else { int oa, ob; better = km_less<W>(canon(F[A.path[0]], oa), canon(F[B.path[0]], ob)); }
                    if (better) best = q;
                }
            }
            for (size_t q = 0; q < t; q++) {
                if (t == d && q == best) continue;
                for (uint32_t r : tips[q].path) doomed.push_back(r);
            }
        }
        uint64_t nodes = 0;
        kill(doomed, nodes);
        return nodes;
    }

    uint64_t bubble_round() {
        std::vector<uint32_t> doomed;
        for (uint32_t S : candidates([&](uint32_t r) { return alive[r] && !R[r].circ && outdeg(r) >= 2; })) {
            uint32_t ob[4];
            const int no = outs(S, ob);
            struct Branch { std::vector<uint32_t> path; uint64_t len = 0, sum = 0; uint32_t end = UG_NIL; bool ok = false; };
            Branch br[4];
            for (int b = 0; b < no; b++) {
                Branch &B = br[b];
                const uint32_t bn = ob[b];
                if (indeg(bn) != 1) continue;
                B.path.push_back(bn); B.len = R[bn].len; B.sum = R[bn].kc;
                if (B.len > T_LEN) continue;                         // too long inside the first chain
                uint32_t cur = bn;
                for (;;) {
                    uint32_t o[4];
                    if (outs(cur, o) != 1) break;                    // dead end or fork
                    const uint32_t nx = o[0];
                    if (indeg(nx) >= 2) { B.end = nx; B.ok = true; break; }
                    B.path.push_back(nx); B.len += R[nx].len; B.sum += R[nx].kc; cur = nx;
                    if (B.len > T_LEN) break;                        // too long
                }
            }
            for (int a = 0; a < no; a++) {
                if (!br[a].ok) continue;
                const uint32_t E = br[a].end;
                {   // evaluated only from the side with key(S) <= key(rc(E)): S = last node of record S, E = first node of record E
                    int os, oe;
                    const Kmer<W> ks = canon(T[S], os), ke = canon(F[E], oe);
                    const int oe_m = 1 - oe;                         // rc(E): the same k-mer, the other orientation
                    bool le;
                    if (km_less<W>(ks, ke)) le = true; else if (km_less<W>(ke, ks)) le = false; else le = os <= oe_m;
                    if (!le) continue;
                }
                int grp = 0; bool best = true;
                for (int b = 0; b < no; b++) {
                    if (!br[b].ok || br[b].end != E) continue;
                    grp++;
                    if (b == a) continue;
                    // is b better than a?  exact means by cross-multiplication (node counts <= 2k: no overflow in 128 bits, nor in 64 in practice)
                    const unsigned __int128 l = (unsigned __int128)br[b].sum * br[a].len, r = (unsigned __int128)br[a].sum * br[b].len;
                    bool better;
                    if (l != r) better = l > r;
                    else if (br[b].len != br[a].len) better = br[b].len < br[a].len;
                    else { int o1, o2; better = km_less<W>(canon(F[br[b].path[0]], o1), canon(F[br[a].path[0]], o2)); }
                    if (better) best = false;
                }
                if (grp >= 2 && !best) for (uint32_t r : br[a].path) doomed.push_back(r);
            }
        }
        uint64_t nodes = 0;
        kill(doomed, nodes);
        return nodes;
    }

    // S10 on what is left: r -> s is simple iff it is r's only out-edge, s's only in-edge and s's first node is neither
    // r's last node nor its reverse complement
    uint32_t simple_succ(uint32_t r) const {
        uint32_t o[4];
        if (outs(r, o) != 1) return UG_NIL;
        const uint32_t s = o[0];
        if (indeg(s) != 1) return UG_NIL;
        if (km_eq<W>(F[s], T[r]) || km_eq<W>(F[s], km_revcomp<W>(T[r], k))) return UG_NIL;
        return s;
    }

    void chains(UnitigGraphResult &out) {
        const bool dbg = getenv("SHK_UG_DEBUG") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!dbg) return;
            const auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "[unitig graph]   chains: %-8s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        };
        std::vector<uint32_t> succ(n, UG_NIL), pred(n, UG_NIL);
        par_ranges(n, [&](size_t a, size_t b, unsigned) { for (size_t r = a; r < b; r++) if (alive[r] && !R[r].circ) succ[r] = simple_succ((uint32_t)r); });
        par_ranges(n, [&](size_t a, size_t b, unsigned) { for (size_t r = a; r < b; r++) if (succ[r] != UG_NIL) pred[succ[r]] = (uint32_t)r; });   // (a record has one simple predecessor)
        lap("succ");
        std::vector<uint8_t> seen(n, 0);
        auto finish = [&](UnitigContig &c, std::vector<UnitigContig> &to) {
            for (uint32_t r : c.recs) { c.len_nodes += R[r].len; c.kc += R[r].kc; }
            to.push_back(std::move(c));
        };
        // Linear chains start at the records without a simple predecessor; every chain exists on both strands, as two chains
        // with two different first nodes.  SPEC S10 emits min(spelling, revcomp(spelling)), which the first k-mers decide:
        // the strand with the smaller first k-mer is the one handed on (the writer never has to reverse-complement a
        // chromosome), its mirror chain is walked — to mark its records — and dropped.  A chain that is its own mirror
        // (it starts at the reverse complement of its last node) is handed on once.  The heads are independent: several
        // threads, each over a range of records, their contigs appended in the order of the ranges.
        const unsigned TH = ug_threads(n);
        std::vector<std::vector<UnitigContig>> part(TH);
        par_ranges(n, [&](size_t a, size_t b, unsigned t) {
            for (size_t r = a; r < b; r++) {
                if (!alive[r] || R[r].circ || pred[r] != UG_NIL) continue;
                uint32_t tail = (uint32_t)r;
                seen[r] = 1;
                for (uint32_t cur = succ[r]; cur != UG_NIL; cur = succ[cur]) { seen[cur] = 1; tail = cur; }
                const uint32_t mirror_first = mirror[tail];
                if (mirror_first != (uint32_t)r && !km_less<W>(F[r], F[mirror_first])) continue;     // the other strand is the smaller one
                UnitigContig c;
                for (uint32_t cur = (uint32_t)r; cur != UG_NIL; cur = succ[cur]) c.recs.push_back(cur);
                finish(c, part[t]);
            }
        });
        lap("heads");
        size_t total = 0;
        for (auto &p : part) total += p.size();
        out.contigs.reserve(out.contigs.size() + total);
        for (auto &p : part) { for (auto &c : p) out.contigs.push_back(std::move(c)); std::vector<UnitigContig>().swap(p); }
        lap("gather");
        for (uint32_t r = 0; r < n; r++) {                           // what is left closes on itself: a ring made of several records
            if (!alive[r] || R[r].circ || seen[r]) continue;
            UnitigContig c; c.ring = true;
            for (uint32_t cur = r;;) { c.recs.push_back(cur); seen[cur] = 1; cur = succ[cur]; if (cur == r || cur == UG_NIL) break; }
            for (uint32_t x : c.recs) { seen[mirror[x]] = 1; out.need_min.push_back(x); out.need_min.push_back(mirror[x]); }
            finish(c, out.contigs);
        }
        for (uint32_t r = 0; r < n; r++) {                           // rings from the start: each strand a record of its own
            if (!alive[r] || !R[r].circ) continue;
            UnitigContig c; c.ring = true; c.recs.push_back(r);
            out.need_min.push_back(r);
            finish(c, out.contigs);
        }
    }
};

template <int W> int assemble_t(int k, const std::vector<UnitigRec> &recs, bool tips, bool bubbles, UnitigGraphResult &out, std::string &err) {
    const bool dbg = getenv("SHK_UG_DEBUG") != nullptr;            // stage times on stderr
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[unitig graph] %-10s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    UG<W> g(k, recs);
    if (int rc = g.init(err)) return rc;
    lap("init");
    out = UnitigGraphResult();
    if (tips || bubbles) {
        for (int round = 0; round < 32; round++) {                   // MAX_ROUNDS (S9)
            const uint64_t a = tips ? g.tip_round() : 0;
            lap("tips");
            const uint64_t b = bubbles ? g.bubble_round() : 0;
            lap("bubbles");
            out.tips_removed += a; out.bubbles_removed += b; out.rounds++;
            if (a + b == 0) break;
        }
    }
    g.chains(out);
    lap("chains");
    out.mirror = g.mirror;
    return 0;
}

template <int W> int resolve_t(int k, const std::vector<UnitigRec> &recs, const std::vector<UnitigMinKey> &mk, UnitigGraphResult &out, std::string &err) {
    auto key_less = [&](const UnitigMinKey &a, const UnitigMinKey &b) {   // (x, o) lexicographic
        for (int i = W - 1; i >= 0; i--) if (a.key[i] != b.key[i]) return a.key[i] < b.key[i];
        return a.o < b.o;
    };
    auto key_same_kmer = [&](const UnitigMinKey &a, const UnitigMinKey &b) {
        for (int i = 0; i < W; i++) if (a.key[i] != b.key[i]) return false;
        return true;
    };
    std::vector<UnitigContig> kept;
    std::vector<UnitigContig> all;                                  // (the partner search below looks at all of them: work on a copy)
    all.swap(out.contigs);
    // rings that were rings from the start come as one contig per strand: the strand that holds its smallest k-mer in
    // orientation 0 is the one SPEC S10 spells; its partner (same k-mer, orientation 1) goes
    kept.reserve(all.size());
    for (UnitigContig &c0 : all) {
        if (!c0.ring) { kept.push_back(std::move(c0)); continue; }     // (millions in a metagenome: moved, not copied; the partner search below looks at rings only)
        UnitigContig c = c0;
        for (uint32_t r : c.recs) if (r >= mk.size() || !mk[r].valid) { err = "unitig graph: a ring without its smallest k-mer"; return -1; }
        if (c.recs.size() == 1 && recs[c.recs[0]].circ) {
            const UnitigMinKey &m = mk[c.recs[0]];
            if (m.o != 0) {
                // the mirror strand; dropped if the other strand exists (it always does when this record was ranked from a
                // sampled node; a ring without one is reported on the strand of its smallest k-mer alone)
                bool partner = false;
                for (const UnitigContig &d : all)
                    if (&d != &c0 && d.ring && d.recs.size() == 1 && recs[d.recs[0]].circ && mk[d.recs[0]].valid &&
                        key_same_kmer(mk[d.recs[0]], m) && mk[d.recs[0]].o == 0) { partner = true; break; }
                if (partner) continue;
                err = "unitig graph: a ring known on its mirror strand only"; return -1;
            }
            c.rot = m.pos;
            kept.push_back(std::move(c));
            continue;
        }
        // a ring of several records: this strand, or the mirror records in reverse order
        UnitigMinKey best; bool best_here = true; uint64_t best_off = 0;
        uint64_t off = 0;
        for (uint32_t r : c.recs) {
            if (!best.valid || key_less(mk[r], best)) { best = mk[r]; best_here = true; best_off = off + mk[r].pos; }
            off += recs[r].len;
        }
        std::vector<uint32_t> mrecs;
        for (size_t i = c.recs.size(); i-- > 0;) mrecs.push_back(out.mirror[c.recs[i]]);
        off = 0;
        for (uint32_t r : mrecs) {
            if (r >= mk.size() || !mk[r].valid) { err = "unitig graph: a ring without its smallest k-mer"; return -1; }
            if (key_less(mk[r], best)) { best = mk[r]; best_here = false; best_off = off + mk[r].pos; }
            off += recs[r].len;
        }
        if (!best_here) c.recs = mrecs;
        c.rot = best_off;
        kept.push_back(std::move(c));
    }
    out.contigs.swap(kept);
    (void)k;
    return 0;
}

}  // namespace

int unitig_assemble(int k, const std::vector<UnitigRec> &recs, bool tips, bool bubbles, UnitigGraphResult &out, std::string &err) {
    switch ((2 * k + 63) / 64) {
        case 1: return assemble_t<1>(k, recs, tips, bubbles, out, err);
        case 2: return assemble_t<2>(k, recs, tips, bubbles, out, err);
        case 3: return assemble_t<3>(k, recs, tips, bubbles, out, err);
        case 4: return assemble_t<4>(k, recs, tips, bubbles, out, err);
        case 5: return assemble_t<5>(k, recs, tips, bubbles, out, err);
        case 6: return assemble_t<6>(k, recs, tips, bubbles, out, err);
        case 7: return assemble_t<7>(k, recs, tips, bubbles, out, err);
        case 8: return assemble_t<8>(k, recs, tips, bubbles, out, err);
    }
    err = "k too large"; return -1;
}
int unitig_resolve_rings(int k, const std::vector<UnitigRec> &recs, const std::vector<UnitigMinKey> &min_of, UnitigGraphResult &out, std::string &err) {
    switch ((2 * k + 63) / 64) {
        case 1: return resolve_t<1>(k, recs, min_of, out, err);
        case 2: return resolve_t<2>(k, recs, min_of, out, err);
        case 3: return resolve_t<3>(k, recs, min_of, out, err);
        case 4: return resolve_t<4>(k, recs, min_of, out, err);
        case 5: return resolve_t<5>(k, recs, min_of, out, err);
        case 6: return resolve_t<6>(k, recs, min_of, out, err);
        case 7: return resolve_t<7>(k, recs, min_of, out, err);
        case 8: return resolve_t<8>(k, recs, min_of, out, err);
    }
    err = "k too large"; return -1;
}

}  // namespace shk
