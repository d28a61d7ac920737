// cpu_mt.cpp — multi-threaded CPU restatement of the assembly path (SPEC.md S3-S10), the timed CPU BASELINE of
// bench.py ("build's CPU restatement — not upstream sparrowhawk-asm": the reference's crate is an empty submodule,
// /root/reference/.gitmodules:1-4, and there is no Rust toolchain; SURVEY.md §8d asks for exactly this stand-in).
//
// TEST / BENCH INFRASTRUCTURE ONLY: built into oracle/libshk_cpu_mt.so, loaded by bench.py's cpu_baseline leg and by
// tests/test_cpu_mt.py (which checks it against the single-threaded oracle); the product never links or loads it.
//
// What a competent CPU implementation of the same path looks like (std::thread over all cores, hash tables):
//   count    reads (2-bit packed, the device layout) -> canonical k-mers, scattered by hash into 512 partitions
//            (thread-local buffers), every partition then counted by one thread in an open-addressing table
//   filter   rows with count > threshold -> solid set; one global open-addressing index key -> node id
//   graph    8 lookups per node in parallel -> adjacency byte (SPEC S8), kept alive-aware
//   correct  SPEC S9 rounds: tips, then bubbles on the graph the tip round left; decisions on a snapshot
//   collapse SPEC S10: maximal chains of simple links, circular ones cut before their smallest k-mer
// k <= 31 uses 64-bit keys, k <= 63 unsigned __int128.  Output: the FASTA text of SPEC S11 (contig order, names).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef unsigned __int128 u128;

template <typename K> struct KeyOps;
template <> struct KeyOps<uint64_t> {
    static uint64_t hash(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
};
template <> struct KeyOps<u128> {
    static uint64_t hash(u128 x) { return KeyOps<uint64_t>::hash((uint64_t)x ^ KeyOps<uint64_t>::hash((uint64_t)(x >> 64))); }
};

template <typename F> void parallel_for(unsigned threads, size_t n, size_t grain, F &&fn) {
    if (n == 0) return;
    std::atomic<size_t> next{0};
    auto body = [&] {
        for (;;) {
            const size_t a = next.fetch_add(grain);
            if (a >= n) return;
            fn(a, std::min(n, a + grain));
        }
    };
    std::vector<std::thread> ts;
    for (unsigned t = 1; t < threads; t++) ts.emplace_back(body);
    body();
    for (auto &t : ts) t.join();
}

struct Contig { std::string seq; uint64_t kc; };

template <typename K> struct Asm {
    int k; unsigned threads;
    K mask;
    // counting
    uint64_t total_instances = 0, histo[500] = {0};
    std::vector<K> rkeys; std::vector<uint32_t> rcnt;          // rows with count > emit threshold
    uint32_t emit_thr = 0;
    // solid set / graph
    std::vector<K> keys; std::vector<uint32_t> cnt;
    std::vector<uint32_t> index; uint64_t imask = 0;           // open addressing: node id + 1, 0 = empty
    std::vector<uint8_t> adj, alive;
    std::vector<Contig> contigs;
    std::string fasta;

    static uint32_t comp(uint32_t b) { return 3u - b; }
    K revcomp(K x) const { K r = 0; for (int i = 0; i < k; i++) { r = (r << 2) | (K)(3u - (uint32_t)(x & 3)); x >>= 2; } return r; }
    K canonical(K f, int *o) const { const K r = revcomp(f); if (r < f) { *o = 1; return r; } *o = 0; return f; }

    // ---------------------------------------------------------------- count
    // Two-pass radix scatter into ONE arena (round 2 kept threads x 512 growing std::vectors: 256 threads fighting the
    // allocator, 67 M k-mers/s).  Pass A: every thread walks its static share of the segments and counts its k-mers per
    // partition (partition = top bits of the key hash).  Prefix sums give every (partition, thread) its exact range of
    // the arena.  Pass B: the same walk again, k-mers written through small per-partition write-combining buffers (one
    // cache line each) to their final place.  Then every partition — contiguous in the arena — is counted by one thread
    // in an open-addressing table that fits its cache.
    template <typename F> void run_threads(unsigned T, F &&fn) {
        std::vector<std::thread> ts;
        for (unsigned t = 1; t < T; t++) ts.emplace_back([&fn, t] { fn(t); });
        fn(0u);
        for (auto &t : ts) t.join();
    }
    template <typename F> void walk_segments(const uint32_t *bases, const uint32_t *seg_off, size_t s0, size_t s1, F &&emit) const {
        const int sh = 2 * (k - 1);
        for (size_t s = s0; s < s1; s++) {
            const uint64_t lo = seg_off[s], hi = seg_off[s + 1];
            if (hi - lo < (uint64_t)k) continue;
            K f = 0, r = 0;
            uint64_t i = lo;
            uint32_t word = bases[i >> 4] >> (2 * (i & 15));
            for (; i < hi; i++) {
                if ((i & 15) == 0) word = bases[i >> 4];
                const uint32_t c = word & 3u; word >>= 2;
                f = ((f << 2) | (K)c) & mask;
                r = (r >> 2) | ((K)(3u - c) << sh);
                if (i - lo + 1 >= (uint64_t)k) emit(f < r ? f : r);
            }
        }
    }
    double t_scatter = 0, t_tables = 0;                          // seconds, of the last count()
    // The arena is kept between calls (a service counts sample after sample) and asked to be backed by huge pages:
    // 3.2 GB of first-touch 4 KB page faults from 256 threads serialise in the kernel (the scatter did not scale
    // beyond 32 threads with a fresh malloc per call).
    void *arena_mem = nullptr; size_t arena_bytes = 0;
    K *get_arena(size_t n) {
        const size_t want = ((n + 1) * sizeof(K) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (want > arena_bytes) {
            if (arena_mem) free(arena_mem);
            arena_mem = aligned_alloc((size_t)2 << 20, want);
            arena_bytes = arena_mem ? want : 0;
            if (arena_mem) {
                (void)madvise(arena_mem, want, MADV_HUGEPAGE);
                // first touch in parallel, one huge page at a time
                const size_t pages = want >> 21;
                std::atomic<size_t> next{0};
                run_threads(std::min<unsigned>(threads, 64u), [&](unsigned) {
                    for (;;) { const size_t i = next.fetch_add(1); if (i >= pages) break; memset((char *)arena_mem + (i << 21), 0, (size_t)1 << 21); }
                });
            }
        }
        return (K *)arena_mem;
    }
    ~Asm() { if (arena_mem) free(arena_mem); }
    void count(const uint32_t *bases, const uint32_t *seg_off, uint64_t n_seg, uint32_t emit_threshold) {
        emit_thr = emit_threshold;
        const auto c0 = std::chrono::steady_clock::now();
        constexpr unsigned PB = 12, P = 1u << PB;                // 4096 partitions: ~100 k instances each at 400 M
        const unsigned T = threads;
        std::vector<uint64_t> cnt_tp((size_t)T * P, 0);         // [thread][partition]
        auto share = [&](unsigned t, size_t &s0, size_t &s1) { s0 = (size_t)(n_seg * t / T); s1 = (size_t)(n_seg * (t + 1) / T); };
        run_threads(T, [&](unsigned t) {
            size_t s0, s1; share(t, s0, s1);
            uint64_t *mine = &cnt_tp[(size_t)t * P];
            walk_segments(bases, seg_off, s0, s1, [&](K x) { mine[KeyOps<K>::hash(x) >> (64 - PB)]++; });
        });
        // arena layout: partition-major, inside a partition thread-major
        std::vector<uint64_t> pbase(P + 1, 0);
        {
            uint64_t at = 0;
            for (unsigned p = 0; p < P; p++) {
                pbase[p] = at;
                for (unsigned t = 0; t < T; t++) { const uint64_t c = cnt_tp[(size_t)t * P + p]; cnt_tp[(size_t)t * P + p] = at; at += c; }
            }
            pbase[P] = at;
            total_instances = at;
        }
        K *arena = get_arena((size_t)total_instances);
        run_threads(T, [&](unsigned t) {
            size_t s0, s1; share(t, s0, s1);
            uint64_t *cur = &cnt_tp[(size_t)t * P];
            constexpr unsigned WC = 64 / sizeof(K);              // one cache line per partition
            std::vector<K> wc((size_t)P * WC);
            std::vector<uint8_t> fill(P, 0);
            walk_segments(bases, seg_off, s0, s1, [&](K x) {
                const unsigned p = (unsigned)(KeyOps<K>::hash(x) >> (64 - PB));
                K *w = &wc[(size_t)p * WC];
                w[fill[p]++] = x;
                if (fill[p] == WC) { memcpy(arena + cur[p], w, sizeof(K) * WC); cur[p] += WC; fill[p] = 0; }
            });
            for (unsigned p = 0; p < P; p++) if (fill[p]) { memcpy(arena + cur[p], &wc[(size_t)p * WC], sizeof(K) * fill[p]); cur[p] += fill[p]; }
        });
        const auto c1 = std::chrono::steady_clock::now();
        // every partition counted by one thread
        std::vector<std::vector<K>> pk(P); std::vector<std::vector<uint32_t>> pc(P);
        std::vector<std::vector<uint64_t>> ph(T, std::vector<uint64_t>(500, 0));
        {
            std::atomic<unsigned> next{0};
            run_threads(T, [&](unsigned me) {
                std::vector<K> tk, nk; std::vector<uint32_t> tc, nc;
                auto insert = [&](std::vector<K> &kk, std::vector<uint32_t> &cc, size_t cp, K x, uint32_t w) -> bool {
                    size_t i = (size_t)((KeyOps<K>::hash(x) * 0x9E3779B97F4A7C15ull) >> 20) & (cp - 1);
                    for (;;) {
                        if (cc[i] == 0) { kk[i] = x; cc[i] = w; return true; }
                        if (kk[i] == x) { const uint64_t v = (uint64_t)cc[i] + w; cc[i] = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; return false; }
                        i = (i + 1) & (cp - 1);
                    }
                };
                for (;;) {
                    const unsigned p = next.fetch_add(1);
                    if (p >= P) break;
                    const uint64_t n = pbase[p + 1] - pbase[p];
                    if (!n) continue;
                    size_t cap = 1024; while (cap < n / 16) cap <<= 1;       // grows when it fills up (error-rich reads)
                    tk.assign(cap, 0); tc.assign(cap, 0);
                    size_t used = 0;
                    const K *src = arena + pbase[p];
                    for (uint64_t j = 0; j < n; j++) {
                        if (insert(tk, tc, cap, src[j], 1)) {
                            if (++used * 10 > cap * 6) {                     // rehash at 60 %
                                nk.assign(cap * 4, 0); nc.assign(cap * 4, 0);
                                for (size_t i = 0; i < cap; i++) if (tc[i]) insert(nk, nc, cap * 4, tk[i], tc[i]);
                                tk.swap(nk); tc.swap(nc); cap *= 4;
                            }
                        }
                    }
                    for (size_t i = 0; i < cap; i++) if (tc[i]) {
                        ph[me][tc[i] >= 500 ? 499 : tc[i] - 1]++;
                        if (tc[i] > emit_thr) { pk[p].push_back(tk[i]); pc[p].push_back(tc[i]); }
                    }
                }
            });
        }
        for (int b = 0; b < 500; b++) histo[b] = 0;
        for (unsigned t = 0; t < T; t++) for (int b = 0; b < 500; b++) histo[b] += ph[t][b];
        size_t tot = 0; for (unsigned p = 0; p < P; p++) tot += pk[p].size();
        rkeys.clear(); rcnt.clear(); rkeys.reserve(tot); rcnt.reserve(tot);
        for (unsigned p = 0; p < P; p++) { rkeys.insert(rkeys.end(), pk[p].begin(), pk[p].end()); rcnt.insert(rcnt.end(), pc[p].begin(), pc[p].end()); }
        const auto c2 = std::chrono::steady_clock::now();
        t_scatter = std::chrono::duration<double>(c1 - c0).count();
        t_tables = std::chrono::duration<double>(c2 - c1).count();
    }

    // ---------------------------------------------------------------- filter + index
    int filter(uint32_t threshold) {
        if (threshold < emit_thr) return -1;
        keys.clear(); cnt.clear();
        for (size_t i = 0; i < rkeys.size(); i++) if (rcnt[i] > threshold) { keys.push_back(rkeys[i]); cnt.push_back(rcnt[i]); }
        std::vector<K>().swap(rkeys); std::vector<uint32_t>().swap(rcnt);
        // rows in key order: node ids (and with them every tie-break below) do not depend on the thread count
        std::vector<uint32_t> ord(keys.size());
        for (size_t i = 0; i < ord.size(); i++) ord[i] = (uint32_t)i;
        {
            // parallel merge sort: chunks sorted by the threads, then merged pairwise
            auto less = [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; };
            unsigned C = 1; while (C * 2 <= threads && C < 64 && ord.size() / (C * 2) >= 65536) C *= 2;
            std::vector<size_t> edge(C + 1);
            for (unsigned c = 0; c <= C; c++) edge[c] = ord.size() * c / C;
            run_threads(C, [&](unsigned c) { std::sort(ord.begin() + edge[c], ord.begin() + edge[c + 1], less); });
            for (unsigned w = 1; w < C; w *= 2)
                run_threads(C / (2 * w), [&](unsigned j) {
                    std::inplace_merge(ord.begin() + edge[2 * w * j], ord.begin() + edge[2 * w * j + w], ord.begin() + edge[2 * w * j + 2 * w], less);
                });
        }
        std::vector<K> k2(keys.size()); std::vector<uint32_t> c2(keys.size());
        for (size_t i = 0; i < ord.size(); i++) { k2[i] = keys[ord[i]]; c2[i] = cnt[ord[i]]; }
        keys.swap(k2); cnt.swap(c2);
        const size_t n = keys.size();
        size_t cap = 16; while (cap < 2 * n + 2) cap <<= 1;
        index.assign(cap, 0); imask = cap - 1;
        parallel_for(threads, n, 1 << 16, [&](size_t a, size_t b) {      // claim the first empty slot (keys are distinct)
            for (size_t i = a; i < b; i++) {
                size_t s = KeyOps<K>::hash(keys[i]) & imask;
                for (;;) {
                    uint32_t expect = 0;
                    if (__atomic_compare_exchange_n(&index[s], &expect, (uint32_t)i + 1, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                    s = (s + 1) & imask;
                }
            }
        });
        return 0;
    }
    int64_t find(K x) const {
        size_t s = KeyOps<K>::hash(x) & imask;
        for (;;) {
            const uint32_t v = index[s];
            if (!v) return -1;
            if (keys[v - 1] == x) return (int64_t)v - 1;
            s = (s + 1) & imask;
        }
    }

    // ---------------------------------------------------------------- graph (SPEC S8)
    typedef uint32_t onode;                                    // idx * 2 + o
    static constexpr onode NONE = 0xFFFFFFFFu;
    K seq(onode v) const { const K x = keys[v >> 1]; return (v & 1) ? revcomp(x) : x; }
    static uint32_t rev4(uint32_t n) { return ((n & 1) << 3) | ((n & 2) << 1) | ((n & 4) >> 1) | ((n & 8) >> 3); }
    uint32_t outmask(onode v) const { const uint32_t a = adj[v >> 1]; return (v & 1) ? rev4(a >> 4) : (a & 15u); }
    int outdeg(onode v) const { return __builtin_popcount(outmask(v)); }
    int indeg(onode v) const { return outdeg(v ^ 1u); }
    onode follow(onode v, uint32_t b) const {                  // out-neighbour by appended base b (must exist)
        const K s = ((seq(v) << 2) | (K)b) & mask;
        int o; const K c = canonical(s, &o);
        const int64_t i = find(c);
        return i < 0 ? NONE : (onode)(i * 2 + o);
    }
    onode only_out(onode v) const { return follow(v, (uint32_t)__builtin_ctz(outmask(v))); }

    void build_graph() {
        const size_t n = keys.size();
        adj.assign(n, 0); alive.assign(n, 1);
        parallel_for(threads, n, 4096, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                const K x = keys[i];
                uint32_t m = 0;
                for (uint32_t c = 0; c < 4; c++) {
                    int o;
                    if (find(canonical(((x << 2) | (K)c) & mask, &o)) >= 0) m |= 1u << c;
                    if (find(canonical((x >> 2) | ((K)c << (2 * (k - 1))), &o)) >= 0) m |= 1u << (4 + c);
                }
                adj[i] = (uint8_t)m;
            }
        });
    }
    void remove_nodes(const std::vector<uint32_t> &rm) {       // alive-aware adjacency (serial: few nodes)
        for (const uint32_t r : rm) {
            if (!alive[r]) continue;
            alive[r] = 0;
            const K x = keys[r];
            const uint32_t a = adj[r], fb = (uint32_t)(x >> (2 * (k - 1))) & 3u, lb = (uint32_t)x & 3u;
            for (uint32_t b = 0; b < 4; b++) {
                if ((a >> b) & 1u) {
                    int o; const int64_t u = find(canonical(((x << 2) | (K)b) & mask, &o));
                    if (u >= 0) adj[u] &= (uint8_t)~(1u << (o == 0 ? 4 + fb : 3 - fb));
                }
                if ((a >> (4 + b)) & 1u) {
                    int o; const int64_t u = find(canonical((x >> 2) | ((K)b << (2 * (k - 1))), &o));
                    if (u >= 0) adj[u] &= (uint8_t)~(1u << (o == 0 ? lb : 4 + (3 - lb)));
                }
            }
            adj[r] = 0;
        }
    }

    // ---------------------------------------------------------------- correction (SPEC S9)
    struct Tip { onode start, junction; uint32_t len; uint64_t sum; };
    size_t tip_round() {
        const uint32_t T = 2u * (uint32_t)k;
        const size_t total = keys.size() * 2;
        std::mutex mu; std::vector<Tip> tips;
        parallel_for(threads, total, 1 << 16, [&](size_t a, size_t b) {
            std::vector<Tip> loc;
            for (size_t vv = a; vv < b; vv++) {
                const onode v = (onode)vv;
                if (!alive[v >> 1] || indeg(v) != 0 || outdeg(v) != 1) continue;
                onode cur = v, J = NONE; uint32_t len = 1; uint64_t sum = cnt[v >> 1];
                for (;;) {
                    if (outdeg(cur) != 1) break;
                    const onode nx = only_out(cur);
                    if (nx == NONE) break;
                    if (indeg(nx) >= 2) { J = nx; break; }
                    len++; sum += cnt[nx >> 1]; cur = nx;
                    if (len > T) break;
                }
                if (J != NONE && len <= T) loc.push_back(Tip{v, J, len, sum});
            }
            if (!loc.empty()) { std::lock_guard<std::mutex> lk(mu); tips.insert(tips.end(), loc.begin(), loc.end()); }
        });
        std::sort(tips.begin(), tips.end(), [](const Tip &x, const Tip &y) { return x.junction != y.junction ? x.junction < y.junction : x.start < y.start; });
        std::vector<uint32_t> rm;
        for (size_t i = 0; i < tips.size();) {
            size_t j = i; while (j < tips.size() && tips[j].junction == tips[i].junction) j++;
            const uint32_t d = (uint32_t)indeg(tips[i].junction), t = (uint32_t)(j - i);
            size_t best = i;
            for (size_t q = i + 1; q < j; q++) {                // best = max (len, sum, smaller first k-mer)
                const Tip &A = tips[q], &B = tips[best];
                bool better;
                if (A.len != B.len) better = A.len > B.len;
                else if (A.sum != B.sum) better = A.sum > B.sum;
                else better = keys[A.start >> 1] < keys[B.start >> 1];
                if (better) best = q;
            }
            for (size_t q = i; q < j; q++) {
                if (t == d && q == best) continue;
                onode cur = tips[q].start;
                for (uint32_t s = 0; s < tips[q].len; s++) { rm.push_back(cur >> 1); if (s + 1 < tips[q].len) cur = only_out(cur); }
            }
            i = j;
        }
        std::sort(rm.begin(), rm.end()); rm.erase(std::unique(rm.begin(), rm.end()), rm.end());
        remove_nodes(rm);
        return rm.size();
    }
    size_t bubble_round() {
        const uint32_t T = 2u * (uint32_t)k;
        const size_t total = keys.size() * 2;
        std::mutex mu; std::vector<uint32_t> rm;
        parallel_for(threads, total, 1 << 16, [&](size_t a, size_t b) {
            std::vector<uint32_t> loc;
            for (size_t vv = a; vv < b; vv++) {
                const onode S = (onode)vv;
                if (!alive[S >> 1] || outdeg(S) < 2) continue;
                const uint32_t om = outmask(S);
                onode first[4], end[4]; uint32_t len[4]; uint64_t sum[4]; bool ok[4];
                for (uint32_t c = 0; c < 4; c++) {
                    ok[c] = false; first[c] = end[c] = NONE; len[c] = 0; sum[c] = 0;
                    if (!((om >> c) & 1u)) continue;
                    const onode bn = follow(S, c);
                    if (bn == NONE || indeg(bn) != 1) continue;
                    first[c] = bn;
                    onode cur = bn; uint32_t l = 1; uint64_t s = cnt[bn >> 1];
                    for (;;) {
                        if (outdeg(cur) != 1) break;
                        const onode nx = only_out(cur);
                        if (nx == NONE) break;
                        if (indeg(nx) >= 2) { end[c] = nx; ok[c] = true; break; }
                        if (l + 1 > T) break;
                        l++; s += cnt[nx >> 1]; cur = nx;
                    }
                    len[c] = l; sum[c] = s;
                }
                for (uint32_t x = 0; x < 4; x++) {
                    if (!ok[x]) continue;
                    const onode E = end[x];
                    {   // evaluated only from the side with key(S) <= key(rc(E)), key(x,o) = (x, o)
                        const K ks = keys[S >> 1], ke = keys[E >> 1];
                        bool le;
                        if (ks < ke) le = true; else if (ke < ks) le = false; else le = (S & 1u) <= ((E ^ 1u) & 1u);
                        if (!le) continue;
                    }
                    uint32_t grp = 0; bool best = true;
                    for (uint32_t y = 0; y < 4; y++) {
                        if (!ok[y] || end[y] != E) continue;
                        grp++;
                        if (y == x) continue;
                        const uint64_t l = sum[y] * len[x], r = sum[x] * len[y];
                        bool better;
                        if (l != r) better = l > r;
                        else if (len[y] != len[x]) better = len[y] < len[x];
                        else better = keys[first[y] >> 1] < keys[first[x] >> 1];
                        if (better) best = false;
                    }
                    if (grp >= 2 && !best) {
                        onode cur = first[x];
                        for (uint32_t s = 0; s < len[x]; s++) { loc.push_back(cur >> 1); if (s + 1 < len[x]) cur = only_out(cur); }
                    }
                }
            }
            if (!loc.empty()) { std::lock_guard<std::mutex> lk(mu); rm.insert(rm.end(), loc.begin(), loc.end()); }
        });
        std::sort(rm.begin(), rm.end()); rm.erase(std::unique(rm.begin(), rm.end()), rm.end());
        remove_nodes(rm);
        return rm.size();
    }
    void correct(bool tips, bool bubbles) {
        if (!tips && !bubbles) return;
        for (int r = 0; r < 32; r++) {
            size_t n1 = tips ? tip_round() : 0, n2 = bubbles ? bubble_round() : 0;
            if (n1 + n2 == 0) break;
        }
    }

    // ---------------------------------------------------------------- collapse (SPEC S10) + FASTA (S11)
    onode succ_simple(onode u) const {
        if (outdeg(u) != 1) return NONE;
        const onode v = only_out(u);
        if (v == NONE || indeg(v) != 1 || v == u || v == (u ^ 1u)) return NONE;
        return v;
    }
    std::string spell(const std::vector<onode> &path) const {
        static const char B[4] = {'A', 'C', 'G', 'T'};
        std::string s(path.size() + k - 1, 'A');
        const K x = seq(path[0]);
        for (int i = 0; i < k; i++) s[i] = B[(uint32_t)(x >> (2 * (k - 1 - i))) & 3u];
        // the last base of an oriented node: of its k-mer, or the complement of the k-mer's first base
        auto last_base = [&](onode v) -> uint32_t { const K y = keys[v >> 1]; return (v & 1) ? 3u - (uint32_t)(y >> (2 * (k - 1))) : (uint32_t)y & 3u; };
        auto fill = [&](size_t a, size_t b) { for (size_t i = std::max<size_t>(a, 1); i < b; i++) s[k - 1 + i] = B[last_base(path[i]) & 3u]; };
        if (path.size() >= ((size_t)1 << 20)) parallel_for(threads, path.size(), 1 << 16, fill);   // (a chromosome: all threads)
        else fill(0, path.size());
        return s;
    }
    static std::string rc_str(const std::string &s) {
        std::string r(s.size(), 'A');
        for (size_t i = 0; i < s.size(); i++) { const char c = s[s.size() - 1 - i]; r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }
        return r;
    }
    void collapse() {
        const size_t n = keys.size(), total = 2 * n;
        std::vector<onode> succ(total, NONE);
        parallel_for(threads, total, 1 << 16, [&](size_t a, size_t b) {
            for (size_t v = a; v < b; v++) if (alive[v >> 1]) succ[v] = succ_simple((onode)v);
        });
        std::vector<uint8_t> visited(total, 0);
        std::mutex mu;
        contigs.clear();
        auto emit = [&](const std::vector<onode> &path, std::vector<Contig> &dst) {
            uint64_t kc = 0;
            for (const onode v : path) kc += cnt[v >> 1];
            std::string s = spell(path), r = rc_str(s);
            dst.push_back(Contig{r < s ? r : s, kc});
        };
        // linear chains: from every head; a unitig exists on both strands — the strand whose head id is smaller
        // (or the only one, for a self-mirror chain) does the emitting, the spelling is canonicalised anyway
        parallel_for(threads, total, 1 << 16, [&](size_t a, size_t b) {
            std::vector<Contig> loc; std::vector<onode> path;
            for (size_t vv = a; vv < b; vv++) {
                const onode v = (onode)vv;
                if (!alive[v >> 1] || succ[v ^ 1u] != NONE) continue;       // has a simple predecessor: not a head
                path.clear();
                for (onode c = v; c != NONE; c = succ[c]) path.push_back(c);
                const onode mirror_head = path.back() ^ 1u;
                for (const onode c : path) visited[c] = 1;
                if (v <= mirror_head) emit(path, loc);
            }
            if (!loc.empty()) { std::lock_guard<std::mutex> lk(mu); for (auto &c : loc) contigs.push_back(std::move(c)); }
        });
        // circular unitigs: alive, never reached from a head; cut before the smallest k-mer, spelled from (x, 0)
        std::vector<onode> path;
        for (size_t vv = 0; vv < total; vv++) {
            const onode v = (onode)vv;
            if (!alive[v >> 1] || visited[v]) continue;
            onode best = v;
            for (onode c = succ[v]; c != v; c = succ[c]) { visited[c] = 1; if (keys[c >> 1] < keys[best >> 1] || (keys[c >> 1] == keys[best >> 1] && c < best)) best = c; }
            visited[v] = 1;
            if (best & 1u) continue;                                         // the mirror strand
            path.clear();
            path.push_back(best);
            for (onode c = succ[best]; c != best; c = succ[c]) path.push_back(c);
            emit(path, contigs);
        }
        std::sort(contigs.begin(), contigs.end(), [](const Contig &a, const Contig &b) {
            if (a.seq.size() != b.seq.size()) return a.seq.size() > b.seq.size();
            return a.seq < b.seq;
        });
        fasta.clear();
        for (size_t i = 0; i < contigs.size(); i++) {
            fasta += ">contig_" + std::to_string(i + 1) + " len=" + std::to_string(contigs[i].seq.size()) + " kc=" + std::to_string(contigs[i].kc) + "\n";
            fasta += contigs[i].seq; fasta += "\n";
        }
    }
};

struct Ctx {
    int k; unsigned threads;
    Asm<uint64_t> *a64 = nullptr; Asm<u128> *a128 = nullptr;
};

}  // namespace

#define DISPATCH(c, expr) ((c)->a64 ? (c)->a64->expr : (c)->a128->expr)

extern "C" {

unsigned cpumt_hardware_threads(void) { const unsigned h = std::thread::hardware_concurrency(); return h ? h : 1; }

void *cpumt_new(uint32_t k, uint32_t threads) {
    if ((k & 1u) == 0 || k < 15 || k > 63) return nullptr;
    Ctx *c = new Ctx();
    c->k = (int)k; c->threads = threads ? threads : cpumt_hardware_threads();
    if (k <= 31) { c->a64 = new Asm<uint64_t>(); c->a64->k = (int)k; c->a64->threads = c->threads; c->a64->mask = ((uint64_t)1 << (2 * k)) - 1; }
    else { c->a128 = new Asm<u128>(); c->a128->k = (int)k; c->a128->threads = c->threads; c->a128->mask = (((u128)1) << (2 * k)) - 1; }
    return c;
}
void cpumt_free(void *p) { Ctx *c = (Ctx *)p; if (!c) return; delete c->a64; delete c->a128; delete c; }
uint32_t cpumt_threads(void *p) { return ((Ctx *)p)->threads; }

// packed reads in the device layout (include/shk.h: shk_preprocess_packed_device); rows with count <= emit_threshold
// are histogrammed and dropped
void cpumt_count(void *p, const uint32_t *bases, const uint32_t *seg_off, uint64_t n_seg, uint32_t emit_threshold) {
    Ctx *c = (Ctx *)p; DISPATCH(c, count(bases, seg_off, n_seg, emit_threshold));
}
uint64_t cpumt_total_instances(void *p) { Ctx *c = (Ctx *)p; return DISPATCH(c, total_instances); }
// seconds the last cpumt_count spent scattering k-mers into partitions / counting the partitions
void cpumt_count_times(void *p, double *scatter_s, double *tables_s) {
    Ctx *c = (Ctx *)p;
    *scatter_s = DISPATCH(c, t_scatter); *tables_s = DISPATCH(c, t_tables);
}
// a new thread count for the next call (bench.py times the count at several)
void cpumt_set_threads(void *p, uint32_t threads) {
    Ctx *c = (Ctx *)p;
    c->threads = threads ? threads : cpumt_hardware_threads();
    if (c->a64) c->a64->threads = c->threads; else c->a128->threads = c->threads;
}
void cpumt_histo(void *p, uint64_t *out500) { Ctx *c = (Ctx *)p; memcpy(out500, c->a64 ? c->a64->histo : c->a128->histo, 500 * 8); }
int cpumt_filter(void *p, uint32_t threshold) { Ctx *c = (Ctx *)p; return DISPATCH(c, filter(threshold)); }
uint64_t cpumt_n_solid(void *p) { Ctx *c = (Ctx *)p; return DISPATCH(c, keys.size()); }
// keys: W = 1 (k <= 31) or 2 words per k-mer, least significant first; rows ascending by key
void cpumt_get_solid(void *p, uint64_t *keys, uint32_t *counts) {
    Ctx *c = (Ctx *)p;
    if (c->a64) { memcpy(keys, c->a64->keys.data(), c->a64->keys.size() * 8); memcpy(counts, c->a64->cnt.data(), c->a64->cnt.size() * 4); }
    else {
        for (size_t i = 0; i < c->a128->keys.size(); i++) { keys[2 * i] = (uint64_t)c->a128->keys[i]; keys[2 * i + 1] = (uint64_t)(c->a128->keys[i] >> 64); }
        memcpy(counts, c->a128->cnt.data(), c->a128->cnt.size() * 4);
    }
}
void cpumt_assemble(void *p, int no_bubble_collapse, int no_dead_end_removal) {
    Ctx *c = (Ctx *)p;
    DISPATCH(c, build_graph());
    DISPATCH(c, correct(!no_dead_end_removal, !no_bubble_collapse));
    DISPATCH(c, collapse());
}
uint64_t cpumt_n_contigs(void *p) { Ctx *c = (Ctx *)p; return DISPATCH(c, contigs.size()); }
const char *cpumt_fasta(void *p) { Ctx *c = (Ctx *)p; return c->a64 ? c->a64->fasta.c_str() : c->a128->fasta.c_str(); }

}  // extern "C"
