// shard_graph.h — the assembly phases with the GRAPH SHARDED over the ranks (one process per GPU; DESIGN.md "Multi-GPU").
// (included by pipeline.hip inside namespace shk, after graph_part.h and collapse.h)
//
// Replaces, for N > 1, the replicated `assembly:create_graph` / `correct_graph` / `collapse_graph` of round 2
// (/root/reference/www/src/components/pages/AssemblyPage.vue:595-602: the reference has one process and no collectives;
// the north_star asks for the k-mer space partitioned over the GPUs "before the per-GPU graph build").
//
// A rank keeps the solid k-mers of the counting partitions it owns — nobody gathers the solid set.  Node ids:
// local i (row index), global gid = gbase[rank] + i, oriented 2 * id + o.
//   1. adjacency   local mini tables and local neighbour candidates as on one GPU (graph_part.h); a candidate whose
//                  minimiser belongs to another rank becomes a QUERY (k-mer + graph partition) routed to its owner,
//                  one pairwise exchange of queries and one of answers (found / global id)
//   2. half links  a link u -> w across ranks is simple iff outdeg(u) = 1 (u's owner knows) and indeg(w) = 1 (w's
//                  owner knows): every rank tells the owner of w about its u's; by mirror symmetry (rc(w) -> rc(u) is
//                  the same edge seen from the other side) the owner of u hears about indeg(w) = 1 in the same exchange
//   3. local chains the single-GPU contraction (collapse.h: splitters, LDS fragments, pointer jumping) over the links
//                  that stay on the rank: every node gets (local chain, position); ~1 node in 10 ends a chain at a rank edge
//   4. stitching   the local chains (32 bytes each) are gathered and ranked by every rank (k_rank_init / k_rank_jump):
//                  unitigs, rings across ranks included
//   5. unitig graph tips, bubbles and the final chains on the host, from the first / last k-mer of every unitig
//                  (unitig_graph.h) — a handful of records for an isolate
//   6. emission    every rank writes the bases of its own nodes at their place in the contig text; one all-reduce
//                  (the ranks' bytes are disjoint) puts the text together
// All integer work, no MFMA.
#pragma once

static constexpr uint32_t XREF = 0x80000000u;              // nb[] entry: the neighbour lives on another rank; low bits = cross query index
static constexpr int ROUTE_CH = 4096;                      // items per workgroup of the router
static constexpr uint32_t ROUTE_MAX_WORLD = 256;

// ---- routing of fixed-size records to destination ranks (dest[i] = rank or NIL: not an item) ---------------------
__global__ __launch_bounds__(256) void k_route_count(const uint32_t *__restrict__ dest, uint32_t n, unsigned long long *__restrict__ counts) {
    __shared__ uint32_t h[ROUTE_MAX_WORLD];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t d = dest[i];
        if (d < ROUTE_MAX_WORLD) atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
// send[cursor[d]...]: the records of destination d, PW words each; sidx[i] = the record index item i got.  One global
// atomic per workgroup and destination (a workgroup owns ROUTE_CH consecutive items).
template <int PW>
__global__ __launch_bounds__(256) void k_route_pack(const uint32_t *__restrict__ dest, const uint64_t *__restrict__ pay, uint32_t n,
                                                    unsigned long long *__restrict__ cursors, uint64_t *__restrict__ send,
                                                    uint32_t *__restrict__ sidx) {
    __shared__ uint32_t h[ROUTE_MAX_WORLD];
    __shared__ unsigned long long base[ROUTE_MAX_WORLD];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i0 = blockIdx.x * ROUTE_CH, i1 = min(n, i0 + (uint32_t)ROUTE_CH);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) { const uint32_t d = dest[i]; if (d < ROUTE_MAX_WORLD) atomicAdd(&h[d], 1u); }
    __syncthreads();
    if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (unsigned long long)h[threadIdx.x]);
    __syncthreads();
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const uint32_t d = dest[i];
        if (d >= ROUTE_MAX_WORLD) continue;
        const unsigned long long at = base[d] + atomicAdd(&h[d], 1u);
#pragma unroll
        for (int j = 0; j < PW; j++) send[at * PW + j] = pay[(uint64_t)i * PW + j];
        if (sidx) sidx[i] = (uint32_t)at;
    }
}

// ---- 1. cross-rank adjacency -----------------------------------------------------------------------------------
// how many of k_graph_local's queued candidates belong to other ranks (bit 63 of the query word)
__global__ __launch_bounds__(256) void k_xq_total(const uint32_t *__restrict__ roff, const unsigned long long *__restrict__ queries,
                                                  const uint32_t *__restrict__ qcnt, unsigned int *__restrict__ total) {
    const unsigned long long *myq = queries + 8ull * roff[blockIdx.x];
    const uint32_t nq = qcnt[blockIdx.x];
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < nq; t += blockDim.x) mine += (uint32_t)(myq[t] >> 63);
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(total, mine);
}
// the cross queries of k_graph_local, staged compactly: dest, {candidate k-mer, graph partition}, and what the answer
// will be applied to: meta = i | j << 32 | orientation << 35 | dest << 36
template <int W>
__global__ __launch_bounds__(256) void k_xq_stage(KeyArr<W> keys, int k, const uint32_t *__restrict__ roff,
                                                  const unsigned long long *__restrict__ queries, const uint32_t *__restrict__ qcnt,
                                                  unsigned int *__restrict__ n_staged, uint32_t cap,
                                                  uint32_t *__restrict__ dest, uint64_t *__restrict__ pay /* [W + 1] */,
                                                  unsigned long long *__restrict__ meta) {
    __shared__ uint32_t blk_n, blk_base;
    const uint32_t P = blockIdx.x;
    const unsigned long long *myq = queries + 8ull * roff[P];
    const uint32_t nq = qcnt[P];
    if (threadIdx.x == 0) blk_n = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < nq; t += blockDim.x) mine += (uint32_t)(myq[t] >> 63);
    if (mine) atomicAdd(&blk_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) blk_base = blk_n ? atomicAdd(n_staged, blk_n) : 0u;
    __syncthreads();
    if (!blk_n) return;
    if (threadIdx.x == 0) blk_n = 0;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < nq; t += blockDim.x) {
        const unsigned long long q = myq[t];
        if (!(q >> 63)) continue;
        const uint32_t at = blk_base + atomicAdd(&blk_n, 1u);
        if (at >= cap) continue;
        const uint32_t i = (uint32_t)q, j = (uint32_t)(q >> 32) & 7u, p = (uint32_t)(q >> 35) & 0x1FFFFu, d = (uint32_t)(q >> 52) & 0xFFu;
        const Kmer<W> x = keys.load(i);
        const Kmer<W> rx = km_revcomp<W>(x, k);
        bool o; const Kmer<W> c = adj_candidate<W>(x, rx, k, j, o);
        dest[at] = d;
#pragma unroll
        for (int w = 0; w < W; w++) pay[(uint64_t)at * (W + 1) + w] = c.w[w];
        pay[(uint64_t)at * (W + 1) + W] = p;
        meta[at] = (unsigned long long)i | ((unsigned long long)j << 32) | ((unsigned long long)(o ? 1u : 0u) << 35) | ((unsigned long long)d << 36);
    }
}
// the owner's side: membership of every received candidate -> its global node id (~0: no such solid k-mer)
template <int W>
__global__ __launch_bounds__(256) void k_xq_answer(KeyArr<W> keys, GraphTable gt, const uint64_t *__restrict__ recv, uint64_t n_recv,
                                                   unsigned long long gbase, unsigned long long *__restrict__ ans) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_recv; t += (uint64_t)gridDim.x * blockDim.x) {
        Kmer<W> c;
#pragma unroll
        for (int w = 0; w < W; w++) c.w[w] = recv[t * (W + 1) + w];
        const uint32_t p = (uint32_t)recv[t * (W + 1) + W] & gt.gp_mask;
        const uint32_t idx = gt_lookup_in<W>(gt, keys, c, p);
        ans[t] = idx == NIL ? ~0ull : gbase + idx;
    }
}
// the asker's side: answers arrive in the order the queries were sent (sidx: item -> send index).  A neighbour found
// on another rank sets its adjacency bit like a local one; nb[] gets XREF | item, and xnb[item] the oriented global id.
__global__ __launch_bounds__(256) void k_xq_apply(const unsigned long long *__restrict__ meta, const uint32_t *__restrict__ sidx,
                                                  const unsigned long long *__restrict__ ans, uint32_t n_items,
                                                  uint8_t *__restrict__ adj, uint32_t *__restrict__ nb,
                                                  unsigned long long *__restrict__ xnb) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_items; t += gridDim.x * blockDim.x) {
        const unsigned long long a = ans[sidx[t]];
        xnb[t] = ~0ull;
        if (a == ~0ull) continue;
        const unsigned long long m = meta[t];
        const uint32_t i = (uint32_t)m, j = (uint32_t)(m >> 32) & 7u, o = (uint32_t)(m >> 35) & 1u;
        atomicOr((uint32_t *)adj + (i >> 2), (1u << j) << (8 * (i & 3u)));
        xnb[t] = a * 2ull + (j < 4 ? o : (o ^ 1u));        // (a predecessor q -> (x,0) is the edge (x,1) -> rc(q): graph_part.h)
        uint32_t *slot = nb + 2ull * i + (j < 4 ? 0 : 1);
        if (atomicCAS(slot, NIL, XREF | t) != NIL) atomicExch(slot, NB_MULTI);
    }
}

// ---- 2. half links ---------------------------------------------------------------------------------------------
// every oriented node v with exactly one out-neighbour, on another rank: {w, v} (oriented global ids) goes to w's owner
__global__ __launch_bounds__(256) void k_hl_stage(const uint8_t *__restrict__ adj, const uint32_t *__restrict__ nb, uint32_t n_nodes,
                                                  const unsigned long long *__restrict__ xnb, const unsigned long long *__restrict__ meta,
                                                  unsigned long long gbase, uint32_t *__restrict__ dest, uint64_t *__restrict__ pay /* [2] */) {
    const uint32_t total = 2u * n_nodes;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
        uint32_t d = NIL;
        const uint32_t u = nb[v];
        if ((u & XREF) && u < NB_MULTI && __popc(outmask_of(adj[v >> 1], v & 1u)) == 1) {
            const uint32_t t = u & ~XREF;
            d = (uint32_t)(meta[t] >> 36) & 0xFFu;
            pay[2ull * v] = xnb[t];
            pay[2ull * v + 1] = 2ull * gbase + v;
        }
        dest[v] = d;
    }
}
// w's owner: the link v -> w is simple iff w has this one in-edge; then xpred[w] = index of the record (it holds v)
__global__ __launch_bounds__(256) void k_hl_apply(const uint64_t *__restrict__ recv, uint64_t n_recv, unsigned long long gbase, uint32_t n_nodes,
                                                  const uint8_t *__restrict__ adj, uint32_t *__restrict__ xpred, uint32_t *__restrict__ flags) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_recv; t += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long w = recv[2 * t];
        if (w < 2ull * gbase || w - 2ull * gbase >= 2ull * n_nodes) { flags[0] = 1; continue; }     // misrouted: the ownership rules disagree
        const uint32_t wl = (uint32_t)(w - 2ull * gbase);
        if (__popc(outmask_of(adj[wl >> 1], (wl & 1u) ^ 1u)) == 1) xpred[wl] = (uint32_t)t;
    }
}

// ---- 3./4. local chains -> records for the stitching ---------------------------------------------------------------
// per local chain (HeadRec of the local ranking): the link that continues it on another rank, if any.
// out: succ gid (~0 none), and SegRec fields; dest / pay = the question "which chain of yours starts at node w?"
__global__ __launch_bounds__(256) void k_lchain_stage(const HeadRec *__restrict__ heads, uint32_t n_heads, const uint32_t *__restrict__ xpred,
                                                      const uint64_t *__restrict__ hl_recv, const unsigned long long *__restrict__ gbases,
                                                      uint32_t world, SegRec *__restrict__ segs, uint32_t *__restrict__ dest,
                                                      uint64_t *__restrict__ pay /* [1] */) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_heads; s += gridDim.x * blockDim.x) {
        const HeadRec h = heads[s];
        SegRec r; r.node = s; r.next_spl = NIL; r.len = (uint32_t)h.len; r.last = h.tail_node; r.sum = h.kc; r.head = 0; r.pad = 0;
        uint32_t d = NIL;
        if (h.circ) r.head = HEAD_ORPHAN;                    // a ring that lives on this rank alone: a unitig of its own
        else {
            const bool has_pred = xpred[h.head_node] != NIL;
            if (!has_pred) r.head = HEAD_LINEAR;
            const uint32_t ref = xpred[h.tail_node ^ 1u];    // the simple predecessor of rc(tail) is rc(successor of tail)
            if (ref != NIL) {
                const unsigned long long w = hl_recv[2ull * ref + 1] ^ 1ull;
                const unsigned long long wn = w >> 1;
                uint32_t lo = 0;
                for (uint32_t q = 1; q < world; q++) if (gbases[q] <= wn) lo = q;
                d = lo;
                pay[s] = w;
            }
        }
        segs[s] = r; dest[s] = d;
    }
}
// the owner of w: its chain (w is a chain's first node), as a global chain index
__global__ __launch_bounds__(256) void k_ls_answer(const uint64_t *__restrict__ recv, uint64_t n_recv, unsigned long long gbase, uint32_t n_nodes,
                                                   const uint2 *__restrict__ ol, uint32_t lbase, unsigned long long *__restrict__ ans, uint32_t *__restrict__ flags) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_recv; t += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long w = recv[t];
        unsigned long long a = NIL;
        if (w >= 2ull * gbase && w - 2ull * gbase < 2ull * n_nodes) {
            const uint2 o = ol[(uint32_t)(w - 2ull * gbase)];
            if (o.x != NIL && o.y == 0u) a = lbase + o.x; else flags[0] = 2;      // not the first node of a chain: the two sides disagree on the link
        } else flags[0] = 1;
        ans[t] = a;
    }
}
__global__ __launch_bounds__(256) void k_ls_apply(const uint32_t *__restrict__ dest, const uint32_t *__restrict__ sidx, const unsigned long long *__restrict__ ans,
                                                  uint32_t n_heads, uint32_t lbase, SegRec *__restrict__ segs) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_heads; s += gridDim.x * blockDim.x) {
        if (dest[s] != NIL) segs[s].next_spl = (uint32_t)ans[sidx[s]];
        segs[s].node = lbase + s;
    }
}

// per unitig (a chain of local chains): reported by its last record — the tail of a linear one, the record in front of the
// smallest index of a ring (collapse.h: k_rank_tails without the strand decision, which the host takes from the k-mers)
struct UHead { uint32_t root, tail, circ, pad; unsigned long long len, kc; };
__global__ __launch_bounds__(256) void k_stitch_tails(const SegRec *__restrict__ segs, const unsigned int *__restrict__ n_p, const RankRec *__restrict__ R,
                                                      UHead *__restrict__ heads, uint32_t *__restrict__ slot_of, unsigned int *__restrict__ n_heads) {
    const uint32_t n = *n_p;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const SegRec r = segs[s];
        const RankRec me = R[s];
        const uint32_t hd = segs[me.P].head;
        UHead h; h.pad = 0;
        if (hd != 0u) {
            if (r.next_spl != NIL) continue;
            h.root = me.P; h.tail = s; h.circ = hd == HEAD_ORPHAN ? 1u : 0u;
            h.len = (unsigned long long)me.A + r.len; h.kc = me.K + r.sum;
        } else {
            if (r.next_spl != me.m) continue;
            h.root = me.m; h.tail = s; h.circ = 1;
            h.len = (unsigned long long)me.d + r.len; h.kc = me.dK + r.sum;
        }
        const uint32_t slot = atomicAdd(n_heads, 1u);
        heads[slot] = h; slot_of[h.root] = slot;
    }
}

// ---- 5. first / last k-mer of the unitigs this rank holds an end of -----------------------------------------------
struct EndReq { uint32_t uid, which /* 0 first, 1 last */, chain /* local chain */, pad; };
template <int W>
__global__ __launch_bounds__(256) void k_fill_ends(Graph<W> g, const HeadRec *__restrict__ heads, const EndReq *__restrict__ req, uint32_t n_req,
                                                   uint64_t *__restrict__ ends /* [uid][2][W] */) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_req; t += gridDim.x * blockDim.x) {
        const EndReq q = req[t];
        const Kmer<W> x = g.seq(q.which ? heads[q.chain].tail_node : heads[q.chain].head_node);
#pragma unroll
        for (int w = 0; w < W; w++) ends[((uint64_t)q.uid * 2 + q.which) * W + w] = x.w[w];
    }
}
// local chain -> dense index of the ring record its unitig is (NIL: not part of a ring)
__global__ __launch_bounds__(256) void k_ring_of(const FinRec *__restrict__ fin, uint32_t lbase, uint32_t n_lch, const uint32_t *__restrict__ uid_of_slot,
                                                 const uint32_t *__restrict__ ring_of_uid, uint32_t *__restrict__ ring_of) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_lch; s += gridDim.x * blockDim.x) {
        const FinRec f = fin[lbase + s];
        ring_of[s] = f.slot == NIL ? NIL : ring_of_uid[uid_of_slot[f.slot]];
    }
}

// ---- rings: the smallest k-mer of the rings the unitig graph ended with (SPEC S10's cut) --------------------------
// ring_of[local chain]: dense index of the ring record (NIL: none).  Pass 1: smallest 64-bit key prefix per ring on this
// rank.  Pass 2 (after the ranks' minima were merged): the node(s) with that prefix; exact order by compare-and-swap.
template <int W>
__global__ __launch_bounds__(256) void k_sring_min1(Graph<W> g, const uint2 *__restrict__ ol, const uint32_t *__restrict__ ring_of,
                                                    unsigned long long *__restrict__ prefix_min) {
    const uint32_t total = g.n * 2;
    const int lane = threadIdx.x & 63;
    for (uint32_t v0 = blockIdx.x * blockDim.x; v0 < total; v0 += gridDim.x * blockDim.x) {
        const uint32_t v = v0 + threadIdx.x;
        uint32_t ring = NIL; unsigned long long pf = ~0ull;
        if (v < total) {
            const uint32_t sl = ol[v].x;
            if (sl != NIL) ring = ring_of[sl];
            if (ring != NIL) pf = km_prefix64<W>(g.keys, v >> 1, g.k);
        }
        unsigned long long todo = __ballot(ring != NIL);
        while (todo) {                                       // one atomic per wave and ring (neighbouring nodes share their ring)
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lr = (uint32_t)__shfl((int)ring, leader);
            const bool mine = ring == lr;
            unsigned long long x = mine ? pf : ~0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(x, o); x = y < x ? y : x; }
            if (lane == leader) atomicMin(&prefix_min[lr], x);
            todo &= ~__ballot(mine);
        }
    }
}
template <int W>
__global__ __launch_bounds__(256) void k_sring_min2(Graph<W> g, const uint2 *__restrict__ ol, const uint32_t *__restrict__ ring_of,
                                                    const unsigned long long *__restrict__ prefix_min, uint32_t *__restrict__ vmin) {
    const uint32_t total = g.n * 2;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
        const uint32_t sl = ol[v].x;
        if (sl == NIL) continue;
        const uint32_t ring = ring_of[sl];
        if (ring == NIL || km_prefix64<W>(g.keys, v >> 1, g.k) != prefix_min[ring]) continue;
        uint32_t cur = atomicCAS(&vmin[ring], NIL, v);
        while (cur != NIL && cur != v && node_key_less<W>(g, v, cur)) {
            const uint32_t prev = atomicCAS(&vmin[ring], cur, v);
            if (prev == cur) break;
            cur = prev;
        }
    }
}
// what this rank offers per ring: {key words, orientation, position in the unitig record} of its smallest node (all ones: none)
template <int W>
__global__ __launch_bounds__(256) void k_sring_report(Graph<W> g, const uint2 *__restrict__ ol, const FinRec *__restrict__ fin, uint32_t lbase,
                                                      const uint32_t *__restrict__ vmin, uint32_t n_rings, uint64_t *__restrict__ out /* [n_rings][W + 2] */) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rings; r += gridDim.x * blockDim.x) {
        const uint32_t v = vmin[r];
        uint64_t *o = out + (uint64_t)r * (W + 2);
        if (v == NIL) { for (int w = 0; w < W + 2; w++) o[w] = ~0ull; continue; }
        const Kmer<W> x = g.keys.load(v >> 1);
#pragma unroll
        for (int w = 0; w < W; w++) o[w] = x.w[w];
        o[W] = v & 1u;
        const uint2 t = ol[v];
        o[W + 1] = (uint64_t)fin[lbase + t.x].base + t.y;
    }
}

// ---- 6. emission: every rank writes the bases of its own nodes ---------------------------------------------------
struct ULayout { unsigned long long off /* byte offset of the contig text; ~0: not emitted */, node_off, ring_len /* 0: linear */, rot; };
template <int W>
__global__ __launch_bounds__(256) void k_shard_emit(Graph<W> g, const uint2 *__restrict__ ol, const FinRec *__restrict__ fin, uint32_t lbase,
                                                    const uint32_t *__restrict__ uid_of_slot, const ULayout *__restrict__ lay,
                                                    char *__restrict__ out) {
    const uint32_t ACGT = 0x54474341u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) {
        const uint4 o2 = *reinterpret_cast<const uint4 *>(&ol[2u * i]);
        if (o2.x == NIL && o2.z == NIL) continue;
        const Kmer<W> x = g.keys.load(i);
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const uint32_t sl = o ? o2.z : o2.x;
            if (sl == NIL) continue;
            const FinRec f = fin[lbase + sl];
            if (f.slot == NIL) continue;
            const ULayout L = lay[uid_of_slot[f.slot]];
            if (L.off == ~0ull) continue;
            unsigned long long pos = L.node_off + f.base + (o ? o2.w : o2.y);
            if (L.ring_len) pos = pos >= L.rot ? pos - L.rot : pos + L.ring_len - L.rot;
            char *dst = out + L.off;
            const uint32_t b = o ? 3u - km_bits2<W>(x, 2 * (g.k - 1)) : km_last_base<W>(x);
            dst[g.k - 1 + pos] = (char)((ACGT >> (8 * b)) & 0xFF);
            if (pos == 0) {
                const Kmer<W> y = o ? km_revcomp<W>(x, g.k) : x;
                for (int j = 0; j + 1 < g.k; j++) dst[j] = (char)((ACGT >> (8 * km_bits2<W>(y, 2 * (g.k - 1 - j)))) & 0xFF);
            }
        }
    }
}

// The contig text crosses the ranks 2 bits per base: every position is written by exactly one rank, the others hold zero
// bytes there; packed (A = 0 = "not mine"), the ranks' words add up to the whole text, a quarter of the bytes on the links.
__global__ __launch_bounds__(256) void k_text_pack2(const char *__restrict__ text, uint64_t n_bytes, uint32_t *__restrict__ words, uint64_t n_words) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + 16 * w);              // (the text buffer is padded to 16 bytes and zeroed)
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t c = (q[j] >> (8 * b)) & 0xFFu;                            // 0, 'A' 0x41, 'C' 0x43, 'G' 0x47, 'T' 0x54
                const uint32_t code = c == 0x43u ? 1u : (c == 0x47u ? 2u : (c == 0x54u ? 3u : 0u));
                out |= code << (2 * (4 * j + b));
            }
        words[w] = out;
    }
    (void)n_bytes;
}
__global__ __launch_bounds__(256) void k_text_unpack2(const uint32_t *__restrict__ words, uint64_t n_words, char *__restrict__ text) {
    const uint32_t ACGT = 0x54474341u;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = words[w];
        uint32_t q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) o |= ((ACGT >> (8 * ((x >> (2 * (4 * j + b))) & 3u))) & 0xFFu) << (8 * b);
            q[j] = o;
        }
        uint4 v; v.x = q[0]; v.y = q[1]; v.z = q[2]; v.w = q[3];
        *reinterpret_cast<uint4 *>(text + 16 * w) = v;
    }
}
