"""A brute-force graph stage in pure Python, written from SPEC.md S7-S11 ALONE: dict of k-mer strings, explicit
oriented nodes, the tip / bubble / unitig / circular-cut rules spelled out literally.  It shares no code and no
data structure with oracle/shk_oracle.c, oracle/cpu_mt.cpp or the product (no adjacency bytes, no hashing, no
integer k-mers: nodes are strings), so a misreading of S8-S10 shared by those three would show up here.
Test infrastructure (tests/test_oracle.py pins the oracle's alive(), adjacency() and contigs with it).

Quadratic in places and slow by design: inputs of a few thousand nodes."""

COMP = str.maketrans("ACGT", "TGCA")
BASES = "ACGT"


def rc(s):
    return s.translate(COMP)[::-1]


class PyGraph:
    def __init__(self, counts, k, threshold):
        """counts: {canonical k-mer string: count}.  S7: solid <=> count > threshold.  S8: nodes = solid k-mers."""
        self.k = k
        self.count = {x: c for x, c in counts.items() if c > threshold}
        for x in self.count:
            assert len(x) == k and x <= rc(x) and x != rc(x)           # canonical; odd k: never its own reverse complement
        self.alive = set(self.count)
        self._memo_out, self._memo_in = {}, {}          # neighbour lists of the CURRENT node set (dropped on every removal)

    # ---- S8: oriented nodes are (x, o); seq(x, 0) = x, seq(x, 1) = rc(x) -------------------------------------
    @staticmethod
    def seq(v):
        return v[0] if v[1] == 0 else rc(v[0])

    @staticmethod
    def mirror(v):
        return (v[0], 1 - v[1])

    def node_of(self, s):
        """the oriented node spelled s, or None when its k-mer is not an (alive) node"""
        r = rc(s)
        x = s if s < r else r
        if x not in self.alive:
            return None
        return (x, 0 if s == x else 1)

    def outs(self, v):
        """u -> w iff seq(w)[0..k-1) == seq(u)[1..k)"""
        res = self._memo_out.get(v)
        if res is None:
            s = self.seq(v)
            res = []
            for b in BASES:
                w = self.node_of(s[1:] + b)
                if w is not None:
                    res.append(w)
            self._memo_out[v] = res
        return res

    def ins(self, v):
        res = self._memo_in.get(v)
        if res is None:
            s = self.seq(v)
            res = []
            for b in BASES:
                w = self.node_of(b + s[:-1])
                if w is not None:
                    res.append(w)
            self._memo_in[v] = res
        return res

    def _remove(self, removed):
        self.alive -= removed
        self._memo_out.clear()
        self._memo_in.clear()

    def adjacency_byte(self, x):
        """S8: bit b = edge from (x,0) to the node spelled x[1:] + base b; bit 4+b = edge into (x,0) from base b + x[:-1]"""
        a = 0
        for i, b in enumerate(BASES):
            if self.node_of(x[1:] + b) is not None:
                a |= 1 << i
            if self.node_of(b + x[:-1]) is not None:
                a |= 1 << (4 + i)
        return a

    def oriented_nodes(self):
        for x in sorted(self.alive):
            yield (x, 0)
            yield (x, 1)

    # ---- S9 ---------------------------------------------------------------------------------------------------
    def tip_round(self):
        T = 2 * self.k
        attached = {}                                   # junction J -> list of tips (lists of oriented nodes)
        for v in self.oriented_nodes():
            if len(self.ins(v)) != 0:
                continue
            P, cur = [v], v
            while True:
                o = self.outs(cur)
                if len(o) != 1:
                    break                               # not a tip
                n = o[0]
                if len(self.ins(n)) >= 2:
                    attached.setdefault(n, []).append(P)    # a tip attached to n
                    break
                P.append(n)
                cur = n
                if len(P) > T:
                    break                               # not a tip
        removed = set()
        for J, tips in attached.items():
            d, t = len(self.ins(J)), len(tips)
            assert t <= d
            if t < d:
                doomed = tips
            else:
                # the best tip is kept: maximum of (|P|, sum of counts, smaller first canonical k-mer)
                def key(P):
                    return (len(P), sum(self.count[u[0]] for u in P), _Rev(P[0][0]))
                best = max(tips, key=key)
                doomed = [P for P in tips if P is not best]
            for P in doomed:
                for u in P:
                    removed.add(u[0])
        self._remove(removed)
        return len(removed)

    def bubble_round(self):
        T = 2 * self.k
        removed = set()
        for S in self.oriented_nodes():
            o = self.outs(S)
            if len(o) < 2:
                continue
            by_end = {}
            for b in o:
                if len(self.ins(b)) != 1:
                    continue
                B, cur, E = [b], b, None
                while True:
                    oc = self.outs(cur)
                    if len(oc) != 1:
                        break                           # dead end or fork: no branch
                    n = oc[0]
                    ind = len(self.ins(n))
                    if ind >= 2:
                        E = n
                        break
                    B.append(n)                         # ind == 1
                    cur = n
                    if len(B) > T:
                        break                           # too long: no branch
                if E is not None:
                    by_end.setdefault(E, []).append(B)
            for E, branches in by_end.items():
                if len(branches) < 2:
                    continue
                if not (S <= self.mirror(E)):           # evaluated from one side only: key(S) <= key(rc(E)), key = (x, o)
                    continue
                best = branches[0]
                for B in branches[1:]:
                    if self._better(B, best):
                        best = B
                for B in branches:
                    if B is not best:
                        for u in B:
                            removed.add(u[0])
        self._remove(removed)
        return len(removed)

    def _better(self, A, B):
        """A beats B: higher mean count (exact cross-multiplication), then fewer nodes, then the smaller canonical
        k-mer of the branch's first node"""
        sa, sb = sum(self.count[u[0]] for u in A), sum(self.count[u[0]] for u in B)
        if sa * len(B) != sb * len(A):
            return sa * len(B) > sb * len(A)
        if len(A) != len(B):
            return len(A) < len(B)
        return A[0][0] < B[0][0]

    def correct(self, tips=True, bubbles=True, max_rounds=32):
        self.tips_removed = self.bubbles_removed = 0
        for _ in range(max_rounds):
            a = self.tip_round() if tips else 0
            b = self.bubble_round() if bubbles else 0
            self.tips_removed += a
            self.bubbles_removed += b
            if a + b == 0:
                break

    # ---- S10 --------------------------------------------------------------------------------------------------
    def simple_succ(self, u):
        o = self.outs(u)
        if len(o) != 1:
            return None
        v = o[0]
        if len(self.ins(v)) != 1 or v == u or v == self.mirror(u):
            return None
        return v

    def simple_pred(self, v):
        i = self.ins(v)
        if len(i) != 1:
            return None
        u = i[0]
        if len(self.outs(u)) != 1 or v == u or v == self.mirror(u):
            return None
        return u

    def unitigs(self):
        """-> list of (sequence, [oriented nodes], circular) with every unitig once, as min(seq, revcomp(seq))"""
        seen = set()
        chains = []
        nodes = list(self.oriented_nodes())
        for v in nodes:                                 # linear chains start at a node without a simple in-link
            if self.simple_pred(v) is None:
                chain, cur = [v], v
                while True:
                    n = self.simple_succ(cur)
                    if n is None:
                        break
                    chain.append(n)
                    cur = n
                seen.update(chain)
                chains.append((chain, False))
        for v in nodes:                                 # what is left closes on itself
            if v in seen:
                continue
            cyc, cur = [v], v
            while True:
                cur = self.simple_succ(cur)
                assert cur is not None
                if cur == v:
                    break
                cyc.append(cur)
            seen.update(cyc)
            # cut before the oriented node with the smallest key (x, o) on the cycle, EITHER strand: the cycle that
            # holds (xmin, 0) is the one spelled; its mirror strand (found later or earlier) is the same unitig
            smallest = min(min(cyc), min(self.mirror(u) for u in cyc))
            if smallest in cyc:
                i = cyc.index(smallest)
                chains.append((cyc[i:] + cyc[:i], True))
            else:
                chains.append((None, True))             # the other strand of this ring carries the cut
        out = []
        for chain, circular in chains:
            if chain is None:
                continue
            xs = [u[0] for u in chain]
            assert len(set(xs)) == len(xs)              # a chain never holds both orientations of a node
            s = self.seq(chain[0]) + "".join(self.seq(u)[-1] for u in chain[1:])
            r = rc(s)
            assert s != r
            if circular:
                out.append((min(s, r), chain if s < r else [self.mirror(u) for u in reversed(chain)], True))
            elif s < r:                                 # the mirror chain is in the list too: emitted once
                out.append((s, chain, False))
        return out

    # ---- S11: order, FASTA, links, GFA1 -----------------------------------------------------------------------
    def assembly(self):
        us = self.unitigs()
        us.sort(key=lambda t: (-len(t[0]), t[0]))
        k = self.k
        contigs = []
        first_of = {}                                   # first oriented node of (contig, orientation) -> (i, '+'/'-')
        for i, (s, chain, _c) in enumerate(us, start=1):
            kc = sum(self.count[u[0]] for u in chain)
            contigs.append((s, kc))
            first_of[chain[0]] = (i, "+")
            first_of[self.mirror(chain[-1])] = (i, "-")
        links = set()
        sign_rank = {"+": 0, "-": 1}
        flip = {"+": "-", "-": "+"}
        for i, (s, chain, _c) in enumerate(us, start=1):
            for o, last in (("+", chain[-1]), ("-", self.mirror(chain[0]))):
                for w in self.outs(last):
                    if w in first_of:
                        j, oj = first_of[w]
                        a = (i, sign_rank[o], j, sign_rank[oj])
                        b = (j, sign_rank[flip[oj]], i, sign_rank[flip[o]])
                        links.add(min(a, b))
        fasta = "".join(f">contig_{i} len={len(s)} kc={kc}\n{s}\n" for i, (s, kc) in enumerate(contigs, start=1))
        sg = "+-"
        gfa = "H\tVN:Z:1.0\n" + "".join(f"S\t{i}\t{s}\tLN:i:{len(s)}\tKC:i:{kc}\n" for i, (s, kc) in enumerate(contigs, start=1))
        gfa += "".join(f"L\t{a}\t{sg[oa]}\t{b}\t{sg[ob]}\t{k - 1}M\n" for (a, oa, b, ob) in sorted(links))
        self.n_rings = sum(1 for _s, _chain, circ in us if circ)
        return contigs, fasta, gfa


class _Rev:
    """orders strings descending inside a tuple that is otherwise compared ascending (max() then prefers the SMALLER k-mer)"""
    __slots__ = ("s",)

    def __init__(self, s):
        self.s = s

    def __lt__(self, o):
        return self.s > o.s

    def __gt__(self, o):
        return self.s < o.s

    def __eq__(self, o):
        return self.s == o.s
