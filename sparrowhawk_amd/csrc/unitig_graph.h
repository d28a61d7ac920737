// unitig_graph.h — SPEC S9 (tips, bubbles) and S10 (maximal chains of simple links) on the graph of UNITIGS instead of
// the graph of k-mers.  Used by the sharded assembly (DESIGN.md "Multi-GPU"): the k-mer-level work — adjacency, the
// contraction of non-branching paths into unitigs — is spread over the ranks on their GPUs; what is left is a graph
// with one vertex per unitig (a handful for an isolate), small enough to be corrected identically on every rank's
// host.  Replaces, for that path, the phases `assembly:correct_graph` and the top level of `assembly:collapse_graph`
// (/root/reference/www/src/components/pages/AssemblyPage.vue:598-602).
//
// Why this is exact.  A unitig is a maximal chain of SIMPLE links (S10); every other edge of the k-mer graph leaves
// the LAST node of a chain and enters the FIRST node of a chain (a node with a simple successor has no other
// out-edge, one with a simple predecessor no other in-edge).  So the whole graph is: chains, plus overlaps between
// chain ends — and those follow from the first and last k-mer of every chain alone.  The walks of S9 stop or go on
// by out-/in-degrees, which inside a chain are 1/1 by construction, and count nodes, which a chain contributes as
// its length; removals take whole chains (a tip starts at a node without predecessor and ends before a junction; a
// bubble branch starts behind a fork and ends before a junction).  Run on chain records the rules are the same rules.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace shk {

static constexpr uint32_t UG_NIL = 0xFFFFFFFFu;

// One STRAND of a unitig: the chain v_1 -> ... -> v_n as it is spelled.  Its mirror strand rc(v_n) -> ... -> rc(v_1) is
// a record of its own (first = revcomp(last of this one)).  K-mers: 2k-bit integers, first base most significant, in
// W = ceil(2k/64) words, w[0] least significant (kmer.h), spelled in the orientation of the strand.
struct UnitigRec {
    uint64_t first[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t len = 0;            // nodes
    uint64_t kc = 0;             // sum of their counts
    uint32_t circ = 0;           // the chain closes on itself: no ends, no edges to anything else
};
// the smallest oriented node key(x, o) = (canonical k-mer, orientation) among the nodes of a record, and where it sits
struct UnitigMinKey {
    uint64_t key[8] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
    uint32_t o = 1;
    uint64_t pos = 0;            // its position in the record's chain (0 = first node)
    bool valid = false;
};
struct UnitigContig {
    std::vector<uint32_t> recs;  // strand records in the order they are spelled
    bool ring = false;
    uint64_t rot = 0;            // rings: the spelling starts at node `rot` of the concatenation (S10: before the smallest k-mer)
    uint64_t len_nodes = 0, kc = 0;
};

struct UnitigGraphResult {
    std::vector<UnitigContig> contigs;      // every contig once, on one strand (the writer picks min(seq, revcomp) later)
    std::vector<uint32_t> mirror;           // per record: its mirror strand's record, UG_NIL for rings / none
    // rings are only settled once the smallest k-mer of their records is known: resolve_rings()
    std::vector<uint32_t> need_min;         // records whose UnitigMinKey resolve_rings() wants
    uint64_t tips_removed = 0, bubbles_removed = 0;   // nodes (as the k-mer-level passes count them)
    int rounds = 0;
};

// S9 rounds on the records (tips unless !tips, bubbles unless !bubbles), then S10 chains over what is left.
// Returns 0, or -1 with err (inconsistent input: a linear record without its mirror strand).
int unitig_assemble(int k, const std::vector<UnitigRec> &recs, bool tips, bool bubbles, UnitigGraphResult &out, std::string &err);
// min_of[r] must be valid for every r in out.need_min.  Fixes strand and rotation of the ring contigs, drops the
// mirror-strand duplicates of rings that were rings from the start.
int unitig_resolve_rings(int k, const std::vector<UnitigRec> &recs, const std::vector<UnitigMinKey> &min_of, UnitigGraphResult &out,
                         std::string &err);

}  // namespace shk
