// outputs.cpp — see outputs.h
#include "outputs.h"

#include <algorithm>
#include <tuple>
#include <unordered_map>

namespace shk {

static std::string revcomp(const std::string &s) {
    std::string r(s.size(), 'A');
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[s.size() - 1 - i];
        r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
    }
    return r;
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

void build_assembly_text(std::vector<RawContig> &contigs, uint32_t k, AssemblyText &out) {
    // SPEC S10: each unitig is emitted as min(seq, revcomp(seq))
    for (auto &c : contigs) {
        std::string r = revcomp(c.seq);
        if (r < c.seq) c.seq.swap(r);
    }
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
        tail_plus[i] = last;                          // last k-mer of '+'
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

    std::string &fa = out.fasta, &g1 = out.gfa1, &g2 = out.gfa2, &dt = out.dot;
    fa.clear(); g1 = "H\tVN:Z:1.0\n"; g2 = "H\tVN:Z:2.0\n"; dt = "digraph sparrowhawk {\n";
    for (size_t i = 0; i < nc; i++) {
        const std::string id = std::to_string(i + 1), len = std::to_string(contigs[i].seq.size()),
                          kc = std::to_string(contigs[i].kc);
        fa += ">contig_" + id + " len=" + len + " kc=" + kc + "\n"; fa += contigs[i].seq; fa += "\n";
        g1 += "S\t" + id + "\t"; g1 += contigs[i].seq; g1 += "\tLN:i:" + len + "\tKC:i:" + kc + "\n";
        g2 += "S\t" + id + "\t" + len + "\t"; g2 += contigs[i].seq; g2 += "\tKC:i:" + kc + "\n";
        dt += "  \"" + id + "\" [label=\"" + id + " len=" + len + " kc=" + kc + "\"];\n";
    }
    const std::string ov = std::to_string(k - 1);
    for (const Link &L : links) {
        const uint32_t a = std::get<0>(L), ao = std::get<1>(L), b = std::get<2>(L), bo = std::get<3>(L);
        const uint64_t la = contigs[a - 1].seq.size(), lb = contigs[b - 1].seq.size();
        g1 += "L\t" + std::to_string(a) + (ao ? "\t-\t" : "\t+\t") + std::to_string(b) + (bo ? "\t-\t" : "\t+\t") + ov + "M\n";
        g2 += "E\t*\t" + std::to_string(a) + (ao ? "-\t" : "+\t") + std::to_string(b) + (bo ? "-\t" : "+\t");
        if (!ao) g2 += std::to_string(la - (k - 1)) + "\t" + std::to_string(la) + "$\t";
        else g2 += "0\t" + ov + ((uint64_t)(k - 1) == la ? "$" : "") + "\t";
        if (!bo) g2 += "0\t" + ov + ((uint64_t)(k - 1) == lb ? "$" : "") + "\t";
        else g2 += std::to_string(lb - (k - 1)) + "\t" + std::to_string(lb) + "$\t";
        g2 += ov + "M\n";
        dt += "  \"" + std::to_string(a) + "\" -> \"" + std::to_string(b) + "\" [label=\"" + (ao ? "-" : "+") + (bo ? "-" : "+") + "\"];\n";
    }
    dt += "}\n";

    std::string &js = out.json;
    js.clear();
    js.reserve(fa.size() + g1.size() + g2.size() + dt.size() + 256);
    js += "{\"outfasta\":"; json_escape_into(js, fa);
    js += ",\"ncontigs\":" + std::to_string(nc);
    js += ",\"outdot\":"; json_escape_into(js, dt);
    js += ",\"outgfa\":"; json_escape_into(js, g1);
    js += ",\"outgfav2\":"; json_escape_into(js, g2);
    js += "}";
}

}  // namespace shk
