// outputs.h — contig canonicalisation, ordering, links and the FASTA / GFA1 / GFA2 / DOT / JSON
// writers (SPEC S10-S11).  Replaces the crate's `assembly:saving` phase and get_assembly()
// (AssemblyPage.vue:604-605; www/src/workers/Assembler.ts:7-13,127-137;
// www/src/components/DownloadButton.vue:46-57).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "bytebuf.h"
#include "pipeline.h"

namespace shk {

struct AssemblyText {
    uint64_t ncontigs = 0;
    std::string fasta, dot, gfa1, gfa2;
    ByteVec json;                              // NUL-terminated (size() = text + 1); uninitialised storage written once by the writer's threads
    std::vector<std::pair<std::string, double>> stage_ms;     // where the writer's time went (host clock)
};

// arrival (optional): the contigs' text (RawContig::ext) is still arriving from the device; the contigs then carry their
// first / last bases (head / tail), and the writer copies every range of text as soon as it is there
void build_assembly_text(std::vector<RawContig> &contigs, uint32_t k, AssemblyText &out, TextArrival *arrival = nullptr);
// the writer's worker threads will be needed within about this long: they wake up now (shk_assemble calls this when
// the graph phases of a megabase assembly start, so that the copies of the contigs do not wait for sleeping threads)
void writer_prewarm(long microseconds);
std::string preprocessing_json(uint64_t nkmers, const uint64_t *histo500, uint32_t used_min_count);
void json_escape_into(std::string &dst, const std::string &s);

}  // namespace shk
