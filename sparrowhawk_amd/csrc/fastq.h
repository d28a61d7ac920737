// fastq.h — host FASTQ/gzip ingestion, quality masking, segmenting and 2-bit packing (SPEC S1-S2).
// Replaces the reader the reference's crate builds over web_sys::File + gz sniff + seq_io
// (/root/reference/AGENTS.md:180-183; sibling in tree: rust/orphos-bridge/src/fastx_wasm.rs:53-70).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <functional>
#include <string>
#include <vector>
#include "bytebuf.h"

namespace shk {

struct PackedReads {
    std::vector<uint32_t> bases;     // 2-bit stream, base i in bits [2*(i%16)+1 : 2*(i%16)] of word i/16
    std::vector<uint32_t> seg_off;   // n_seg+1 base offsets
    uint64_t n_bases = 0;            // bases in the packed stream (valid segments >= k only)
    uint64_t n_reads = 0;            // FASTQ records seen
    uint64_t n_input_bases = 0;      // bases in the FASTQ records
    uint32_t cur = 0;                // partial word being filled
    void clear();
    void reset_stream();             // drop the packed stream (a batch was handed on), keep the read counters
    void finish();                   // flush the partial word, pad one spare word
    uint64_t n_seg() const { return seg_off.empty() ? 0 : seg_off.size() - 1; }
};

// progress(reads_so_far, bytes_consumed, bytes_total) is called every `every` reads (0 = never)
using ProgressFn = std::function<void(uint64_t, uint64_t, uint64_t)>;

// Appends the reads of one FASTQ buffer (plain or gzip) to `out`.  0 or SHK_E_PARSE(-3)/-4.
// flush(out) — optional — is called after a record when out.n_reads is a multiple of flush_reads (if non-zero)
// or the stream holds >= flush_bases bases: the caller takes the batch (PackedReads::finish + upload) and
// calls out.reset_stream().  Non-zero return aborts the parse with that code.
using FlushFn = std::function<int(PackedReads &)>;
int pack_fastq(const uint8_t *buf, size_t n, uint32_t k, uint32_t min_qual, PackedReads &out,
               std::string &err, uint64_t every = 0, const ProgressFn &progress = nullptr,
               uint64_t flush_reads = 0, uint64_t flush_bases = 0, const FlushFn &flush = nullptr,
               uint64_t rec_base = 0 /* records of this file that came before buf: numbering of the error messages */);

// gzip sniff (1F 8B) + multi-member inflate; plain input is passed through (p/n point at buf or at `storage`)
int maybe_inflate(const uint8_t *buf, size_t n, ByteVec &storage, const uint8_t *&p, size_t &pn,
                  std::string &err);
// the two files of a pair in two threads (b2 may be null); BGZF input is inflated block-parallel either way
int maybe_inflate_pair(const uint8_t *b1, size_t n1, const uint8_t *b2, size_t n2, ByteVec &s1,
                       ByteVec &s2, const uint8_t *&p1, size_t &l1, const uint8_t *&p2, size_t &l2,
                       std::string &err);

}  // namespace shk
