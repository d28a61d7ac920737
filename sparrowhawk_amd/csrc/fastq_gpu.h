// fastq_gpu.h — device-side FASTQ parser/packer (fastq_gpu.hip); same packed layout as fastq.h.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace shk {

struct GpuPacked {
    uint32_t *d_bases = nullptr;      // device: ceil(n_bases/16)+1 words
    uint32_t *d_seg_off = nullptr;    // device: n_seg+1
    size_t bases_bytes = 0, seg_off_bytes = 0;      // pool block sizes
    double h2d_ms = 0, kernels_ms = 0;
    uint64_t n_seg = 0, n_bases = 0, n_reads = 0, n_input_bases = 0;
    // one entry per `every` reads: bytes consumed inside the file the batch ends in; bit 63 = second file
    std::vector<unsigned long long> progress_bytes;
    uint64_t first_mark = 0;          // entry j belongs to read number every * (first_mark + j + 1)
};

// a FASTQ text already in device memory (trailing blank lines cut off, 32 zero bytes behind it): uploaded by
// gpu_upload_text, possibly from a helper thread while the previous piece is being parsed
struct GpuText {
    uint8_t *d = nullptr; size_t pool_bytes = 0;
    size_t e = 0;                     // bytes of text
    bool unterminated = false;        // the last line lacks its newline
    double h2d_ms = 0;
};
// blocking; uses a stream of its own on `device` (callable from any thread)
int gpu_upload_text(const uint8_t *t, size_t n, int device, GpuText &out, std::string &err);
void gpu_text_free(GpuText &t);

// t1/t2: plain FASTQ texts in host memory (t2 may be null).  Returns 0 = packed on the device,
// 1 = input is not regular 4-line FASTQ (the caller runs the host parser, which owns the error
// messages and the blank-line rules), < 0 = error (-4 memory, -5 HIP, -1 too large).
// read_base: records that came before this text (earlier pieces of the same input) — progress marks
// stay at global multiples of `every`.
int gpu_pack_fastq(const uint8_t *t1, size_t n1, const uint8_t *t2, size_t n2, uint32_t k, uint32_t min_qual,
                   uint64_t every, void *stream, GpuPacked &out, std::string &err, uint64_t read_base = 0,
                   const GpuText *uploaded = nullptr /* instead of t1 (t2 must be null): the text is on the device already */);
void gpu_packed_free(GpuPacked &p);

}  // namespace shk
