// inflate_gpu.h — one gzip member inflated ON THE DEVICE (inflate_gpu.hip): the compressed bytes are what crosses PCIe,
// the FASTQ text is born in HBM and goes straight to the device parser (fastq_gpu.h).  Same two-pass scheme as the host
// reader (inflate_mt.cpp, after pugz / rapidgzip) with thousands of chunks instead of one per host thread.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>
#include "fastq_gpu.h"

namespace shk {

struct GpuInflateStats {
    double h2d_ms = 0, search_ms = 0, decode_ms = 0, windows_ms = 0, resolve_ms = 0, total_ms = 0;
    uint64_t chunks = 0, text_bytes = 0;
    const char *why_not = "";            // when the member was not taken: the reason (for the logs and the tests)
};

// gz[0..n): ONE plain gzip member (not BGZF, nothing behind its trailer) of FASTQ-like text.  Returns
//   0  the text is on `device` in out (a GpuText as gpu_upload_text makes them: trailing blank lines cut, 32 zero bytes
//      behind it); the bytes are exactly what zlib would produce — CRC-32 and ISIZE of the trailer verified;
//   1  not taken: too small, several members, stored / binary data, no block starts found, a chunk that does not end where
//      the next begins, more output than the room, a checksum that does not match — the caller inflates on the host
//      (which also owns the error messages of a damaged stream);
//  <0  -4 out of device memory, -5 HIP error.
// raw: every byte of the member stays as it is (out.e = the member's size; the tests compare with zlib) — otherwise the text
// is made ready for the parser (trailing blank lines cut, see above).
int gpu_inflate_member(const uint8_t *gz, size_t n, int device, void *stream, GpuText &out, std::string &err,
                       GpuInflateStats *stats = nullptr, bool raw = false);

}  // namespace shk
