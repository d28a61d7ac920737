// inflate_mt.h — one gzip member inflated by many host threads (two-pass speculative decoding: see inflate_mt.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "bytebuf.h"

namespace shk {

// Inflates the gzip member at in[0..n) into out[out_at...) (out is resized) with up to `threads` host threads.
// Returns 0 and sets `consumed` (header + deflate data + trailer of that member) when it did — the bytes are then exactly
// what zlib would produce (CRC-32 and ISIZE of the member's trailer verified) —, 1 when it did not apply or anything looked
// unexpected (out is left at out_at bytes): the caller inflates with zlib instead.
int inflate_member_parallel(const uint8_t *in, size_t n, ByteVec &out, size_t out_at, size_t &consumed, unsigned threads);

// members the multi-threaded inflater has handled in this process (statistics; the tests assert the path was taken)
uint64_t inflate_mt_members();

}  // namespace shk
