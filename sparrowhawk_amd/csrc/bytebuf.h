// bytebuf.h — std::vector storage for large host buffers that are WRITTEN IN FULL right after they are sized: elements are
// left uninitialised on resize (no serial zero-fill of a gigabyte before many threads overwrite it) and large blocks are
// 2 MiB-aligned and advised to use huge pages (first-touch page faults from many threads serialise in the kernel; with
// huge pages there are 512 times fewer).  Used by the gzip reader (fastq.cpp, inflate_mt.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <new>
#include <utility>
#include <vector>

namespace shk {

// Blocks of >= 4 MiB are 2 MiB-aligned, advised to use huge pages and — up to a few gigabytes — kept for the next caller
// when they are freed: a fresh gigabyte costs the kernel a gigabyte of page zeroing on first touch (0.3 s per .fastq.gz of
// the bench isolate), a recycled one costs nothing.  big_trim() returns the kept blocks (shk_release_cached_memory).
void *big_alloc(size_t bytes);
void big_free(void *p, size_t bytes);
void big_trim();

template <typename T> struct NoInitAlloc {
    using value_type = T;
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    T *allocate(size_t n) {
        void *p = big_alloc(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t n) { big_free(p, n * sizeof(T)); }
    template <typename U, typename... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) (void)p;          // default construction: leave the memory as it is
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
    template <typename U> bool operator==(const NoInitAlloc<U> &) const { return true; }
    template <typename U> bool operator!=(const NoInitAlloc<U> &) const { return false; }
};
using ByteVec = std::vector<uint8_t, NoInitAlloc<uint8_t>>;

}  // namespace shk
