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

template <typename T> struct NoInitAlloc {
    using value_type = T;
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        void *p;
        if (bytes >= ((size_t)4 << 20)) {
            const size_t want = (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
            p = aligned_alloc((size_t)2 << 20, want);
            if (p) (void)madvise(p, want, MADV_HUGEPAGE);
        } else p = malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t) { free(p); }
    template <typename U, typename... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) (void)p;          // default construction: leave the memory as it is
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
    template <typename U> bool operator==(const NoInitAlloc<U> &) const { return true; }
    template <typename U> bool operator!=(const NoInitAlloc<U> &) const { return false; }
};
using ByteVec = std::vector<uint8_t, NoInitAlloc<uint8_t>>;

}  // namespace shk
