// hugepages.hpp -- transparent huge pages for the schedule builder's large host arrays.
//
// At 100 M ratings build_schedule allocates, zero-fills and frees some 400 MB of vectors; with 4 KiB pages that is
// 100 K page faults and as many page-table entries to tear down (38 ms of a 0.41 s set_ratings went into the frees
// alone).  Where the system offers transparent huge pages on request (`madvise` in
// /sys/kernel/mm/transparent_hugepage/enabled, as on the MI355X hosts) a vector is reserved, its untouched memory is
// advised, and only then filled: the 2 MiB-aligned part of it is faulted in and freed 512 pages at a time.  Elsewhere
// the advice is refused or ignored and nothing changes.
#pragma once

#include <sys/mman.h>

#include <cstddef>
#include <cstdint>

namespace mfsgd {

inline void advise_huge(const void* p, size_t bytes) {
#ifdef MADV_HUGEPAGE
    constexpr uintptr_t kPage = 4096, kHuge = (uintptr_t)2 << 20;
    if (!p || bytes < 4 * kHuge) return;  // small arrays: nothing to gain
    const uintptr_t lo = ((uintptr_t)p + kPage - 1) & ~(kPage - 1), hi = ((uintptr_t)p + bytes) & ~(kPage - 1);
    if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), (size_t)(hi - lo), MADV_HUGEPAGE);
#else
    (void)p;
    (void)bytes;
#endif
}

// v.reserve(n) with the advice given before anything touches the memory; the caller resizes / assigns afterwards.
template <class V>
inline void reserve_huge(V& v, size_t n) {
    v.reserve(n);
    advise_huge(v.data(), v.capacity() * sizeof(typename V::value_type));
}

}  // namespace mfsgd
