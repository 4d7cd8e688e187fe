// host_ranges.h -- which caller host memory may reach the HIP runtime as a raw pointer (pure host logic, no HIP: unit-tested on
// the CPU by tests/san/ranges_driver.cpp).
//
// Rule of the library since round 4: an asynchronous copy is handed a caller's host pointer ONLY while that memory
// lies inside a range the caller page-locked through nvca_host_register and has not released.  Everything else -- pageable
// memory, and memory that WAS registered once -- crosses through page-locked memory of the context's own (api.cpp, BounceRing).
// The table also remembers which streams carried a copy out of / into a registered range since it was registered:
// nvca_host_unregister drains exactly those before the pages are released (a copy still in flight on the copy stream or on a
// lane must never lose its pages).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace nvca {

struct HostRangeTable {
    struct Range { uintptr_t lo, hi; uint64_t streams; };     // [lo, hi); streams: bit per stream id that carried a copy of it
    std::vector<Range> live;            // registered now
    std::vector<Range> retired;         // released since (kept for diagnostics: NVCA_ALLOC_LOG names a fault address's history)
    static constexpr size_t kRetiredMax = 256;

    // a range overlapping a live one is refused (the runtime would refuse it too): false
    bool add(const void *p, size_t bytes)
    {
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        if (!p || !bytes || hi < lo) return false;
        for (const Range &r : live) if (lo < r.hi && r.lo < hi) return false;
        live.push_back(Range{lo, hi, 0});
        return true;
    }
    // index of the live range that starts at p, -1: none (hipHostUnregister takes the pointer the range was registered with)
    int find(const void *p) const
    {
        for (size_t i = 0; i < live.size(); i++) if (live[i].lo == (uintptr_t)p) return (int)i;
        return -1;
    }
    // the whole of [p, p + bytes) inside ONE live range?  (a copy that straddles the end of a registered range is not registered memory)
    int covering(const void *p, size_t bytes) const
    {
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        if (!p || hi < lo) return -1;
        for (size_t i = 0; i < live.size(); i++) if (lo >= live[i].lo && hi <= live[i].hi) return (int)i;
        return -1;
    }
    bool registered(const void *p, size_t bytes) const { return covering(p, bytes) >= 0; }
    // a direct copy of [p, p + bytes) is about to be queued on stream `id` (0 .. 63): true = allowed (and remembered), false = bounce it
    bool note_copy(const void *p, size_t bytes, int id)
    {
        const int i = covering(p, bytes);
        if (i < 0) return false;
        live[(size_t)i].streams |= 1ull << (id & 63);
        return true;
    }
    // release the range registered at p: the streams that must be drained BEFORE the pages go (bit mask); found = false: not registered
    uint64_t remove(const void *p, bool *found)
    {
        const int i = find(p);
        if (found) *found = i >= 0;
        if (i < 0) return 0;
        const Range r = live[(size_t)i];
        live.erase(live.begin() + i);
        if (retired.size() >= kRetiredMax) retired.erase(retired.begin());
        retired.push_back(r);
        return r.streams;
    }
    // was any byte of [p, p + bytes) registered once and released since?  (diagnostics only: such memory is bounced like any other)
    bool was_registered(const void *p, size_t bytes) const
    {
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        for (const Range &r : retired) if (lo < r.hi && r.lo < hi) return true;
        return false;
    }
};

} // namespace nvca
