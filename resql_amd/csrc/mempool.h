// mempool.h — the context's memory arenas (device memory, pinned host memory).
//
// The reference allocates a query's hash tables with malloc and frees them with the plan (reference src/qlib/hash.h:225-287,
// operators/aggregation.h:66-70): microseconds.  hipMalloc / hipFree / hipHostMalloc are not that: a few hundred MB of device
// memory cost milliseconds (the driver maps and clears the pages), hipFree waits for the whole device, a pinned allocation is a
// system call that locks pages.  A statement that is compiled, executed ONCE and destroyed - all a ReSQL host ever does
// (reference src/execute.h:213-247) - would pay them on every execution.  So the context owns slabs and hands out ranges of them:
//   * alloc: best fit among the free ranges, split; a new slab (the only driver call) when nothing fits;
//   * free:  the range goes to a PENDING list - kernels enqueued before the free may still use it - and becomes reusable when the
//            owner says the stream has drained (promote); neighbours coalesce then;
//   * trim:  wholly free slabs go back to the driver above a cap.
// One instance per memory kind; not thread safe (one query at a time per context, like the reference's JitContextFlounder).
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <unordered_map>
#include <vector>

namespace rsq {

class Arena {
  public:
    typedef std::function<void*(size_t)> RawAlloc;      // returns nullptr when the driver refuses
    typedef std::function<void(void*)> RawFree;
    Arena(RawAlloc a, RawFree f, size_t minSlabBytes, size_t alignBytes) : rawAlloc(a), rawFree(f), minSlab(minSlabBytes), align(alignBytes) {}
    ~Arena() { releaseAll(); }
    Arena(const Arena&) = delete;
    Arena& operator=(const Arena&) = delete;

    void* alloc(size_t bytes);              // nullptr: nothing fits and the driver gave no new slab
    bool free(void* p);                     // false: not one of ours
    bool owns(const void* p) const { return used.count(const_cast<void*>(p)) != 0; }
    bool hasPending() const { return !pending.empty(); }
    bool fitsWithoutGrowing(size_t bytes) const;      // a free range of that size exists now
    void promote();                         // every pending range is free from now on
    void reserve(size_t bytes);             // make sure one free range of that size exists (a slab allocated up front)
    void trim(size_t keepFreeBytes);        // release wholly free slabs while more than keepFreeBytes are free
    void releaseAll();
    size_t slabBytes() const { return totalSlab; }
    size_t usedBytes() const { return totalUsed; }
    size_t freeBytes() const { return totalSlab - totalUsed - totalPending; }
    // counters (rsq trace / tests)
    uint64_t nAllocs = 0, nSlabAllocs = 0, nFrees = 0;
    double slabAllocMs = 0;

  private:
    struct Range { size_t size; int state; };        // state: 0 free, 1 used, 2 pending
    struct Slab { char* base; size_t size; std::map<size_t, Range> ranges; };      // by offset
    RawAlloc rawAlloc; RawFree rawFree;
    size_t minSlab, align;
    std::vector<Slab> slabs;
    std::multimap<size_t, std::pair<int, size_t>> freeBySize;       // size -> (slab, offset)
    std::unordered_map<void*, std::pair<int, size_t>> used;         // pointer -> (slab, offset)
    std::vector<std::pair<int, size_t>> pending;
    size_t totalSlab = 0, totalUsed = 0, totalPending = 0;

    void insertFree(int s, size_t off, size_t size);
    void eraseFree(int s, size_t off, size_t size);
    int newSlab(size_t bytes);
};

}  // namespace rsq
