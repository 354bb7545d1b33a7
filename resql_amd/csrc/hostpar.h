// hostpar.h — the host tail's worker pool and its two parallel primitives.
//
// The part of a query above the last pipeline breaker runs on the host over #groups rows (tail.cpp).  With a million groups
// that is tens of MB of data and — until round 3 — 85 ms of one execution whose kernels take 8: the tail is now split over a
// persistent pool of host threads (starting threads per phase cost more than the phases), and the one step that looks
// sequential, the replay of the reference's hash table (hostref.h), is cut into its independent probe clusters.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

namespace rsq {

// workers of the pool + the calling thread; at most 16 (a GPU box gives one GPU's process a share of the host's cores),
// RSQ_TAIL_THREADS overrides
int hostThreads();

// fn(part) for part = 0 .. parts-1 on the pool (the caller takes parts too).  One parallel region at a time per process;
// a region started while another runs (shards of a multi-GPU plan finishing together) runs on the calling thread alone.
// An exception in any part is re-thrown on the calling thread.
void parallelRun(int parts, const std::function<void(int)>& fn);

// fn(begin, end, part) over [0, n) in `parts` contiguous ranges
void parallelRanges(size_t n, int parts, const std::function<void(size_t, size_t, int)>& fn);
// how many parts a loop over n cheap items deserves (1 below ~16 K items)
int partsFor(size_t n);

// idx[0..n) = the permutation that sorts keys ascending, stable (equal keys keep their index order): parallel LSD radix sort,
// 11-bit digits over the bits the keys use.  scratch vectors are grown as needed and can be kept between calls.
struct SortScratch { std::vector<uint64_t> k0, k1; std::vector<uint32_t> i0, i1; std::vector<size_t> hist; };
void parallelSortIndex(const uint64_t* keys, size_t n, std::vector<uint32_t>& idx, SortScratch& scratch);

}  // namespace rsq
