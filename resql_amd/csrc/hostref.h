// hostref.h — host-side pieces of the result path that run over the (few) group rows after the
// device pipelines: the reference's group emission order, its quicksort, and scalar evaluation.
//
// Why the emission order matters: without ORDER BY the reference emits groups in the slot order of
// its aggregation hash table, and with ORDER BY its (unstable) quicksort starts from that order,
// so ties come out in an order that depends on it.  To hand back a result relation that is
// byte-identical to ReSQL's — not just equal as a multiset — the engine replays the reference's
// insertion sequence (groups ordered by the first input row that produced them) through the same
// hash function, prime table sizes, 60 % growth rule and linear probing
// (reference src/ValuesJitFlounder.h:65-142, src/qlib/hash.h:32-95, 225-287, 330-419).
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "expr.h"

namespace rsq {

// Values::hash for one value (reference src/ValuesJitFlounder.h:65-142)
uint64_t refHashValue(uint64_t h, Val v, const Type& t);

// Slot order of the reference's aggregation hash table after inserting `hashes` in this order into a
// table allocated for `minSize` (allocateHashTable(getSize(), ..), aggregation.h:253).  Returns, for
// every slot in ascending slot order, the index into `hashes` of the group stored there.
std::vector<size_t> refEmissionOrder(const std::vector<uint64_t>& hashes, uint64_t minSize);

// The same slot order computed on the host's worker pool (hostpar.h): the table is cut into its independent probe clusters
// (hostref.cpp explains).  hashes[i] in insertion order; order[k] = index into hashes of the group in the k-th occupied slot.
// The scratch buffers are grown as needed; keep them between calls (a query's executions) to avoid fresh pages every time.
struct ReplayScratch {
    std::vector<uint64_t> ts, byChunk, sorted;
    std::vector<uint32_t> start, slotWho;
    std::vector<size_t> chunkStart, hist;
    std::vector<int64_t> chunkA, chunkB, chunkIn;
};
void refEmissionOrderParallel(const uint64_t* hashes, size_t n, uint64_t minSize, std::vector<uint32_t>& order, ReplayScratch& scratch);

// Quicksorter (reference src/qlib/sort.h:21-173): Lomuto partition, pivot = last element, over
// packed tuples.
struct OrderRequest { int offset; Type type; bool asc; };
// compare<> per type as the Quicksorter uses it (reference src/types.h:264-353, src/qlib/sort.h:138-160)
int compareTyped(const Type& t, const uint8_t* l, const uint8_t* r);
void refQuicksort(uint8_t* tuples, int64_t n, size_t tupleSize, const std::vector<OrderRequest>& order);

// stringLikeCheck (reference src/qlib/scalar.h:49-118) on NUL-terminated strings: '%' any run, '_' any one character; the
// host-side twin of rsq_device.h like() for LIKE expressions evaluated above an aggregation
bool refLike(const char* s, const char* pattern);

// packed tuple access (reference src/values.h:151-232)
void storeValue(uint8_t* addr, Val v, const Type& t);      // strings by value, NUL terminated
Val loadValue(const uint8_t* addr, const Type& t);

}  // namespace rsq
