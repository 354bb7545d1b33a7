// mempool.cpp — see mempool.h
#include "mempool.h"

#include <algorithm>
#include <chrono>

namespace rsq {

static inline size_t roundUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

void Arena::insertFree(int s, size_t off, size_t size) { freeBySize.emplace(size, std::make_pair(s, off)); }

void Arena::eraseFree(int s, size_t off, size_t size) {
    auto r = freeBySize.equal_range(size);
    for (auto it = r.first; it != r.second; ++it)
        if (it->second.first == s && it->second.second == off) { freeBySize.erase(it); return; }
}

int Arena::newSlab(size_t bytes) {
    const size_t size = roundUp(std::max(bytes, minSlab), std::max<size_t>(align, 2u << 20));
    const auto t0 = std::chrono::steady_clock::now();
    void* base = rawAlloc(size);
    slabAllocMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!base) return -1;
    nSlabAllocs++;
    int s = -1;
    for (size_t i = 0; i < slabs.size(); i++) if (!slabs[i].base) { s = (int)i; break; }
    if (s < 0) { slabs.emplace_back(); s = (int)slabs.size() - 1; }
    slabs[(size_t)s].base = (char*)base; slabs[(size_t)s].size = size; slabs[(size_t)s].ranges.clear();
    slabs[(size_t)s].ranges[0] = Range{size, 0};
    insertFree(s, 0, size);
    totalSlab += size;
    return s;
}

bool Arena::fitsWithoutGrowing(size_t bytes) const {
    return freeBySize.lower_bound(roundUp(std::max<size_t>(bytes, 1), align)) != freeBySize.end();
}

void Arena::reserve(size_t bytes) { if (bytes && !fitsWithoutGrowing(bytes)) (void)newSlab(bytes); }

void* Arena::alloc(size_t bytes) {
    const size_t need = roundUp(std::max<size_t>(bytes, 1), align);
    auto it = freeBySize.lower_bound(need);
    if (it == freeBySize.end()) {
        if (newSlab(need) < 0) return nullptr;
        it = freeBySize.lower_bound(need);
        if (it == freeBySize.end()) return nullptr;
    }
    const int s = it->second.first; const size_t off = it->second.second; const size_t have = it->first;
    freeBySize.erase(it);
    Slab& sl = slabs[(size_t)s];
    Range& r = sl.ranges[off];
    r.state = 1;
    if (have > need) {            // the rest stays free
        r.size = need;
        sl.ranges[off + need] = Range{have - need, 0};
        insertFree(s, off + need, have - need);
    }
    void* p = sl.base + off;
    used[p] = std::make_pair(s, off);
    totalUsed += need;
    nAllocs++;
    return p;
}

bool Arena::free(void* p) {
    auto it = used.find(p);
    if (it == used.end()) return false;
    const int s = it->second.first; const size_t off = it->second.second;
    used.erase(it);
    Range& r = slabs[(size_t)s].ranges[off];
    r.state = 2;
    totalUsed -= r.size; totalPending += r.size;
    pending.emplace_back(s, off);
    nFrees++;
    return true;
}

void Arena::promote() {
    for (auto& pr : pending) {
        const int s = pr.first; size_t off = pr.second;
        Slab& sl = slabs[(size_t)s];
        auto it = sl.ranges.find(off);
        size_t size = it->second.size;
        totalPending -= size;
        // coalesce with the free neighbour behind, then with the one in front
        auto nx = std::next(it);
        if (nx != sl.ranges.end() && nx->second.state == 0) { eraseFree(s, nx->first, nx->second.size); size += nx->second.size; sl.ranges.erase(nx); }
        if (it != sl.ranges.begin()) {
            auto pv = std::prev(it);
            if (pv->second.state == 0) { eraseFree(s, pv->first, pv->second.size); size += pv->second.size; off = pv->first; sl.ranges.erase(it); it = pv; }
        }
        it->second.size = size; it->second.state = 0;
        insertFree(s, off, size);
    }
    pending.clear();
}

void Arena::trim(size_t keepFreeBytes) {
    for (size_t s = 0; s < slabs.size() && freeBytes() > keepFreeBytes; s++) {
        Slab& sl = slabs[s];
        if (!sl.base || sl.ranges.size() != 1 || sl.ranges.begin()->second.state != 0) continue;
        eraseFree((int)s, 0, sl.size);
        rawFree(sl.base);
        totalSlab -= sl.size;
        sl.base = nullptr; sl.size = 0; sl.ranges.clear();
    }
}

void Arena::releaseAll() {
    for (auto& sl : slabs) if (sl.base) rawFree(sl.base);
    slabs.clear(); freeBySize.clear(); used.clear(); pending.clear();
    totalSlab = totalUsed = totalPending = 0;
}

}  // namespace rsq
