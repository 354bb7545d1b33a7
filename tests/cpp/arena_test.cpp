// Exercises resql_amd/csrc/mempool.cpp on the host (malloc slabs): ranges never overlap, a freed range is not handed out before promote(),
// everything coalesces back into whole slabs, trim releases them.  Built and run by tests/test_mempool.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "mempool.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "arena_test: %s failed at line %d\n", #c, __LINE__); return 1; } } while (0)

int main() {
    size_t rawLive = 0, rawCalls = 0;
    std::map<void*, size_t> raw;
    rsq::Arena a([&](size_t b) -> void* { if (b > (64u << 20)) return nullptr; void* p = malloc(b); raw[p] = b; rawLive += b; rawCalls++; return p; },
                 [&](void* p) { rawLive -= raw[p]; raw.erase(p); free(p); }, 1u << 20, 256);
    a.reserve(1u << 20);
    CHECK(rawCalls == 1 && a.slabBytes() >= (1u << 20));
    std::mt19937_64 rng(7);
    std::map<char*, size_t> live;           // what we hold
    std::vector<std::pair<char*, size_t>> freedNotPromoted;
    auto overlaps = [&](char* p, size_t n, const std::map<char*, size_t>& m) {
        for (auto& kv : m) if (p < kv.first + kv.second && kv.first < p + n) return true;
        return false;
    };
    for (int step = 0; step < 20000; step++) {
        const unsigned op = (unsigned)(rng() % 100);
        if (op < 55 || live.empty()) {
            const size_t n = 1 + (size_t)(rng() % ((rng() % 8 == 0) ? (3u << 20) : 5000u));
            char* p = (char*)a.alloc(n);
            CHECK(p != nullptr);
            CHECK(((uintptr_t)p & 255) == 0);
            CHECK(!overlaps(p, n, live));
            for (auto& f : freedNotPromoted) CHECK(!(p < f.first + f.second && f.first < p + n));      // pending ranges are not reused
            memset(p, (int)(step & 255), n);
            live[p] = n;
        } else if (op < 95) {
            auto it = live.begin(); std::advance(it, (long)(rng() % live.size()));
            CHECK(a.free(it->first));
            freedNotPromoted.emplace_back(it->first, it->second);
            live.erase(it);
        } else { a.promote(); freedNotPromoted.clear(); }
    }
    CHECK(!a.free((void*)&rng));            // a foreign pointer is refused
    for (auto& kv : live) CHECK(a.free(kv.first));
    a.promote();
    CHECK(a.usedBytes() == 0 && a.freeBytes() == a.slabBytes());
    CHECK(a.alloc((size_t)65 << 20) == nullptr);      // the driver refuses: nullptr, nothing leaks
    const size_t slabs = a.slabBytes();
    a.trim(slabs);                          // nothing above the cap
    CHECK(a.slabBytes() == slabs);
    a.trim(0);
    CHECK(a.slabBytes() == 0 && rawLive == 0);
    void* p = a.alloc(100);                 // grows again afterwards
    CHECK(p && a.free(p));
    a.releaseAll();
    CHECK(rawLive == 0);
    printf("arena_test ok: %zu driver allocations for 20000 steps\n", rawCalls);
    return 0;
}
