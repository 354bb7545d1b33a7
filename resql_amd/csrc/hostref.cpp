// hostref.cpp — see hostref.h.
#include <cstring>
#include "hostref.h"
#include "hostpar.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>

namespace rsq {

static const uint64_t kPrimes[] = {
    5ull, 11ull, 23ull, 47ull, 97ull, 199ull, 409ull, 823ull, 1741ull, 3469ull, 6949ull, 14033ull,
    28411ull, 57557ull, 116731ull, 236897ull, 480881ull, 976369ull, 1982627ull, 4026031ull,
    8175383ull, 16601593ull, 33712729ull, 68460391ull, 139022417ull, 282312799ull, 573292817ull,
    1164186217ull, 2364114217ull, 4294967291ull, 8589934583ull, 17179869143ull, 34359738337ull,
    68719476731ull, 137438953447ull, 274877906899ull, 549755813881ull, 1099511627689ull,
    2199023255531ull, 4398046511093ull, 8796093022151ull, 17592186044399ull, 35184372088777ull,
    70368744177643ull, 140737488355213ull, 281474976710597ull, 562949953421231ull,
    1125899906842597ull, 2251799813685119ull, 4503599627370449ull, 9007199254740881ull,
    18014398509481951ull, 36028797018963913ull, 72057594037927931ull, 144115188075855859ull,
    288230376151711717ull, 576460752303423433ull, 1152921504606846883ull, 2305843009213693951ull,
    4611686018427387847ull, 9223372036854775783ull};   // the 61 sizes the reference's upper_bound searches

static uint64_t primeAbove(uint64_t minSize) {
    if (minSize < 2) minSize = 2;
    for (uint64_t p : kPrimes) if (minSize < p) return p;
    return 18446744073709551557ull;
}

uint64_t refHashValue(uint64_t h, Val v, const Type& t) {
    const uint64_t A = 1710227316115945415ull, B = 741332713408129251ull;
    switch (t.tag) {
        case RSQ_BIGINT: case RSQ_DECIMAL: return h + ((uint64_t)v.i * A + B);
        case RSQ_INT: return h + ((uint64_t)(int64_t)(int32_t)v.i + B) * A;
        case RSQ_DATE: return h + ((uint64_t)(int64_t)(int32_t)(uint32_t)v.i + B) * A;
        case RSQ_BOOL: return ((uint8_t)v.i == 0) ? h + 31636373ull : h;
        case RSQ_CHAR:
            if (t.len == 1) { h += (uint64_t)(uint8_t)v.i; return h + h; }
            {   // hashChar: fixed length, missing characters count as ' '
                const char* s = v.s;
                for (int i = 0; i < t.len; i++) {
                    char c; if (*s != '\0') { c = *s; s++; } else c = ' ';
                    int32_t m = (int32_t)((uint32_t)(int)c * 31636373u);
                    h = h + (uint64_t)(int64_t)m + (uint64_t)(int64_t)c;
                }
                return h;
            }
        case RSQ_VARCHAR: {
            const char* s = v.s;
            for (int i = 0; i < t.len && *s != '\0'; i++, s++) {
                int c = *s;
                int32_t m = (int32_t)((uint32_t)c * 31636373u);
                h = h + (uint64_t)(int64_t)m + (uint64_t)(int64_t)c;
            }
            return h;
        }
        default: failType("Values::hash(..) not implemented for datatype");
    }
}

namespace {
// The reference's table has numEntries slots, but only the occupied ones matter: keep them in an
// ordered map (slot -> {hash, group}) so replaying a handful of groups does not touch a table sized
// for millions of input rows.
struct Sim {
    uint64_t numEntries, threshold, numInserts = 0;
    std::map<uint64_t, std::pair<uint64_t, size_t>> slots;
    explicit Sim(uint64_t minSize) {
        numEntries = primeAbove(minSize);
        threshold = numEntries * 6 / 10;
    }
    void put(uint64_t h, size_t id) {
        numInserts++;
        if (numInserts > threshold) grow();
        uint64_t loc = h % numEntries;
        for (uint64_t n = 0; n < numEntries; n++) {
            if (slots.find(loc) == slots.end()) { slots[loc] = {h, id}; return; }
            if (++loc >= numEntries) loc = 0;
        }
        failRuntime("Hash table full");
    }
    void grow() {
        Sim bigger(numEntries + 1);
        for (auto& kv : slots) bigger.put(kv.second.first, kv.second.second);   // old table in slot order
        *this = std::move(bigger);
    }
};
}  // namespace

namespace {
// Same replay with the table held as arrays: used when there are many groups (an ordered map of a million slots
// costs far more than the arrays of the table it simulates).
struct DenseSim {
    struct Slot { uint64_t hash; uint32_t who; uint32_t used; };     // one cache line touch per probe step
    uint64_t numEntries, threshold, numInserts = 0, magic = 0;
    std::vector<Slot> slots;
    explicit DenseSim(uint64_t minSize) {
        numEntries = primeAbove(minSize);
        threshold = numEntries * 6 / 10;
        magic = (uint64_t)((((__uint128_t)1) << 64) / numEntries);      // floor(2^64 / d), d >= 5
        slots.assign(numEntries, Slot{0, 0, 0});
    }
    // h % numEntries without the 64-bit divide (≈35 cycles, and there are two per insert): the quotient estimate
    // floor(h * floor(2^64/d) / 2^64) is at most 1 short, fixed by the conditional subtractions
    uint64_t mod(uint64_t h) const {
        uint64_t q = (uint64_t)(((__uint128_t)h * magic) >> 64);
        uint64_t r = h - q * numEntries;
        while (r >= numEntries) r -= numEntries;
        return r;
    }
    void put(uint64_t h, uint32_t id) {
        numInserts++;
        if (numInserts > threshold) grow();
        uint64_t loc = mod(h);
        for (uint64_t n = 0; n < numEntries; n++) {
            Slot& s = slots[loc];
            if (!s.used) { s.used = 1; s.hash = h; s.who = id; return; }
            if (++loc >= numEntries) loc = 0;
        }
        failRuntime("Hash table full");
    }
    void prefetch(uint64_t h) const { __builtin_prefetch(&slots[mod(h)], 1, 1); }
    void grow() {
        DenseSim bigger(numEntries + 1);
        // old table in slot order (growHashTable, qlib/hash.h:330-365); the occupied slots are gathered first so that
        // the target slot of an insert can be prefetched a few inserts ahead (the table is far larger than the caches)
        std::vector<Slot> live;
        live.reserve(numInserts);
        for (uint64_t i = 0; i < numEntries; i++) if (slots[i].used) live.push_back(slots[i]);
        for (size_t i = 0; i < live.size(); i++) {
            if (i + 12 < live.size()) bigger.prefetch(live[i + 12].hash);
            bigger.put(live[i].hash, live[i].who);
        }
        *this = std::move(bigger);
    }
};
}  // namespace

std::vector<size_t> refEmissionOrder(const std::vector<uint64_t>& hashes, uint64_t minSize) {
    if (hashes.size() >= 2048 && hashes.size() < 0xffffffffull && primeAbove(minSize) <= (1ull << 28)) {
        DenseSim sim(minSize);
        for (size_t i = 0; i < hashes.size(); i++) {
            if (i + 12 < hashes.size()) sim.prefetch(hashes[i + 12]);
            sim.put(hashes[i], (uint32_t)i);
        }
        std::vector<size_t> order;
        order.reserve(hashes.size());
        for (uint64_t s = 0; s < sim.numEntries; s++) if (sim.slots[s].used) order.push_back(sim.slots[s].who);
        return order;
    }
    Sim sim(minSize);
    for (size_t i = 0; i < hashes.size(); i++) sim.put(hashes[i], i);
    std::vector<size_t> order;
    order.reserve(hashes.size());
    for (auto& kv : sim.slots) order.push_back(kv.second.second);
    return order;
}

// ---- the same replay, cut into its independent probe clusters ------------------------------------------------------
// Linear probing with first-come-first-served inserts: WHICH slots end up occupied does not depend on the insertion order
// (slot s is occupied iff carry(s) + home-count(s) > 0 with carry(s+1) = max(0, carry(s) + count(s) - 1)), and an empty slot
// is a wall no probe sequence crosses.  So the table falls into clusters — maximal runs of occupied slots — that can be
// replayed independently, each in the order of its items' timestamps.  One level of the reference's table (between two
// growths) is then: bucket the items by slot chunk, counting-sort every chunk by home slot (cache-sized), reduce every
// chunk to its carry function x -> max(A, x + B) and chain those (the only sequential step: one multiply-free pass over a
// few hundred chunk summaries; the cyclic carry into slot 0 is the fixed point A of the whole chain since fewer items than
// slots make B negative), then walk the chunks in parallel and replay every cluster where it starts.  Growth (a full rehash
// in slot order of the old table, qlib/hash.h:330-365) makes the next level's timestamps: old entries by their old slot,
// newer groups behind them in input order.  A million groups: 55 ms as a sequential replay on one core of the GPU box
// (the table is far larger than the caches and every insert is a dependent miss), a few ms this way.
namespace {

struct ReplayLevel {
    uint64_t N = 0;
    size_t cnt = 0;
    static constexpr uint64_t CHUNK_BITS = 12, CHUNK = 1ull << CHUNK_BITS;
};

// items [0, cnt) with hashes h[] and timestamps ts[] into a table of N slots: slotWho[s] = item in slot s or EMPTY
void replayLevelByClusters(const uint64_t* h, const uint64_t* ts, size_t cnt, uint64_t N, ReplayScratch& S) {
    constexpr uint32_t EMPTY = 0xffffffffu;
    constexpr uint64_t CB = ReplayLevel::CHUNK_BITS, CH = ReplayLevel::CHUNK;
    const size_t nChunks = (size_t)((N + CH - 1) >> CB);
    const int parts = partsFor(std::max<size_t>(cnt, (size_t)N / 4));
    if (S.slotWho.size() < N) S.slotWho.resize(N);
    if (S.start.size() < N + 1) S.start.resize(N + 1);
    if (S.byChunk.size() < cnt) { S.byChunk.resize(cnt); S.sorted.resize(cnt); }
    S.chunkStart.assign(nChunks + 1, 0);
    S.chunkA.assign(nChunks, 0); S.chunkB.assign(nChunks, 0); S.chunkIn.assign(nChunks + 1, 0);
    const uint64_t magic = (uint64_t)((((__uint128_t)1) << 64) / N);
    auto mod = [&](uint64_t x) { uint64_t q = (uint64_t)(((__uint128_t)x * magic) >> 64); uint64_t r = x - q * N; while (r >= N) r -= N; return r; };
    static const bool traceReplay = getenv("RSQ_TRACE") && atoi(getenv("RSQ_TRACE")) >= 2;
    auto nowUs = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tStep = traceReplay ? nowUs() : 0;
    auto step = [&](const char* what) {
        if (!traceReplay) return;
        const double t = nowUs();
        fprintf(stderr, "[rsq replay]   N=%llu items=%zu  %8.0f us  %s\n", (unsigned long long)N, cnt, t - tStep, what);
        tStep = t;
    };
    // (1) items -> chunk buckets: (home << 32 | item), stable within a bucket is not needed (timestamps order a cluster's items)
    std::vector<size_t>& hist = S.hist;
    hist.assign((size_t)parts * nChunks, 0);
    parallelRanges(cnt, parts, [&](size_t b, size_t e, int p) {
        size_t* hc = hist.data() + (size_t)p * nChunks;
        for (size_t i = b; i < e; i++) { const uint64_t home = mod(h[i]); S.byChunk[i] = (home << 32) | (uint64_t)i; hc[home >> CB]++; }
    });
    {
        size_t pos = 0;
        for (size_t c = 0; c < nChunks; c++) {
            S.chunkStart[c] = pos;
            for (int p = 0; p < parts; p++) { size_t k = hist[(size_t)p * nChunks + c]; hist[(size_t)p * nChunks + c] = pos; pos += k; }
        }
        S.chunkStart[nChunks] = pos;
    }
    uint64_t* bucketed = S.sorted.data();          // first use of `sorted`: the bucketed pairs; the per-chunk sort writes back into byChunk
    parallelRanges(cnt, parts, [&](size_t b, size_t e, int p) {
        size_t* hc = hist.data() + (size_t)p * nChunks;
        for (size_t i = b; i < e; i++) { const uint64_t v = S.byChunk[i]; bucketed[hc[v >> (32 + CB)]++] = v; }
    });
    step("items bucketed by slot chunk");
    // (2) every chunk: counting sort by home slot, start[] offsets, carry summary
    uint64_t* sorted = S.byChunk.data();
    const int cparts = (int)std::min<size_t>((size_t)hostThreads(), std::max<size_t>(1, nChunks / 4));
    parallelRanges(nChunks, cparts, [&](size_t cb, size_t ce, int) {
        std::vector<uint32_t> local(CH + 1);
        for (size_t c = cb; c < ce; c++) {
            const uint64_t s0 = (uint64_t)c << CB, s1 = std::min<uint64_t>(N, s0 + CH);
            const size_t ib = S.chunkStart[c], ie = S.chunkStart[c + 1];
            std::fill(local.begin(), local.end(), 0u);
            for (size_t i = ib; i < ie; i++) local[(size_t)((bucketed[i] >> 32) - s0) + 1]++;
            int64_t A = 0, B = 0;                   // x -> max(A, x + B) of the slots so far
            for (uint64_t s = s0; s < s1; s++) {
                const int64_t k = (int64_t)local[(size_t)(s - s0) + 1] - 1;
                A = std::max<int64_t>(0, A + k); B += k;
            }
            S.chunkA[c] = A; S.chunkB[c] = B;
            uint32_t run = (uint32_t)ib;
            for (uint64_t s = s0; s < s1; s++) { const uint32_t k = local[(size_t)(s - s0) + 1]; local[(size_t)(s - s0)] = run; S.start[s] = run; run += k; }
            for (size_t i = ib; i < ie; i++) { const uint64_t v = bucketed[i]; sorted[local[(size_t)((v >> 32) - s0)]++] = v; }
        }
    });
    S.start[N] = (uint32_t)cnt;
    step("chunks sorted by home slot");
    // (3) the carry into every chunk: the cyclic carry into slot 0 is the fixed point of the whole chain
    {
        int64_t A = 0;
        for (size_t c = 0; c < nChunks; c++) A = std::max(S.chunkA[c], A + S.chunkB[c]);
        int64_t x = A;                               // F(A) = max(A, A + B) = A because B = items - slots < 0
        for (size_t c = 0; c < nChunks; c++) { S.chunkIn[c] = x; x = std::max(S.chunkA[c], x + S.chunkB[c]); }
    }
    step("carries chained");
    // (4) walk the chunks: empty slots, slots of clusters begun earlier, clusters that begin here (replayed by timestamp)
    uint32_t* who = S.slotWho.data();
    const uint32_t* start = S.start.data();
    auto count = [&](uint64_t s) { return (int64_t)(start[s + 1] - start[s]); };
    parallelRanges(nChunks, cparts, [&](size_t cb, size_t ce, int) {
        struct Item { uint64_t ts; uint32_t off; uint32_t item; };
        std::vector<Item> items;
        std::vector<uint32_t> occ;
        for (size_t c = cb; c < ce; c++) {
            const uint64_t s0 = (uint64_t)c << CB, s1 = std::min<uint64_t>(N, s0 + CH);
            int64_t x = S.chunkIn[c];
            uint64_t s = s0;
            while (s < s1) {
                const int64_t k = count(s);
                if (x > 0) { x += k - 1; s++; continue; }
                if (k == 0) { who[s] = EMPTY; s++; continue; }
                // a cluster begins at s: it ends at the first slot where as many items have their home in [s, e] as there are slots
                uint64_t e = s, ew = s; int64_t tot = 0;                  // ew = e modulo N (clusters may run around the table's end)
                for (;;) { tot += count(ew); if (tot == (int64_t)(e - s + 1)) break; e++; if (++ew == N) ew = 0; }
                const size_t len = (size_t)(e - s + 1);
                if (len == 1) { who[s] = (uint32_t)sorted[start[s]]; s = e + 1; continue; }
                if (len <= 24) {
                    // the common case: a handful of items — insertion sort by timestamp in registers / stack, no allocation
                    Item small[24]; uint32_t occS[24];
                    size_t k = 0;
                    uint64_t slot = s;
                    for (uint64_t t = s; t <= e; t++) {
                        for (uint32_t i = start[slot]; i < start[slot + 1]; i++) {
                            const uint32_t it = (uint32_t)sorted[i];
                            const Item x{ts[it], (uint32_t)(t - s), it};
                            size_t j = k++;
                            while (j > 0 && small[j - 1].ts > x.ts) { small[j] = small[j - 1]; j--; }
                            small[j] = x;
                        }
                        if (++slot == N) slot = 0;
                    }
                    for (size_t j = 0; j < len; j++) occS[j] = EMPTY;
                    for (size_t i = 0; i < k; i++) { size_t j = small[i].off; while (occS[j] != EMPTY) j++; occS[j] = small[i].item; }
                    slot = s;
                    for (size_t j = 0; j < len; j++) { who[slot] = occS[j]; if (++slot == N) slot = 0; }
                    s = e + 1;
                    continue;
                }
                items.clear();
                {
                    uint64_t slot = s;
                    for (uint64_t t = s; t <= e; t++) {
                        for (uint32_t i = start[slot]; i < start[slot + 1]; i++) { const uint32_t it = (uint32_t)sorted[i]; items.push_back({ts[it], (uint32_t)(t - s), it}); }
                        if (++slot == N) slot = 0;
                    }
                }
                std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.ts < b.ts; });
                occ.assign(len, EMPTY);
                for (const Item& it : items) { size_t j = it.off; while (occ[j] != EMPTY) j++; occ[j] = it.item; }
                { uint64_t slot = s; for (size_t j = 0; j < len; j++) { who[slot] = occ[j]; if (++slot == N) slot = 0; } }
                s = e + 1;                           // (the carry behind a cluster is 0 by construction)
            }
        }
    });
    step("clusters replayed");
}

}  // namespace

void refEmissionOrderParallel(const uint64_t* hashes, size_t n, uint64_t minSize, std::vector<uint32_t>& order, ReplayScratch& S) {
    constexpr uint32_t EMPTY = 0xffffffffu;
    order.resize(n);
    if (n == 0) return;
    uint64_t N = primeAbove(minSize);
    bool fallback = n < 8192 || n >= 0xfffffff0ull;
    // the level sizes of the reference's counter (DenseSim::put above: the insert that triggers a growth is not counted again)
    if (!fallback) {
        uint64_t c = 0, live = 0, M = N;
        for (;;) {
            const uint64_t th = M * 6 / 10;
            if (th < c || M >= (1ull << 32)) { fallback = true; break; }
            const uint64_t atGrowth = live + (th - c);
            if (n <= atGrowth) break;
            c = atGrowth; live = atGrowth + 1; M = primeAbove(M + 1);
        }
    }
    if (fallback) {
        std::vector<uint64_t> hv(hashes, hashes + n);
        std::vector<size_t> o = refEmissionOrder(hv, minSize);
        for (size_t i = 0; i < n; i++) order[i] = (uint32_t)o[i];
        return;
    }
    if (S.ts.size() < n) S.ts.resize(n);
    uint64_t* ts = S.ts.data();
    parallelRanges(n, partsFor(n), [&](size_t b, size_t e, int) { for (size_t i = b; i < e; i++) ts[i] = (uint64_t)i; });
    uint64_t c = 0, live = 0;
    for (;;) {
        const uint64_t th = N * 6 / 10;
        const uint64_t atGrowth = live + (th - c);
        const size_t cnt = (size_t)std::min<uint64_t>(n, atGrowth);
        replayLevelByClusters(hashes, ts, cnt, N, S);
        const uint32_t* who = S.slotWho.data();
        if (n <= atGrowth) {
            // final table: the items in slot order
            constexpr uint64_t CH = ReplayLevel::CHUNK;
            const size_t nChunks = (size_t)((N + CH - 1) / CH);
            const int parts = (int)std::min<size_t>((size_t)hostThreads(), std::max<size_t>(1, nChunks / 4));
            std::vector<size_t> occupied(nChunks + 1, 0);
            parallelRanges(nChunks, parts, [&](size_t cb, size_t ce, int) {
                for (size_t ch = cb; ch < ce; ch++) {
                    size_t k = 0;
                    for (uint64_t s = ch * CH, s1 = std::min<uint64_t>(N, s + CH); s < s1; s++) k += who[s] != EMPTY;
                    occupied[ch + 1] = k;
                }
            });
            for (size_t ch = 0; ch < nChunks; ch++) occupied[ch + 1] += occupied[ch];
            if (occupied[nChunks] != n) failRuntime("internal error: the replayed hash table holds a different number of groups");
            parallelRanges(nChunks, parts, [&](size_t cb, size_t ce, int) {
                for (size_t ch = cb; ch < ce; ch++) {
                    size_t o = occupied[ch];
                    for (uint64_t s = ch * CH, s1 = std::min<uint64_t>(N, s + CH); s < s1; s++) if (who[s] != EMPTY) order[o++] = who[s];
                }
            });
            return;
        }
        // growth: the old table's entries re-enter in slot order, the groups behind them in input order
        parallelRanges((size_t)N, partsFor((size_t)N), [&](size_t b, size_t e, int) { for (size_t s = b; s < e; s++) if (who[s] != EMPTY) ts[who[s]] = (uint64_t)s; });
        parallelRanges(n - cnt, partsFor(n - cnt), [&](size_t b, size_t e, int) { for (size_t i = b; i < e; i++) ts[cnt + i] = N + (uint64_t)(cnt + i); });
        c = atGrowth; live = atGrowth + 1;
        N = primeAbove(N + 1);
    }
}

void storeValue(uint8_t* addr, Val v, const Type& t) {
    switch (t.tag) {
        case RSQ_DATE: { uint32_t x = (uint32_t)v.i; memcpy(addr, &x, 4); break; }
        case RSQ_BOOL: { uint8_t x = (uint8_t)v.i; memcpy(addr, &x, 1); break; }
        case RSQ_INT: { int32_t x = (int32_t)v.i; memcpy(addr, &x, 4); break; }
        case RSQ_BIGINT: case RSQ_DECIMAL: memcpy(addr, &v.i, 8); break;
        case RSQ_CHAR:
            if (t.len == 1) { addr[0] = (uint8_t)v.i; break; }
            [[fallthrough]];
        case RSQ_VARCHAR: {
            size_t i = 0, max = (size_t)t.len;
            for (; i < max; i++) { addr[i] = (uint8_t)v.s[i]; if (v.s[i] == '\0') break; }
            addr[i] = '\0';
            break;
        }
        default: failType("storeToMem(..) not implemented for datatype");
    }
}

Val loadValue(const uint8_t* addr, const Type& t) {
    Val v; v.i = 0;
    switch (t.tag) {
        case RSQ_DATE: { uint32_t x; memcpy(&x, addr, 4); v.i = x; break; }
        case RSQ_BOOL: v.i = addr[0]; break;
        case RSQ_INT: { int32_t x; memcpy(&x, addr, 4); v.i = x; break; }
        case RSQ_BIGINT: case RSQ_DECIMAL: memcpy(&v.i, addr, 8); break;
        case RSQ_CHAR: if (t.len == 1) { v.i = addr[0]; break; } [[fallthrough]];
        case RSQ_VARCHAR: v.s = (const char*)addr; break;
        default: failType("loadAttributeToReg(..) not implemented for datatype");
    }
    return v;
}

int compareTyped(const Type& t, const uint8_t* l, const uint8_t* r) {
    switch (t.tag) {
        case RSQ_BIGINT: case RSQ_DECIMAL: { int64_t a, b; memcpy(&a, l, 8); memcpy(&b, r, 8); return a < b ? -1 : a > b; }
        case RSQ_INT: case RSQ_DATE: { int32_t a, b; memcpy(&a, l, 4); memcpy(&b, r, 4); return a < b ? -1 : a > b; }
        case RSQ_BOOL: { uint8_t a = l[0] != 0, b = r[0] != 0; return a < b ? -1 : a > b; }
        case RSQ_CHAR: case RSQ_VARCHAR: return (int)(int8_t)strcmp((const char*)l, (const char*)r);
        default: return 0;
    }
}

// LIKE with the reference's results (stringLikeCheck, src/qlib/scalar.h:49-118; pinned by tests/golden/like_reference.json).
// Same three steps as the device version (kernels/rsq_device.h like()): the pattern's literal head, its literal tail
// compared from the ends (it may reach into characters the head used), then the '%'-separated segments in between,
// each at its leftmost place behind the one before.
namespace {
struct LikeText {
    const char* p; int n;
    char operator[](int i) const { return i >= 0 && i < n ? p[i] : '\0'; }
};
inline bool likeSame(char c, char pat) { return pat == '_' || c == pat; }
}
bool refLike(const char* str, const char* pattern) {
    const LikeText S{str, (int)strlen(str)}, P{pattern, (int)strlen(pattern)};
    int head = 0;
    if (P[0] != '%') {
        while (head < P.n && head < S.n && P[head] != '%') { if (!likeSame(S[head], P[head])) return false; head++; }
        if (head == P.n) return head == S.n;
    }
    int patEnd = P.n, strEnd = S.n;
    if (P[P.n - 1] != '%') {
        int k = 0;
        while (k < P.n && k < S.n && P[P.n - 1 - k] != '%') { if (!likeSame(S[S.n - 1 - k], P[P.n - 1 - k])) return false; k++; }
        patEnd = P.n - 1 - k; strEnd = S.n - k;
        if (head >= patEnd) return true;
    }
    int seg = head + 1;
    for (int at = head; at < strEnd && seg < patEnd; at++) {
        int j = 0;
        while (at + j < strEnd && likeSame(S[at + j], P[seg + j])) {
            if (P[seg + j + 1] == '%') { at += j; seg += j + 2; break; }
            j++;
        }
    }
    return seg >= patEnd;
}

void refQuicksort(uint8_t* data, int64_t n, size_t ts, const std::vector<OrderRequest>& order) {
    if (n < 2) return;
    std::vector<uint8_t> tmp(ts);
    auto before = [&](const uint8_t* a, const uint8_t* b) {
        for (const auto& o : order) {
            int c = compareTyped(o.type, a + o.offset, b + o.offset);
            if (o.asc) { if (c < 0) return true; if (c > 0) return false; }
            else { if (c > 0) return true; if (c < 0) return false; }
        }
        return false;
    };
    auto swp = [&](int64_t i, int64_t j) {
        if (i == j) return;
        memcpy(tmp.data(), data + (size_t)i * ts, ts);
        memcpy(data + (size_t)i * ts, data + (size_t)j * ts, ts);
        memcpy(data + (size_t)j * ts, tmp.data(), ts);
    };
    std::vector<std::pair<int64_t, int64_t>> stack;
    stack.emplace_back(0, n - 1);
    while (!stack.empty()) {
        auto [lo, hi] = stack.back(); stack.pop_back();
        if (!(lo < hi)) continue;
        const uint8_t* pivot = data + (size_t)hi * ts;
        int64_t i = lo;
        for (int64_t j = lo; j < hi; ++j) if (before(data + (size_t)j * ts, pivot)) { swp(i, j); ++i; }
        swp(i, hi);
        stack.emplace_back(i + 1, hi);
        stack.emplace_back(lo, i - 1);
    }
}

}  // namespace rsq
