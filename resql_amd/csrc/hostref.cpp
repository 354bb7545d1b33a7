// hostref.cpp — see hostref.h.
#include <cstring>
#include "hostref.h"

#include <algorithm>
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
