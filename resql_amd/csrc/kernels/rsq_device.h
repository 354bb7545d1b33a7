// rsq_device.h — hand-written CDNA4 (gfx950) device-side building blocks of the ReSQL pipelines.
//
// Every pipeline kernel the engine launches is this header plus a short query-specific row
// function emitted by codegen.cpp (the analogue of the reference's operators emitting Flounder IR
// around a hand-written runtime library, reference src/qlib/*.h).  Compiled with hiprtc (or
// hipcc --genco at build time) for --offload-arch=gfx950 only; wave size is 64.
//
// Contents
//   * integer arithmetic with the reference's x86 semantics (64-bit wrap-around add/sub/imul,
//     truncating idiv; reference src/ExpressionsJitFlounder.h:298-440)
//   * coalesced tile loads: one wave owns a tile of 128 consecutive rows, lane l owns rows
//     2l and 2l+1, so an 8-byte column is ONE global_load_dwordx4 per lane (1 KiB per wave
//     instruction, unit stride), a 4-byte column one dwordx2, a 1-byte column one ushort
//   * wave64 reductions in the VALU (DPP row shifts + row broadcasts, wave_reduce_to_lane63; a
//     __shfl_xor butterfly only where one value per wave is reduced) and the flush of per-thread
//     accumulators: registers -> wave reduce -> LDS cell -> one global atomic per cell and workgroup
//   * fixed-width CHAR(n)/VARCHAR(n) comparison (reference src/qlib/scalar.h:16-46)
//   * open-addressing hash tables in HBM for join build/probe and large group-by
#pragma once

typedef long long i64;
typedef unsigned long long u64;
typedef int i32;
typedef unsigned int u32;
typedef unsigned char u8;
typedef unsigned short u16;

#define RSQ_WAVE 64
#define RSQ_TILE_ROWS 128          /* rows per wave tile: 64 lanes x 2 rows */
#define RSQ_DEV __device__ inline __attribute__((always_inline))

namespace rsq {

// error word bits (set by kernels, read by the host after the final sync)
enum { ERR_DIV_ZERO = 1, ERR_HT_FULL = 2, ERR_DUP_KEY = 4, ERR_GROUP_OVERFLOW = 8, ERR_STUCK = 16,
       NOTE_CHAR_GROUP_ENDS_WITH_SPACE = 32 /* not an error: a CHAR(n) group value ends with a space (host merges groups) */,
       NOTE_BUILD_KEYS_NOT_UNIQUE = 64 /* not an error: two build rows of a join declared single-match share a key (the table stays a hash table) */ };

// ---- arithmetic: x86-64 add / sub / imul wrap, cqo+idiv truncates ---------------------------
RSQ_DEV i64 add(i64 a, i64 b) { return (i64)((u64)a + (u64)b); }
RSQ_DEV i64 sub(i64 a, i64 b) { return (i64)((u64)a - (u64)b); }
RSQ_DEV i64 mul(i64 a, i64 b) { return (i64)((u64)a * (u64)b); }
RSQ_DEV i64 div(i64 a, i64 b, u32* err) {
    // the reference's generated code traps (SIGFPE) on /0 and INT64_MIN / -1; we flag and carry on
    if (b == 0 || (a == (i64)0x8000000000000000ull && b == -1)) { atomicOr(err, (u32)ERR_DIV_ZERO); return 0; }
    return a / b;
}

// ---- tile loads -----------------------------------------------------------------------------
struct __attribute__((aligned(16))) i64x2 { i64 x, y; };
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };
struct __attribute__((aligned(8))) i32x2 { i32 x, y; };
struct __attribute__((aligned(2))) u8x2 { u8 x, y; };

// `p` points at the first of this lane's two consecutive rows; 16 / 8 / 2 byte aligned because
// tiles start at multiples of 128 rows and column bases are 256-byte aligned.
#ifdef RSQ_NT_LOADS
// streamed-once data: non-temporal loads (global_load_dwordx4 ... nt) leave the caches to the tables that are re-read
typedef i64 i64v2 __attribute__((ext_vector_type(2)));
typedef i32 i32v2 __attribute__((ext_vector_type(2)));
RSQ_DEV void ld2(const i64* p, i64 (&v)[2]) { i64v2 t = __builtin_nontemporal_load(reinterpret_cast<const i64v2*>(p)); v[0] = t.x; v[1] = t.y; }
RSQ_DEV void ld2(const i32* p, i32 (&v)[2]) { i32v2 t = __builtin_nontemporal_load(reinterpret_cast<const i32v2*>(p)); v[0] = t.x; v[1] = t.y; }
RSQ_DEV void ld2(const u8* p, u8 (&v)[2]) { u16 t = __builtin_nontemporal_load(reinterpret_cast<const u16*>(p)); v[0] = (u8)(t & 0xff); v[1] = (u8)(t >> 8); }
#else
RSQ_DEV void ld2(const i64* p, i64 (&v)[2]) { i64x2 t = *reinterpret_cast<const i64x2*>(p); v[0] = t.x; v[1] = t.y; }
RSQ_DEV void ld2(const i32* p, i32 (&v)[2]) { i32x2 t = *reinterpret_cast<const i32x2*>(p); v[0] = t.x; v[1] = t.y; }
RSQ_DEV void ld2(const u8* p, u8 (&v)[2]) { u8x2 t = *reinterpret_cast<const u8x2*>(p); v[0] = t.x; v[1] = t.y; }
#endif

// One 128-row tile of a W-byte string column = 8 * W chunks of 16 bytes; chunk 64 * R + lane is the lane's load number R (the lanes past the
// last chunk repeat it: an unconditional load, which the compiler's wait counting needs of every load of the tile).  Tiles start at
// multiples of 128 rows, so the chunks are 16-byte aligned wherever the column base is.
typedef u32 u32v4 __attribute__((ext_vector_type(4)));
template <int W, int R>
RSQ_DEV u32v4 ld_str_chunk(const char* tile, int lane) {
    int c = R * 64 + lane;
    if ((R + 1) * 64 > 8 * W) c = c < 8 * W ? c : 8 * W - 1;
    return __builtin_nontemporal_load(reinterpret_cast<const u32v4*>(tile) + c);
}
template <int W, int R>
RSQ_DEV void st_str_chunk(char* ldsTile, int lane, u32v4 v) {
    const int c = R * 64 + lane;
    if ((R + 1) * 64 <= 8 * W || c < 8 * W) reinterpret_cast<u32v4*>(ldsTile)[c] = v;
}

// Between the chunk stores and the word loads, and behind the loads: the compiler reasons per thread, proves that a lane's own chunk stores
// never touch the bytes it reads back (they are other lanes' chunks) and would keep the previous tile's words in registers.  The hardware
// needs nothing here - the LDS operations of one wave execute in order.
RSQ_DEV void wave_lds_order() { asm volatile("" ::: "memory"); }

// The records of one tile (128 rows x NW words, written by the lanes into the wave's LDS region) leave as the wave's 16-byte stores.
template <int NW>
RSQ_DEV void flush_tile_records(i64* ldsTile, i64* dst, int lane) {
    wave_lds_order();
    const u32v4* src = reinterpret_cast<const u32v4*>(ldsTile);
    u32v4* out = reinterpret_cast<u32v4*>(dst);
#pragma unroll
    for (int r = 0; r < NW; r++) out[r * 64 + lane] = src[r * 64 + lane];
    wave_lds_order();
}

// ---- wave64 reductions ----------------------------------------------------------------------
RSQ_DEV u64 shfl_xor_u64(u64 v, int mask) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = (u32)__shfl_xor((int)lo, mask, 64);
    hi = (u32)__shfl_xor((int)hi, mask, 64);
    return ((u64)hi << 32) | lo;
}
RSQ_DEV u64 wave_sum(u64 v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
    return v;
}
// bit d of a plain bitmap, 0 outside [0, bits)
RSQ_DEV bool bit_in(const u32* bm, u64 d, u64 bits) { return d < bits && ((bm[d >> 5] >> (d & 31)) & 1u) != 0u; }
// ... of the bitmap word that holds it, fetched earlier (bm_word)
RSQ_DEV bool bit_of_word(u32 word, u64 d, u64 bits) { return d < bits && ((word >> (d & 31)) & 1u) != 0u; }
// exclusive prefix sum over the lanes of the wave (all 64 lanes must be here)
RSQ_DEV u32 wave_excl_sum_u32(u32 v) {
    u32 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 y = (u32)__shfl_up((int)x, d, 64); if ((int)(threadIdx.x & 63) >= d) x += y; }
    return x - v;
}
RSQ_DEV u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { u64 o = shfl_xor_u64(v, m); v = o < v ? o : v; }
    return v;
}
RSQ_DEV i64 wave_min_i64(i64 v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { i64 o = (i64)shfl_xor_u64((u64)v, m); v = o < v ? o : v; }
    return v;
}
RSQ_DEV i64 wave_max_i64(i64 v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { i64 o = (i64)shfl_xor_u64((u64)v, m); v = o > v ? o : v; }
    return v;
}

// Flush of one per-thread accumulator: wave butterfly, then lane 0 of every wave merges into the
// block's LDS slot, later one lane per slot issues the global atomic (flush_block).
enum Merge { M_SUM = 0, M_MIN_U64 = 1, M_MIN_I64 = 2, M_MAX_I64 = 3 };

template <int OP>
RSQ_DEV void lds_merge(u64* slot, u64 v) {
    if (OP == M_SUM) atomicAdd(slot, v);
    else if (OP == M_MIN_U64) atomicMin(slot, v);
    else if (OP == M_MIN_I64) atomicMin(reinterpret_cast<i64*>(slot), (i64)v);
    else atomicMax(reinterpret_cast<i64*>(slot), (i64)v);
}
// Reduction of one 64-bit value per lane to LANE 63, in the VALU: DPP row shifts (an inclusive scan inside each row of 16 lanes),
// then the two row broadcasts of the GCN / CDNA lineage (lane 15 of rows 0 and 2 into rows 1 and 3, lane 31 into rows 2 and 3).
// Lanes whose source lane does not exist take the identity.  The butterfly above goes through __shfl_xor = ds_bpermute_b32, two
// per step and value: the 42 accumulators of TPC-H Q1 cost every launch 14-20 us of LDS crossbar (device timestamps,
// RSQ_DEBUG_TAIL) - a third of the kernel at SF1, 5 % at SF10.
template <int OP, int CTRL, int ROW_MASK>
RSQ_DEV u64 dpp_step(u64 v) {
    constexpr u64 idv = OP == M_SUM ? 0ull : OP == M_MIN_U64 ? ~0ull : OP == M_MIN_I64 ? 0x7fffffffffffffffull : 0x8000000000000000ull;
    const u32 lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)idv, (int)(u32)v, CTRL, ROW_MASK, 0xf, false);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)(idv >> 32), (int)(u32)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    const u64 o = ((u64)hi << 32) | lo;
    if (OP == M_SUM) return v + o;
    if (OP == M_MIN_U64) return o < v ? o : v;
    if (OP == M_MIN_I64) return (i64)o < (i64)v ? o : v;
    return (i64)o > (i64)v ? o : v;
}
template <int OP>
RSQ_DEV u64 wave_reduce_to_lane63(u64 v) {
    v = dpp_step<OP, 0x111, 0xf>(v);          // row_shr:1
    v = dpp_step<OP, 0x112, 0xf>(v);          // row_shr:2
    v = dpp_step<OP, 0x114, 0xf>(v);          // row_shr:4
    v = dpp_step<OP, 0x118, 0xf>(v);          // row_shr:8   -> lane 15 of every row holds the row
    v = dpp_step<OP, 0x142, 0xa>(v);          // row_bcast:15 into rows 1 and 3
    v = dpp_step<OP, 0x143, 0xc>(v);          // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave
    return v;
}
template <int OP>
RSQ_DEV void wave_to_lds(u64* slot, u64 v) {
#ifdef RSQ_BUTTERFLY_FLUSH
    u64 r;
    if (OP == M_SUM) r = wave_sum(v);
    else if (OP == M_MIN_U64) r = wave_min_u64(v);
    else if (OP == M_MIN_I64) r = (u64)wave_min_i64((i64)v);
    else r = (u64)wave_max_i64((i64)v);
    if ((threadIdx.x & 63) == 0) lds_merge<OP>(slot, r);
#else
    const u64 r = wave_reduce_to_lane63<OP>(v);
    if ((threadIdx.x & 63) == 63) lds_merge<OP>(slot, r);
#endif
}
// HBM atomics execute at the memory side, one 64-byte request each (MI355X_MICROARCH.md "Global float atomics"; the
// int64 forms measure the same ≈25 G requests/s chip-wide on scattered addresses), so a min / max that cannot change
// the word is filtered by a load first: a word only ever moves towards the merged value, so a stale (L2) read can
// cause a superfluous atomic but never suppress a needed one.  For the first-row tracker (rows arrive roughly in
// ascending order) this removes nearly all of its atomics.
RSQ_DEV i64 peek_i64(const u64* p) { return (i64)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int OP>
RSQ_DEV void global_merge(u64* dst, u64 v) {
    if (OP == M_SUM) { if (v != 0) atomicAdd(dst, v); }
    else if (OP == M_MIN_U64) { if (v != ~0ull) atomicMin(dst, v); }
    else if (OP == M_MIN_I64) { if ((i64)v < peek_i64(dst)) atomicMin(reinterpret_cast<i64*>(dst), (i64)v); }
    else { if ((i64)v > peek_i64(dst)) atomicMax(reinterpret_cast<i64*>(dst), (i64)v); }
}
// unconditional form (aggregates kept beside a hash-table entry: every matching row updates its entry)
template <int OP>
RSQ_DEV void global_merge_always(u64* dst, u64 v) {
    if (OP == M_SUM) atomicAdd(dst, v);
    else if (OP == M_MIN_U64) atomicMin(dst, v);
    else if (OP == M_MIN_I64) { if ((i64)v < peek_i64(dst)) atomicMin(reinterpret_cast<i64*>(dst), (i64)v); }
    else { if ((i64)v > peek_i64(dst)) atomicMax(reinterpret_cast<i64*>(dst), (i64)v); }
}
// Wave-level pre-aggregation for hash aggregation: the lanes of a wave that update the SAME table slot are folded
// into one update by their first lane.  Atomics on one word serialise at the memory side (~11 ns each), so a group-by
// with a handful of groups (TPC-H Q12: 2, Q5: 5) otherwise spends its time queueing on a few addresses.  The fold reads
// the members' values with v_readlane (uniform lane index, members are active lanes by construction), no LDS, no
// assumptions about inactive lanes.
RSQ_DEV u64 readlane_u64(u64 v, int lane) {
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}
template <int OP>
RSQ_DEV u64 subset_reduce(u64 v, u64 members) {
    u64 acc = OP == M_SUM ? 0ull : OP == M_MIN_U64 ? ~0ull : OP == M_MIN_I64 ? 0x7fffffffffffffffull : 0x8000000000000000ull;
    while (members) {
        const int m = __ffsll((long long)members) - 1;
        members &= members - 1;
        const u64 x = readlane_u64(v, m);
        if (OP == M_SUM) acc += x;
        else if (OP == M_MIN_U64) acc = x < acc ? x : acc;
        else if (OP == M_MIN_I64) acc = (i64)x < (i64)acc ? x : acc;
        else acc = (i64)x > (i64)acc ? x : acc;
    }
    return acc;
}

RSQ_DEV u64 merge_identity(int op) {
    return op == M_SUM ? 0ull : op == M_MIN_U64 ? ~0ull : op == M_MIN_I64 ? 0x7fffffffffffffffull : 0x8000000000000000ull;
}

// ---- strings: columns hold CHAR(n)/VARCHAR(n) as n bytes per row, NUL padded ------------------
struct Str { const char* p; int cap; };   // at most `cap` bytes, ends at the first NUL if any
RSQ_DEV Str str(const char* p, int cap) { Str s; s.p = p; s.cap = cap; return s; }
RSQ_DEV char str_at(const Str& s, int i) { return i < s.cap ? s.p[i] : '\0'; }

// compareChar (reference src/qlib/scalar.h:27-46): equal up to trailing spaces
RSQ_DEV u8 compare_char(const Str& a, const Str& b) {
    int i = 0;
    while (str_at(a, i) != '\0' && str_at(b, i) != '\0') { if (str_at(a, i) != str_at(b, i)) return 0; i++; }
    for (int j = i; str_at(a, j) != '\0'; j++) if (str_at(a, j) != ' ') return 0;
    for (int j = i; str_at(b, j) != '\0'; j++) if (str_at(b, j) != ' ') return 0;
    return 1;
}
// compareVarchar (reference src/qlib/scalar.h:16-24): exact
RSQ_DEV u8 compare_varchar(const Str& a, const Str& b) {
    int i = 0;
    while (str_at(a, i) != '\0' && str_at(b, i) != '\0') { if (str_at(a, i) != str_at(b, i)) return 0; i++; }
    return str_at(a, i) == str_at(b, i);
}

// Comparison with a string CONSTANT, word-wise: R (1..8) bytes at p as a little-endian word.  Column rows are not
// 8-byte aligned (CHAR(10): stride 10); gfx950 global loads take any alignment and the compiler emits one
// global_load_dwordx2 for the 8-byte form.  A byte-wise compare_char costs one dependent global_load_ubyte per character
// (TPC-H Q3's customer filter ran at 0.44 TB/s, Q19's predicates at 0.5 TB/s); the generated form is
//   ((word_k ^ CONST_k) & MASK_k) == 0 for every word,  MASK byte = 0xFF inside the constant, 0xDF behind it for CHAR
// (a byte behind the constant may be NUL padding or a trailing space, which CHAR equality ignores), 0xFF for VARCHAR.
// Relies on the column contract of resql_plan.h: values are NUL padded to their declared width.
template <int R>
RSQ_DEV u64 ld_bytes(const char* p) {
    u64 v = 0;
    if (R == 8) { __builtin_memcpy(&v, p, 8); return v; }
    int o = 0;
    if (R & 4) { u32 t; __builtin_memcpy(&t, p + o, 4); v |= (u64)t << (8 * o); o += 4; }
    if (R & 2) { u16 t; __builtin_memcpy(&t, p + o, 2); v |= (u64)t << (8 * o); o += 2; }
    if (R & 1) { v |= (u64)(u8)p[o] << (8 * o); }
    return v;
}

// Strings as hash-table KEY words: bytes [8w, 8w+8) of the string, little-endian, NUL padded, so that word-wise
// equality is compareVarchar (exact) — or compareChar when the effective length excludes trailing spaces
// (str_len_char): the reference's CHAR(n) equality ignores them (qlib/scalar.h:27-46).
// Strings as hash-table PAYLOAD are not copied at all: the word is the device address of the column bytes.
RSQ_DEV int str_len_exact(const Str& s) { int n = 0; while (n < s.cap && s.p[n] != '\0') n++; return n; }
RSQ_DEV int str_len_char(const Str& s) { int n = str_len_exact(s); while (n > 0 && s.p[n - 1] == ' ') n--; return n; }
RSQ_DEV i64 str_word(const Str& s, int n, int w) {
    u64 v = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { const int k = w * 8 + i; v |= (u64)(k < n ? (u8)s.p[k] : (u8)0) << (8 * i); }
    return (i64)v;
}
// the last character of a string held as little-endian key words = the top non-zero byte of its last non-zero word
RSQ_DEV bool top_byte_is_space(i64 w) {
    const u64 v = (u64)w;
    if (v == 0) return false;
    const int top = (63 - __clzll((long long)v)) & ~7;
    return ((v >> top) & 0xffull) == 0x20ull;
}
RSQ_DEV bool ends_with_space(const Str& s) { const int n = str_len_exact(s); return n > 0 && s.p[n - 1] == ' '; }
RSQ_DEV i64 str_addr(const Str& s) { return (i64)(u64)reinterpret_cast<unsigned long long>(s.p); }
RSQ_DEV Str str_from_addr(i64 w, int cap) { return str(reinterpret_cast<const char*>((unsigned long long)(u64)w), cap); }

// LIKE ('%' any run, '_' any one character) with the reference's results (stringLikeCheck, src/qlib/scalar.h:49-118;
// pinned by tests/golden/like_reference.json), peculiar ones included: the pattern's literal head and tail are matched
// independently and may use the same characters ('ab' LIKE 'a%ab'); the '%'-separated segments between them are
// placed leftmost-first; an empty segment ("%%") only matches a literal '%'.
struct LikeText {
    const Str& s; int n;
    RSQ_DEV char operator[](int i) const { return i >= 0 && i < n ? s.p[i] : '\0'; }
};
RSQ_DEV int str_len(const Str& s) { int n = 0; while (n < s.cap && s.p[n] != '\0') n++; return n; }
RSQ_DEV bool like_same(char c, char pat) { return pat == '_' || c == pat; }
RSQ_DEV u8 like(const Str& str, const Str& pattern) {
    const LikeText S{str, str_len(str)}, P{pattern, str_len(pattern)};
    int head = 0;                                               // literal head, as far as the string reaches
    if (P[0] != '%') {
        while (head < P.n && head < S.n && P[head] != '%') { if (!like_same(S[head], P[head])) return 0; head++; }
        if (head == P.n) return (u8)(head == S.n);              // a pattern without '%': equal or nothing
    }
    int patEnd = P.n, strEnd = S.n;                             // literal tail, compared from the ends
    if (P[P.n - 1] != '%') {
        int k = 0;
        while (k < P.n && k < S.n && P[P.n - 1 - k] != '%') { if (!like_same(S[S.n - 1 - k], P[P.n - 1 - k])) return 0; k++; }
        patEnd = P.n - 1 - k; strEnd = S.n - k;
        if (head >= patEnd) return 1;                           // nothing between head and tail
    }
    int seg = head + 1;                                         // first pattern character behind the head's '%'
    for (int at = head; at < strEnd && seg < patEnd; at++) {    // each segment at its leftmost place behind the last
        int j = 0;
        while (at + j < strEnd && like_same(S[at + j], P[seg + j])) {
            if (P[seg + j + 1] == '%') { at += j; seg += j + 2; break; }
            j++;
        }
    }
    return (u8)(seg >= patEnd);
}

// ---- hash tables in HBM ----------------------------------------------------------------------
// Open addressing, linear probing, capacity a power of two.  Slot state lives in `state`
// (0 empty, 1 being written, 2 ready); keys and payload are struct-of-arrays beside it, so a
// probe touches one 4-byte state word and one key word per step.
// Slots that are inserted AND looked up inside one launch (hash aggregation) are published with agent-scope
// accesses: a CU's vector L1 is never refreshed by another CU's stores, so plain loads of a freshly written key
// could be stale (MI355X_MICROARCH.md, inter-workgroup visibility).  Writer: key stores (sc1, write-through) ->
// s_waitcnt vmcnt(0) (the stores are acknowledged at the level every XCD sees) -> state = 2.  Reader: state load (sc1) == 2 ->
// key loads (sc1).
RSQ_DEV u32 ld_agent(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RSQ_DEV i64 ld_agent(const i64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RSQ_DEV void st_agent(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RSQ_DEV void st_agent(i64* p, i64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Count one event per calling lane with ONE atomic per wave: the active lanes vote, the lowest one adds the
// population count.  (A plain atomicAdd(p, 1) from inside a divergent probe loop is not folded by the compiler:
// 1.4 M inserts into one counter word cost 4 ms on MI355X.)
RSQ_DEV void wave_count(u32* p) {
    const u64 active = __ballot(1);
    const int lane = (int)(threadIdx.x & 63);
    if (lane == __ffsll((long long)active) - 1) atomicAdd(p, (u32)__popcll(active));
}

// Reserve one output position per calling lane with ONE returning atomic per wave: the lowest active lane adds the
// population count, every lane takes base + its rank among the active lanes.
RSQ_DEV u32 wave_reserve(u32* ctr) {
    const u64 active = __ballot(1);
    const int lane = (int)(threadIdx.x & 63);
    const int leader = __ffsll((long long)active) - 1;
    u32 base = 0;
    if (lane == leader) base = atomicAdd(ctr, (u32)__popcll(active));
    base = (u32)__shfl((int)base, leader, 64);
    return base + (u32)__popcll(active & ((1ull << lane) - 1ull));
}


// the 32-bit word of a key-domain bitmap that holds `key`'s bit (0 for keys outside the domain)
RSQ_DEV u32 bm_word(const u32* bm, i64 bmmin, u64 bmbits, i64 key) {
    const u64 d = (u64)(key - bmmin);
    return d < bmbits ? bm[d >> 5] : 0u;
}

// ---- bitmap-rank dictionary (join tables over unique integer keys of a known range) -------------------------------
// The key-domain bitmap of a join table (one bit per possible key value) doubles as the table's index: the entry of key k
// is entry number rank(k) = the number of set bits below k's bit.  The bitmap of such a table is laid out in 32-byte
// blocks of [rank word | 7 words = 224 bits]: the rank word holds the number of set bits in front of the block (written by
// the index kernels, aot_kernels.hip), so rank(k) = that word + the popcount inside k's own block — ONE aligned 32-byte
// fetch, from the block whose bit the probe has just tested.  A probe that finds its bit set HAS found its entry: no key
// compare, no walk.  (Prefix words in a separate array measured 25 % slower on TPC-H Q3's lineitem pipeline.)
#define RSQ_RANK_BLOCK_BITS 224
#define RSQ_RANK_CHUNK_BLOCKS 4096
/* entry r of a dictionary that carries aggregates keeps them at rank_mix(r): a bijection of [0, capacity), capacity = 2^k */
#define RSQ_RANK_MIX_C1 0x9E3779B97F4A7C15ull
#define RSQ_RANK_MIX_C2 0xBF58476D1CE4E5B9ull
RSQ_DEV u32 bmi_word(u64 d) { const u32 w = (u32)(d >> 5); return (w / 7u) * 8u + 1u + (w % 7u); }
RSQ_DEV u32 bmi_load(const u32* bm, i64 bmmin, u64 bmbits, i64 key) {
    const u64 d = (u64)(key - bmmin);
    return d < bmbits ? bm[bmi_word(d)] : 0u;
}
// Set bits `mask` of word `w` of a key bitmap together with the neighbouring lanes that target the same word: a segmented OR over
// the wave (runs of active lanes with equal w), and the last lane of every run issues the run's one atomic.  For a table scanned
// in key order a wave's rows fall into a few words - TPC-H orders: eight keys per 32-bit word - and memory-side atomics are what a
// build over such a table waits for.  Called under any control flow: only the lanes that are here take part.
RSQ_DEV void bm_set_combined(u32* bm, u32 w, u32 mask) {
    const u64 act = __ballot(1);
    const int ln = (int)(threadIdx.x & 63);
    const u32 wp = (u32)__shfl_up((int)w, 1, 64);
    const u32 head = (ln == 0 || !((act >> (ln - 1)) & 1ull) || wp != w) ? 1u : 0u;
    u32 m = mask, f = head;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 mo = (u32)__shfl_up((int)m, d, 64), fo = (u32)__shfl_up((int)f, d, 64);
        if (ln >= d && !f) { m |= mo; f |= fo; }      // (f == 0: the run began at or before lane ln - d, so that lane is here and in the run)
    }
    const u32 nextHead = (u32)__shfl_down((int)head, 1, 64);
    const bool last = ln == 63 || !((act >> (ln + 1)) & 1ull) || nextHead != 0u;
    if (last) atomicOr(&bm[w], m);
}
struct __attribute__((aligned(16))) u32x4 { u32 x, y, z, w; };
RSQ_DEV u64 rank_of(const u32* bm, u64 d) {
    const u32 w = (u32)(d >> 5), blk = w / 7u, wi = 1u + (w % 7u), bit = (u32)d & 31u;
    const u32x4* p = reinterpret_cast<const u32x4*>(bm + blk * 8u);
    const u32x4 lo = p[0], hi = p[1];
    const u32 v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u32 r = v[0];
#pragma unroll
    for (u32 j = 1; j < 8; j++) r += j < wi ? (u32)__popc(v[j]) : (j == wi ? (u32)__popc(v[j] & ((1u << bit) - 1u)) : 0u);
    return (u64)r;
}

// Where entry r keeps its aggregates.  Entries are in key order; rows clustered by the key update neighbouring entries, and
// atomics on one cache line serialise at the memory side like atomics on one word (measured: TPC-H Q3's lineitem pipeline
// 0.14 -> 0.19 ms with the aggregates in entry order).  A single multiplication (r * odd mod 2^k) spreads neighbours but
// measured just as slow — a fixed stride through the DRAM banks — while a hashed position is fast: so two multiply /
// xor-shift rounds on k bits, each of them invertible (odd multipliers; x ^= x >> h is its own inverse for 2h >= k), which
// the host-side compaction inverts (aot_kernels.hip).
RSQ_DEV u64 rank_mix_round(u64 v, u64 key) {      // the round function of the Feistel network below (any function will do)
    v = (v + key) * RSQ_RANK_MIX_C1; v ^= v >> 29; v *= RSQ_RANK_MIX_C2; v ^= v >> 32;
    return v;
}
RSQ_DEV u64 rank_mix(u64 r, u64 cap) {
    // three Feistel rounds over the k bits of r (halves of hi = k/2 and lo = k - k/2 bits): a bijection of [0, 2^k) whatever
    // the round function is, inverted by running the rounds backwards (aot_kernels.hip k_compact_entries)
    const int k = 63 - __builtin_clzll(cap), lo = k - (k >> 1), hi = k >> 1;
    const u64 mlo = (1ull << lo) - 1, mhi = (1ull << hi) - 1;
    u64 L = (r >> lo) & mhi, R = r & mlo;
    L ^= rank_mix_round(R, 1) & mhi;
    R ^= rank_mix_round(L, 2) & mlo;
    L ^= rank_mix_round(R, 3) & mhi;
    return (L << lo) | R;
}

RSQ_DEV u64 hash64(u64 x) {     // splitmix64 finaliser; the engine's own table layout, not the reference's
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// ---- staged partitioning (large dense aggregations: codegen.cpp emitDenseAggregation, form 3) ------------------------------
// One pass over the rows turns every passing row into a packed record in the region of its partition (partition = a
// contiguous range of groups small enough for an LDS table); a second kernel aggregates each partition in LDS.  The pass is
// HBM-bound, so records must reach HBM as whole 128-byte lines and must not be counted first:
//  * a workgroup numbers the records of each partition with an LDS ticket counter; record k of partition p belongs at
//    region(workgroup, p) + k.  It first goes to slot k % 32 of the partition's LDS ring; once per round (one tile of rows per
//    thread) the workgroup meets at a barrier and writes every complete line of 16 / RECW records with one 128-byte store.
//    A record whose slot still holds an unflushed record waits for the next flush (skewed keys; uniform keys never wait).
//  * regions are sized from a sample of the table (engine.cpp), not from a counting pass; a region that runs full raises
//    StageCtl::overflow and the engine repeats the pass with exact sizes.
//  * the row index is not part of the record.  The first-row tracker (block 0 of the table: the reference emits groups in
//    the order of their first rows) is kept with HBM atomics, filtered by a load — and only for rows below the WATERMARK:
//    once every one of the D groups has been seen, no row at or behind 1 + (largest first sighting) can be a group's
//    first row.  Workgroups report their first sightings every few rounds (largest row, then count, in this order), the
//    report that completes the count publishes the watermark, and from then on rows behind it skip the tracker.
struct StageCtl { u64 maxFirst; u64 watermark; u32 seen; u32 overflow; };        // zeroed before a launch; watermark 0: not known yet

template <int RECW, int P>
struct StageLds {
    static constexpr int S = 32;                     // ring slots per partition (two lines of 8-byte records)
    alignas(16) u64 ring[P * S * RECW];
    u64 start[P];
    u32 cap[P], tail[P], head[P];
    u64 maxFirst, wm;
    u32 newSeen, rounds;
    u32 waiting[2];
};

template <int RECW, int P>
RSQ_DEV void stage_init(StageLds<RECW, P>& L, const u64* partBase, const u32* partCap, const StageCtl* ctl) {
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        L.cap[i] = partCap[i]; L.start[i] = partBase[i] + (u64)blockIdx.x * partCap[i];
        L.tail[i] = 0u; L.head[i] = 0u;
    }
    if (threadIdx.x == 0) {
        L.newSeen = 0u; L.rounds = 0u; L.maxFirst = 0ull; L.waiting[0] = L.waiting[1] = 0u;
        const u64 w = __hip_atomic_load(&ctl->watermark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L.wm = w ? w : ~0ull;
    }
    __syncthreads();
}

// first-row tracker of one passing row in front of the watermark
template <int RECW, int P>
RSQ_DEV void stage_track(StageLds<RECW, P>& L, u64* cell, i64 row) {
    if (row < peek_i64(cell)) {
        const i64 old = atomicMin(reinterpret_cast<i64*>(cell), row);
        if (old == (i64)0x7fffffffffffffffll) { atomicAdd(&L.newSeen, 1u); atomicMax(&L.maxFirst, (u64)row); }
    }
}

template <int RECW, int P>
RSQ_DEV void stage_flush_lines(StageLds<RECW, P>& L, u64* recOut, StageCtl* ctl) {
    constexpr int S = StageLds<RECW, P>::S, LPR = 16 / RECW;
    const int j = threadIdx.x & 7;                   // 8 lanes per partition, 16 bytes each: one 128-byte line per step
    for (int p = threadIdx.x >> 3; p < P; p += blockDim.x >> 3) {
        u32 h = L.head[p];
        const u32 tl = min(L.tail[p], h + (u32)S);
        const u32 full = tl / LPR * LPR;
        if (h < full) {
            const u32 cap = L.cap[p];
            for (; h < full; h += LPR) {
                const u64x2 v = *reinterpret_cast<const u64x2*>(&L.ring[((u32)p * S + (h & (S - 1))) * RECW + j * 2]);
                if (h + LPR <= cap) *reinterpret_cast<u64x2*>(&recOut[(L.start[p] + h) * RECW + j * 2]) = v;
                else if (j == 0) atomicOr(&ctl->overflow, 1u);
            }
            if (j == 0) L.head[p] = h;
        }
    }
}

// Workgroup barrier that orders LDS only.  __syncthreads() also waits for every outstanding global load and store
// (s_waitcnt vmcnt(0)); the staging rounds keep their column loads and line stores in flight across the barrier.
RSQ_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// end of a round: the round's records into the rings, complete lines out, first sightings reported
template <int RECW, int P, int RPT>
RSQ_DEV void stage_commit(StageLds<RECW, P>& L, const u64 (&rec)[RPT * RECW], const u32 (&k)[RPT], const u32 (&pp)[RPT], u32 pending,
                          u64* recOut, StageCtl* ctl, u32 D) {
    constexpr int S = StageLds<RECW, P>::S;
    for (int it = 0;; it++) {
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            if ((pending >> i) & 1u) {
                if (k[i] - L.head[pp[i]] < (u32)S) {
                    u64* slot = &L.ring[(pp[i] * S + (k[i] & (S - 1))) * RECW];
#pragma unroll
                    for (int w = 0; w < RECW; w++) slot[w] = rec[i * RECW + w];
                    pending &= ~(1u << i);
                }
            }
        }
        // "does anyone still hold a record?" is voted through two LDS words used in turn: this iteration's word is set before
        // the first barrier and read behind it; the other word is cleared between the barriers, when nobody writes it
        if (pending) L.waiting[it & 1] = 1u;
        lds_barrier();
        stage_flush_lines<RECW, P>(L, recOut, ctl);
        const u32 any = L.waiting[it & 1];
        if (threadIdx.x == blockDim.x - 1) {
            L.waiting[(it + 1) & 1] = 0u;
            // first sightings are reported every 8th round: the two dependent atomics keep the whole workgroup at the barrier
            if (L.newSeen && (++L.rounds & 7u) == 0u) {
                // the largest first sighting must have been performed before the count that may complete D is added
                const u64 m = atomicMax(&ctl->maxFirst, L.maxFirst);
                asm volatile("" :: "v"(m));
                const u32 seen = atomicAdd(&ctl->seen, L.newSeen + (u32)(m & 0ull)) + L.newSeen;
                if (seen == D)
                    __hip_atomic_store(&ctl->watermark, __hip_atomic_load(&ctl->maxFirst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                L.newSeen = 0u; L.maxFirst = 0ull;
            }
            if (L.wm == ~0ull) {
                const u64 w = __hip_atomic_load(&ctl->watermark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (w) L.wm = w;
            }
        }
        lds_barrier();
        if (!any) break;
    }
}

// end of the pass: the incomplete last line of every partition, and the number of records per (workgroup, partition)
template <int RECW, int P>
RSQ_DEV void stage_finish(StageLds<RECW, P>& L, u64* recOut, u32* counts, StageCtl* ctl, bool countOnly) {
    constexpr int S = StageLds<RECW, P>::S;
    __syncthreads();
    if (!countOnly) {
        const int j = threadIdx.x & 7;
        for (int p = threadIdx.x >> 3; p < P; p += blockDim.x >> 3) {
            const u32 h = L.head[p], tl = L.tail[p];
            if (h < tl) {
                if (tl <= L.cap[p]) {
                    if ((u32)(j * 2 / RECW) < tl - h)
                        *reinterpret_cast<u64x2*>(&recOut[(L.start[p] + h) * RECW + j * 2]) = *reinterpret_cast<const u64x2*>(&L.ring[((u32)p * S + (h & (S - 1))) * RECW + j * 2]);
                } else if (j == 0) atomicOr(&ctl->overflow, 1u);
            }
        }
    }
    for (int i = threadIdx.x; i < P; i += blockDim.x) counts[(u64)blockIdx.x * P + i] = L.tail[i];
}

}  // namespace rsq
