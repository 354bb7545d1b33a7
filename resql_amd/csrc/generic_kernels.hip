// generic_kernels.hip — the pre-compiled interpreter for WHOLE pipelines: scan -> selections / projections / join probes ->
// { join build | dense aggregation | aggregation at a join entry | hash aggregation | materialisation }.
//
// ReSQL is "a query compilation-based database system with low compilation times": its Flounder back end makes machine code of
// a plan in 0.6-3 ms (reference README:1-6, src/JitContextFlounder.h:410-456).  hiprtc needs 0.25-0.55 s per pipeline shape it
// has not seen.  A plan whose specialised kernels are not in the code-object cache therefore starts on THIS kernel — one AOT
// kernel that interprets a register program per row (generic2.cpp builds it from the same operator tree, types and table layouts
// codegen.cpp uses) — while hiprtc builds the specialised kernels on a host thread.  Round 2 had the interpreter for
// scan -> selection -> dense aggregation only (aot_kernels.hip k_generic_aggregate, still the path for that shape); this one adds
// joins (build, single- and all-matches probes), strings (compared where they lie, carried by address), hash aggregation,
// aggregation at a join entry and materialisation.
//
// Semantics are the ones ExprGen emits as C++ for the specialised kernels: 64-bit wrap-around arithmetic, truncating division with
// the /0 flag, byte-wise AND / OR without short circuit, CASE as nested selects, CHAR equality up to trailing spaces
// (rsq::compare_char), LIKE (rsq::like).  Tables are open addressing / linear probing in HBM: state[cap] (0 empty, 1 being
// written, 2 ready), words[cap][nWords] (a slot's key words, then its payload words), acc[block][cap].  Throughput is not the
// point (one row per thread, registers in LDS, a switch per instruction): a cold TPC-H Q3 at SF1 runs in a few milliseconds and
// the second execution usually finds the specialised kernels ready.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "engine_internal.h"
#undef RSQ_RANK_CHUNK_BLOCKS          /* (the device header has its own, unused here) */
#include "kernels/rsq_device.h"

namespace rsq {

struct GenericTableDev { u32* state; i64* words; u64* acc; u64 cap; u32* count; int nWords; int pad; };

struct GenericPipelineArgs {
    const void* col[G2_MAX_COLS]; int colWidth[G2_MAX_COLS];
    const GenericInstr* code; int nInstr;
    const GenericProbeDesc* probes;
    const char* constPool;
    GenericTableDev tab[G2_MAX_TABLES];
    GenericSinkDesc sink;
    i64 nRows, row0;
    u32* err;
    // materialisation
    u32* matCnt; const u64* matOffs; u64 matLimit; void* matOut[G2_MAX_OUT]; int matPass;
    // dense aggregation
    u64* dense; i64 denseGroups;
};

__device__ __forceinline__ u64 g2_hash(const i64* k, int n) {
    u64 h = hash64((u64)k[0]);
    for (int i = 1; i < n; i++) h = hash64(h ^ ((u64)k[i] * 0x9E3779B97F4A7C15ull));
    return h;
}
__device__ __forceinline__ void g2_merge(u64* p, u64 v, int merge) {
    if (merge == 0) atomicAdd(p, v);
    else if (merge == 2) { if ((i64)v < peek_i64(p)) atomicMin(reinterpret_cast<i64*>(p), (i64)v); }
    else { if ((i64)v > peek_i64(p)) atomicMax(reinterpret_cast<i64*>(p), (i64)v); }
}

// n (1..8) bytes at p as a little-endian word: one load for 8 (global loads take any alignment), the pieces of ld_bytes otherwise
__device__ __forceinline__ u64 g2_ld_bytes(const char* p, int n) {
    u64 v = 0;
    if (n >= 8) { __builtin_memcpy(&v, p, 8); return v; }
    int o = 0;
    if (n & 4) { u32 t; __builtin_memcpy(&t, p, 4); v = t; o = 4; }
    if (n & 2) { u16 t; __builtin_memcpy(&t, p + o, 2); v |= (u64)t << (8 * o); o += 2; }
    if (n & 1) v |= (u64)(u8)p[o] << (8 * o);
    return v;
}
// compareChar / compareVarchar (rsq_device.h compare_char, compare_varchar; reference src/qlib/scalar.h:16-46) with the strings read a word at a
// time: the byte-wise forms cost one dependent global load per character and side - TPC-H Q19's three string predicates over 60 M rows were
// most of the interpreter's 74 ms.  Same walk, byte for byte: while both run they must agree; once one has ended (NUL, or its capacity) the
// other may only hold spaces (CHAR) or must have ended too (VARCHAR).
__device__ __noinline__ u8 g2_compare_str(const char* pa, int capA, const char* pb, int capB, bool charSemantics) {
    {   // most comparisons end at the first character: one byte of each side decides them
        const u8 a0 = capA > 0 ? (u8)pa[0] : (u8)0, b0 = capB > 0 ? (u8)pb[0] : (u8)0;
        if (a0 != 0 && b0 != 0 && a0 != b0) return 0;
    }
    int state = 0;                              // 0: both running, 1: only a is left, 2: only b is left
    const int n = capA > capB ? capA : capB;
    for (int w0 = 0; w0 < n; w0 += 8) {
        const u64 wa = w0 < capA ? g2_ld_bytes(pa + w0, capA - w0) : 0ull;
        const u64 wb = w0 < capB ? g2_ld_bytes(pb + w0, capB - w0) : 0ull;
        if (state == 0 && wa == wb && !((wa - 0x0101010101010101ull) & ~wa & 0x8080808080808080ull)) continue;      // eight equal bytes, none of them NUL
#pragma unroll 1
        for (int i = 0; i < 8; i++) {
            const u8 ca = (u8)(wa >> (8 * i)), cb = (u8)(wb >> (8 * i));
            if (state == 0) {
                if (ca != 0 && cb != 0) { if (ca != cb) return 0; continue; }
                if (!charSemantics) return ca == cb;
                if (ca == 0 && cb == 0) return 1;
                state = ca != 0 ? 1 : 2;
            }
            if (state == 1) { if (ca == 0) return 1; if (ca != ' ') return 0; }
            else { if (cb == 0) return 1; if (cb != ' ') return 0; }
        }
    }
    return 1;                                   // (both ran to their capacities)
}

#define G2_REG(r) reg[(int)(r) * 256]

__global__ void __launch_bounds__(256) k_generic_pipeline(GenericPipelineArgs a) {
    // the register file: as many registers as the program uses (launchGenericPipeline sizes it) - with all 40 a workgroup held 80 KB of
    // LDS and a CU two workgroups: eight waves for chains of dependent loads
    extern __shared__ __attribute__((aligned(16))) i64 s_reg[];
    __shared__ GenericInstr s_code[256];       // the program, where every wave reads it with one broadcast
    for (int i = threadIdx.x; i < a.nInstr && i < 256; i += 256) s_code[i] = a.code[i];
    const GenericSinkDesc& S = a.sink;
    // a small dense aggregate table is kept per workgroup in LDS and merged into the device table once, at the end: rows that all want the
    // same few words (TPC-H Q14: ONE group, 720 K rows that pass) otherwise queue up at the memory side, 11-13 ns per atomic on one address
    __shared__ u64 s_dense[1024];
    const bool ldsDense = S.kind == G2_SINK_DENSE && a.denseGroups * (i64)S.nAccs <= 1024;
    if (ldsDense)
        for (int i = threadIdx.x; i < (int)(a.denseGroups * S.nAccs); i += 256) {
            const int m = S.accMerge[i / (int)a.denseGroups];
            s_dense[i] = m == 0 ? 0ull : m == 2 ? 0x7fffffffffffffffull : 0x8000000000000000ull;
        }
    __syncthreads();
    i64* reg = s_reg + threadIdx.x;            // register r of this thread: reg[r * 256] (conflict-free across the lanes)
    u32 created = 0;                           // groups / entries this thread created (one atomic per thread at the end)
    for (i64 base = (i64)blockIdx.x * 256; base < a.nRows; base += (i64)gridDim.x * 256) {
        const i64 r = base + threadIdx.x;
        if (r >= a.nRows) continue;
        const i64 row = a.row0 + r;
        u32 emitted = 0;                       // tuples this row has materialised so far
        // backtracking over the probes for all matches: the continuation of a match is the rest of the program
        int sp = 0;
        u8 stPc[G2_MAX_DEPTH]; u64 stSlot[G2_MAX_DEPTH]; u64 stSteps[G2_MAX_DEPTH];
        int pc = 0;
        bool resume = false;                   // re-enter the probe at pc with (slot, steps) from the stack
        u64 rsSlot = 0, rsSteps = 0;
        for (;;) {
            bool fail = false;
            // one instruction.  The lanes of a wave are nearly always AT THE SAME instruction (a row that fails a filter leaves the loop; only the
            // continuations of probes for all matches set lanes apart): the instruction is then fetched once for the wave and the switch over
            // its operation is a scalar branch - with a per-lane instruction word every case of the switch was tested under an execution mask.
            auto stepInstr = [&](const GenericInstr in) {
                if (in.op == G_PROBE) {
                    const GenericProbeDesc& P = a.probes[in.c];
                    const GenericTableDev& T = a.tab[P.table];
                    i64 k[G2_MAX_KEYW];
                    for (int i = 0; i < P.nKeys; i++) k[i] = G2_REG(P.keyReg[i]);
                    const u64 mask = T.cap - 1;
                    u64 s = resume ? rsSlot : (g2_hash(k, P.nKeys) & mask), steps = resume ? rsSteps : 0;
                    resume = false;
                    bool hit = false;
                    for (; steps <= mask; steps++, s = (s + 1) & mask) {
                        if (T.state[s] == 0u) break;
                        bool eq = true;
                        for (int i = 0; i < P.nKeys; i++) eq = eq && T.words[s * T.nWords + i] == k[i];
                        if (eq) { hit = true; break; }
                    }
                    if (!hit) fail = true;
                    else {
                        for (int i = 0; i < P.nPayload; i++) G2_REG(P.payloadReg[i]) = T.words[s * T.nWords + P.nKeys + i];
                        if (P.slotReg >= 0) G2_REG(P.slotReg) = (i64)s;
                        if (!P.single && steps < mask) {
                            if (sp >= G2_MAX_DEPTH) { atomicOr(a.err, (u32)ERR_STUCK); }
                            else { stPc[sp] = (u8)pc; stSlot[sp] = (s + 1) & mask; stSteps[sp] = steps + 1; sp++; }
                        }
                        pc++;
                    }
                } else {
                    const i64 x = G2_REG(in.a), y = G2_REG(in.b);
                    i64 v = 0;
                    switch (in.op) {
                        case G_COL: {
                            const int w = a.colWidth[in.a];
                            if (w == 8) v = reinterpret_cast<const i64*>(a.col[in.a])[r];
                            else if (w == 4) v = (i64)reinterpret_cast<const int*>(a.col[in.a])[r];
                            else v = (i64)reinterpret_cast<const unsigned char*>(a.col[in.a])[r];
                            break;
                        }
                        case G_COLADDR: v = (i64)(u64)(reinterpret_cast<unsigned long long>(a.col[in.a]) + (u64)r * (u64)a.colWidth[in.a]); break;
                        case G_CONST: v = in.imm; break;
                        case G_CONSTADDR: v = (i64)(u64)(reinterpret_cast<unsigned long long>(a.constPool) + (u64)in.imm); break;
                        case G_MOV: v = x; break;
                        case G_ADD: v = (i64)((u64)x + (u64)y); break;
                        case G_SUB: v = (i64)((u64)x - (u64)y); break;
                        case G_MUL: v = (i64)((u64)x * (u64)y); break;
                        case G_DIV:
                            if (y == 0 || (x == (i64)0x8000000000000000ull && y == -1)) { atomicOr(a.err, 1u); v = 0; }
                            else v = x / y;
                            break;
                        case G_LT: v = x < y; break;
                        case G_LE: v = x <= y; break;
                        case G_GT: v = x > y; break;
                        case G_GE: v = x >= y; break;
                        case G_EQ: v = x == y; break;
                        case G_NE: v = x != y; break;
                        case G_AND: v = (x & y) & 0xff; break;
                        case G_OR: v = (x | y) & 0xff; break;
                        case G_MULI: v = (i64)((u64)x * (u64)in.imm); break;
                        case G_DIVI: v = x / in.imm; break;
                        case G_CAST16: v = (i64)(short)x; break;
                        case G_SELECT: v = x ? y : G2_REG(in.c); break;
                        case G_STREQ: {      // imm: capA | capB << 16 | charSemantics << 32 | negate << 33
                            const u8 e = g2_compare_str(reinterpret_cast<const char*>((unsigned long long)(u64)x), (int)(in.imm & 0xffff),
                                                        reinterpret_cast<const char*>((unsigned long long)(u64)y), (int)((in.imm >> 16) & 0xffff), ((in.imm >> 32) & 1) != 0);
                            v = ((in.imm >> 33) & 1) ? 1 - (i64)e : (i64)e;
                            break;
                        }
                        case G_LIKE: {
                            const Str sa = str(reinterpret_cast<const char*>((unsigned long long)(u64)x), (int)(in.imm & 0xffff));
                            const Str sb = str(reinterpret_cast<const char*>((unsigned long long)(u64)y), (int)((in.imm >> 16) & 0xffff));
                            v = (i64)like(sa, sb);
                            break;
                        }
                        case G_STRWORD: {    // bytes [8w, 8w + n) of the string at x as a little-endian word; imm = offset | n << 16
                            const char* p = reinterpret_cast<const char*>((unsigned long long)(u64)x) + (in.imm & 0xffff);
                            const int n = (int)((in.imm >> 16) & 0xff);
                            v = n > 0 ? (i64)g2_ld_bytes(p, n) : 0;
                            break;
                        }
                        case G_FILTER: if (x == 0) fail = true; break;
                        default: break;
                    }
                    if (in.op != G_FILTER) G2_REG(in.dst) = v;
                    if (!fail) pc++;
                }
            };
            if (pc < a.nInstr) {
                const int pcU = __builtin_amdgcn_readfirstlane(pc);
                if (__ballot(pc != pcU) == 0ull) stepInstr(s_code[pcU]);
                else stepInstr(s_code[pc]);
            } else {
                // ---- the sink ----
                switch (S.kind) {
                    case G2_SINK_BUILD: {
                        const GenericTableDev& T = a.tab[S.table];
                        i64 k[G2_MAX_KEYW];
                        for (int i = 0; i < S.nKeys; i++) k[i] = G2_REG(S.keyReg[i]);
                        const u64 mask = T.cap - 1;
                        u64 s = g2_hash(k, S.nKeys) & mask;
                        bool done = false;
                        for (u64 n = 0; n <= mask && !done; n++, s = (s + 1) & mask) {
                            if (T.state[s] == 0u && atomicCAS(&T.state[s], 0u, 1u) == 0u) {
                                for (int i = 0; i < S.nKeys; i++) T.words[s * T.nWords + i] = k[i];
                                for (int i = 0; i < S.nPayload; i++) T.words[s * T.nWords + S.nKeys + i] = G2_REG(S.payloadReg[i]);
                                created++; done = true;
                            }
                        }
                        if (!done) atomicOr(a.err, (u32)ERR_HT_FULL);
                        break;
                    }
                    case G2_SINK_DENSE: {
                        i64 gid = 0;
                        for (int kk = 0; kk < S.nKeys; kk++) {
                            const i64 v = G2_REG(S.keyReg[kk]);
                            i64 rank = 0; bool ok;
                            if (S.keyByteSet[kk]) {
                                ok = false;
                                for (int d = 0; d < S.keyNValues[kk]; d++) { if (d && (u8)v >= S.keyValues[kk][d]) rank++; ok = ok || (u8)v == S.keyValues[kk][d]; }
                            } else { rank = v - S.keyMin[kk]; ok = (u64)rank < (u64)S.keyCard[kk]; }
                            if (!ok) { atomicOr(a.err, 8u); rank = 0; }
                            gid += rank * S.keyStride[kk];
                        }
                        for (int w = 0; w < S.nAccs; w++) {
                            const int rg = S.accReg[w];
                            const u64 v = rg == -1 ? (u64)row : rg == -2 ? 1ull : (u64)G2_REG(rg);
                            if (ldsDense) {
                                u64* p = s_dense + (i64)w * a.denseGroups + gid;
                                if (S.accMerge[w] == 0) { if (v) atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); }
                                else if (S.accMerge[w] == 2) atomicMin(reinterpret_cast<long long*>(p), (long long)v);
                                else atomicMax(reinterpret_cast<long long*>(p), (long long)v);
                                continue;
                            }
                            u64* p = a.dense + (i64)S.accBlock[w] * a.denseGroups + gid;
                            if (S.accMerge[w] == 0) { if (v) atomicAdd(p, v); } else g2_merge(p, v, S.accMerge[w]);
                        }
                        break;
                    }
                    case G2_SINK_ENTRY: {
                        const GenericTableDev& T = a.tab[S.table];
                        const u64 s = (u64)G2_REG(S.slotReg);
                        for (int w = 0; w < S.nAccs; w++) {
                            const int rg = S.accReg[w];
                            const u64 v = rg == -1 ? (u64)row : rg == -2 ? 1ull : (u64)G2_REG(rg);
                            g2_merge(T.acc + (u64)S.accBlock[w] * T.cap + s, v, S.accMerge[w]);
                        }
                        break;
                    }
                    case G2_SINK_HASH: {
                        const GenericTableDev& T = a.tab[S.table];
                        const int NW = T.nWords;          // compared key words, then carried words: ALL are compared here (the
                        i64 k[G2_MAX_KEYW];               // dependencies codegen.cpp relies on need a rank dictionary, which this path has not)
                        for (int i = 0; i < NW; i++) {
                            const i64 v = G2_REG(S.keyReg[i]);
                            if (S.wordStr[i]) {      // a string group value: its bytes where they lie (NUL padded to the column's width)
                                const char* sp8 = reinterpret_cast<const char*>((unsigned long long)(u64)v) + S.wordOff[i];
                                k[i] = S.wordN[i] ? (i64)g2_ld_bytes(sp8, (int)S.wordN[i]) : 0;
                            } else k[i] = v;
                        }
                        const u64 mask = T.cap - 1;
                        u64 s = g2_hash(k, S.nKeys) & mask;
                        u64 adv = 0; u32 spin = 0; bool found = false;
                        if (!(ld_agent(a.err) & (u32)ERR_HT_FULL)) {
                            for (;;) {
                                u32 stt = ld_agent(&T.state[s]);
                                if (stt == 0u) {
                                    if (atomicCAS(&T.state[s], 0u, 1u) == 0u) {
                                        for (int i = 0; i < NW; i++) st_agent(&T.words[s * NW + i], k[i]);
                                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                        st_agent(&T.state[s], 2u);
                                        created++;
                                        for (int c = 0; c < S.nCharKeys; c++) {
                                            i64 last = k[S.charFirst[c]];
                                            for (int w = S.charFirst[c] + 1; w <= S.charLast[c]; w++) if (k[w] != 0) last = k[w];
                                            if (top_byte_is_space(last)) atomicOr(a.err, (u32)NOTE_CHAR_GROUP_ENDS_WITH_SPACE);
                                        }
                                    }
                                    stt = ld_agent(&T.state[s]);
                                }
                                if (stt == 2u) {
                                    bool eq = true;
                                    for (int i = 0; i < NW; i++) eq = eq && ld_agent(&T.words[s * NW + i]) == k[i];
                                    if (eq) { found = true; break; }
                                    s = (s + 1) & mask;
                                    if (++adv > (mask < 4096 ? mask : 4096)) { atomicOr(a.err, (u32)ERR_HT_FULL); break; }
                                } else if (++spin > (1u << 22)) { atomicOr(a.err, (u32)ERR_STUCK); break; }
                            }
                        }
                        if (found)
                            for (int w = 0; w < S.nAccs; w++) {
                                const int rg = S.accReg[w];
                                const u64 v = rg == -1 ? (u64)row : rg == -2 ? 1ull : (u64)G2_REG(rg);
                                g2_merge(T.acc + (u64)S.accBlock[w] * T.cap + s, v, S.accMerge[w]);
                            }
                        break;
                    }
                    case G2_SINK_MATERIALIZE: {
                        if (a.matPass == 1) { emitted++; break; }
                        const u64 pos = a.matOffs[r] + emitted;
                        emitted++;
                        if (pos >= a.matLimit) break;
                        for (int c = 0; c < S.nOut; c++) {
                            const i64 v = G2_REG(S.outReg[c]);
                            const int w = S.outWidth[c];
                            if (S.outString[c]) {
                                const char* src = reinterpret_cast<const char*>((unsigned long long)(u64)v);
                                char* dst = reinterpret_cast<char*>(a.matOut[c]) + pos * (u64)w;
                                for (int i = 0; i < w; i++) dst[i] = i < S.outSrcCap[c] ? src[i] : '\0';
                            } else if (w == 8) reinterpret_cast<i64*>(a.matOut[c])[pos] = v;
                            else if (w == 4) reinterpret_cast<int*>(a.matOut[c])[pos] = (int)v;
                            else reinterpret_cast<unsigned char*>(a.matOut[c])[pos] = (unsigned char)v;
                        }
                        break;
                    }
                    default: break;
                }
                fail = true;                   // the row's continuation ends here: on to the next match, if any
            }
            if (fail) {
                if (sp == 0) break;
                sp--;
                pc = stPc[sp]; rsSlot = stSlot[sp]; rsSteps = stSteps[sp]; resume = true;
            }
        }
        if (S.kind == G2_SINK_MATERIALIZE && a.matPass == 1) a.matCnt[r] = emitted;
    }
    if (created && (S.kind == G2_SINK_BUILD || S.kind == G2_SINK_HASH)) atomicAdd(a.tab[S.table].count, created);
    if (ldsDense) {
        __syncthreads();
        for (int i = threadIdx.x; i < (int)(a.denseGroups * S.nAccs); i += 256) {
            const int w = i / (int)a.denseGroups, m = S.accMerge[w];
            const u64 v = s_dense[i];
            if (v == (m == 0 ? 0ull : m == 2 ? 0x7fffffffffffffffull : 0x8000000000000000ull)) continue;      // (nothing arrived)
            u64* p = a.dense + (i64)S.accBlock[w] * a.denseGroups + (i - w * (int)a.denseGroups);
            if (m == 0) atomicAdd(p, v); else g2_merge(p, v, m);
        }
    }
}

void launchGenericPipeline(Context& ctx, const GenericPipelineLaunch& L) {
    GenericPipelineArgs a;
    memset(&a, 0, sizeof a);
    const GenericProgram2& p = *L.prog;
    for (size_t i = 0; i < p.cols.size(); i++) { a.col[i] = p.cols[i].ptr; a.colWidth[i] = p.cols[i].width; }
    a.code = L.dCode; a.nInstr = (int)p.code.size();
    a.probes = L.dProbes; a.constPool = L.dConstPool;
    for (int t = 0; t < G2_MAX_TABLES; t++) {
        a.tab[t].state = L.tables[t].state; a.tab[t].words = (i64*)L.tables[t].words; a.tab[t].acc = (u64*)L.tables[t].acc;
        a.tab[t].cap = L.tables[t].cap; a.tab[t].count = L.tables[t].count; a.tab[t].nWords = L.tables[t].nWords;
    }
    a.sink = p.sink;
    a.nRows = L.nRows; a.row0 = L.row0; a.err = (u32*)ctx.dErr;
    a.matCnt = L.matCnt; a.matOffs = (const u64*)L.matOffs; a.matLimit = L.matLimit; a.matPass = L.matPass;
    for (int c = 0; c < G2_MAX_OUT; c++) a.matOut[c] = L.matOut[c];
    a.dense = (u64*)L.dense; a.denseGroups = L.denseGroups;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8 * (int64_t)ctx.numCUs, (L.nRows + 255) / 256));
    // registers the program names: the builder hands out the lowest free one and keeps the high-water mark (generic2.cpp Builder2::alloc)
    const size_t regBytes = (size_t)std::min<int>(G2_REGS, std::max<int>(p.nRegs, 2)) * 256 * 8;
    if (regBytes > (48u << 10)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_generic_pipeline), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(G2_REGS * 256 * 8));
    hipLaunchKernelGGL(k_generic_pipeline, dim3(grid), dim3(256), regBytes, ctx.stream, a);
    RSQ_HIP(hipGetLastError());
}

}  // namespace rsq
