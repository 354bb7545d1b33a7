// devtail.hip — the tail of a large dense aggregation, on the device.
//
// What the reference does behind its aggregation hash table (reference src/operators/aggregation.h:298-343: scan the table in
// slot order, AVG = sum * 100 / count; projection.h:62-72; materialize.h:78-220) touches every group once.  With a million
// groups that was 85 ms of host work behind 8 ms of kernels (round 2): reading 32 MB of table back, building group arrays,
// sorting them by first row, hashing, replaying the reference's table, writing a million packed tuples.  Here the per-group
// work stays on the GPU:
//   k_present_flags / k_present_scatter   the groups that occur (first-row word != +inf), compacted in group-id order
//   radixSortPairs                        those groups ordered by their first input row (LSD radix sort, 8-bit digits,
//                                         stable: wave-level digit matching + per-wave digit counts in LDS)
//   k_dense_hashes                        Values::hash of every group's key values (reference src/ValuesJitFlounder.h:65-142),
//                                         in that order — the only thing the host's replay of the reference's table needs
//   k_dense_rows                          the result relation's packed tuples (reference src/schema.h:76-106), gathered in
//                                         the emission order the host hands back
// The host keeps the one step that is a chain of data-dependent decisions: the slot order of the reference's table
// (hostref.cpp refEmissionOrderParallel, itself cut into independent probe clusters).  8 bytes per group go up, 4 come
// back, the finished tuples go up once.
#include <algorithm>
#include <utility>
#include <vector>

#include "engine.h"

namespace rsq {

typedef long long i64;
typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned char u8;

// ---- present groups, in group-id order ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_present_flags(const i64* __restrict__ first, i64 D, u32* __restrict__ flags) {
    // (flags has D + 1 entries: the scan's trailing zero slot)
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i <= D; i += (i64)gridDim.x * blockDim.x)
        flags[i] = (i < D && first[i] != 0x7fffffffffffffffll) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_present_scatter(const i64* __restrict__ first, i64 D, const u64* __restrict__ offs, u64* __restrict__ outFirst,
                                                         u32* __restrict__ outGid) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < D; i += (i64)gridDim.x * blockDim.x) {
        const i64 f = first[i];
        if (f != 0x7fffffffffffffffll) { const u64 o = offs[i]; outFirst[o] = (u64)f; outGid[o] = (u32)i; }
    }
}

// ---- LSD radix sort of (key u64, value u32) pairs, 8 bits per pass ----------------------------------------------------
// One workgroup owns a tile of RS_TILE consecutive pairs in both kernels of a pass, so the offsets the scan makes from the
// histograms ([digit][tile], digit-major) are exactly where the tile's pairs of each digit go.  Inside a tile the pairs keep
// their order (LSD needs a stable pass): the tile is taken in rounds of 256 pairs; in a round every lane finds the lanes of
// its wave that hold the same digit (eight ballots), its rank among them, and the leader of each digit leaves the wave's
// count in LDS; the counts of the waves in front give the rest.
#define RS_TILE 2048
__global__ void __launch_bounds__(256) k_rs_hist(const u64* __restrict__ keys, i64 n, int shift, u32* __restrict__ hist, u32 nTiles) {
    __shared__ u32 s_h[256];
    s_h[threadIdx.x] = 0u;
    __syncthreads();
    const i64 t0 = (i64)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_TILE / 256; r++) {
        const i64 i = t0 + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(u32)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nTiles + blockIdx.x] = s_h[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_rs_scatter(const u64* __restrict__ keysIn, const u32* __restrict__ valsIn, u64* __restrict__ keysOut,
                                                    u32* __restrict__ valsOut, i64 n, int shift, const u64* __restrict__ offs, u32 nTiles) {
    __shared__ u64 s_base[256];
    __shared__ u32 s_wcnt[4][256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    s_base[t] = offs[(size_t)t * nTiles + blockIdx.x];
    const i64 t0 = (i64)blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_TILE / 256; r++) {
#pragma unroll
        for (int w = 0; w < 4; w++) s_wcnt[w][t] = 0u;
        __syncthreads();
        const i64 i = t0 + r * 256 + t;
        const bool valid = i < n;
        const u64 k = valid ? keysIn[i] : 0ull;
        const u32 v = valid ? valsIn[i] : 0u;
        const u32 d = (u32)(k >> shift) & 255u;
        u64 same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) { const u64 m = __ballot((d >> b) & 1u); same &= ((d >> b) & 1u) ? m : ~m; }
        const u32 rankInWave = (u32)__popcll(same & ((1ull << lane) - 1ull));
        if (valid && rankInWave == 0u) s_wcnt[wave][d] = (u32)__popcll(same);
        __syncthreads();
        if (valid) {
            u32 before = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) if (w < wave) before += s_wcnt[w][d];
            const u64 pos = s_base[d] + before + rankInWave;
            keysOut[pos] = k; valsOut[pos] = v;
        }
        __syncthreads();
        s_base[t] += (u64)s_wcnt[0][t] + s_wcnt[1][t] + s_wcnt[2][t] + s_wcnt[3][t];
        // (the next round's zeroing of s_wcnt comes after this read: same thread, same column)
    }
}

size_t radixSortTempBytes(int64_t n) {
    const int64_t nTiles = (n + RS_TILE - 1) / RS_TILE;
    const int64_t cells = 256 * nTiles + 1;
    return (size_t)cells * 4 + 64 + (size_t)cells * 8 + 64 + scanTempBytes(cells);
}

// sorts n pairs by the low `keyBits` bits of the keys; the result is in (keysA, valsA) after an even number of passes, in
// (keysB, valsB) after an odd one: returns true when it is in the B buffers
bool radixSortPairs(Context& ctx, uint64_t* keysA, uint32_t* valsA, uint64_t* keysB, uint32_t* valsB, int64_t n, int keyBits, void* temp, size_t tempBytes) {
    if (n <= 1 || keyBits <= 0) return false;
    if (tempBytes < radixSortTempBytes(n)) throw Error(RSQ_ERR_DEVICE, "radixSortPairs: temporary buffer too small");
    const u32 nTiles = (u32)((n + RS_TILE - 1) / RS_TILE);
    const int64_t cells = 256 * (int64_t)nTiles + 1;
    u32* hist = (u32*)temp;
    u64* offs = (u64*)((char*)temp + (((size_t)cells * 4 + 63) & ~(size_t)63));
    void* scanTemp = (char*)offs + (((size_t)cells * 8 + 63) & ~(size_t)63);
    bool inB = false;
    RSQ_HIP(hipMemsetAsync(hist + 256 * (size_t)nTiles, 0, 4, ctx.stream));      // the scan's trailing zero slot
    for (int shift = 0; shift < keyBits; shift += 8) {
        const u64* kin = (const u64*)(inB ? keysB : keysA); const u32* vin = inB ? valsB : valsA;
        u64* kout = (u64*)(inB ? keysA : keysB); u32* vout = inB ? valsA : valsB;
        hipLaunchKernelGGL(k_rs_hist, dim3(nTiles), dim3(256), 0, ctx.stream, kin, (i64)n, shift, hist, nTiles);
        exclusiveScanCounts(ctx, hist, (uint64_t*)offs, cells, scanTemp, scanTempBytes(cells));
        hipLaunchKernelGGL(k_rs_scatter, dim3(nTiles), dim3(256), 0, ctx.stream, kin, vin, kout, vout, (i64)n, shift, (const u64*)offs, nTiles);
        inB = !inB;
    }
    RSQ_HIP(hipGetLastError());
    return inB;
}

// ---- Values::hash of dense group keys -----------------------------------------------------------------------------------
// a dense group id is sum over the keys of rank_k * stride_k (codegen.cpp groupIdExpr); the key value is min_k + rank_k, or
// the rank-th byte of the column's value set
__device__ __forceinline__ i64 dense_key_value(const DenseTailKey& k, u32 gid) {
    const i64 rank = ((i64)gid / k.stride) % k.card;
    return k.byteSet ? (i64)k.values[rank] : k.min + rank;
}
__global__ void __launch_bounds__(256) k_dense_hashes(const u32* __restrict__ gids, i64 n, DenseTailKeys keys, u64* __restrict__ hashes) {
    const u64 A = 1710227316115945415ull, B = 741332713408129251ull;
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const u32 g = gids[i];
        u64 h = 0;
        for (int k = 0; k < keys.n; k++) {
            const i64 v = dense_key_value(keys.k[k], g);
            switch (keys.k[k].typeTag) {
                case RSQ_BIGINT: case RSQ_DECIMAL: h += (u64)v * A + B; break;
                case RSQ_INT: case RSQ_DATE: h += ((u64)(i64)(int)(u32)v + B) * A; break;
                case RSQ_BOOL: if ((u8)v == 0) h += 31636373ull; break;
                default: h += (u64)(u8)v; h += h; break;                      // CHAR(1)
            }
        }
        hashes[i] = h;
    }
}

// ---- the result relation's packed tuples --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dense_rows(const u64* __restrict__ table, i64 D, const u32* __restrict__ gids, const u32* __restrict__ order,
                                                    i64 nRows, DenseTailKeys keys, DenseTailCols cols, int tupleSize, u8* __restrict__ out, u32* err) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < nRows; i += (i64)gridDim.x * blockDim.x) {
        const u32 g = gids[order ? order[i] : (u32)i];
        u8* dst = out + (size_t)i * (size_t)tupleSize;
        for (int c = 0; c < cols.n; c++) {
            const DenseTailCol& col = cols.c[c];
            i64 v;
            if (col.kind == 0) v = dense_key_value(keys.k[col.a], g);
            else if (col.kind == 1) v = (i64)table[(size_t)col.a * (size_t)D + g];
            else {
                const i64 s = (i64)((u64)table[(size_t)col.a * (size_t)D + g] * 100ull), n = (i64)table[(size_t)col.b * (size_t)D + g];
                if (n == 0 || (s == (i64)0x8000000000000000ull && n == -1)) { atomicOr(err, 1u); v = 0; } else v = s / n;
            }
            // packed tuples have no alignment: byte stores
            u8* p = dst + col.offset;
            for (int b = 0; b < col.width; b++) p[b] = (u8)((u64)v >> (8 * b));
        }
    }
}

// ---- the same tail over the group rows of a hash / join-entry aggregation -----------------------------------------------------------
__global__ void __launch_bounds__(256) k_row_first_keys(const i64* __restrict__ rows, int stride, i64 n, u64* __restrict__ keys, u32* __restrict__ idx) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) { keys[i] = (u64)rows[(size_t)i * (size_t)stride]; idx[i] = (u32)i; }
}
__device__ __forceinline__ int row_str_byte(const i64* w, int i) { return (int)(signed char)(u8)((u64)w[i >> 3] >> (8 * (i & 7))); }
// Values::hash of the group's values (reference src/ValuesJitFlounder.h:65-162; hashChar / hashVarchar src/qlib/hash.h:116-147), as hostref.cpp refHashValue
__global__ void __launch_bounds__(256) k_row_hashes(const i64* __restrict__ rows, int stride, const u32* __restrict__ idx, i64 n, RowTailKeys keys, u64* __restrict__ hashes) {
    const u64 A = 1710227316115945415ull, B = 741332713408129251ull;
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64* r = rows + (size_t)idx[i] * (size_t)stride;
        u64 h = 0;
        for (int k = 0; k < keys.n; k++) {
            const RowTailKey& key = keys.k[k];
            const i64 v = r[key.word];
            switch (key.typeTag) {
                case RSQ_BIGINT: case RSQ_DECIMAL: h += (u64)v * A + B; break;
                case RSQ_INT: case RSQ_DATE: h += ((u64)(i64)(int)(u32)v + B) * A; break;
                case RSQ_BOOL: if ((u8)v == 0) h += 31636373ull; break;
                case RSQ_CHAR:
                    if (key.len <= 1) { h += (u64)(u8)v; h += h; break; }
                    {   // hashChar: the declared length, missing characters count as ' '
                        bool ended = false;
                        for (int c = 0; c < key.len; c++) {
                            int ch = ended ? 0 : row_str_byte(r + key.word, c);
                            if (ch == 0) { ended = true; ch = ' '; }
                            const int m = (int)((u32)ch * 31636373u);
                            h = h + (u64)(i64)m + (u64)(i64)ch;
                        }
                    }
                    break;
                default:      // VARCHAR: the characters up to the first NUL
                    for (int c = 0; c < key.len; c++) {
                        const int ch = row_str_byte(r + key.word, c);
                        if (ch == 0) break;
                        const int m = (int)((u32)ch * 31636373u);
                        h = h + (u64)(i64)m + (u64)(i64)ch;
                    }
                    break;
            }
        }
        hashes[i] = h;
    }
}
__global__ void __launch_bounds__(256) k_row_result_rows(const i64* __restrict__ rows, int stride, const u32* __restrict__ idx, const u32* __restrict__ order, i64 nRows,
                                                         RowTailCols cols, int tupleSize, u8* __restrict__ out, u32* err) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < nRows; i += (i64)gridDim.x * blockDim.x) {
        const u32 at = order ? order[i] : (u32)i;
        const i64* r = rows + (size_t)(idx ? idx[at] : at) * (size_t)stride;
        u8* dst = out + (size_t)i * (size_t)tupleSize;
        for (int c = 0; c < cols.n; c++) {
            const RowTailCol& col = cols.c[c];
            u8* p = dst + col.offset;
            if (col.kind == 0 && col.len > 0) {      // a string by value, NUL terminated / padded to its place in the tuple (ValueMoves::writeString)
                bool ended = false;
                for (int b = 0; b < col.width; b++) {
                    int ch = (ended || b >= col.len) ? 0 : row_str_byte(r + col.a, b);
                    if (ch == 0) ended = true;
                    p[b] = (u8)ch;
                }
                continue;
            }
            i64 v;
            if (col.kind != 2) v = r[col.a];
            else {
                const i64 s = (i64)((u64)r[col.a] * 100ull), n = r[col.b];
                if (n == 0 || (s == (i64)0x8000000000000000ull && n == -1)) { atomicOr(err, 1u); v = 0; } else v = s / n;
            }
            for (int b = 0; b < col.width; b++) p[b] = (u8)((u64)v >> (8 * b));      // packed tuples have no alignment: byte stores
        }
    }
}
void rowTailFirstKeys(Context& ctx, const int64_t* rows, int stride, int64_t n, uint64_t* keys, uint32_t* idx) {
    if (n <= 0) return;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (n + 255) / 256));
    hipLaunchKernelGGL(k_row_first_keys, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)rows, stride, (i64)n, (u64*)keys, idx);
    RSQ_HIP(hipGetLastError());
}
void rowTailHashes(Context& ctx, const int64_t* rows, int stride, const uint32_t* idx, int64_t n, const RowTailKeys& keys, uint64_t* hashes) {
    if (n <= 0) return;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (n + 255) / 256));
    hipLaunchKernelGGL(k_row_hashes, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)rows, stride, idx, (i64)n, keys, (u64*)hashes);
    RSQ_HIP(hipGetLastError());
}
void rowTailResultRows(Context& ctx, const int64_t* rows, int stride, const uint32_t* idx, const uint32_t* order, int64_t nRows, const RowTailCols& cols,
                       int tupleSize, uint8_t* out) {
    if (nRows <= 0) return;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, (nRows + 255) / 256));
    hipLaunchKernelGGL(k_row_result_rows, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)rows, stride, idx, order, (i64)nRows, cols, tupleSize, out, ctx.dErr);
    RSQ_HIP(hipGetLastError());
}

void densePresentGroups(Context& ctx, const int64_t* firstBlock, int64_t D, uint32_t* flags, uint64_t* offs, void* scanTemp, uint64_t* outFirst, uint32_t* outGid) {
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (D + 256) / 256));
    hipLaunchKernelGGL(k_present_flags, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)firstBlock, (i64)D, flags);
    exclusiveScanCounts(ctx, flags, offs, D + 1, scanTemp, scanTempBytes(D + 1));
    hipLaunchKernelGGL(k_present_scatter, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)firstBlock, (i64)D, (const u64*)offs, (u64*)outFirst, outGid);
    RSQ_HIP(hipGetLastError());
}

void denseGroupHashes(Context& ctx, const uint32_t* gids, int64_t n, const DenseTailKeys& keys, uint64_t* hashes) {
    if (n <= 0) return;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (n + 255) / 256));
    hipLaunchKernelGGL(k_dense_hashes, dim3(grid), dim3(256), 0, ctx.stream, gids, (i64)n, keys, (u64*)hashes);
    RSQ_HIP(hipGetLastError());
}

void denseResultRows(Context& ctx, const uint64_t* table, int64_t D, const uint32_t* gids, const uint32_t* order, int64_t nRows, const DenseTailKeys& keys,
                     const DenseTailCols& cols, int tupleSize, uint8_t* out) {
    if (nRows <= 0) return;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, (nRows + 255) / 256));
    hipLaunchKernelGGL(k_dense_rows, dim3(grid), dim3(256), 0, ctx.stream, (const u64*)table, (i64)D, gids, order, (i64)nRows, keys, cols, tupleSize, out, ctx.dErr);
    RSQ_HIP(hipGetLastError());
}


// ------------------------------------------------------------------------------------------------------------------------------
// The replay of the reference's aggregation hash table ON THE DEVICE (round 3, second step): hostref.cpp's decomposition into
// independent probe clusters, one level (the table between two growths) at a time, the host only deciding the level sizes —
// which follow from the number of groups alone (qlib/hash.h:271, 387-390: grow when numInserts > 0.6 * numEntries).
//   k_rp_home      home slot of every item (hash % N, N the level's prime) and the number of items per home slot
//   scan           start[s] = items with a home in front of s  (exclusiveScanCounts)
//   k_rp_scatter   the items grouped by home slot
//   k_rp_excess    S(s) = start[s] - s over TWO laps of the table; its running minimum (k_scanmin_*) gives the carry into every
//                  slot, carry(s) = S(s) - min_{t <= s} S(t): the second lap's values are the cyclic ones (fewer items than slots)
//   k_rp_clusters  one thread per slot; the thread of a slot that starts a cluster (carry 0, items present) replays it: its items in
//                  timestamp order (repeated minimum search: clusters are a handful of items), linear probing inside the cluster
//   k_rp_next      the next level's timestamps (old entries by slot, newer groups behind them) — or the final order
// A million groups, one level: ~0.3 ms of device time against 4-6 ms on 16 host threads (and 55 ms as a sequential replay).
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 rp_mod(u64 x, u64 N, u64 magic) {
    const u64 qd = __umul64hi(x, magic);
    u64 r = x - qd * N;
    while (r >= N) r -= N;
    return r;
}
__global__ void __launch_bounds__(256) k_rp_home(const u64* __restrict__ hashes, i64 cnt, u64 N, u64 magic, u32* __restrict__ home, u32* __restrict__ count) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < cnt; i += (i64)gridDim.x * blockDim.x) {
        const u32 h = (u32)rp_mod(hashes[i], N, magic);
        home[i] = h;
        atomicAdd(&count[h], 1u);
    }
}
__global__ void __launch_bounds__(256) k_rp_scatter(const u32* __restrict__ home, i64 cnt, const u64* __restrict__ start, u32* __restrict__ count,
                                                    u32* __restrict__ items) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < cnt; i += (i64)gridDim.x * blockDim.x) {
        const u32 h = home[i];
        const u32 k = atomicSub(&count[h], 1u) - 1u;              // (leaves count[] zero again: the next level starts clean)
        items[start[h] + k] = (u32)i;
    }
}
// S over two laps: S(s) = start[s mod N] + (s >= N ? cnt : 0) - s
__global__ void __launch_bounds__(256) k_rp_excess(const u64* __restrict__ start, u64 N, i64 cnt, i64* __restrict__ S) {
    for (u64 s = blockIdx.x * (u64)blockDim.x + threadIdx.x; s < 2 * N; s += (u64)gridDim.x * blockDim.x)
        S[s] = (i64)start[s < N ? s : s - N] + (s >= N ? cnt : 0) - (i64)s;
}
// inclusive running minimum of an i64 array, three launches like the sum scan of aot_kernels.hip (chunks of 4096)
#define SM_CHUNK 4096
__global__ void __launch_bounds__(256) k_scanmin_chunks(i64* __restrict__ v, i64 n, i64* __restrict__ chunkMin) {
    __shared__ i64 s_wave[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const i64 b0 = (i64)blockIdx.x * SM_CHUNK + (i64)t * 16;
    i64 c[16];
    i64 mine = 0x7fffffffffffffffll;
#pragma unroll
    for (int j = 0; j < 16; j++) { c[j] = b0 + j < n ? v[b0 + j] : 0x7fffffffffffffffll; mine = c[j] < mine ? c[j] : mine; c[j] = mine; }      // c[j] = thread-local inclusive min
    i64 incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const i64 o = __shfl_up(incl, d, 64); if (lane >= d) incl = o < incl ? o : incl; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    i64 before = 0x7fffffffffffffffll;
    for (int w = 0; w < wave; w++) before = s_wave[w] < before ? s_wave[w] : before;
    const i64 up = __shfl_up(incl, 1, 64);
    const i64 excl = lane == 0 ? before : (up < before ? up : before);      // min of everything in front of this thread within the chunk
#pragma unroll
    for (int j = 0; j < 16; j++) if (b0 + j < n) v[b0 + j] = c[j] < excl ? c[j] : excl;
    if (t == 255) { i64 m = incl; for (int w = 0; w < 4; w++) m = s_wave[w] < m ? s_wave[w] : m; chunkMin[blockIdx.x] = m; }
}
__global__ void __launch_bounds__(1024) k_scanmin_totals(i64* __restrict__ chunkMin, i64 nChunks) {
    // exclusive running minimum of the chunk minima, in place (one workgroup; a few thousand chunks at most)
    __shared__ i64 s[1024];
    i64 carry = 0x7fffffffffffffffll;
    for (i64 base = 0; base < nChunks; base += 1024) {
        const i64 i = base + threadIdx.x;
        const i64 x = i < nChunks ? chunkMin[i] : 0x7fffffffffffffffll;
        s[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const i64 o = threadIdx.x >= (unsigned)off ? s[threadIdx.x - off] : 0x7fffffffffffffffll;
            __syncthreads();
            if (o < s[threadIdx.x]) s[threadIdx.x] = o;
            __syncthreads();
        }
        const i64 prev = threadIdx.x ? s[threadIdx.x - 1] : 0x7fffffffffffffffll;
        if (i < nChunks) chunkMin[i] = prev < carry ? prev : carry;
        const i64 last = s[1023];
        __syncthreads();
        carry = last < carry ? last : carry;
    }
}
__global__ void __launch_bounds__(256) k_scanmin_apply(i64* __restrict__ v, i64 n, const i64* __restrict__ chunkBase) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64 b = chunkBase[i / SM_CHUNK];
        if (b < v[i]) v[i] = b;
    }
}
// one thread per slot; a cluster is replayed by the thread of its first slot
__global__ void __launch_bounds__(256) k_rp_clusters(const u64* __restrict__ start, const i64* __restrict__ minS, u64 N, i64 cnt, const u32* __restrict__ items,
                                                     const u32* __restrict__ home, const u64* __restrict__ ts /* null: item index */, u32* __restrict__ who) {
    for (u64 s = blockIdx.x * (u64)blockDim.x + threadIdx.x; s < N; s += (u64)gridDim.x * blockDim.x) {
        const i64 c = (i64)(start[s + 1] - start[s]);
        const i64 carry = ((i64)start[s] + cnt - (i64)(s + N)) - minS[s + N];        // the second lap's value: cyclic
        if (carry != 0 || c == 0) continue;                        // not the first slot of a cluster (an empty slot keeps its EMPTY fill)
        // extent: the cluster ends where as many items have their home in [s, e] as there are slots
        u64 e = s, ew = s; i64 tot = 0;
        for (;;) { tot += (i64)(start[ew + 1] - start[ew]); if (tot == (i64)(e - s + 1)) break; e++; if (++ew == N) ew = 0; }
        const i64 K = tot;
        if (K == 1) { who[s] = items[start[s]]; continue; }
        // the items in timestamp order: K times the smallest timestamp above the last one taken
        u64 last = 0; bool first = true;
        for (i64 k = 0; k < K; k++) {
            u64 best = ~0ull; u32 bestItem = 0;
            u64 slot = s;
            for (u64 t = s; t <= e; t++) {
                for (u64 i = start[slot]; i < start[slot + 1]; i++) {
                    const u32 it = items[i];
                    const u64 tv = ts ? ts[it] : (u64)it;
                    if ((first || tv > last) && tv < best) { best = tv; bestItem = it; }
                }
                if (++slot == N) slot = 0;
            }
            last = best; first = false;
            u64 j = home[bestItem];
            while (who[j] != 0xffffffffu) { if (++j == N) j = 0; }
            who[j] = bestItem;
        }
    }
}
// growth: the next level's timestamps.  Old entries re-enter in slot order, the groups behind them in input order.
__global__ void __launch_bounds__(256) k_rp_next_ts(const u32* __restrict__ who, u64 N, i64 cntNow, i64 n, u64* __restrict__ ts) {
    const u64 total = N + (u64)(n - cntNow);
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        if (i < N) { const u32 w = who[i]; if (w != 0xffffffffu) ts[w] = i; }
        else { const u64 it = (u64)cntNow + (i - N); ts[it] = N + it; }
    }
}
__global__ void __launch_bounds__(256) k_rp_flags(const u32* __restrict__ who, u64 N, u32* __restrict__ flags) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i <= N; i += (u64)gridDim.x * blockDim.x) flags[i] = (i < N && who[i] != 0xffffffffu) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_rp_order(const u32* __restrict__ who, u64 N, const u64* __restrict__ offs, u32* __restrict__ order) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) { const u32 w = who[i]; if (w != 0xffffffffu) order[offs[i]] = w; }
}

static const uint64_t kReplayPrimes[] = {
    5ull, 11ull, 23ull, 47ull, 97ull, 199ull, 409ull, 823ull, 1741ull, 3469ull, 6949ull, 14033ull, 28411ull, 57557ull, 116731ull, 236897ull, 480881ull, 976369ull,
    1982627ull, 4026031ull, 8175383ull, 16601593ull, 33712729ull, 68460391ull, 139022417ull, 282312799ull, 573292817ull, 1164186217ull, 2364114217ull, 4294967291ull};
static uint64_t replayPrimeAbove(uint64_t minSize) {       // the reference's prime table up to 2^32 (qlib/hash.h:32-95, upper_bound); 0: beyond it
    if (minSize < 2) minSize = 2;
    for (uint64_t p : kReplayPrimes) if (minSize < p) return p;
    return 0;
}

// the level sizes; false when the device path does not take the case (tables beyond 2^31 slots, the counter corner of hostref.cpp)
bool replayLevels(uint64_t n, uint64_t minSize, std::vector<std::pair<uint64_t, uint64_t>>& levels /* (N, items) */) {
    levels.clear();
    uint64_t N = replayPrimeAbove(minSize), c = 0, live = 0;
    for (;;) {
        if (N == 0 || N >= (1ull << 31)) return false;
        const uint64_t th = N * 6 / 10;
        if (th < c) return false;
        const uint64_t atGrowth = live + (th - c);
        levels.push_back({N, std::min<uint64_t>(n, atGrowth)});
        if (n <= atGrowth) return true;
        c = atGrowth; live = atGrowth + 1;
        N = replayPrimeAbove(N + 1);
    }
}
size_t replayDeviceBytes(uint64_t n, uint64_t nMax) {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return up(n * 4) + up(n * 4) + up(n * 8) + up((nMax + 1) * 4) + up((nMax + 1) * 8) + up(2 * nMax * 8) + up(nMax * 4) + up(scanTempBytes((int64_t)nMax + 1)) +
           up(((2 * nMax + SM_CHUNK - 1) / SM_CHUNK + 1) * 8);
}

// order[k] = index (into hashes) of the group in the k-th occupied slot of the reference's table; everything on ctx.stream, no
// synchronisation.  `work` = replayDeviceBytes(n, N of the last level) bytes of device memory.
void replayEmissionOrderDevice(Context& ctx, const uint64_t* hashes, uint64_t n, const std::vector<std::pair<uint64_t, uint64_t>>& levels, void* work, uint32_t* order) {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const uint64_t nMax = levels.back().first;
    char* p = (char*)work;
    u32* home = (u32*)p; p += up(n * 4);
    u32* items = (u32*)p; p += up(n * 4);
    u64* ts = (u64*)p; p += up(n * 8);
    u32* count = (u32*)p; p += up((nMax + 1) * 4);
    u64* start = (u64*)p; p += up((nMax + 1) * 8);
    i64* S = (i64*)p; p += up(2 * nMax * 8);
    u32* who = (u32*)p; p += up(nMax * 4);
    void* scanTemp = p; p += up(scanTempBytes((int64_t)nMax + 1));
    i64* chunkMin = (i64*)p;
    RSQ_HIP(hipMemsetAsync(count, 0, (size_t)(nMax + 1) * 4, ctx.stream));
    auto grid = [](uint64_t work) { return dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(4096, (work + 255) / 256))); };
    for (size_t L = 0; L < levels.size(); L++) {
        const u64 N = levels[L].first; const i64 cnt = (i64)levels[L].second;
        const u64 magic = (u64)((((__uint128_t)1) << 64) / N);
        hipLaunchKernelGGL(k_rp_home, grid((u64)cnt), dim3(256), 0, ctx.stream, (const u64*)hashes, cnt, N, magic, home, count);
        exclusiveScanCounts(ctx, count, (uint64_t*)start, (int64_t)N + 1, scanTemp, scanTempBytes((int64_t)N + 1));       // (count[N] is 0: the trailing slot)
        hipLaunchKernelGGL(k_rp_scatter, grid((u64)cnt), dim3(256), 0, ctx.stream, (const u32*)home, cnt, (const u64*)start, count, items);
        hipLaunchKernelGGL(k_rp_excess, grid(2 * N), dim3(256), 0, ctx.stream, (const u64*)start, N, cnt, S);
        const i64 n2 = (i64)(2 * N), nChunks = (n2 + SM_CHUNK - 1) / SM_CHUNK;
        hipLaunchKernelGGL(k_scanmin_chunks, dim3((unsigned)nChunks), dim3(256), 0, ctx.stream, S, n2, chunkMin);
        hipLaunchKernelGGL(k_scanmin_totals, dim3(1), dim3(1024), 0, ctx.stream, chunkMin, nChunks);
        hipLaunchKernelGGL(k_scanmin_apply, grid((u64)n2), dim3(256), 0, ctx.stream, S, n2, (const i64*)chunkMin);
        RSQ_HIP(hipMemsetAsync(who, 0xff, (size_t)N * 4, ctx.stream));
        hipLaunchKernelGGL(k_rp_clusters, grid(N), dim3(256), 0, ctx.stream, (const u64*)start, (const i64*)S, N, cnt, (const u32*)items, (const u32*)home,
                           L == 0 ? (const u64*)nullptr : (const u64*)ts, who);
        if (L + 1 < levels.size())
            hipLaunchKernelGGL(k_rp_next_ts, grid(N + (n - (u64)cnt)), dim3(256), 0, ctx.stream, (const u32*)who, N, cnt, (i64)n, ts);
        else {
            // final table: the items in slot order (count[] doubles as the flag array: it is zero again behind the scatter)
            hipLaunchKernelGGL(k_rp_flags, grid(N + 1), dim3(256), 0, ctx.stream, (const u32*)who, N, count);
            exclusiveScanCounts(ctx, count, (uint64_t*)start, (int64_t)N + 1, scanTemp, scanTempBytes((int64_t)N + 1));
            hipLaunchKernelGGL(k_rp_order, grid(N), dim3(256), 0, ctx.stream, (const u32*)who, N, (const u64*)start, order);
            RSQ_HIP(hipMemsetAsync(count, 0, (size_t)(N + 1) * 4, ctx.stream));
        }
    }
    RSQ_HIP(hipGetLastError());
}

}  // namespace rsq
