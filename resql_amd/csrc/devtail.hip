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

}  // namespace rsq
