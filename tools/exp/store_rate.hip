// What do 5 GB of stores cost NEXT TO a 40 GB streaming read on this GPU?  (VERDICT r04 "next round" 6: the staged partitioning of
// BASELINE config 5 at 50 % selectivity reads 40 GB, writes 5 GB of 8-byte records as whole 128-byte lines and takes 8.3 ms where the
// same kernel without its stores takes 6.1 ms - DESIGN.md §4.  Is that the price of ANY 5 GB of stores beside such a read?)
// Stand-alone, not part of the product.   build: hipcc -O3 --offload-arch=gfx950 -o store_rate store_rate.hip ; run: ./store_rate [rows]
//   read   : every wave streams its 128-row tiles of four int64 columns (16 B per lane and column, non-temporal), sums them
//   plain  : the same, and every wave writes one 16-byte word per lane for every K-th tile to ITS OWN contiguous output stream
//            (perfectly coalesced 1 KiB stores, sequential per wave: the friendliest store pattern there is), 1/8 of the bytes read
//   regions: whole 128-byte lines, every line to another of the wave's 256 region streams (the staged form's shape: many open write streams,
//            one line at a time).  NOT 5 GB of unique lines: a region's position advances once per 32 stores of the wave whatever region they
//            went to, so lines are rewritten while they sit in the L2 and far fewer bytes leave the chip - this line shows what write
//            COMBINING buys, the four forms below it (every line written exactly once) what unique lines cost
// Prints ms and GB/s per form, and what the stores cost per GB beside the read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef long long i64; typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef i64 ll2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const ll2* __restrict__ a, const ll2* __restrict__ b, const ll2* __restrict__ c, const ll2* __restrict__ d, i64 ntiles,
                                         ll2* __restrict__ out, i64 outPerWave, u64* __restrict__ sink, u64 lineMask) {
    const int lane = threadIdx.x & 63;
    const i64 wave = (i64)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (i64)gridDim.x * 4;
    u64 s = 0;
    i64 w = 0;                       // 16-byte words this wave has stored per lane
    for (i64 t = wave; t < ntiles; t += nwaves) {
        const i64 i = t * 64 + lane;
        const ll2 va = __builtin_nontemporal_load(a + i), vb = __builtin_nontemporal_load(b + i), vc = __builtin_nontemporal_load(c + i), vd = __builtin_nontemporal_load(d + i);
        s += (u64)va.x + (u64)va.y + (u64)vb.x + (u64)vb.y + (u64)vc.x + (u64)vc.y + (u64)vd.x + (u64)vd.y;
        if (MODE != 0 && ((t / nwaves) & 1) == 0) {          // every second tile: 1 KiB of 8 KiB read... see main(): bytes written = 1/8 of the bytes read
            ll2 v; v.x = va.x ^ vc.x; v.y = vb.y ^ vd.y;
            if (MODE == 1) out[(wave * outPerWave + w) * 64 + lane] = v;
            else if (MODE == 2) {      // eight 128-byte lines per store instruction, each to another of the wave's 256 region streams
                const i64 r = ((t * 2654435761ll) + (lane >> 3) * 31) & 255;
                out[((r * nwaves + wave) * (outPerWave / 32 + 2) + (w >> 5)) * 8 + (lane & 7)] = v;
            } else if (MODE == 3) {    // the wave's own stream, its lines in a scrambled order (every line written exactly once)
                const i64 L = (w * 8 + (lane >> 3)) * 40503ll % (outPerWave * 8);      // 40503 is coprime with outPerWave * 8 (odd x 8) only if ... see main
                out[(wave * outPerWave * 8 + L) * 8 + (lane & 7)] = v;
            } else if (MODE == 4) {    // every line written exactly once, anywhere in the buffer: a bijection of the global line number modulo 2^k
                const u64 L = ((u64)((wave * outPerWave + w) * 8 + (lane >> 3)) * 0x9E3779B1ull) & lineMask;
                out[L * 8 + (lane & 7)] = v;
            } else {                   // MODE 5: lines interleaved ACROSS the waves: line k of wave i at (k * nwaves + i) - streams that advance in step fill memory densely
                const i64 L = (w * 8 + (lane >> 3)) * nwaves + wave;
                out[L * 8 + (lane & 7)] = v;
            }
            w++;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0 && s == 0x1234567) atomicAdd(sink, s);
}

int main(int argc, char** argv) {
    const i64 rows = argc > 1 ? atoll(argv[1]) : 1250000000ll;
    const i64 ntiles = rows / 128;
    ll2 *a, *b, *c, *d, *out; u64* sink;
    const size_t colBytes = (size_t)ntiles * 128 * 8;
    CK(hipMalloc(&a, colBytes)); CK(hipMalloc(&b, colBytes)); CK(hipMalloc(&c, colBytes)); CK(hipMalloc(&d, colBytes)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(a, 1, colBytes)); CK(hipMemset(b, 2, colBytes)); CK(hipMemset(c, 3, colBytes)); CK(hipMemset(d, 4, colBytes)); CK(hipMemset(sink, 0, 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const unsigned grid = 2 * (unsigned)prop.multiProcessorCount;
    const i64 nwaves = (i64)grid * 4;
    // 16-byte words per lane a wave may store (every second of its tiles); ODD, so that the waves' streams - which advance in step - do not
    // start a multiple of 16 KiB apart and meet in the same memory channels (the first version of this file measured exactly that: 8.2 ms)
    const i64 outPerWave = ((ntiles / nwaves + 2) / 2 + 512) | 1;
    size_t outBytes = (size_t)nwaves * (size_t)(outPerWave + 256) * 64 * 16 * 2;
    { size_t pow2 = 1; while (pow2 < (size_t)nwaves * (size_t)outPerWave * 8) pow2 <<= 1; if (pow2 * 128 > outBytes) outBytes = pow2 * 128; }
    CK(hipMalloc(&out, outBytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double readGB = 4.0 * colBytes / 1e9, writeGB = (double)(ntiles / 2) * 1024 / 1e9;
    printf("rows %lld: %.1f GB read per launch, %.2f GB written by the store forms (%.3f of the read), grid %u x 256\n", rows, readGB, writeGB, writeGB / readGB, grid);
    double base = 0;
    u64 lineMask = 1; while (lineMask < (u64)nwaves * (u64)outPerWave * 8) lineMask <<= 1;      // lines of 128 B; the buffer holds 2 x that
    lineMask -= 1;
    for (int mode = 0; mode < 6; mode++) {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, b, c, d, ntiles, out, outPerWave, sink, lineMask);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, a, b, c, d, ntiles, out, outPerWave, sink, lineMask);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, a, b, c, d, ntiles, out, outPerWave, sink, lineMask);
            else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, a, b, c, d, ntiles, out, outPerWave, sink, lineMask);
            else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, a, b, c, d, ntiles, out, outPerWave, sink, lineMask);
            else hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, a, b, c, d, ntiles, out, outPerWave, sink, lineMask);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        if (mode == 0) base = best;
        printf("%-8s %8.3f ms  read %.0f GB/s", mode == 0 ? "read" : mode == 1 ? "plain" : mode == 2 ? "regions" : mode == 3 ? "own-scr" : mode == 4 ? "any-line" : "interlv", best, readGB / (best * 1e-3));
        if (mode) printf("   stores cost %.3f ms = %.2f ms per GB written = what %.1f GB of reads cost", best - base, (best - base) / writeGB, (best - base) * (readGB / base));
        printf("\n");
    }
    return 0;
}
