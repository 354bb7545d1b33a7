// Experiment (not part of the product): LDS write-combining partition scatter + per-partition LDS aggregation for
//   select b, sum(c), sum(d), count(*), min(row) from t where a < thr group by b      (G = 2^20 dense groups)
// build: hipcc -O3 --offload-arch=gfx950 -o part_bench part_bench.hip ; run: ./part_bench ROWS SEL
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef long long i64; typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int P = 256, GPP = 4096, G = P * GPP;
#ifdef NOCOUNT
#define AGG_COUNT(g)
#else
#define AGG_COUNT(g) atomicAdd(&tab[3 * GPP + g], 1ull);
#endif
#ifdef LIGHT
__device__ inline void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else
__device__ inline void wg_barrier() { __syncthreads(); }
#endif

__device__ inline u64 mix(u64 x) { x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }
__global__ void gen(i64* a, i64* b, i64* c, i64* d, i64 n) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
        const u64 x = mix((u64)r), y = mix(x);
        a[r] = (i64)(x & 0x7fffffffull); b[r] = (i64)((x >> 31) & (G - 1)); c[r] = (i64)(y & 0xfffff); d[r] = (i64)((y >> 20) & 0xfffff);
    }
}
// reference: direct atomics
__global__ void ref_agg(const i64* a, const i64* b, const i64* c, const i64* d, i64 n, i64 thr, u64* out) {
    for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x)
        if (a[r] < thr) { const i64 g = b[r]; atomicMin((i64*)&out[g], r); atomicAdd(&out[G + g], (u64)c[r]); atomicAdd(&out[2 * G + g], (u64)d[r]); atomicAdd(&out[3 * G + g], 1ull); }
}
__global__ void init_tab(u64* t) { for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < 4ll * G; i += (i64)gridDim.x * blockDim.x) t[i] = i < G ? 0x7fffffffffffffffull : 0ull; }
__global__ void cmp_tab(const u64* x, const u64* y, u32* bad, int from) { for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x + (i64)from * G; i < 4ll * G; i += (i64)gridDim.x * blockDim.x) if (x[i] != y[i]) atomicAdd(bad, 1u); }

// ---------------------------------------------------------------------------------------------------------------------
// scatter: RECW u64 words per record.  RECW == 1: [gip:12 | c:20 | d:20]   RECW == 2: word0 = gip<<40 | row, word1 = c | d<<32
struct Ctl { u64 maxFirst; u64 watermark; u32 seen; u32 pad; };
template <int RECW, int BLOCK, int RPT, int COUNT_ONLY, int TRACK, int NT, int PF>
__global__ void __launch_bounds__(BLOCK) scatter(const i64* __restrict__ ca, const i64* __restrict__ cb, const i64* __restrict__ cc, const i64* __restrict__ cd,
                                                 i64 n, i64 thr, u64* rec, const u64* __restrict__ regStart, const u32* __restrict__ regCap, u32* regCount, u32* flags, int roundStep, u64* trk, Ctl* ctl, u32 D, int diag) {
    constexpr int S = 32;                         // ring slots per partition
    constexpr int LPR = 16 / RECW;                // records per 128-byte line
    __shared__ u64 ring[COUNT_ONLY ? 1 : P * S * RECW];
    __shared__ u32 tail[P], head[P];
    __shared__ u32 sNew; __shared__ u64 sMax, sWm; __shared__ u32 sAny[2];
    if (threadIdx.x == 0) { sAny[0] = sAny[1] = 0; sNew = 0; sMax = 0; sWm = TRACK ? __hip_atomic_load(&ctl->watermark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; }
    for (int i = threadIdx.x; i < P; i += BLOCK) { tail[i] = 0; head[i] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int ROWS = BLOCK * RPT;             // rows per round
    const i64 nRounds = n / ROWS;                 // (experiment: n is a multiple of ROWS)
    const u64* myStart = regStart + (u64)blockIdx.x * P;
    const u32* myCap = regCap + (u64)blockIdx.x * P;
    i64 va[RPT], vb[RPT], vc[RPT], vd[RPT];
    i64 na[RPT], nb[RPT], nc[RPT], nd[RPT];
    auto loadRound = [&](i64 round, i64 (&va)[RPT], i64 (&vb)[RPT], i64 (&vc)[RPT], i64 (&vd)[RPT]) {
        const i64 base = round * ROWS + (i64)wv * 64 * RPT;      // a wave covers 64*RPT consecutive rows, 2 consecutive rows per lane per 128-row tile
#pragma unroll
        for (int j = 0; j < RPT / 2; j++) {
            const i64 r = base + j * 128 + lane * 2;
            typedef i64 v2 __attribute__((ext_vector_type(2)));
            v2 A, B, C, D;
            if (NT) { A = __builtin_nontemporal_load((const v2*)(ca + r)); B = __builtin_nontemporal_load((const v2*)(cb + r)); C = __builtin_nontemporal_load((const v2*)(cc + r)); D = __builtin_nontemporal_load((const v2*)(cd + r)); }
            else { A = *(const v2*)(ca + r); B = *(const v2*)(cb + r); C = *(const v2*)(cc + r); D = *(const v2*)(cd + r); }
            va[2 * j] = A.x; va[2 * j + 1] = A.y; vb[2 * j] = B.x; vb[2 * j + 1] = B.y; vc[2 * j] = C.x; vc[2 * j + 1] = C.y; vd[2 * j] = D.x; vd[2 * j + 1] = D.y;
        }
    };
    if (PF && (i64)blockIdx.x * roundStep < nRounds) loadRound((i64)blockIdx.x * roundStep, na, nb, nc, nd);
    for (i64 round = (i64)blockIdx.x * roundStep; round < nRounds; round += (i64)gridDim.x * roundStep) {
        if (PF) {
#pragma unroll
            for (int i = 0; i < RPT; i++) { va[i] = na[i]; vb[i] = nb[i]; vc[i] = nc[i]; vd[i] = nd[i]; }
            const i64 nx = round + (i64)gridDim.x * roundStep;
            if (nx < nRounds) loadRound(nx, na, nb, nc, nd);
        } else loadRound(round, va, vb, vc, vd);
        const i64 base = round * ROWS + (i64)wv * 64 * RPT;
        u32 k[RPT]; int pp[RPT]; u32 pending = 0;
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            pp[i] = (int)(vb[i] >> 12);
            if (va[i] < thr) { k[i] = atomicAdd(&tail[pp[i]], 1u); pending |= 1u << i; }
        }
        if (COUNT_ONLY) continue;
        if (diag == 2) { u64 f = 0;
#pragma unroll
            for (int i = 0; i < RPT; i++) f ^= (u64)vc[i] ^ (u64)vd[i];
            if (f == 0x123456789abcdefull) flags[1] = 1; continue; }
        if (TRACK) {
            const u64 wm = sWm;
            if ((u64)base < wm) {
#pragma unroll
                for (int i = 0; i < RPT; i++) {
                    const u64 row = (u64)(base + (i >> 1) * 128 + lane * 2 + (i & 1));
                    if (((pending >> i) & 1u) && row < wm) {
                        if (row < trk[vb[i]]) {
                            const u64 old = (u64)atomicMin((i64*)&trk[vb[i]], (i64)row);
                            if (old == 0x7fffffffffffffffull) { atomicAdd(&sNew, 1u); atomicMax(&sMax, row); }
                        }
                    }
                }
            }
        }
        int any; int it = 0;
        do {
#pragma unroll
            for (int i = 0; i < RPT; i++) {
                if ((pending >> i) & 1u) {
                    if (k[i] - head[pp[i]] < (u32)S) {
                        u64* slot = &ring[((u32)pp[i] * S + (k[i] & (S - 1))) * RECW];
                        const i64 row = base + (i >> 1) * 128 + lane * 2 + (i & 1);
                        if (RECW == 1) slot[0] = ((u64)(vb[i] & (GPP - 1)) << 40) | ((u64)vc[i] << 20) | (u64)vd[i];
                        else { slot[0] = ((u64)(vb[i] & (GPP - 1)) << 40) | (u64)row; slot[1] = (u64)vc[i] | ((u64)vd[i] << 32); }
                        pending &= ~(1u << i);
                    }
                }
            }
#ifdef LIGHT
            if (pending) sAny[it & 1] = 1u;
            wg_barrier();
#else
            __syncthreads();
#endif
            // flush complete lines: 8 lanes per partition, 16 bytes each
            for (int p = threadIdx.x >> 3; p < P; p += BLOCK >> 3) {
                u32 h = head[p];
                const u32 tl = min(tail[p], h + (u32)S);
                const u32 full = tl / LPR * LPR;
                if (h < full) {
                    const int j = threadIdx.x & 7;
                    for (; h < full; h += LPR) {
                        const ulonglong2 v = *(const ulonglong2*)&ring[((u32)p * S + (h & (S - 1))) * RECW + j * 2];
                        if (diag == 1) { if (v.x == 0x123456789abcdefull) flags[1] = 1; }
                        else if (diag == 3) *(ulonglong2*)&rec[(((u64)blockIdx.x << 16) + (((u64)p << 8) + h) % 65536) * RECW + j * 2] = v;
                        else if (diag == 4 && h + LPR <= myCap[p]) { typedef u64 uv2 __attribute__((ext_vector_type(2))); uv2 t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, (uv2*)&rec[(myStart[p] + h) * RECW + j * 2]); }
                        else if (h + LPR <= myCap[p]) *(ulonglong2*)&rec[(myStart[p] + h) * RECW + j * 2] = v;
                        else if (j == 0) atomicOr(flags, 1u);
                    }
                    if (j == 0) head[p] = h;
                }
            }
            if (TRACK && threadIdx.x == BLOCK - 1) {
                if (sNew) {
                    const u64 m = atomicMax(&ctl->maxFirst, sMax);
                    asm volatile("" :: "v"(m));
                    const u32 seen = atomicAdd(&ctl->seen, sNew + (u32)(m & 0)) + sNew;
                    if (TRACK == 1 && seen == D) __hip_atomic_store(&ctl->watermark, __hip_atomic_load(&ctl->maxFirst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sNew = 0; sMax = 0;
                }
                sWm = __hip_atomic_load(&ctl->watermark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#ifdef LIGHT
            any = (int)sAny[it & 1];
            if (threadIdx.x == 0) sAny[(it + 1) & 1] = 0u;
            wg_barrier();
            it++;
#else
            any = __syncthreads_or(pending != 0);
#endif
        } while (any);
    }
    __syncthreads();
    if (!COUNT_ONLY) {
        // the incomplete last line of every partition
        for (int p = threadIdx.x >> 3; p < P; p += BLOCK >> 3) {
            const u32 h = head[p], tl = tail[p];
            const int j = threadIdx.x & 7;
            if (h < tl) {
                if (tl <= myCap[p]) { if ((u32)(j * 2 / RECW) < tl - h) *(ulonglong2*)&rec[(myStart[p] + h) * RECW + j * 2] = *(const ulonglong2*)&ring[((u32)p * S + (h & (S - 1))) * RECW + j * 2]; }
                else if (j == 0) atomicOr(flags, 1u);
            }
        }
    }
    for (int i = threadIdx.x; i < P; i += BLOCK) regCount[(u64)blockIdx.x * P + i] = tail[i];
}

// aggregation of one partition: one workgroup, LDS table [4][GPP]
template <int RECW>
__global__ void __launch_bounds__(1024) part_agg(const u64* __restrict__ rec, const u64* __restrict__ regStart, const u32* __restrict__ regCount, int nwg, u64* out) {
    __shared__ u64 tab[4 * GPP];
    for (int i = threadIdx.x; i < 4 * GPP; i += 1024) tab[i] = i < GPP ? 0x7fffffffffffffffull : 0ull;
    __syncthreads();
    const int p = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int w = wv; w < nwg; w += 16) {
        const u64 st = regStart[(u64)w * P + p]; const u32 cnt = regCount[(u64)w * P + p];
        if (RECW == 1) {
            for (u32 i = lane * 2; i < cnt; i += 128) {
                const ulonglong2 v = *(const ulonglong2*)&rec[st + i];
                { const int g = (int)(v.x >> 40); atomicAdd(&tab[GPP + g], (v.x >> 20) & 0xfffff); atomicAdd(&tab[2 * GPP + g], v.x & 0xfffff); AGG_COUNT(g) }
                if (i + 1 < cnt) { const int g = (int)(v.y >> 40); atomicAdd(&tab[GPP + g], (v.y >> 20) & 0xfffff); atomicAdd(&tab[2 * GPP + g], v.y & 0xfffff); AGG_COUNT(g) }
            }
        } else {
            for (u32 i = lane; i < cnt; i += 64) {
                const ulonglong2 v = *(const ulonglong2*)&rec[(st + i) * 2];
                const int g = (int)(v.x >> 40);
                atomicMin((i64*)&tab[g], (i64)(v.x & ((1ull << 40) - 1)));
                atomicAdd(&tab[GPP + g], v.y & 0xffffffffull); atomicAdd(&tab[2 * GPP + g], v.y >> 32); atomicAdd(&tab[3 * GPP + g], 1ull);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * GPP; i += 1024) {
        if (RECW == 1 && i < GPP) continue;
        out[(u64)(i / GPP) * G + (u64)p * GPP + (i & (GPP - 1))] = tab[i];
    }
}

template <int RECW, int BLOCK, int RPT, int TRACK, int NT, int PF>
static void runVariant(const char* name, const i64* a, const i64* b, const i64* c, const i64* d, i64 n, i64 thr, double sel, u64* refTab, int wgPerCU) {
    const int nwg = 256 * wgPerCU;
    u64 *regStart, *tab; u32 *regCap, *regCount, *flags, *bad;
    CK(hipMalloc(&regStart, (size_t)nwg * P * 8)); CK(hipMalloc(&regCap, (size_t)nwg * P * 4)); CK(hipMalloc(&regCount, (size_t)nwg * P * 4));
    CK(hipMalloc(&flags, 8)); bad = flags + 1; CK(hipMalloc(&tab, 4ull * G * 8)); Ctl* ctl; CK(hipMalloc(&ctl, sizeof(Ctl)));
    hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    // sample: every 32nd round, counts only
    CK(hipEventRecord(e0));
    scatter<RECW, BLOCK, RPT, 1, 0, NT, 0><<<nwg, BLOCK>>>(a, b, c, d, n, thr, nullptr, regStart, regCap, regCount, flags, 32, nullptr, nullptr, 0, 0);
    CK(hipEventRecord(e1));
    std::vector<u32> cnt((size_t)nwg * P);
    CK(hipMemcpy(cnt.data(), regCount, cnt.size() * 4, hipMemcpyDeviceToHost));
    // capacities: per-partition estimate spread evenly over the workgroups, + 15 % + 256, rounded to lines
    std::vector<u64> start((size_t)nwg * P); std::vector<u32> cap((size_t)nwg * P);
    u64 pos = 0;
    {
        std::vector<u64> capv(P);
        for (int p = 0; p < P; p++) {
            u64 tot = 0; for (int w = 0; w < nwg; w++) tot += cnt[(size_t)w * P + p];
            const u64 per = (u64)((double)tot * 32.0 / nwg * 1.15) + 256;
            capv[p] = (per + 15) / 16 * 16;
        }
#ifdef WGMAJOR
        for (int w = 0; w < nwg; w++) for (int p = 0; p < P; p++) { start[(size_t)w * P + p] = pos; cap[(size_t)w * P + p] = (u32)capv[p]; pos += capv[p]; }
#else
        for (int p = 0; p < P; p++) for (int w = 0; w < nwg; w++) { start[(size_t)w * P + p] = pos; cap[(size_t)w * P + p] = (u32)capv[p]; pos += capv[p]; }
#endif
    }
    u64* rec; CK(hipMalloc(&rec, pos * 8 * RECW));
    CK(hipMemcpy(regStart, start.data(), start.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(regCap, cap.data(), cap.size() * 4, hipMemcpyHostToDevice));
    float best[3] = {1e9f, 1e9f, 1e9f}; float dg[3] = {1e9f, 1e9f, 1e9f};
    {   // floor: the same loop, tickets only
        CK(hipEventRecord(e1)); scatter<RECW, BLOCK, RPT, 1, 0, NT, PF><<<nwg, BLOCK>>>(a, b, c, d, n, thr, nullptr, regStart, regCap, regCount, flags, 1, nullptr, nullptr, 0, 0); CK(hipEventRecord(e2)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&best[0], e1, e2));
    }
    std::vector<float> tn, tt, ta;
    for (int it = 0; it < 12; it++) {
        CK(hipMemset(flags, 0, 8));
        init_tab<<<1024, 256>>>(tab); { Ctl h = {0, 0x7fffffffffffffffull, 0, 0}; CK(hipMemcpy(ctl, &h, sizeof h, hipMemcpyHostToDevice)); }
        CK(hipEventRecord(e1));
        scatter<RECW, BLOCK, RPT, 0, TRACK, NT, PF><<<nwg, BLOCK>>>(a, b, c, d, n, thr, rec, regStart, regCap, regCount, flags, 1, tab, ctl, (u32)G, (it & 1) ? 4 : 0);
        CK(hipEventRecord(e2));
        part_agg<RECW><<<P, 1024>>>(rec, regStart, regCount, nwg, tab);
        CK(hipEventRecord(e3));
        CK(hipDeviceSynchronize());
        float t; CK(hipEventElapsedTime(&t, e1, e2)); ((it & 1) ? tt : tn).push_back(t); CK(hipEventElapsedTime(&t, e2, e3)); ta.push_back(t);
    }
    std::sort(tn.begin(), tn.end()); std::sort(tt.begin(), tt.end()); std::sort(ta.begin(), ta.end());
    best[1] = tn[0]; best[2] = ta[0]; dg[1] = tt[0]; dg[2] = tt[tt.size() / 2]; best[0] = tn[tn.size() / 2];
    cmp_tab<<<1024, 256>>>(tab, refTab, bad, (RECW == 1 && !TRACK) ? 1 : 0);
    Ctl hc; CK(hipMemcpy(&hc, ctl, sizeof hc, hipMemcpyDeviceToHost));
    u32 fl[2]; CK(hipMemcpy(fl, flags, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)n * 32.0;
    printf("%-28s sel %.2f  ntstore min %6.3f med %6.3f | plain med %6.3f min %7.3f ms  agg %6.3f ms  total %7.3f ms  = %.2f TB/s algorithmic (%.2f of 8)  overflow %u mismatches %u  records buffer %.2f GB  seen %u watermark %lld\n",
           name, sel, dg[1], dg[2], best[0], best[1], best[2], best[1] + best[2], bytes / ((best[1] + best[2]) * 1e-3) / 1e12, bytes / ((best[1] + best[2]) * 1e-3) / 8e12, fl[0], fl[1], (double)pos * 8 * RECW / 1e9, hc.seen, (long long)hc.watermark);
    fflush(stdout);
    CK(hipFree(rec)); CK(hipFree(regStart)); CK(hipFree(regCap)); CK(hipFree(regCount)); CK(hipFree(flags)); CK(hipFree(tab));
}

int main(int argc, char** argv) {
    const i64 n = argc > 1 ? atoll(argv[1]) : (1ll << 24);
    i64 *a, *b, *c, *d; u64* refTab;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8)); CK(hipMalloc(&d, n * 8)); CK(hipMalloc(&refTab, 4ull * G * 8));
    gen<<<4096, 256>>>(a, b, c, d, n); CK(hipDeviceSynchronize());
    for (int si = 2; si < argc; si++) {
        const double sel = atof(argv[si]); const i64 thr = (i64)(sel * 2147483648.0);
        init_tab<<<1024, 256>>>(refTab); ref_agg<<<4096, 256>>>(a, b, c, d, n, thr, refTab); CK(hipDeviceSynchronize());
        runVariant<1, 1024, 4, 1, 1, 0>("8B 1024 rpt4 track", a, b, c, d, n, thr, sel, refTab, 1);
        runVariant<1, 1024, 4, 1, 1, 1>("8B 1024 rpt4 track pf", a, b, c, d, n, thr, sel, refTab, 1);
        runVariant<1, 1024, 8, 1, 1, 0>("8B 1024 rpt8 track", a, b, c, d, n, thr, sel, refTab, 1);
        runVariant<1, 1024, 4, 1, 1, 0>("8B 1024 rpt4 track (again)", a, b, c, d, n, thr, sel, refTab, 1);
    }
    return 0;
}
