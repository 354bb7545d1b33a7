// aot_kernels.hip — hand-written gfx950 kernels that do not depend on the query:
//   * column statistics (min / max, byte-value sets) gathered when a table is created
//   * the deterministic TPC-H-shaped generator (bit-identical to resql_amd/datagen.py)
//   * the read-only streaming bandwidth probe (the measured roofline of SURVEY.md §8d)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "engine_internal.h"

namespace rsq {

typedef long long i64;
typedef unsigned long long u64;

// ------------------------------------------------------------------------------------------------
// statistics
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_minmax(const T* __restrict__ p, i64 n, i64* out /* [min, max, descents, repeats] */) {
    i64 mn = 0x7fffffffffffffffll, mx = (i64)0x8000000000000000ull;
    u64 desc = 0;          // rows smaller than the row before them: 0 = the column is in ascending order
    u64 same = 0;          // rows equal to the row before them: 0 (in an ascending column) = its values are unique
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        i64 v = (i64)p[i];
        mn = v < mn ? v : mn; mx = v > mx ? v : mx;
        if (i > 0) { const i64 pv = (i64)p[i - 1]; if (v < pv) desc++; if (v == pv) same++; }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        i64 a = __shfl_xor(mn, m, 64), b = __shfl_xor(mx, m, 64);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        desc += (u64)__shfl_xor((long long)desc, m, 64);
        same += (u64)__shfl_xor((long long)same, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], mn); atomicMax(&out[1], mx);
        if (desc) atomicAdd(reinterpret_cast<u64*>(&out[2]), desc);
        if (same) atomicAdd(reinterpret_cast<u64*>(&out[3]), same);
    }
}

__global__ void __launch_bounds__(256) k_byteset(const unsigned char* __restrict__ p, i64 n, unsigned* out /* 8 words */) {
    __shared__ unsigned s[8];
    if (threadIdx.x < 8) s[threadIdx.x] = 0;
    __syncthreads();
    unsigned loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        unsigned v = p[i];
#pragma unroll
        for (int w = 0; w < 8; w++) loc[w] |= ((v >> 5) == (unsigned)w) ? (1u << (v & 31)) : 0u;
    }
#pragma unroll
    for (int w = 0; w < 8; w++) if (loc[w]) atomicOr(&s[w], loc[w]);
    __syncthreads();
    if (threadIdx.x < 8 && s[threadIdx.x]) atomicOr(&out[threadIdx.x], s[threadIdx.x]);
}

static void hostStats(Table& t) {
    for (auto& c : t.cols) {
        c.stats = ColumnStats();
        if (!c.dptr || t.nRows == 0) continue;
        int w = columnWidth(c.type);
        if (c.type.tag == RSQ_BOOL || (c.type.tag == RSQ_CHAR && c.type.len == 1)) {
            bool seen[256] = {false};
            const unsigned char* p = (const unsigned char*)c.dptr;
            for (int64_t i = 0; i < t.nRows; i++) seen[p[i]] = true;
            for (int v = 0; v < 256; v++) if (seen[v]) c.stats.distinctBytes.push_back((uint8_t)v);
            c.stats.min = c.stats.distinctBytes.front(); c.stats.max = c.stats.distinctBytes.back();
            c.stats.valid = true;
        } else if (w == 4 || w == 8) {
            if (c.type.isString()) continue;
            int64_t mn = INT64_MAX, mx = INT64_MIN, prev = INT64_MIN;
            bool asc = true, strict = true;
            for (int64_t i = 0; i < t.nRows; i++) {
                int64_t v = (w == 8) ? ((const int64_t*)c.dptr)[i]
                          : (c.type.tag == RSQ_DATE ? (int64_t)((const uint32_t*)c.dptr)[i] : (int64_t)((const int32_t*)c.dptr)[i]);
                mn = std::min(mn, v); mx = std::max(mx, v);
                if (v < prev) asc = false;
                if (i > 0 && v <= prev) strict = false;
                prev = v;
            }
            c.stats.min = mn; c.stats.max = mx; c.stats.valid = true; c.stats.ascending = asc; c.stats.strictlyAscending = asc && strict;
        }
    }
}

void computeColumnStats(Context& ctx, Table& t) {
    if (ctx.device < 0) { hostStats(t); return; }
    if (t.nRows == 0) return;
    RSQ_HIP(hipSetDevice(ctx.device));
    i64* dmm = (i64*)ctx.alloc(4 * sizeof(i64));
    unsigned* dset = (unsigned*)ctx.alloc(8 * sizeof(unsigned));
    const unsigned grid = 1024;
    for (auto& c : t.cols) {
        c.stats = ColumnStats();                // (a table that grew or was refreshed: nothing of the old statistics stays)
        if (!c.dptr) continue;
        if (c.type.tag == RSQ_BOOL || (c.type.tag == RSQ_CHAR && c.type.len == 1)) {
            RSQ_HIP(hipMemsetAsync(dset, 0, 32, ctx.stream));
            hipLaunchKernelGGL(k_byteset, dim3(grid), dim3(256), 0, ctx.stream, (const unsigned char*)c.dptr, (i64)t.nRows, dset);
            unsigned h[8];
            RSQ_HIP(hipMemcpyAsync(h, dset, 32, hipMemcpyDeviceToHost, ctx.stream));
            RSQ_HIP(hipStreamSynchronize(ctx.stream));
            for (int v = 0; v < 256; v++) if (h[v >> 5] & (1u << (v & 31))) c.stats.distinctBytes.push_back((uint8_t)v);
            if (!c.stats.distinctBytes.empty()) { c.stats.min = c.stats.distinctBytes.front(); c.stats.max = c.stats.distinctBytes.back(); c.stats.valid = true; }
        } else if (!c.type.isString()) {
            i64 init[4] = {0x7fffffffffffffffll, (i64)0x8000000000000000ull, 0, 0};
            RSQ_HIP(hipMemcpyAsync(dmm, init, 32, hipMemcpyHostToDevice, ctx.stream));
            if (c.type.tag == RSQ_INT) hipLaunchKernelGGL(k_minmax<int>, dim3(grid), dim3(256), 0, ctx.stream, (const int*)c.dptr, (i64)t.nRows, dmm);
            else if (c.type.tag == RSQ_DATE) hipLaunchKernelGGL(k_minmax<unsigned>, dim3(grid), dim3(256), 0, ctx.stream, (const unsigned*)c.dptr, (i64)t.nRows, dmm);
            else hipLaunchKernelGGL(k_minmax<i64>, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)c.dptr, (i64)t.nRows, dmm);
            i64 h[4];
            RSQ_HIP(hipMemcpyAsync(h, dmm, 32, hipMemcpyDeviceToHost, ctx.stream));
            RSQ_HIP(hipStreamSynchronize(ctx.stream));
            c.stats.min = h[0]; c.stats.max = h[1]; c.stats.valid = true; c.stats.ascending = h[2] == 0; c.stats.strictlyAscending = h[2] == 0 && h[3] == 0;
        }
    }
    ctx.free(dmm); ctx.free(dset);
}

// ------------------------------------------------------------------------------------------------
// generator (mirror of resql_amd/datagen.py; keep the two in lock step)
// ------------------------------------------------------------------------------------------------
enum { S_QTY = 1, S_PKEY, S_DISC, S_TAX, S_SHIP, S_RCPT, S_RFLG, S_ODATE, S_OCUST, S_PERM, S_CSEG,
       S_A = 21, S_B, S_C, S_D };

__host__ __device__ inline u64 g_mix(u64 seed, u64 stream, u64 idx) {
    u64 z = seed + stream * 0x9E3779B97F4A7C15ull + idx * 0xD1342543DE82EF95ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline i64 g_uniform(u64 seed, u64 stream, u64 idx, u64 n) {
    return (i64)(((g_mix(seed, stream, idx) >> 32) * n) >> 32);
}
__host__ __device__ inline unsigned g_yyyymmdd(i64 daysSince1992) {
    i64 z = daysSince1992 + 8035 + 719468;
    i64 era = z / 146097;
    i64 doe = z - era * 146097;
    i64 yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    i64 y = yoe + era * 400;
    i64 doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    i64 mp = (5 * doy + 2) / 153;
    i64 d = doy - (153 * mp + 2) / 5 + 1;
    i64 m = mp < 10 ? mp + 3 : mp - 9;
    if (m <= 2) y += 1;
    return (unsigned)(y * 10000 + m * 100 + d);
}

struct GenTables { signed char orderOf[14][28]; };

static GenTables makeGenTables() {
    GenTables g;
    const int base[7] = {4, 1, 7, 3, 5, 2, 6};
    for (int p = 0; p < 14; p++) {
        int counts[7];
        for (int i = 0; i < 7; i++) {
            int k = p % 7;
            counts[i] = (p < 7) ? base[(i + k) % 7] : base[6 - ((i + k) % 7)];
        }
        int pos = 0;
        for (int j = 0; j < 7; j++) for (int l = 0; l < counts[j]; l++) g.orderOf[p][pos++] = (signed char)j;
    }
    return g;
}

struct LineitemCols {
    int* l_orderkey; i64* l_quantity; i64* l_extendedprice; i64* l_discount; i64* l_tax;
    unsigned char* l_returnflag; unsigned char* l_linestatus; unsigned* l_shipdate;
};

__global__ void __launch_bounds__(256) k_gen_lineitem(LineitemCols c, GenTables gt, i64 row0, i64 n, u64 seed, u64 nPart) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u64 r = (u64)(row0 + i);
        u64 block = r / 28; int off = (int)(r % 28);
        int pat = (int)g_uniform(seed, S_PERM, block, 14);
        u64 o = block * 7 + (u64)gt.orderOf[pat][off];
        i64 qty = 1 + g_uniform(seed, S_QTY, r, 50);
        i64 pk = 1 + g_uniform(seed, S_PKEY, r, nPart);
        i64 retail = 90000 + ((pk / 10) % 20001) + 100 * (pk % 1000);
        i64 ship = g_uniform(seed, S_ODATE, o, 2406) + 1 + g_uniform(seed, S_SHIP, r, 121);
        i64 rcpt = ship + 1 + g_uniform(seed, S_RCPT, r, 30);
        if (c.l_orderkey) c.l_orderkey[i] = (int)((o / 8) * 32 + (o % 8) + 1);
        c.l_quantity[i] = qty;
        c.l_extendedprice[i] = qty * retail;
        c.l_discount[i] = g_uniform(seed, S_DISC, r, 11);
        c.l_tax[i] = g_uniform(seed, S_TAX, r, 9);
        c.l_shipdate[i] = g_yyyymmdd(ship);
        unsigned char ra = g_uniform(seed, S_RFLG, r, 2) == 0 ? 'R' : 'A';
        c.l_returnflag[i] = rcpt <= 1263 ? ra : 'N';
        c.l_linestatus[i] = ship > 1263 ? 'O' : 'F';
    }
}

__global__ void __launch_bounds__(256) k_gen_orders(int* okey, int* ckey, unsigned* odate, int* prio, i64 o0, i64 n, u64 seed, u64 nCust) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u64 o = (u64)(o0 + i);
        i64 k = 1 + g_uniform(seed, S_OCUST, o, nCust);
        if (k % 3 == 0) k -= 1;
        if (k < 1) k = 1;
        okey[i] = (int)((o / 8) * 32 + (o % 8) + 1);
        ckey[i] = (int)k;
        odate[i] = g_yyyymmdd(g_uniform(seed, S_ODATE, o, 2406));
        prio[i] = 0;
    }
}

__global__ void __launch_bounds__(256) k_gen_customer(int* ckey, char* seg /* 10 B per row */, i64 c0, i64 n, u64 seed) {
    const char names[5][11] = {"AUTOMOBILE", "BUILDING\0\0", "FURNITURE\0", "MACHINERY\0", "HOUSEHOLD\0"};
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u64 c = (u64)(c0 + i);
        int s = (int)g_uniform(seed, S_CSEG, c, 5);
        ckey[i] = (int)(c + 1);
        for (int b = 0; b < 10; b++) seg[i * 10 + b] = names[s][b];
    }
}

__global__ void __launch_bounds__(256) k_gen_synth(i64* a, i64* b, i64* c, i64* d, i64 row0, i64 n, u64 seed, u64 groups) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u64 r = (u64)(row0 + i);
        a[i] = g_uniform(seed, S_A, r, 1ull << 31);
        b[i] = g_uniform(seed, S_B, r, groups);
        c[i] = g_uniform(seed, S_C, r, 1ull << 20);
        d[i] = g_uniform(seed, S_D, r, 1ull << 20);
    }
}

static void addCol(Context& ctx, Table& t, const char* name, Type type, bool withData) {
    TableColumn c; c.name = name; c.type = type;
    if (withData) { c.dptr = ctx.allocRaw((size_t)t.nRows * (size_t)columnWidth(type)); c.owned = true; }
    t.cols.push_back(c);
}
static void* colPtr(Table& t, const char* n) { int i = t.findCol(n); return i < 0 ? nullptr : t.cols[i].dptr; }

void generateTable(Context& ctx, Table& t, int kind, int64_t row0, int64_t nRows, double sf, int64_t param, uint64_t seed) {
    if (ctx.device < 0) throw Error(RSQ_ERR_DEVICE, "rsq_table_generate needs a device context");
    RSQ_HIP(hipSetDevice(ctx.device));
    t.ctx = &ctx; t.nRows = nRows; t.row0 = row0;
    Type dec0 = Type::decimal(12, 0), dec2 = Type::decimal(12, 2);
    auto CH = [](int n) { Type x(RSQ_CHAR); x.len = n; return x; };
    auto VC = [](int n) { Type x(RSQ_VARCHAR); x.len = n; return x; };
    const unsigned grid = 2048;
    if (kind == 0) {
        t.name = "lineitem";
        bool withKey = param != 0;
        addCol(ctx, t, "l_orderkey", Type(RSQ_INT), withKey);
        addCol(ctx, t, "l_partkey", Type(RSQ_INT), false); addCol(ctx, t, "l_suppkey", Type(RSQ_INT), false);
        addCol(ctx, t, "l_linenumber", Type(RSQ_INT), false);
        addCol(ctx, t, "l_quantity", dec0, true); addCol(ctx, t, "l_extendedprice", dec2, true);
        addCol(ctx, t, "l_discount", dec2, true); addCol(ctx, t, "l_tax", dec2, true);
        addCol(ctx, t, "l_returnflag", CH(1), true); addCol(ctx, t, "l_linestatus", CH(1), true);
        addCol(ctx, t, "l_shipdate", Type(RSQ_DATE), true);
        addCol(ctx, t, "l_commitdate", Type(RSQ_DATE), false); addCol(ctx, t, "l_receiptdate", Type(RSQ_DATE), false);
        addCol(ctx, t, "l_shipinstruct", CH(25), false); addCol(ctx, t, "l_shipmode", CH(10), false);
        addCol(ctx, t, "l_comment", VC(44), false);
        LineitemCols c{(int*)colPtr(t, "l_orderkey"), (i64*)colPtr(t, "l_quantity"), (i64*)colPtr(t, "l_extendedprice"),
                       (i64*)colPtr(t, "l_discount"), (i64*)colPtr(t, "l_tax"), (unsigned char*)colPtr(t, "l_returnflag"),
                       (unsigned char*)colPtr(t, "l_linestatus"), (unsigned*)colPtr(t, "l_shipdate")};
        u64 nPart = (u64)std::max<int64_t>(1, (int64_t)(200000.0 * sf + 0.5));
        hipLaunchKernelGGL(k_gen_lineitem, dim3(grid), dim3(256), 0, ctx.stream, c, makeGenTables(), (i64)row0, (i64)nRows, (u64)seed, nPart);
    } else if (kind == 1) {
        t.name = "orders";
        addCol(ctx, t, "o_orderkey", Type(RSQ_INT), true); addCol(ctx, t, "o_custkey", Type(RSQ_INT), true);
        addCol(ctx, t, "o_orderstatus", CH(1), false); addCol(ctx, t, "o_totalprice", dec2, false);
        addCol(ctx, t, "o_orderdate", Type(RSQ_DATE), true); addCol(ctx, t, "o_orderpriority", CH(15), false);
        addCol(ctx, t, "o_clerk", CH(15), false); addCol(ctx, t, "o_shippriority", Type(RSQ_INT), true);
        addCol(ctx, t, "o_comment", VC(79), false);
        u64 nCust = (u64)std::max<int64_t>(3, (int64_t)(150000.0 * sf + 0.5));
        hipLaunchKernelGGL(k_gen_orders, dim3(grid), dim3(256), 0, ctx.stream, (int*)colPtr(t, "o_orderkey"), (int*)colPtr(t, "o_custkey"),
                           (unsigned*)colPtr(t, "o_orderdate"), (int*)colPtr(t, "o_shippriority"), (i64)row0, (i64)nRows, (u64)seed, nCust);
    } else if (kind == 2) {
        t.name = "customer";
        addCol(ctx, t, "c_custkey", Type(RSQ_INT), true); addCol(ctx, t, "c_name", VC(25), false);
        addCol(ctx, t, "c_address", VC(40), false); addCol(ctx, t, "c_nationkey", Type(RSQ_INT), false);
        addCol(ctx, t, "c_phone", CH(15), false); addCol(ctx, t, "c_acctbal", dec2, false);
        addCol(ctx, t, "c_mktsegment", CH(10), true); addCol(ctx, t, "c_comment", VC(117), false);
        hipLaunchKernelGGL(k_gen_customer, dim3(grid), dim3(256), 0, ctx.stream, (int*)colPtr(t, "c_custkey"), (char*)colPtr(t, "c_mktsegment"),
                           (i64)row0, (i64)nRows, (u64)seed);
    } else if (kind == 3) {
        t.name = "t";
        for (const char* n : {"a", "b", "c", "d"}) addCol(ctx, t, n, Type(RSQ_BIGINT), true);
        hipLaunchKernelGGL(k_gen_synth, dim3(grid), dim3(256), 0, ctx.stream, (i64*)colPtr(t, "a"), (i64*)colPtr(t, "b"), (i64*)colPtr(t, "c"),
                           (i64*)colPtr(t, "d"), (i64)row0, (i64)nRows, (u64)seed, (u64)std::max<int64_t>(1, param));
    } else throw Error(RSQ_ERR_INVALID, "unknown table kind");
    RSQ_HIP(hipGetLastError());
    RSQ_HIP(hipStreamSynchronize(ctx.stream));
    computeColumnStats(ctx, t);
}

// ------------------------------------------------------------------------------------------------
// hash-table helpers
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };

__global__ void __launch_bounds__(256) k_fill_u64(u64* __restrict__ p, i64 n, u64 v) {
    // 16-byte stores where aligned (hipMalloc pointers + even offsets); tail with 8-byte stores
    i64 n2 = n >> 1;
    u64x2 vv; vv.x = v; vv.y = v;
    u64x2* p2 = reinterpret_cast<u64x2*>(p);
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n2; i += (i64)gridDim.x * blockDim.x) p2[i] = vv;
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) p[n - 1] = v;
}

// Several fills in ONE launch (blockIdx.y picks the item).  An execution clears a handful of small things — the error word,
// counters, the candidate selection's scratch, accumulator blocks — and a stream operation each costs ~5 us of device time
// whatever its size (TPC-H Q3 at SF10: 7 of them = 30 of 445 us).  Units are 4-byte words; the value is a 64-bit pattern.
#define FILL_BATCH_MAX 24
// (round 3) one flat grid: item i owns workgroups [first[i], first[i + 1]) — as many as its size deserves (one per 64 KB, at most
// 1024).  The earlier form gave EVERY item the grid of the largest one (blockIdx.y = item): with a 16 MB accumulator block beside
// a dozen 4-byte counters that was 24 000 workgroups of which 23 000 had nothing to do, ~25 us of a 300 us query.
struct FillBatchArgs { unsigned* p[FILL_BATCH_MAX]; u64 n32[FILL_BATCH_MAX]; u64 v[FILL_BATCH_MAX]; unsigned first[FILL_BATCH_MAX + 1]; int n; };
__global__ void __launch_bounds__(256) k_fill_batch(FillBatchArgs a) {
    int it = 0;
    while (it + 1 < a.n && blockIdx.x >= a.first[it + 1]) it++;
    unsigned* p = a.p[it];
    const u64 n = a.n32[it], v = a.v[it];
    const u64 gtid = (u64)(blockIdx.x - a.first[it]) * blockDim.x + threadIdx.x, gsz = (u64)(a.first[it + 1] - a.first[it]) * blockDim.x;
    if ((reinterpret_cast<unsigned long long>(p) & 15ull) != 0 || n < 256) {
        for (u64 i = gtid; i < n; i += gsz) p[i] = (unsigned)((i & 1) ? (v >> 32) : v);
        return;
    }
    u64x2 vv; vv.x = v; vv.y = v;
    u64x2* p4 = reinterpret_cast<u64x2*>(p);
    const u64 n4 = n >> 2;
    for (u64 i = gtid; i < n4; i += gsz) p4[i] = vv;
    if (gtid == 0) for (u64 i = n4 << 2; i < n; i++) p[i] = (unsigned)((i & 1) ? (v >> 32) : v);
}

void fillBatchAsync(Context& ctx, const FillItem* items, int count) {
    for (int base = 0; base < count; base += FILL_BATCH_MAX) {
        FillBatchArgs a;
        memset(&a, 0, sizeof a);
        const int n = std::min(FILL_BATCH_MAX, count - base);
        int k = 0;
        unsigned blocks = 0;
        for (int i = 0; i < n; i++) {
            const FillItem& f = items[base + i];
            if (!f.p || f.bytes == 0) continue;
            if ((f.bytes & 3) || ((uintptr_t)f.p & 3)) throw Error(RSQ_ERR_DEVICE, "fillBatchAsync: a fill must cover whole 4-byte words");
            a.p[k] = (unsigned*)f.p; a.n32[k] = f.bytes / 4; a.v[k] = f.value;
            a.first[k] = blocks;
            blocks += (unsigned)std::max<u64>(1, std::min<u64>(1024, (f.bytes + 65535) / 65536));
            k++;
        }
        if (k == 0) continue;
        a.first[k] = blocks; a.n = k;
        hipLaunchKernelGGL(k_fill_batch, dim3(blocks), dim3(256), 0, ctx.stream, a);
        RSQ_HIP(hipGetLastError());
    }
}

__global__ void k_publish_status(u64* __restrict__ host, const unsigned* __restrict__ err, const unsigned* __restrict__ groupCount,
                                 const unsigned* __restrict__ candCount, const u64* __restrict__ pipeStats, int nPipelines,
                                 const i64* __restrict__ rows, int rowWords, unsigned maxInline, i64* __restrict__ hostRows, const u64* __restrict__ word3,
                                 const u64* __restrict__ copySrc, u64* __restrict__ copyDst, unsigned copyWords) {
    const int t = threadIdx.x;
    if (t == 3 && word3) host[3] = *word3;
    for (unsigned i = (unsigned)t; i < copyWords; i += blockDim.x) copyDst[i] = copySrc[i];      // (a small dense aggregate table: no read-back copy of its own)
    // a handful of group rows travel with the status words (TPC-H Q5: five, Q12: two): the host then has them when the stream reports
    // completion, instead of asking for them with a blocking copy afterwards (a blit kernel and another round trip, ~20 us)
    if (rows && groupCount) {
        const unsigned n = *groupCount;
        if (n <= maxInline) for (unsigned i = (unsigned)t; i < n * (unsigned)rowWords; i += blockDim.x) hostRows[i] = rows[i];
    }
    if (t == 0) host[0] = (u64)*err;
    if (t == 1 && groupCount) host[1] = (u64)*groupCount;
    if (t == 2 && candCount) host[2] = (u64)*candCount;
    if (pipeStats && t >= 8 && t < 8 + nPipelines) host[t] = pipeStats[t - 8];
}
void publishStatusAsync(Context& ctx, uint64_t* hostWords, const uint32_t* err, const uint32_t* groupCount, const uint32_t* candCount,
                        const uint64_t* pipeStats, int nPipelines, const int64_t* rows, int rowWords, uint32_t maxInline, int64_t* hostRows, const uint64_t* word3,
                        const uint64_t* copySrc, uint64_t* copyDst, uint32_t copyWords) {
    if (nPipelines > 56) throw Error(RSQ_ERR_UNSUPPORTED, "more than 56 pipelines in one query");
    hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(64), 0, ctx.stream, (u64*)hostWords, (const unsigned*)err, (const unsigned*)groupCount,
                       (const unsigned*)candCount, (const u64*)pipeStats, nPipelines, (const i64*)rows, rowWords, (unsigned)maxInline, (i64*)hostRows, (const u64*)word3, (const u64*)copySrc, (u64*)copyDst, (unsigned)copyWords);
    RSQ_HIP(hipGetLastError());
}

// One launch that readies a join table for its build: the fill of the key / state words, the clear of the key bitmap and of
// the entry counter.  (Three stream operations before every build add up: TPC-H Q5 runs five builds over tiny tables and
// spent a third of its 0.3 ms on clears and launches.)
__global__ void __launch_bounds__(256) k_prepare_table(u64* __restrict__ fill, i64 nFill, u64 fillValue, unsigned* __restrict__ zeroA, i64 nZeroA,
                                                       unsigned* __restrict__ zeroB, i64 nZeroB, unsigned* count) {
    const i64 tid = blockIdx.x * (i64)blockDim.x + threadIdx.x, stride = (i64)gridDim.x * blockDim.x;
    {   // 16-byte stores where the fill is long and aligned
        u64x2 vv; vv.x = fillValue; vv.y = fillValue;
        u64x2* f2 = reinterpret_cast<u64x2*>(fill);
        const i64 n2 = nFill >> 1;
        for (i64 i = tid; i < n2; i += stride) f2[i] = vv;
        if (tid == 0 && (nFill & 1)) fill[nFill - 1] = fillValue;
    }
    for (i64 i = tid; i < nZeroA; i += stride) zeroA[i] = 0u;
    for (i64 i = tid; i < nZeroB; i += stride) zeroB[i] = 0u;
    if (tid == 0 && count) *count = 0u;
}

void prepareTableAsync(Context& ctx, uint64_t* fill, size_t nFill, uint64_t fillValue, uint32_t* zeroA, size_t nZeroA, uint32_t* zeroB, size_t nZeroB,
                       uint32_t* count) {
    if (nFill && ((uintptr_t)fill & 15) != 0) throw Error(RSQ_ERR_DEVICE, "prepareTableAsync: unaligned destination");
    const size_t work = std::max<size_t>(std::max<size_t>(nFill / 2, nZeroA), nZeroB);
    unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(2048, (work + 255) / 256));
    hipLaunchKernelGGL(k_prepare_table, dim3(grid), dim3(256), 0, ctx.stream, (u64*)fill, (i64)nFill, (u64)fillValue, (unsigned*)zeroA, (i64)nZeroA,
                       (unsigned*)zeroB, (i64)nZeroB, (unsigned*)count);
    RSQ_HIP(hipGetLastError());
}

void fillU64Async(Context& ctx, uint64_t* dptr, size_t n, uint64_t value) {
    if (n == 0) return;
    if (value == 0) { RSQ_HIP(hipMemsetAsync(dptr, 0, n * 8, ctx.stream)); return; }
    if (((uintptr_t)dptr & 15) != 0) throw Error(RSQ_ERR_DEVICE, "fillU64Async: unaligned destination");
    unsigned grid = (unsigned)std::min<size_t>(2048, (n / 2 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_fill_u64, dim3(grid), dim3(256), 0, ctx.stream, (u64*)dptr, (i64)n, (u64)value);
    RSQ_HIP(hipGetLastError());
}

// Each workgroup owns chunks of 256 x COMPACT_PER_THREAD consecutive slots.  All first-row words of a thread are loaded up front
// (independent, coalesced across the workgroup), the occupied ones are counted, an LDS scan gives every thread its
// offset and ONE global atomic per workgroup and chunk reserves the output rows: returning atomics on a single word
// serialise at ~11 ns each on MI355X (MI355X_MICROARCH.md, "fanin"), so one per occupied slot — or one per wave and
// round, 54 K of them for TPC-H Q3 at SF10 — cost 620 us.  The table's first-row block is read once (the earlier
// count-then-write form read it twice with 32 dependent rounds per thread: 73 us for 4 M slots; this form: 19 us).
// Rows beyond `maxRows` are counted but not written (the host re-runs with a larger buffer).
// order-preserving unsigned image of a sort key ("earlier in the requested order" = "larger"), see the top-k section below
__device__ __forceinline__ u64 topk_image(i64 w, int is32, int desc) {
    const i64 v = is32 ? (i64)(int)(unsigned)w : w;
    const u64 u = (u64)v ^ 0x8000000000000000ull;
    return desc ? u : ~u;
}

// inverse of rsq::rank_mix (kernels/rsq_device.h — keep the two in lock step): the Feistel rounds backwards
__device__ inline u64 rank_mix_round(u64 v, u64 key) {
    v = (v + key) * 0x9E3779B97F4A7C15ull; v ^= v >> 29; v *= 0xBF58476D1CE4E5B9ull; v ^= v >> 32;
    return v;
}
__device__ inline u64 rank_unmix(u64 x, u64 cap) {
    const int k = 63 - __builtin_clzll(cap), lo = k - (k >> 1), hi = k >> 1;
    const u64 mlo = (1ull << lo) - 1, mhi = (1ull << hi) - 1;
    u64 L = (x >> lo) & mhi, R = x & mlo;
    L ^= rank_mix_round(R, 3) & mhi;
    R ^= rank_mix_round(L, 2) & mlo;
    L ^= rank_mix_round(R, 1) & mhi;
    return (L << lo) | R;
}

// where the words of a table entry live (the packed group row [first row | table words | accumulator blocks] read in place)
struct EntrySource {
    const i64* first; i64 cap; const i64* words; int nWords; int wordsAos; const i64* acc; int nAcc; int unmix;
    const int* deref;        // per table word of the ROW: 0 = the word itself, else engine.h entryDerefCode / entryPlainCode (group values kept by address)
    int tabStride;           // words between two entries of `words` (the row's nWords unless the entries keep one word per carried value)
};
// Table word w of entry e.  A hash aggregation whose string group values are functions of its key (TPC-H Q10: name, address, phone,
// comment, nation behind c_custkey) keeps their ADDRESS in the value's first word instead of copying 31 words into every new group
// (codegen_agg.cpp); whoever makes group rows for the host rebuilds the words here, for the rows that are really delivered.
__device__ __forceinline__ i64 table_word(const i64* __restrict__ words, int stride, int aos, i64 cap, i64 e, int w, const int* __restrict__ deref) {
    const int d = deref ? deref[w] : 0;
    if (d == 0) return aos ? words[(size_t)e * stride + w] : words[(size_t)w * cap + e];
    const int src = (d >> 16) & 0xff, off = (d >> 4) & 0xfff, len = d & 0xf;
    const i64 sw = aos ? words[(size_t)e * stride + src] : words[(size_t)src * cap + e];
    if (!(d & 0x40000000)) return sw;          // (entryPlainCode: the value itself, at another word number)
    const char* p = reinterpret_cast<const char*>((uintptr_t)sw) + off;
    u64 v = 0;
    if (len == 8) __builtin_memcpy(&v, p, 8);
    else for (int i = 0; i < len; i++) v |= (u64)(unsigned char)p[i] << (8 * i);
    return (i64)v;
}
__device__ __forceinline__ i64 entry_word(const EntrySource& es, i64 s, i64 e, int k) {
    if (k == 0) return es.first[s];
    if (k - 1 < es.nWords) return !es.words ? s : table_word(es.words, es.tabStride, es.wordsAos, es.cap, e, k - 1, es.deref);
    return es.acc[(size_t)(k - 1 - es.nWords) * es.cap + s];
}

template <int COMPACT_PER_THREAD>
__global__ void __launch_bounds__(256) k_compact_entries(const i64* __restrict__ first, i64 cap, const i64* __restrict__ words, int nWords,
                                                         int wordsAos, const i64* __restrict__ acc, int nAcc, i64* __restrict__ out,
                                                         unsigned maxRows, unsigned* count, int unmix, int keyWord, int keyIs32, int keyDesc,
                                                         u64* __restrict__ imageRange, u64* __restrict__ chain, unsigned launchNo, int narrow,
                                                         const int* __restrict__ deref, int tabStride) {
    const int stride = 1 + nWords + nAcc;
    u64 imgMax = 0, imgMaxInv = 0;      // range of the sort-key images of the rows written (keyWord >= 0): max(u), max(~u)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __shared__ unsigned s_wave[4];
    __shared__ unsigned s_base;
    __shared__ unsigned s_list[256 * COMPACT_PER_THREAD];      // chunk-relative indices of the occupied slots
    const i64 chunkSlots = 256 * COMPACT_PER_THREAD;
    for (i64 lo = (i64)blockIdx.x * chunkSlots; lo < cap; lo += (i64)gridDim.x * chunkSlots) {
        i64 f[COMPACT_PER_THREAD];
        unsigned mine = 0;
#pragma unroll
        for (int r = 0; r < COMPACT_PER_THREAD; r++) {
            const i64 s = lo + r * 256 + t;
            f[r] = s < cap ? first[s] : 0x7fffffffffffffffll;
            mine += f[r] != 0x7fffffffffffffffll ? 1u : 0u;
        }
        // exclusive prefix of `mine` over the workgroup: wave scan (shuffles), then the four wave totals
        unsigned incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned v = (unsigned)__shfl_up((int)incl, d, 64); if (lane >= d) incl += v; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const unsigned c = s_wave[w]; if (w < wave) before += c; total += c; }
        // where the chunk's rows go.  With a chain (RSQ_COMPACT_CHAINED=1, one chunk per workgroup): the rows of the chunks in front, by
        // decoupled look-back (a wave reads 64 predecessors per round trip; see k_rank_blocks_chained); the rows then come out in slot
        // order, whatever the workgroups' timing.  Default: one returning atomic per chunk on the row counter.
        if (chain) {
            if (t < 64) {
                const u64 tag = ((u64)(launchNo & 0x3fffffffu)) << 2;
                auto stateOf = [&](u64 v) -> unsigned { return ((v >> 2) & 0x3fffffffull) == (u64)(launchNo & 0x3fffffffu) ? (unsigned)(v & 3ull) : 0u; };
                const int ln = t;
                const i64 me = lo / chunkSlots;
                unsigned base = 0;
                if (me == 0) { if (ln == 0) __hip_atomic_store(&chain[0], ((u64)total << 32) | tag | 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else {
                    if (ln == 0) __hip_atomic_store(&chain[me], ((u64)total << 32) | tag | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const long long t0 = wall_clock64();
                    i64 hi = me - 1;
                    for (;;) {
                        const i64 idx = hi - ln;
                        const u64 v = idx >= 0 ? __hip_atomic_load(&chain[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tag | 2ull);      // (in front of chunk 0: nothing, inclusive)
                        const unsigned st = stateOf(v);
                        const u64 ready = __ballot(st != 0u), incl = __ballot(st == 2u);
                        unsigned take = 0; bool done = false, moved = false;
                        if (incl) {
                            const int f = __ffsll((long long)incl) - 1;
                            const u64 below = (1ull << f) - 1ull;
                            if ((ready & below) == below) { take = ln <= f ? (unsigned)(v >> 32) : 0u; done = true; }
                        } else if (ready == ~0ull) { take = (unsigned)(v >> 32); moved = true; }
                        if (done || moved) {
#pragma unroll
                            for (int m = 32; m >= 1; m >>= 1) take += (unsigned)__shfl_xor((int)take, m, 64);
                            base += take;
                            if (done) break;
                            hi -= 64;
                            continue;
                        }
                        // (not reachable - workgroups start in index order; the rows would overlap, so the count is made to say so: the host fails the execution)
                        if (wall_clock64() - t0 > 2000000ll) { base = 0; if (ln == 0) atomicMax(count, 0xffffffffu); break; }      // 20 ms
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (ln == 0) __hip_atomic_store(&chain[me], ((u64)(base + total) << 32) | tag | 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (ln == 0) {
                    s_base = base;
                    if (lo + chunkSlots >= cap) atomicMax(count, base + total);          // the last chunk knows the number of rows (0xffffffff: a look-back gave up)
                }
            }
        } else
        if (t == 0) s_base = total ? atomicAdd(count, total) : 0u;
        __syncthreads();
        if (total) {
            // the occupied slots are sparse (a few per cent): gather them into an LDS list first so that the row copies run
            // with every lane busy — copying straight from the per-thread slots executed the (divergent) copy body once per
            // slot position of the wave, 13 of 16 times for Q3, each a chain of dependent loads and stores: 42 us
            unsigned lp = before + incl - mine;
#pragma unroll
            for (int r = 0; r < COMPACT_PER_THREAD; r++)
                if (f[r] != 0x7fffffffffffffffll) s_list[lp++] = (unsigned)(r * 256 + t);
            __syncthreads();
            const unsigned base = s_base;
            if (narrow) {
                // rows [slot | sort key] only: the candidate selection reads the keys and fetches the few rows it takes from the table itself
                // (k_topk_range_select).  TPC-H Q10 at SF10: 380 K groups of 40 words, 20 of them wanted - 108 us of row copies.
                const EntrySource es{first, cap, words, nWords, wordsAos, acc, nAcc, unmix, deref, tabStride};
                for (unsigned i = (unsigned)t; i < total; i += 256u) {
                    const unsigned pos = base + i;
                    if (pos >= maxRows) continue;
                    const i64 s = lo + (i64)s_list[i];
                    const i64 e = unmix ? (i64)rank_unmix((u64)s, (u64)cap) : s;
                    const i64 kv = entry_word(es, s, e, keyWord);
                    out[(size_t)pos * 2] = s;
                    out[(size_t)pos * 2 + 1] = kv;
                    const u64 u = topk_image(kv, keyIs32, keyDesc); imgMax = u > imgMax ? u : imgMax; imgMaxInv = ~u > imgMaxInv ? ~u : imgMaxInv;
                }
            } else
            if (stride <= 8) {
                // three rows per thread at a time, all their loads first, then their stores.  (Word by word, every load waited for
                // its predecessor's store, and the sort key's image was read back from the row just written: chains of up to ten
                // round trips per row - 18 us for TPC-H Q3's 114 K groups.)
                for (unsigned i0 = (unsigned)t; i0 < total; i0 += 3u * 256u) {
                    i64 v[3][8];
                    bool ok[3];
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const unsigned i = i0 + (unsigned)r * 256u;
                        ok[r] = i < total && base + i < maxRows;
                        const i64 s = lo + (i64)s_list[ok[r] ? i : 0u];
                        // (no word arrays: the one word is the slot index itself — dense aggregate tables, whose slot IS the group id)
                        // (rank dictionaries keep the aggregates of entry r at rank_mix(r), kernels/rsq_device.h: the entry of
                        // accumulator slot s is the inverse)
                        i64 e = s;
                        if (unmix) e = (i64)rank_unmix((u64)s, (u64)cap);
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            v[r][k] = 0;
                            if (!ok[r]) continue;
                            if (k == 0) v[r][k] = first[s];
                            else if (k - 1 < nWords) v[r][k] = !words ? s : table_word(words, tabStride, wordsAos, cap, e, k - 1, deref);
                            else if (k < stride) v[r][k] = acc[(size_t)(k - 1 - nWords) * cap + s];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        if (!ok[r]) continue;
                        i64* o = out + (size_t)(base + i0 + (unsigned)r * 256u) * stride;
                        i64 kv = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) { if (k < stride) o[k] = v[r][k]; if (k == keyWord) kv = v[r][k]; }
                        if (keyWord >= 0) { const u64 u = topk_image(kv, keyIs32, keyDesc); imgMax = u > imgMax ? u : imgMax; imgMaxInv = ~u > imgMaxInv ? ~u : imgMaxInv; }
                    }
                }
            } else {
                // wide rows (string group values: TPC-H Q10's 380 K groups of 40 words): EIGHT LANES per row, lane j of the eight copies words
                // j, j + 8, ... - four of them at a time, their loads first.  A load instruction of the wave then covers 8 rows x 64
                // contiguous bytes; one thread per row asked for 8 bytes at a stride of 320 (64 memory lines per instruction: 133 us at SF10).
                const int sub = t & 7;
                for (unsigned i0 = (unsigned)t >> 3; i0 < total; i0 += 64u) {          // two rows per eight lanes at a time
                    i64 s2[2], e2[2];
                    bool ok[2];
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const unsigned i = i0 + 32u * (unsigned)r;
                        ok[r] = i < total && base + i < maxRows;
                        s2[r] = lo + (i64)s_list[ok[r] ? i : 0u];
                        e2[r] = unmix ? (i64)rank_unmix((u64)s2[r], (u64)cap) : s2[r];
                    }
                    for (int k0 = 0; k0 < stride; k0 += 32) {
                        i64 v[2][4];
#pragma unroll
                        for (int r = 0; r < 2; r++)
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int k = k0 + j * 8 + sub;
                                v[r][j] = 0;
                                if (k >= stride || !ok[r]) continue;
                                if (k == 0) v[r][j] = first[s2[r]];
                                else if (k - 1 < nWords) v[r][j] = !words ? s2[r] : table_word(words, tabStride, wordsAos, cap, e2[r], k - 1, deref);
                                else v[r][j] = acc[(size_t)(k - 1 - nWords) * cap + s2[r]];
                            }
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            i64* o = out + (size_t)(base + i0 + 32u * (unsigned)r) * stride;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int k = k0 + j * 8 + sub;
                                if (k >= stride || !ok[r]) continue;
                                o[k] = v[r][j];
                                if (k == keyWord) { const u64 u = topk_image(v[r][j], keyIs32, keyDesc); imgMax = u > imgMax ? u : imgMax; imgMaxInv = ~u > imgMaxInv ? ~u : imgMaxInv; }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();          // s_wave / s_base are rewritten by the next chunk
    }
    if (keyWord >= 0) {          // one pair of atomics per workgroup (atomics on one word serialise)
        __shared__ u64 s_img[8];
        for (int m = 32; m >= 1; m >>= 1) {
            const u64 a = (u64)__shfl_xor((long long)imgMax, m, 64), b = (u64)__shfl_xor((long long)imgMaxInv, m, 64);
            imgMax = a > imgMax ? a : imgMax; imgMaxInv = b > imgMaxInv ? b : imgMaxInv;
        }
        if (lane == 0) { s_img[wave] = imgMax; s_img[4 + wave] = imgMaxInv; }
        __syncthreads();
        if (t == 0) {
            u64 a = 0, b = 0;
            for (int w = 0; w < 4; w++) { a = s_img[w] > a ? s_img[w] : a; b = s_img[4 + w] > b ? s_img[4 + w] : b; }
            if (a | b) { atomicMax(&imageRange[0], a); atomicMax(&imageRange[1], b); }
        }
    }
}

void compactEntries(Context& ctx, const int64_t* firstRow, int64_t capacity, const int64_t* words, int nWords, bool wordsAos,
                    const int64_t* acc, int nAcc, int64_t* outRows, uint32_t maxRows, uint32_t* count, bool unmix, int keyWord, bool keyIs32,
                    bool keyDesc, uint64_t* imageRange, bool narrow, const int* deref, int tabStride) {
    if (tabStride <= 0) tabStride = nWords;
    if (narrow && (!imageRange || keyWord < 0)) throw Error(RSQ_ERR_DEVICE, "compactEntries: narrow rows carry the sort key");
    // slots per thread: every chunk costs one reservation atomic on the same word (they serialise), so large tables take
    // large chunks; swept on the box for a 4 M-slot table: 16 -> 26 us, 32 -> 22 us, 64 -> 19 us.  (Row positions by look-back through
    // a chain of chunk totals instead of the reservation atomics were tried and measured no gain - TPC-H Q3 at SF10 0.305 against
    // 0.300 ms, Q10 0.939 either way: the 177 / 256 same-word atomics are not what these 19 / 119 us kernels wait for.)
    // (wide rows - more than eight words - are copied by eight lanes each: their kernel wants waves, not long chunks; TPC-H Q10 at SF10,
    // 380 K rows of 40 words out of 4 M slots: 133 us with one thread per row and 64 slots per thread - one workgroup per CU)
    const int perThread = 1 + nWords + nAcc > 8 && !narrow ? 16 : capacity >= (1 << 22) ? 64 : capacity >= (1 << 20) ? 32 : 16;
    const int64_t chunkSlots = 256 * (int64_t)perThread;
    const int64_t nChunks = std::max<int64_t>(1, (capacity + chunkSlots - 1) / chunkSlots);
    unsigned grid = (unsigned)std::min<int64_t>(8 * (int64_t)ctx.numCUs, nChunks);
    u64* chain = nullptr;
    unsigned launchNo = 0;
#define RSQ_LAUNCH_COMPACT(PT) hipLaunchKernelGGL(k_compact_entries<PT>, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)firstRow, (i64)capacity, \
                       (const i64*)words, nWords, wordsAos ? 1 : 0, (const i64*)acc, nAcc, (i64*)outRows, (unsigned)maxRows, count, unmix ? 1 : 0, imageRange ? keyWord : -1, keyIs32 ? 1 : 0, keyDesc ? 1 : 0, (u64*)imageRange, chain, launchNo, narrow ? 1 : 0, deref, tabStride)
    if (perThread >= 64) RSQ_LAUNCH_COMPACT(64); else if (perThread >= 32) RSQ_LAUNCH_COMPACT(32); else RSQ_LAUNCH_COMPACT(16);
#undef RSQ_LAUNCH_COMPACT
    RSQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// ORDER BY ... LIMIT k over many groups: pre-selection of the candidate rows on the device.
// The first sort key of every group row is mapped to an unsigned image in which "earlier in the requested order" is
// "larger"; an MSB-first radix select (11-bit digits, one histogram launch per digit, no host round trip: every
// workgroup re-derives the prefix chosen so far from the previous histograms, 2048 bins each) finds T, the `want`-th
// largest image; the gather writes the rows with image >= T.  The rows that lead the full lexicographic order are among
// them, so the host tail sorts a few dozen rows instead of every group.
// ------------------------------------------------------------------------------------------------
enum { TOPK_PASSES = 6, TOPK_BINS = 2048 };
__device__ __forceinline__ int topk_shift(int p) { return p < 5 ? 53 - 11 * p : 0; }
__device__ __forceinline__ int topk_bits(int p) { return p < 5 ? 11 : 9; }

// Replays the digit choices of passes [0, upto): returns the chosen prefix (the top bits of T, right-aligned) and, in
// *remOut, how many rows with exactly that prefix are still wanted.  *allOut is set when fewer than `want` rows exist
// (then every row qualifies).  Called by all 256 threads of a workgroup.
__device__ u64 topk_prefix(const unsigned* __restrict__ hists, int upto, unsigned want, unsigned* remOut, int* allOut) {
    __shared__ unsigned s_above[256];
    __shared__ unsigned s_digit, s_rem;
    __shared__ int s_found;
    const int t = threadIdx.x;
    u64 prefix = 0;
    unsigned rem = want;
    int all = 0;
    for (int p = 0; p < upto && !all; p++) {
        const unsigned* h = hists + (size_t)p * TOPK_BINS;
        unsigned c[8], local = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) { c[b] = h[t * 8 + b]; local += c[b]; }
        if (t == 0) s_found = 0;
        s_above[t] = local;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {                  // inclusive suffix sums over the threads
            const unsigned v = t + d < 256 ? s_above[t + d] : 0u;
            __syncthreads();
            s_above[t] += v;
            __syncthreads();
        }
        unsigned running = s_above[t] - local;               // rows in bins above this thread's eight
#pragma unroll
        for (int b = 7; b >= 0; b--) {
            if (running < rem && rem <= running + c[b]) { s_digit = (unsigned)(t * 8 + b); s_rem = rem - running; s_found = 1; }
            running += c[b];
        }
        __syncthreads();
        if (!s_found) all = 1;
        else { prefix = (prefix << topk_bits(p)) | (u64)s_digit; rem = s_rem; }
        __syncthreads();
    }
    *remOut = rem; *allOut = all;
    return prefix;
}

// pass p: histogram of digit p over the rows whose higher digits equal the prefix chosen so far.  Pass 0 also
// extracts the key images from the group rows.
__global__ void __launch_bounds__(256) k_topk_hist(const i64* __restrict__ rows, int stride, int keyWord, int is32, int desc,
                                                   const unsigned* __restrict__ nRows, unsigned maxRows, u64* __restrict__ images,
                                                   unsigned* __restrict__ hists, int pass, unsigned want) {
    __shared__ unsigned s_hist[TOPK_BINS];
    for (int b = threadIdx.x; b < TOPK_BINS; b += 256) s_hist[b] = 0;
    unsigned rem; int all;
    const u64 prefix = topk_prefix(hists, pass, want, &rem, &all);      // contains the barriers that publish s_hist = 0
    if (all) return;
    const unsigned n = *nRows < maxRows ? *nRows : maxRows;
    const int shift = topk_shift(pass);
    const unsigned mask = (1u << topk_bits(pass)) - 1u;
    const int above = shift + topk_bits(pass);                          // bits above this digit
    __syncthreads();
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        u64 u;
        if (pass == 0) { u = topk_image(rows[(size_t)i * stride + keyWord], is32, desc); images[i] = u; }
        else u = images[i];
        if (pass == 0 || (u >> above) == prefix) atomicAdd(&s_hist[(unsigned)(u >> shift) & mask], 1u);
    }
    __syncthreads();
    unsigned* h = hists + (size_t)pass * TOPK_BINS;
    for (int b = threadIdx.x; b < TOPK_BINS; b += 256) { const unsigned c = s_hist[b]; if (c) atomicAdd(&h[b], c); }
}

// rows whose image is >= T, in no particular order; *candCount counts all of them, rows beyond `capacity` are dropped
// (the host then falls back to reading every group)
__global__ void __launch_bounds__(256) k_topk_gather(const i64* __restrict__ rows, int stride, const unsigned* __restrict__ nRows, unsigned maxRows,
                                                     const u64* __restrict__ images, const unsigned* __restrict__ hists, unsigned want,
                                                     i64* __restrict__ cand, unsigned capacity, unsigned* candCount) {
    unsigned rem; int all;
    const u64 T = topk_prefix(hists, TOPK_PASSES, want, &rem, &all);
    const unsigned n = *nRows < maxRows ? *nRows : maxRows;
    const int lane = threadIdx.x & 63;
    const unsigned rounds = (n + gridDim.x * 256u - 1) / (gridDim.x * 256u);
    for (unsigned r = 0; r < rounds; r++) {
        const unsigned i = (r * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
        const bool take = i < n && (all || images[i] >= T);
        const unsigned long long vote = __ballot(take);
        if (vote == 0) continue;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(candCount, (unsigned)__popcll(vote));
        base = (unsigned)__shfl((int)base, 0, 64);
        if (!take) continue;
        const unsigned pos = base + (unsigned)__popcll(vote & ((1ull << lane) - 1ull));
        if (pos >= capacity) continue;
        const i64* src = rows + (size_t)i * stride;
        i64* dst = cand + (size_t)pos * stride;
        for (int w = 0; w < stride; w++) dst[w] = src[w];
    }
}

// ---- the short form: ONE histogram over the images' actual range ------------------------------------------------
// With the smallest and largest image known (the compaction kernel collects them while it writes the rows), the images are
// stretched to that range, so the very first 11-bit digit already tells rows apart: the bin that holds the `want`-th
// largest image and the bins above it together hold the candidates — a few dozen rows for TPC-H Q3's 114 K groups — and the
// remaining five digit passes are not needed: two launches instead of seven.  The candidates are a superset of the exact
// selection's (every row at or above the `want`-th largest image is among them); should they not fit the candidate buffer,
// the host runs the exact selection.
__device__ __forceinline__ unsigned topk_range_digit(u64 u, u64 lo, int shift) { return (unsigned)(((u - lo) << shift) >> 53); }

__global__ void __launch_bounds__(256) k_topk_range_hist(const i64* __restrict__ rows, int stride, int keyWord, int is32, int desc,
                                                         const unsigned* __restrict__ nRows, unsigned maxRows, const u64* __restrict__ imageRange,
                                                         unsigned* __restrict__ hist) {
    __shared__ unsigned s_hist[TOPK_BINS];
    for (int b = threadIdx.x; b < TOPK_BINS; b += 256) s_hist[b] = 0;
    __syncthreads();
    const u64 hi = imageRange[0], lo = ~imageRange[1];
    const int shift = hi > lo ? __builtin_clzll(hi - lo) : 0;
    const unsigned n = *nRows < maxRows ? *nRows : maxRows;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        atomicAdd(&s_hist[topk_range_digit(topk_image(rows[(size_t)i * stride + keyWord], is32, desc), lo, shift)], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < TOPK_BINS; b += 256) { const unsigned c = s_hist[b]; if (c) atomicAdd(&hist[b], c); }
}

__global__ void __launch_bounds__(256) k_topk_range_gather(const i64* __restrict__ rows, int stride, int keyWord, int is32, int desc,
                                                           const unsigned* __restrict__ nRows, unsigned maxRows, const u64* __restrict__ imageRange,
                                                           const unsigned* __restrict__ hist, unsigned want, i64* __restrict__ cand, unsigned capacity,
                                                           unsigned* candCount) {
    // the lowest bin that still belongs to the candidates: the highest b with (rows in bins >= b) >= want
    __shared__ unsigned s_above[256];
    __shared__ unsigned s_bin;
    const int t = threadIdx.x;
    unsigned c[8], local = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) { c[b] = hist[t * 8 + b]; local += c[b]; }
    if (t == 0) s_bin = 0;                                  // fewer than `want` rows in all: every row qualifies
    s_above[t] = local;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                     // inclusive suffix sums over the threads
        const unsigned v = t + d < 256 ? s_above[t + d] : 0u;
        __syncthreads();
        s_above[t] += v;
        __syncthreads();
    }
    unsigned running = s_above[t] - local;                  // rows in bins above this thread's eight
#pragma unroll
    for (int b = 7; b >= 0; b--) {
        if (running < want && want <= running + c[b]) s_bin = (unsigned)(t * 8 + b);
        running += c[b];
    }
    __syncthreads();
    const unsigned bin = s_bin;
    const u64 hi = imageRange[0], lo = ~imageRange[1];
    const int shift = hi > lo ? __builtin_clzll(hi - lo) : 0;
    const unsigned n = *nRows < maxRows ? *nRows : maxRows;
    const int lane = threadIdx.x & 63;
    const unsigned rounds = (n + gridDim.x * 256u - 1) / (gridDim.x * 256u);
    for (unsigned r = 0; r < rounds; r++) {
        const unsigned i = (r * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
        const bool take = i < n && topk_range_digit(topk_image(rows[(size_t)i * stride + keyWord], is32, desc), lo, shift) >= bin;
        const unsigned long long vote = __ballot(take);
        if (vote == 0) continue;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(candCount, (unsigned)__popcll(vote));
        base = (unsigned)__shfl((int)base, 0, 64);
        if (!take) continue;
        const unsigned pos = base + (unsigned)__popcll(vote & ((1ull << lane) - 1ull));
        if (pos >= capacity) continue;
        const i64* src = rows + (size_t)i * stride;
        i64* dst = cand + (size_t)pos * stride;
        for (int w = 0; w < stride; w++) dst[w] = src[w];
    }
}

// ---- the short form in ONE launch, answer delivered ------------------------------------------------------------------
// histogram -> [all workgroups have added theirs] -> gather -> [all workgroups have gathered] -> status words.  The grid is at most
// one 256-thread workgroup per CU with 8 KB of LDS, so every workgroup is on the chip at once and the two meeting points are a
// counter in device memory the workgroups watch (agent-scope loads; the histogram itself is summed with device-scope atomics, which
// execute at the memory side, and is read back with agent-scope loads).  The candidates go straight into host-mapped pinned memory,
// and the last workgroup publishes the execution's status words there too: this launch replaces the histogram, the gather, the
// status kernel and the device-to-host copy of the candidates.  A workgroup that waits longer than 1 ms gives up and sets bit 256 of
// the error word (the engine then repeats the execution with the separate launches) - every wave has its way out.
__global__ void __launch_bounds__(256) k_topk_range_select(const i64* __restrict__ rows, int stride, int keyWord, int is32, int desc,
                                                           const unsigned* __restrict__ nRows, unsigned maxRows, const u64* __restrict__ imageRange,
                                                           unsigned* __restrict__ hist, unsigned want, i64* __restrict__ cand, unsigned capacity,
                                                           unsigned* candCount, unsigned* ticket1, unsigned* ticket2, unsigned* err, u64* __restrict__ host, u64 seq,
                                                           const unsigned* __restrict__ groupCount, const u64* __restrict__ pipeStats, int nPipelines,
                                                           EntrySource es, int fullStride) {
    __shared__ unsigned s_hist[TOPK_BINS];
    __shared__ unsigned s_above[256];
    __shared__ unsigned s_bin, s_flag;
    const int t = threadIdx.x;
    for (int b = t; b < TOPK_BINS; b += 256) s_hist[b] = 0;
    __syncthreads();
    const u64 hi = imageRange[0], lo = ~imageRange[1];
    const int shift = hi > lo ? __builtin_clzll(hi - lo) : 0;
    const unsigned n = *nRows < maxRows ? *nRows : maxRows;
    // a thread's rows are (round * gridDim.x + blockIdx.x) * 256 + t; the digits of its first RSEL_KEEP rounds stay in registers for
    // the gather (their loads are issued together: one round trip instead of one per row - a loop of load, wait, LDS atomic took
    // 2 us per round)
    enum { RSEL_KEEP = 8 };
    const unsigned rounds = (n + gridDim.x * 256u - 1) / (gridDim.x * 256u);
    unsigned dig[RSEL_KEEP];
    {
        i64 key[RSEL_KEEP];
#pragma unroll
        for (int r = 0; r < RSEL_KEEP; r++) {
            const unsigned i = ((unsigned)r * gridDim.x + blockIdx.x) * 256u + t;
            key[r] = (unsigned)r < rounds && i < n ? rows[(size_t)i * stride + keyWord] : 0;
        }
#pragma unroll
        for (int r = 0; r < RSEL_KEEP; r++) {
            const unsigned i = ((unsigned)r * gridDim.x + blockIdx.x) * 256u + t;
            dig[r] = 0xffffffffu;                         // no row
            if ((unsigned)r < rounds && i < n) { dig[r] = topk_range_digit(topk_image(key[r], is32, desc), lo, shift); atomicAdd(&s_hist[dig[r]], 1u); }
        }
    }
    for (unsigned r = RSEL_KEEP; r < rounds; r++) {
        const unsigned i = (r * gridDim.x + blockIdx.x) * 256u + t;
        if (i < n) atomicAdd(&s_hist[topk_range_digit(topk_image(rows[(size_t)i * stride + keyWord], is32, desc), lo, shift)], 1u);
    }
    __syncthreads();
    for (int b = t; b < TOPK_BINS; b += 256) { const unsigned c = s_hist[b]; if (c) atomicAdd(&hist[b], c); }
    // meeting point 1: this workgroup's histogram atomics have been performed; wait for everybody's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __hip_atomic_fetch_add(ticket1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 t0 = (u64)wall_clock64();
        unsigned ok = 1;
        while (__hip_atomic_load(ticket1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            if ((u64)wall_clock64() - t0 > 100000ull) { ok = 0; atomicOr(err, 256u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        s_flag = ok;
    }
    __syncthreads();
    // the lowest bin that still belongs to the candidates: the highest b with (rows in bins >= b) >= want
    unsigned c[8], local = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) { c[b] = __hip_atomic_load(&hist[t * 8 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); local += c[b]; }
    if (t == 0) s_bin = 0;                                  // fewer than `want` rows in all: every row qualifies
    s_above[t] = local;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                     // inclusive suffix sums over the threads
        const unsigned v = t + d < 256 ? s_above[t + d] : 0u;
        __syncthreads();
        s_above[t] += v;
        __syncthreads();
    }
    unsigned running = s_above[t] - local;                  // rows in bins above this thread's eight
#pragma unroll
    for (int b = 7; b >= 0; b--) {
        if (running < want && want <= running + c[b]) s_bin = (unsigned)(t * 8 + b);
        running += c[b];
    }
    __syncthreads();
    const unsigned bin = s_bin;
    const int lane = t & 63;
    auto gather = [&](unsigned i, bool take) {
        const unsigned long long vote = __ballot(take);
        if (vote == 0) return;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(candCount, (unsigned)__popcll(vote));
        base = (unsigned)__shfl((int)base, 0, 64);
        if (!take) return;
        const unsigned pos = base + (unsigned)__popcll(vote & ((1ull << lane) - 1ull));
        if (pos >= capacity) return;
        const i64* src = rows + (size_t)i * stride;
        // (host-mapped memory, read by the host as soon as the sequence number arrives: system-scope stores, waited for before this
        // workgroup takes its second ticket)
        if (es.first) {          // narrow rows [slot | key] (k_compact_entries): the candidate's words come from the table entry
            const i64 s = src[0];
            const i64 e = es.unmix ? (i64)rank_unmix((u64)s, (u64)es.cap) : s;
            i64* dst = cand + (size_t)pos * fullStride;
            for (int w0 = 0; w0 < fullStride; w0 += 8) {
                i64 v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = w0 + j < fullStride ? entry_word(es, s, e, w0 + j) : 0;
#pragma unroll
                for (int j = 0; j < 8; j++) if (w0 + j < fullStride) __hip_atomic_store(dst + w0 + j, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        i64* dst = cand + (size_t)pos * stride;
        for (int w = 0; w < stride; w++) __hip_atomic_store(dst + w, src[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
#pragma unroll
    for (int r = 0; r < RSEL_KEEP; r++)
        if ((unsigned)r < rounds) gather(((unsigned)r * gridDim.x + blockIdx.x) * 256u + t, dig[r] != 0xffffffffu && dig[r] >= bin);
    for (unsigned r = RSEL_KEEP; r < rounds; r++) {
        const unsigned i = (r * gridDim.x + blockIdx.x) * 256u + t;
        gather(i, i < n && topk_range_digit(topk_image(rows[(size_t)i * stride + keyWord], is32, desc), lo, shift) >= bin);
    }
    // meeting point 2: the holder of the last ticket publishes the status words (nobody waits here)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_flag = __hip_atomic_fetch_add(ticket2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (s_flag) {
        if (t == 0) __hip_atomic_store(host + 0, (u64)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (t == 1 && groupCount) __hip_atomic_store(host + 1, (u64)*groupCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (t == 2) __hip_atomic_store(host + 2, (u64)__hip_atomic_load(candCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (pipeStats && t >= 8 && t < 8 + nPipelines) __hip_atomic_store(host + t, pipeStats[t - 8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // the sequence number behind them: stored once every store above has been acknowledged (every workgroup waited for its
        // candidate stores before it took its ticket), system-scope release - the host watches this word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store(host + 4, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// scratch of both forms: [image range: 2 x u64][candidate count: u32][ticket: u32][TOPK_PASSES histograms]; the one-launch short form
// keeps its second ticket behind the first histogram
size_t topkHistBytes() { return 24 + (size_t)TOPK_PASSES * TOPK_BINS * sizeof(unsigned); }

// to be enqueued BEFORE the compaction that collects the image range: clears range, candidate count and the histogram
size_t topkRangeScratchBytes() { return 24 + TOPK_BINS * sizeof(unsigned) + 8; }
void prepareTopCandidatesRange(Context& ctx, void* scratch) { RSQ_HIP(hipMemsetAsync(scratch, 0, topkRangeScratchBytes(), ctx.stream)); }

void selectTopCandidatesRange(Context& ctx, const int64_t* rows, int stride, int keyWord, bool is32, bool desc, const uint32_t* nRows,
                              uint32_t rowsUpperBound, uint32_t want, void* scratch, int64_t* cand, uint32_t capacity) {
    const u64* range = (const u64*)scratch;
    unsigned* candCount = (unsigned*)((char*)scratch + 16);
    unsigned* hist = (unsigned*)((char*)scratch + 24);
    const unsigned grid = (unsigned)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)ctx.numCUs, (rowsUpperBound + 2047) / 2048));
    hipLaunchKernelGGL(k_topk_range_hist, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)rows, stride, keyWord, is32 ? 1 : 0, desc ? 1 : 0,
                       (const unsigned*)nRows, (unsigned)rowsUpperBound, range, hist);
    hipLaunchKernelGGL(k_topk_range_gather, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)rows, stride, keyWord, is32 ? 1 : 0, desc ? 1 : 0,
                       (const unsigned*)nRows, (unsigned)rowsUpperBound, range, (const unsigned*)hist, (unsigned)want, (i64*)cand, (unsigned)capacity, candCount);
    RSQ_HIP(hipGetLastError());
}

void selectTopCandidatesRangePublish(Context& ctx, const int64_t* rows, int stride, int keyWord, bool is32, bool desc, const uint32_t* nRows,
                                     uint32_t rowsUpperBound, uint32_t want, void* scratch, int64_t* candHostMapped, uint32_t capacity,
                                     uint64_t* hostWords, uint64_t seq, uint32_t* err, const uint32_t* groupCount, const uint64_t* pipeStats, int nPipelines,
                                     const TableEntries* entries) {
    if (nPipelines > 56) throw Error(RSQ_ERR_UNSUPPORTED, "more than 56 pipelines in one query");
    const u64* range = (const u64*)scratch;
    unsigned* candCount = (unsigned*)((char*)scratch + 16);
    unsigned* hist = (unsigned*)((char*)scratch + 24);
    EntrySource es{};
    int fullStride = stride;
    if (entries) {
        if (stride != 2 || keyWord != 1) throw Error(RSQ_ERR_DEVICE, "selectTopCandidatesRangePublish: narrow rows are [slot | key]");
        es = EntrySource{(const i64*)entries->firstRow, (i64)entries->capacity, (const i64*)entries->words, entries->nWords, entries->wordsAos ? 1 : 0,
                         (const i64*)entries->acc, entries->nAcc, entries->unmix ? 1 : 0, entries->deref, entries->tabStride > 0 ? entries->tabStride : entries->nWords};
        fullStride = 1 + entries->nWords + entries->nAcc;
    }
    // (the grid must be on the chip as a whole: at most one workgroup per CU)
    const unsigned grid = (unsigned)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)ctx.numCUs, (rowsUpperBound + 1023) / 1024));
    hipLaunchKernelGGL(k_topk_range_select, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)rows, stride, keyWord, is32 ? 1 : 0, desc ? 1 : 0,
                       (const unsigned*)nRows, (unsigned)rowsUpperBound, range, hist, (unsigned)want, (i64*)candHostMapped, (unsigned)capacity, candCount,
                       (unsigned*)((char*)scratch + 20), (unsigned*)((char*)scratch + 24 + TOPK_BINS * sizeof(unsigned)), err, (u64*)hostWords, (u64)seq, (const unsigned*)groupCount, (const u64*)pipeStats, nPipelines, es, fullStride);
    RSQ_HIP(hipGetLastError());
}

void selectTopCandidates(Context& ctx, const int64_t* rows, int stride, int keyWord, bool is32, bool desc, const uint32_t* nRows,
                         uint32_t rowsUpperBound, uint32_t want, uint64_t* images, void* scratch, int64_t* cand, uint32_t capacity) {
    unsigned* candCount = (unsigned*)((char*)scratch + 16);
    unsigned* hists = (unsigned*)((char*)scratch + 24);
    const unsigned grid = (unsigned)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)ctx.numCUs, (rowsUpperBound + 2047) / 2048));
    RSQ_HIP(hipMemsetAsync(scratch, 0, topkHistBytes(), ctx.stream));
    for (int p = 0; p < TOPK_PASSES; p++)
        hipLaunchKernelGGL(k_topk_hist, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)rows, stride, keyWord, is32 ? 1 : 0, desc ? 1 : 0,
                           (const unsigned*)nRows, (unsigned)rowsUpperBound, (u64*)images, (unsigned*)hists, p, (unsigned)want);
    hipLaunchKernelGGL(k_topk_gather, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)rows, stride, (const unsigned*)nRows, (unsigned)rowsUpperBound,
                       (const u64*)images, (const unsigned*)hists, (unsigned)want, (i64*)cand, (unsigned)capacity, (unsigned*)candCount);
    RSQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// partitioned aggregation: per-(workgroup, partition) record counts -> record positions.
// m[wg][p] becomes the position of workgroup wg's first record of partition p when partitions are laid out one after
// the other and, inside a partition, workgroups in order; partStart[p] / partStart[P] are the partition bounds.
// One workgroup: the matrix is small (workgroups x partitions words) and every column walk is coalesced across threads.
// ------------------------------------------------------------------------------------------------
// step 1, one workgroup per partition p: exclusive prefix of column p over the workgroups (block scan in LDS, chunks of
// 1024 workgroups with a carry), the column total goes to totals[p]
__global__ void __launch_bounds__(1024) k_part_within(unsigned* __restrict__ m, int nWG, int P, u64* __restrict__ totals) {
    __shared__ u64 s[1024];          // 64-bit: the host rejects totals >= 2^32 and must see them unwrapped
    __shared__ u64 carry;
    const int p = blockIdx.x, t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nWG; base += 1024) {
        const int wg = base + t;
        const u64 c = wg < nWG ? (u64)m[(size_t)wg * P + p] : 0ull;
        s[t] = c;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                  // Hillis-Steele inclusive scan
            const u64 v = t >= d ? s[t - d] : 0ull;
            __syncthreads();
            s[t] += v;
            __syncthreads();
        }
        const u64 before = carry;
        if (wg < nWG) m[(size_t)wg * P + p] = (unsigned)(before + s[t] - c);
        __syncthreads();
        if (t == 1023) carry = before + s[1023];
        __syncthreads();
    }
    if (t == 0) totals[p] = carry;
}
// step 2, one workgroup: exclusive scan of the partition totals -> partStart[0..P], grand total
__global__ void __launch_bounds__(1024) k_part_starts(const u64* __restrict__ totals, int P, unsigned* __restrict__ partStart, u64* total) {
    __shared__ u64 s_tot[4097];
    for (int p = threadIdx.x; p < P; p += blockDim.x) s_tot[p] = totals[p];
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 acc = 0;
        for (int p = 0; p < P; p++) { const u64 v = s_tot[p]; s_tot[p] = acc; acc += v; }
        s_tot[P] = acc;
        *total = acc;
    }
    __syncthreads();
    for (int p = threadIdx.x; p <= P; p += blockDim.x) partStart[p] = (unsigned)s_tot[p];
}

void partitionOffsets(Context& ctx, uint32_t* counts, int nWorkgroups, int nPartitions, uint64_t* totals, uint32_t* partStart, uint64_t* total) {
    if (nPartitions > 4096) throw Error(RSQ_ERR_UNSUPPORTED, "more than 4096 aggregation partitions");
    hipLaunchKernelGGL(k_part_within, dim3((unsigned)nPartitions), dim3(1024), 0, ctx.stream, (unsigned*)counts, nWorkgroups, nPartitions, (u64*)totals);
    hipLaunchKernelGGL(k_part_starts, dim3(1), dim3(1024), 0, ctx.stream, (const u64*)totals, nPartitions, (unsigned*)partStart, (u64*)total);
    RSQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of per-slot tuple counts (u32) into output offsets (u64); `n` includes one trailing zero slot so that
// offs[n - 1] is the total.  Hand-written (round 1 went through hipCUB): three launches — (1) every workgroup scans a chunk
// of 4096 counts (16 per thread, wave shuffles + the four wave totals) and leaves chunk-relative offsets and its total,
// (2) one workgroup scans the chunk totals (1024 at a time with a carry), (3) the chunk bases are added.  Reads the counts
// once and writes the offsets twice: 4 + 16 bytes per slot, bandwidth-bound like the rocPRIM scan it replaces.
// `temp` holds the chunk totals and bases: scanTempBytes(n).
// ------------------------------------------------------------------------------------------------
#define SCAN_CHUNK 4096
// (four sub-blocks of 1024 counts per chunk, a thread takes four neighbouring counts of each: one 16-byte load and two 16-byte stores per
// lane, unit stride across the wave.  A thread walking 16 consecutive counts - stores 128 bytes apart across the lanes - took 168 us for
// the 30 M lane slots of a 60 M-row materialisation; this form: about a third.)
__global__ void __launch_bounds__(256) k_scan_chunks(const unsigned* __restrict__ counts, u64* __restrict__ offs, i64 n, u64* __restrict__ chunkTotal) {
    __shared__ u64 s_wave[2][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    u64 carry = 0;
#pragma unroll
    for (int sb = 0; sb < 4; sb++) {
        const i64 i0 = (i64)blockIdx.x * SCAN_CHUNK + (i64)sb * 1024 + (i64)t * 4;
        unsigned c[4] = {0u, 0u, 0u, 0u};
        if (i0 + 3 < n) { const uint4 v = *reinterpret_cast<const uint4*>(counts + i0); c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w; }
        else for (int j = 0; j < 4; j++) if (i0 + j < n) c[j] = counts[i0 + j];
        const u64 mine = (u64)c[0] + c[1] + c[2] + c[3];
        u64 incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u64 v = (u64)__shfl_up((long long)incl, d, 64); if (lane >= d) incl += v; }
        if (lane == 63) s_wave[sb & 1][wave] = incl;
        __syncthreads();                                  // (two sets of wave totals in turn: one barrier per sub-block)
        u64 before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const u64 x = s_wave[sb & 1][w]; if (w < wave) before += x; total += x; }
        const u64 r0 = carry + before + incl - mine, r1 = r0 + c[0], r2 = r1 + c[1], r3 = r2 + c[2];
        if (i0 + 3 < n) {
            ulonglong2* o = reinterpret_cast<ulonglong2*>(offs + i0);
            o[0] = make_ulonglong2(r0, r1); o[1] = make_ulonglong2(r2, r3);
        } else {
            const u64 r[4] = {r0, r1, r2, r3};
            for (int j = 0; j < 4; j++) if (i0 + j < n) offs[i0 + j] = r[j];
        }
        carry += total;
    }
    if (t == 0) chunkTotal[blockIdx.x] = carry;
}
// The same scan in ONE launch, offsets absolute from the start: the chunks' totals travel through chain[chunk] = (value << 2) | state
// (1: the chunk's own total, 2: the total of all chunks up to and including it) by decoupled look-back, a wave reading 64 predecessors
// per round trip - as in k_rank_blocks_chained below.  Workgroups are dispatched in the order of their index, so whatever a chunk
// waits for is running or done; a wave that nevertheless waits longer than 1 ms sets bit 512 of *stuck and leaves (the host repeats
// the scan with the three launches).  `chain` must be zero before the launch.  Saves the pass that adds the chunk bases (87 us of
// reading and re-writing the 240 MB of offsets of a 60 M-row materialisation) and the single-workgroup scan of the totals.
__global__ void __launch_bounds__(256) k_scan_chained(const unsigned* __restrict__ counts, u64* __restrict__ offs, i64 n, u64* __restrict__ chain,
                                                      unsigned* __restrict__ stuck) {
    __shared__ u64 s_wave[2][4];
    __shared__ u64 s_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned c[4][4];
    u64 loc[4];
    u64 carry = 0;
#pragma unroll
    for (int sb = 0; sb < 4; sb++) {
        const i64 i0 = (i64)blockIdx.x * SCAN_CHUNK + (i64)sb * 1024 + (i64)t * 4;
        c[sb][0] = c[sb][1] = c[sb][2] = c[sb][3] = 0u;
        if (i0 + 3 < n) { const uint4 v = *reinterpret_cast<const uint4*>(counts + i0); c[sb][0] = v.x; c[sb][1] = v.y; c[sb][2] = v.z; c[sb][3] = v.w; }
        else for (int j = 0; j < 4; j++) if (i0 + j < n) c[sb][j] = counts[i0 + j];
        const u64 mine = (u64)c[sb][0] + c[sb][1] + c[sb][2] + c[sb][3];
        u64 incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u64 v = (u64)__shfl_up((long long)incl, d, 64); if (lane >= d) incl += v; }
        if (lane == 63) s_wave[sb & 1][wave] = incl;
        __syncthreads();
        u64 before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const u64 x = s_wave[sb & 1][w]; if (w < wave) before += x; total += x; }
        loc[sb] = carry + before + incl - mine;
        carry += total;
    }
    if (t < 64) {
        const u64 total = carry;
        u64 base = 0;
        if (blockIdx.x == 0) { if (lane == 0) __hip_atomic_store(&chain[0], (total << 2) | 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else {
            if (lane == 0) __hip_atomic_store(&chain[blockIdx.x], (total << 2) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            i64 hi = (i64)blockIdx.x - 1;
            for (;;) {
                const i64 idx = hi - lane;
                const u64 v = idx >= 0 ? __hip_atomic_load(&chain[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 2ull;      // (in front of chunk 0: nothing, inclusive)
                const u64 ready = __ballot((v & 3ull) != 0ull), incl = __ballot((v & 3ull) == 2ull);
                u64 take = 0; bool done = false, moved = false;
                if (incl) {
                    const int f = __ffsll((long long)incl) - 1;
                    const u64 below = (1ull << f) - 1ull;
                    if ((ready & below) == below) { take = lane <= f ? v >> 2 : 0ull; done = true; }
                } else if (ready == ~0ull) { take = v >> 2; moved = true; }
                if (done || moved) {
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) take += (u64)__shfl_xor((long long)take, m, 64);
                    base += take;
                    if (done) break;
                    hi -= 64;
                    continue;
                }
                if (wall_clock64() - t0 > 100000ll) { if (lane == 0) atomicOr(stuck, 512u); base = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) __hip_atomic_store(&chain[blockIdx.x], ((base + total) << 2) | 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_base = base;
    }
    __syncthreads();
    const u64 base = s_base;
#pragma unroll
    for (int sb = 0; sb < 4; sb++) {
        const i64 i0 = (i64)blockIdx.x * SCAN_CHUNK + (i64)sb * 1024 + (i64)t * 4;
        const u64 r0 = base + loc[sb], r1 = r0 + c[sb][0], r2 = r1 + c[sb][1], r3 = r2 + c[sb][2];
        if (i0 + 3 < n) {
            ulonglong2* o = reinterpret_cast<ulonglong2*>(offs + i0);
            o[0] = make_ulonglong2(r0, r1); o[1] = make_ulonglong2(r2, r3);
        } else {
            const u64 r[4] = {r0, r1, r2, r3};
            for (int j = 0; j < 4; j++) if (i0 + j < n) offs[i0 + j] = r[j];
        }
    }
}
__global__ void __launch_bounds__(1024) k_scan_chunk_totals(const u64* __restrict__ chunkTotal, i64 nChunks, u64* __restrict__ chunkBase) {
    __shared__ u64 s[1024];
    u64 carry = 0;
    for (i64 base = 0; base < nChunks; base += 1024) {
        const i64 i = base + threadIdx.x;
        const u64 v = i < nChunks ? chunkTotal[i] : 0ull;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const u64 x = threadIdx.x >= (unsigned)off ? s[threadIdx.x - off] : 0ull;
            __syncthreads();
            s[threadIdx.x] += x;
            __syncthreads();
        }
        if (i < nChunks) chunkBase[i] = carry + s[threadIdx.x] - v;
        carry += s[1023];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_scan_add_base(u64* __restrict__ offs, i64 n, const u64* __restrict__ chunkBase) {
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) offs[i] += chunkBase[i / SCAN_CHUNK];
}

size_t scanTempBytes(int64_t n) { return (size_t)((n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1) * 16; }

// one launch (k_scan_chained); `temp` as for exclusiveScanCounts.  A look-back that timed out sets bit 512 of the context's error word:
// the caller then repeats with exclusiveScanCounts.
void exclusiveScanCountsChained(Context& ctx, const uint32_t* counts, uint64_t* offs, int64_t n, void* temp, size_t tempBytes) {
    if (n > 0x7fffffff) throw Error(RSQ_ERR_UNSUPPORTED, "materialisation of more than 2^31 lane slots");
    if (n <= 0) return;
    const i64 nChunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (tempBytes < (size_t)(nChunks + 1) * 16) throw Error(RSQ_ERR_DEVICE, "exclusiveScanCountsChained: temporary buffer too small");
    RSQ_HIP(hipMemsetAsync(temp, 0, (size_t)nChunks * 8, ctx.stream));
    hipLaunchKernelGGL(k_scan_chained, dim3((unsigned)nChunks), dim3(256), 0, ctx.stream, (const unsigned*)counts, (u64*)offs, (i64)n, (u64*)temp, (unsigned*)ctx.dErr);
    RSQ_HIP(hipGetLastError());
}
void exclusiveScanCounts(Context& ctx, const uint32_t* counts, uint64_t* offs, int64_t n, void* temp, size_t tempBytes) {
    if (n > 0x7fffffff) throw Error(RSQ_ERR_UNSUPPORTED, "materialisation of more than 2^31 lane slots");
    if (n <= 0) return;
    const i64 nChunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (tempBytes < (size_t)(nChunks + 1) * 16) throw Error(RSQ_ERR_DEVICE, "exclusiveScanCounts: temporary buffer too small");
    u64* chunkTotal = (u64*)temp;
    u64* chunkBase = chunkTotal + nChunks;
    hipLaunchKernelGGL(k_scan_chunks, dim3((unsigned)nChunks), dim3(256), 0, ctx.stream, (const unsigned*)counts, (u64*)offs, (i64)n, chunkTotal);
    hipLaunchKernelGGL(k_scan_chunk_totals, dim3(1), dim3(1024), 0, ctx.stream, (const u64*)chunkTotal, nChunks, chunkBase);
    const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>(4096, (n + 255) / 256));
    hipLaunchKernelGGL(k_scan_add_base, dim3(grid), dim3(256), 0, ctx.stream, (u64*)offs, (i64)n, (const u64*)chunkBase);
    RSQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// bitmap-rank dictionary (kernels/rsq_device.h rank_of): prefix arrays over a join table's key bitmap, and the placement
// of the build pipeline's records (appended in arrival order) at the rank of their key.  The bitmap is allocated in whole
// 256-bit blocks.  Input that is clustered by the key (TPC-H orders by o_orderkey) arrives nearly in rank order, so the
// placement writes walk the entry array almost sequentially; input in random order is still placed correctly.
// ------------------------------------------------------------------------------------------------
#define RANK_CHUNK_BLOCKS RSQ_RANK_CHUNK_BLOCKS          /* 32-byte blocks per workgroup (engine.h): 1024 — with 4096 a 60 M-bit bitmap
                                                           gave 66 workgroups to 256 CUs: 10.6 us instead of ~4 */
#define RANK_PT (RANK_CHUNK_BLOCKS / 256)          /* blocks per thread */
// blocks are [rank word | 7 bitmap words]; this pass writes every block's rank relative to its chunk
__global__ void __launch_bounds__(256) k_rank_blocks(unsigned* __restrict__ bm, i64 nBlocks, unsigned* __restrict__ chunkTotal) {
    __shared__ unsigned s_tot[256];
    const i64 b0 = (i64)blockIdx.x * RANK_CHUNK_BLOCKS + (i64)threadIdx.x * RANK_PT;
    unsigned c[RANK_PT];
    unsigned mine = 0;
#pragma unroll
    for (int j = 0; j < RANK_PT; j++) {
        unsigned n = 0;
        if (b0 + j < nBlocks) {
            const uint4* w = reinterpret_cast<const uint4*>(bm + (b0 + j) * 8);
            const uint4 lo = w[0], hi = w[1];
            n = __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
        }
        c[j] = n; mine += n;
    }
    s_tot[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                  // inclusive scan of the thread totals
        unsigned v = threadIdx.x >= (unsigned)off ? s_tot[threadIdx.x - off] : 0u;
        __syncthreads();
        s_tot[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = s_tot[threadIdx.x] - mine;
#pragma unroll
    for (int j = 0; j < RANK_PT; j++) { if (b0 + j < nBlocks) bm[(b0 + j) * 8] = run; run += c[j]; }
    if (threadIdx.x == 255) chunkTotal[blockIdx.x] = s_tot[255];
}

// the rank words become absolute (chunk base added): a probe then needs nothing but the block it tested
// one workgroup per chunk: it sums the totals of the chunks before it (a few hundred words; a separate single-workgroup scan of
// them was one more launch, ~5 us of a 0.45 ms query), adds that base to its blocks' rank words and leaves it in chunkBase
__global__ void __launch_bounds__(256) k_rank_absolute(unsigned* __restrict__ bm, i64 nBlocks, const unsigned* __restrict__ chunkTotal, int nChunks,
                                                       unsigned* __restrict__ chunkBase /* [nChunks + 1] */) {
    __shared__ unsigned s_part[4];
    unsigned v = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += blockDim.x) v += chunkTotal[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    const unsigned base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (threadIdx.x == 0) {
        chunkBase[blockIdx.x] = base;
        if ((int)blockIdx.x == nChunks - 1) chunkBase[nChunks] = base + chunkTotal[blockIdx.x];          // number of distinct keys
    }
    const i64 b0 = (i64)blockIdx.x * RANK_CHUNK_BLOCKS;
    for (i64 b = b0 + threadIdx.x; b < b0 + RANK_CHUNK_BLOCKS && b < nBlocks; b += blockDim.x) bm[b * 8] += base;
}

// The whole index in ONE launch (round 3): the chunks' totals travel through `chain[chunk]` (below) and a workgroup writes its
// rank words absolute from the start.  The grid is at most 1171 workgroups of 256 threads (a key domain of 2^28 bits), fewer than the chip holds at a time,
// so a predecessor is always running; a workgroup that nevertheless waits longer than ~1 ms raises `*stuck` and leaves (the host
// then repeats the index with the two-launch form above — never a hang).  `chain` must be zero before the launch.
__global__ void __launch_bounds__(256) k_rank_blocks_chained(unsigned* __restrict__ bm, i64 nBlocks, unsigned* __restrict__ chain, int nChunks,
                                                             unsigned* __restrict__ chunkBase /* [nChunks + 1] */, unsigned* __restrict__ stuck) {
    __shared__ unsigned s_tot[256];
    __shared__ unsigned s_base;
    const i64 b0 = (i64)blockIdx.x * RANK_CHUNK_BLOCKS + (i64)threadIdx.x * RANK_PT;
    unsigned c[RANK_PT];
    unsigned mine = 0;
#pragma unroll
    for (int j = 0; j < RANK_PT; j++) {
        unsigned n = 0;
        if (b0 + j < nBlocks) {
            const uint4* w = reinterpret_cast<const uint4*>(bm + (b0 + j) * 8);
            const uint4 lo = w[0], hi = w[1];
            n = __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
        }
        c[j] = n; mine += n;
    }
    s_tot[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                  // inclusive scan of the thread totals
        unsigned v = threadIdx.x >= (unsigned)off ? s_tot[threadIdx.x - off] : 0u;
        __syncthreads();
        s_tot[threadIdx.x] += v;
        __syncthreads();
    }
    if (threadIdx.x < 64) {
        // chain word = (value << 2) | state: 1 = this chunk's own total, 2 = the total of all chunks up to and including it.
        // A chunk publishes its own total at once and then looks BACK over its predecessors, adding own totals until it meets an
        // inclusive one (decoupled look-back): no chunk waits for a chain of 260 hand-overs, which a plain "wait for the chunk
        // before me" turned into 130 us of serial latency on TPC-H Q3's 60 M-bit orders bitmap.  The look-back is done by one WAVE,
        // lane l reading the l-th predecessor: 64 chain words per round trip (one thread walking them one by one: 19 us for 262 chunks).
        const int lane = (int)threadIdx.x;
        const unsigned total = s_tot[255];
        unsigned base = 0;
        if (blockIdx.x == 0) { if (lane == 0) __hip_atomic_store(&chain[0], (total << 2) | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else {
            if (lane == 0) __hip_atomic_store(&chain[blockIdx.x], (total << 2) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            int hi = (int)blockIdx.x - 1;                 // the nearest predecessor not yet accounted for
            for (;;) {
                const int idx = hi - lane;
                // (in front of chunk 0: nothing, inclusive)
                const unsigned v = idx >= 0 ? __hip_atomic_load(&chain[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 2u;
                const u64 ready = __ballot((v & 3u) != 0u), incl = __ballot((v & 3u) == 2u);
                unsigned take = 0; bool done = false, moved = false;
                if (incl) {
                    const int f = __ffsll((long long)incl) - 1;               // the nearest inclusive total
                    const u64 below = (1ull << f) - 1ull;
                    if ((ready & below) == below) { take = lane <= f ? v >> 2 : 0u; done = true; }
                } else if (ready == ~0ull) { take = v >> 2; moved = true; }           // 64 own totals: add them, look further back
                if (done || moved) {
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) take += (unsigned)__shfl_xor((int)take, m, 64);
                    base += take;
                    if (done) break;
                    hi -= 64;
                    continue;
                }
                if (wall_clock64() - t0 > 100000ll) { if (lane == 0) atomicOr(stuck, 128u); base = 0; break; }          // 100 MHz clock: 1 ms
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) __hip_atomic_store(&chain[blockIdx.x], ((base + total) << 2) | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            chunkBase[blockIdx.x] = base;
            if ((int)blockIdx.x == nChunks - 1) chunkBase[nChunks] = base + total;
            s_base = base;
        }
    }
    __syncthreads();
    unsigned run = s_base + s_tot[threadIdx.x] - mine;
#pragma unroll
    for (int j = 0; j < RANK_PT; j++) { if (b0 + j < nBlocks) bm[(b0 + j) * 8] = run; run += c[j]; }
}

void rankTableIndex(Context& ctx, uint32_t* bitmap, int64_t nBlocks, uint32_t* chunkTotal, uint32_t* chunkBase) {
    const int nChunks = (int)((nBlocks + RANK_CHUNK_BLOCKS - 1) / RANK_CHUNK_BLOCKS);
    hipLaunchKernelGGL(k_rank_blocks, dim3((unsigned)nChunks), dim3(256), 0, ctx.stream, (unsigned*)bitmap, (i64)nBlocks, (unsigned*)chunkTotal);
    hipLaunchKernelGGL(k_rank_absolute, dim3((unsigned)nChunks), dim3(256), 0, ctx.stream, (unsigned*)bitmap, (i64)nBlocks, (const unsigned*)chunkTotal, nChunks,
                       (unsigned*)chunkBase);
    RSQ_HIP(hipGetLastError());
}
// `chain` = the chunkTotal scratch, zeroed before the launch (the execution's first fill does that)
void rankTableIndexChained(Context& ctx, uint32_t* bitmap, int64_t nBlocks, uint32_t* chain, uint32_t* chunkBase) {
    const int nChunks = (int)((nBlocks + RANK_CHUNK_BLOCKS - 1) / RANK_CHUNK_BLOCKS);
    hipLaunchKernelGGL(k_rank_blocks_chained, dim3((unsigned)nChunks), dim3(256), 0, ctx.stream, (unsigned*)bitmap, (i64)nBlocks, (unsigned*)chain, nChunks,
                       (unsigned*)chunkBase, (unsigned*)ctx.dErr);
    RSQ_HIP(hipGetLastError());
}

// One workgroup per wave of the build pipeline: its region of the arrival-order buffer holds used[wave] records.
// Every thread takes two records at a time and issues all their loads before it uses any: the record words, then the 32-byte bitmap
// block of each key as two 16-byte loads (the rank is the block's rank word + the popcount below the key's bit, computed without
// branches).  The first form walked the block word by word behind a branch each - eight dependent round trips per record, 27.7 us for
// TPC-H Q3's 1.45 M records at SF10.
template <int NW>      // words per record (0: any, at most 8)
__global__ void __launch_bounds__(256) k_rank_place(const i64* __restrict__ temp, const unsigned* __restrict__ used, unsigned region,
                                                    const unsigned* __restrict__ nRecords, int nWordsArg, const unsigned* __restrict__ bm, i64 bmMin,
                                                    u64 bmBits, const unsigned* __restrict__ chunkBase, int nChunks, i64* __restrict__ words,
                                                    i64 capacity, unsigned* __restrict__ err) {
    // as many records as distinct keys, and no more than the table was sized for — anything else means the build side changed
    // since the sizing pass (two rows with one key, more rows): the host then falls back to the hash table
    if (blockIdx.x == 0 && threadIdx.x == 0 && (*nRecords != chunkBase[nChunks] || (i64)*nRecords > capacity)) atomicOr(err, 64u);
    const int nWords = NW ? NW : nWordsArg;
    constexpr int MAXW = NW ? NW : 8;
    const unsigned n = used[blockIdx.x];
    const i64* base = temp + (i64)blockIdx.x * region * nWords;
    for (unsigned i0 = threadIdx.x; i0 < n; i0 += 2u * blockDim.x) {
        i64 rec[2][MAXW];
        bool have[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const unsigned i = i0 + (unsigned)u * blockDim.x;
            have[u] = i < n;
#pragma unroll
            for (int k = 0; k < MAXW; k++) rec[u][k] = have[u] && k < nWords ? base[(i64)i * nWords + k] : 0;
        }
        uint4 lo[2], hi[2];
        unsigned wi[2], bit[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const u64 d = (u64)(rec[u][0] - bmMin);
            have[u] = have[u] && d < bmBits;          // a key outside the bitmap's domain: the build kernel has raised ERR_GROUP_OVERFLOW for it
            const unsigned w = (unsigned)(d >> 5), blkI = have[u] ? w / 7u : 0u;
            wi[u] = 1u + (w % 7u); bit[u] = (unsigned)d & 31u;
            const uint4* blk = reinterpret_cast<const uint4*>(bm + (i64)blkI * 8);
            lo[u] = blk[0]; hi[u] = blk[1];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (!have[u]) continue;
            const unsigned x[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
            unsigned r = x[0];
#pragma unroll
            for (unsigned j = 1; j < 8; j++) {
                const unsigned m = j < wi[u] ? 0xffffffffu : (j == wi[u] ? (1u << bit[u]) - 1u : 0u);
                r += __popc(x[j] & m);
            }
            if ((i64)r < capacity) {
#pragma unroll
                for (int k = 0; k < MAXW; k++) if (k < nWords) words[(i64)r * nWords + k] = rec[u][k];
            }
        }
    }
}

// ... records wider than 8 words (string payloads): word by word
__global__ void __launch_bounds__(256) k_rank_place_wide(const i64* __restrict__ temp, const unsigned* __restrict__ used, unsigned region,
                                                         const unsigned* __restrict__ nRecords, int nWords, const unsigned* __restrict__ bm, i64 bmMin,
                                                         u64 bmBits, const unsigned* __restrict__ chunkBase, int nChunks, i64* __restrict__ words,
                                                         i64 capacity, unsigned* __restrict__ err) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && (*nRecords != chunkBase[nChunks] || (i64)*nRecords > capacity)) atomicOr(err, 64u);
    const unsigned n = used[blockIdx.x];
    const i64* base = temp + (i64)blockIdx.x * region * nWords;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        const i64* rec = base + (i64)i * nWords;
        const u64 d = (u64)(rec[0] - bmMin);
        if (d >= bmBits) continue;
        const unsigned w = (unsigned)(d >> 5), blkI = w / 7u, wi = 1u + (w % 7u), bit = (unsigned)d & 31u;
        const uint4* blk = reinterpret_cast<const uint4*>(bm + (i64)blkI * 8);
        const uint4 lo = blk[0], hi = blk[1];
        const unsigned x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        unsigned r = x[0];
#pragma unroll
        for (unsigned j = 1; j < 8; j++) r += __popc(x[j] & (j < wi ? 0xffffffffu : (j == wi ? (1u << bit) - 1u : 0u)));
        if ((i64)r < capacity) for (int k = 0; k < nWords; k++) words[(i64)r * nWords + k] = rec[k];
    }
}

void rankTablePlace(Context& ctx, const int64_t* temp, const uint32_t* used, uint32_t nWaves, uint32_t region, const uint32_t* nRecords, int nWords,
                    const uint32_t* bitmap, int64_t bmMin, int64_t bmBits, const uint32_t* chunkBase, int64_t nBlocks, int64_t* words,
                    int64_t capacity) {
    const int nChunks = (int)((nBlocks + RANK_CHUNK_BLOCKS - 1) / RANK_CHUNK_BLOCKS);
    if (nWords > 8) {
        hipLaunchKernelGGL(k_rank_place_wide, dim3(std::max(1u, nWaves)), dim3(256), 0, ctx.stream, (const i64*)temp, (const unsigned*)used, (unsigned)region,
                           (const unsigned*)nRecords, nWords, (const unsigned*)bitmap, (i64)bmMin, (u64)bmBits, (const unsigned*)chunkBase, nChunks,
                           (i64*)words, (i64)capacity, (unsigned*)ctx.dErr);
        RSQ_HIP(hipGetLastError());
        return;
    }
#define RSQ_LAUNCH_PLACE(NW) hipLaunchKernelGGL(k_rank_place<NW>, dim3(std::max(1u, nWaves)), dim3(256), 0, ctx.stream, (const i64*)temp, (const unsigned*)used, \
                       (unsigned)region, (const unsigned*)nRecords, nWords, (const unsigned*)bitmap, (i64)bmMin, (u64)bmBits, (const unsigned*)chunkBase, nChunks, \
                       (i64*)words, (i64)capacity, (unsigned*)ctx.dErr)
    switch (nWords) { case 1: RSQ_LAUNCH_PLACE(1); break; case 2: RSQ_LAUNCH_PLACE(2); break; case 3: RSQ_LAUNCH_PLACE(3); break; case 4: RSQ_LAUNCH_PLACE(4); break;
                      default: RSQ_LAUNCH_PLACE(0); }
#undef RSQ_LAUNCH_PLACE
    RSQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// group-by merge of partial aggregate tables (multi-GPU): `parts` holds nParts tables of `words` int64 words, each laid
// out [ nMin | nMax | nSum ] (include/resql_hip.h, rsq_query_execute_partial), either back to back (one all-gather
// buffer: stride = words) or at `stride` words from each other.  out[w] = min / max / sum over the parts — the one
// kernel behind the collective, instead of one reduction launch per segment.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_merge_partials(const i64* __restrict__ parts, int nParts, i64 stride, i64 nMin, i64 nMax, i64 nSum,
                                                        i64* __restrict__ out, i64* __restrict__ hostOut, u64* hostFlag, u64 seq) {
    const i64 words = nMin + nMax + nSum;
    for (i64 w = blockIdx.x * (i64)blockDim.x + threadIdx.x; w < words; w += (i64)gridDim.x * blockDim.x) {
        i64 v = parts[w];
        if (w < nMin) { for (int r = 1; r < nParts; r++) { i64 x = parts[(i64)r * stride + w]; v = x < v ? x : v; } }
        else if (w < nMin + nMax) { for (int r = 1; r < nParts; r++) { i64 x = parts[(i64)r * stride + w]; v = x > v ? x : v; } }
        else { u64 s = (u64)v; for (int r = 1; r < nParts; r++) s += (u64)parts[(i64)r * stride + w]; v = (i64)s; }
        out[w] = v;
        if (hostOut) hostOut[w] = v;
    }
    // small tables (one workgroup): the merged table also goes straight into host-mapped memory, announced by a sequence
    // number the finalising host polls for — no read-back copy, no stream synchronisation on the step's critical path
    if (hostOut && gridDim.x == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(hostFlag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

void mergePartialsAsync(Context& ctx, const int64_t* parts, int nParts, int64_t stride, int64_t nMin, int64_t nMax, int64_t nSum, int64_t* out,
                        int64_t* hostOut, uint64_t* hostFlag, uint64_t seq) {
    const int64_t words = nMin + nMax + nSum;
    if (words <= 0 || nParts <= 0) return;
    unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (words + 255) / 256));
    if (hostOut && words <= 2048) grid = 1; else hostOut = nullptr;        // publication needs the single-workgroup form
    hipLaunchKernelGGL(k_merge_partials, dim3(grid), dim3(256), 0, ctx.stream, (const i64*)parts, nParts, (i64)stride, (i64)nMin, (i64)nMax, (i64)nSum,
                       (i64*)out, (i64*)hostOut, (u64*)hostFlag, (u64)seq);
    RSQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// the pre-compiled generic pipeline: scan -> filter -> dense aggregation, interpreted (generic.cpp builds the program).
// Registers live in LDS, [register][thread] (conflict-free: a wave's 64 accesses to one register are 64 consecutive words);
// the program is read through uniform loads.  Values are int64: INT / DATE columns sign-extend from 32 bits (the reference
// compares dates as signed 32-bit values), BOOL / CHAR(1) zero-extend from a byte.  Arithmetic wraps, division truncates and
// flags /0 like rsq::div, AND / OR are bit-wise on 0 / 1 values (no short circuit, as in the reference).
// Group ranks are checked against the statistics (ERR_GROUP_OVERFLOW) like the specialised kernels'.  Small aggregate tables
// are kept per workgroup in LDS and flushed once; larger ones take HBM atomics.
// ------------------------------------------------------------------------------------------------
struct GenericArgs {
    const void* col[G_MAX_COLS]; int colKind[G_MAX_COLS];
    const GenericInstr* code; int nInstr;
    int nKeys; int keyReg[G_MAX_KEYS]; int keyByteSet[G_MAX_KEYS]; i64 keyMin[G_MAX_KEYS]; i64 keyCard[G_MAX_KEYS]; i64 keyStride[G_MAX_KEYS];
    unsigned char keyValues[G_MAX_KEYS][G_MAX_SET]; int keyNValues[G_MAX_KEYS];
    int nAccs; int accReg[G_MAX_ACCS]; int accMerge[G_MAX_ACCS]; i64 accBlock[G_MAX_ACCS];
    i64 nRows, row0, groups, tableWords;
    u64* table; unsigned* err;
    int ldsTable;            // the [block][group] table fits the workgroup's LDS copy
};
#define GENERIC_LDS_TABLE_WORDS 3072

__device__ __forceinline__ void generic_merge_lds(u64* p, u64 v, int merge) {
    if (merge == 0) atomicAdd(p, v);
    else if (merge == 2) atomicMin(reinterpret_cast<i64*>(p), (i64)v);
    else atomicMax(reinterpret_cast<i64*>(p), (i64)v);
}
__device__ __forceinline__ void generic_merge_global(u64* p, u64 v, int merge) {
    if (merge == 0) { if (v) atomicAdd(p, v); }
    else if (merge == 2) { if ((i64)v < (i64)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(reinterpret_cast<i64*>(p), (i64)v); }
    else { if ((i64)v > (i64)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<i64*>(p), (i64)v); }
}

__global__ void __launch_bounds__(256) k_generic_aggregate(GenericArgs a) {
    // (the register file: as many registers as the program names, launchGenericAggregate - all 32 were 64 KB of LDS, two workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) i64 s_reg[];
    __shared__ u64 s_tab[GENERIC_LDS_TABLE_WORDS];
    i64* reg = s_reg + threadIdx.x;            // register r of this thread: reg[r * 256]
    if (a.ldsTable) {
        for (i64 i = threadIdx.x; i < a.tableWords; i += 256) {
            const i64 blk = i / a.groups;
            int merge = 0;
            for (int w = 0; w < a.nAccs; w++) if (a.accBlock[w] == blk) merge = a.accMerge[w];
            s_tab[i] = merge == 0 ? 0ull : merge == 2 ? 0x7fffffffffffffffull : 0x8000000000000000ull;
        }
        __syncthreads();
    }
    for (i64 base = (i64)blockIdx.x * 256; base < a.nRows; base += (i64)gridDim.x * 256) {
        const i64 r = base + threadIdx.x;
        bool alive = r < a.nRows;
        for (int pc = 0; pc < a.nInstr; pc++) {
            const GenericInstr in = a.code[pc];
            i64 x = reg[(int)in.a * 256], y = reg[(int)in.b * 256], v = 0;
            switch (in.op) {
                case G_COL: {
                    const int k = a.colKind[in.a];
                    if (alive) {
                        if (k == 3) v = reinterpret_cast<const i64*>(a.col[in.a])[r];
                        else if (k == 2) v = (i64)reinterpret_cast<const int*>(a.col[in.a])[r];
                        else v = (i64)reinterpret_cast<const unsigned char*>(a.col[in.a])[r];
                    }
                    break;
                }
                case G_CONST: v = in.imm; break;
                case G_ADD: v = (i64)((u64)x + (u64)y); break;
                case G_SUB: v = (i64)((u64)x - (u64)y); break;
                case G_MUL: v = (i64)((u64)x * (u64)y); break;
                case G_DIV:
                    if (y == 0 || (x == (i64)0x8000000000000000ull && y == -1)) { if (alive) atomicOr(a.err, 1u); v = 0; }
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
                case G_SELECT: v = x ? y : reg[(int)in.c * 256]; break;
                case G_FILTER: alive = alive && x != 0; break;
                default: break;
            }
            if (in.op != G_FILTER) reg[(int)in.dst * 256] = v;
            else if (!__any(alive)) break;                 // the whole wave is filtered out
        }
        if (!alive) continue;
        i64 gid = 0;
        for (int k = 0; k < a.nKeys; k++) {
            const i64 v = reg[a.keyReg[k] * 256];
            i64 rank = 0;
            bool ok;
            if (a.keyByteSet[k]) {
                ok = false;
                for (int d = 0; d < a.keyNValues[k]; d++) {
                    if (d && (unsigned char)v >= a.keyValues[k][d]) rank++;
                    ok = ok || (unsigned char)v == a.keyValues[k][d];
                }
            } else { rank = v - a.keyMin[k]; ok = (u64)rank < (u64)a.keyCard[k]; }
            if (!ok) { atomicOr(a.err, 8u); rank = 0; }
            gid += rank * a.keyStride[k];
        }
        for (int w = 0; w < a.nAccs; w++) {
            const int rg = a.accReg[w];
            const u64 v = rg == -1 ? (u64)(a.row0 + r) : rg == -2 ? 1ull : (u64)reg[rg * 256];
            const i64 cell = a.accBlock[w] * a.groups + gid;
            if (a.ldsTable) generic_merge_lds(&s_tab[cell], v, a.accMerge[w]);
            else generic_merge_global(a.table + cell, v, a.accMerge[w]);
        }
    }
    if (a.ldsTable) {
        __syncthreads();
        for (i64 i = threadIdx.x; i < a.tableWords; i += 256) {
            const i64 blk = i / a.groups;
            int merge = 0;
            for (int w = 0; w < a.nAccs; w++) if (a.accBlock[w] == blk) merge = a.accMerge[w];
            const u64 v = s_tab[i];
            const u64 idv = merge == 0 ? 0ull : merge == 2 ? 0x7fffffffffffffffull : 0x8000000000000000ull;
            if (v != idv) generic_merge_global(a.table + i, v, merge);
        }
    }
}

void launchGenericAggregate(Context& ctx, const GenericProgram& prog, const GenericInstr* dCode, int64_t nRows, int64_t row0, uint64_t* dTable,
                            int64_t denseGroups, int64_t tableWords) {
    GenericArgs a;
    memset(&a, 0, sizeof a);
    for (size_t i = 0; i < prog.cols.size(); i++) { a.col[i] = prog.cols[i].ptr; a.colKind[i] = prog.cols[i].kind; }
    a.code = dCode; a.nInstr = (int)prog.code.size();
    a.nKeys = (int)prog.keys.size();
    for (size_t k = 0; k < prog.keys.size(); k++) {
        a.keyReg[k] = prog.keys[k].reg; a.keyByteSet[k] = prog.keys[k].byteSet; a.keyMin[k] = prog.keys[k].min; a.keyCard[k] = prog.keys[k].card;
        a.keyStride[k] = prog.keys[k].stride; a.keyNValues[k] = prog.keys[k].nValues;
        memcpy(a.keyValues[k], prog.keys[k].values, G_MAX_SET);
    }
    a.nAccs = (int)prog.accs.size();
    for (size_t w = 0; w < prog.accs.size(); w++) { a.accReg[w] = prog.accs[w].reg; a.accMerge[w] = prog.accs[w].merge; a.accBlock[w] = prog.accs[w].block; }
    a.nRows = nRows; a.row0 = row0; a.groups = denseGroups; a.tableWords = tableWords;
    a.table = (u64*)dTable; a.err = (unsigned*)ctx.dErr;
    a.ldsTable = tableWords <= GENERIC_LDS_TABLE_WORDS ? 1 : 0;
    int nRegs = 2;
    for (auto& in : prog.code) nRegs = std::max(nRegs, std::max((int)in.dst, std::max((int)in.a, std::max((int)in.b, in.op == G_SELECT ? (int)in.c : 0))) + 1);
    for (auto& k : prog.keys) nRegs = std::max(nRegs, k.reg + 1);
    for (auto& w : prog.accs) nRegs = std::max(nRegs, w.reg + 1);
    nRegs = std::min<int>(nRegs, G_REGS);      // (G_COL / G_CONST name a column or nothing in `a`: an over-estimate at worst)
    const size_t regBytes = (size_t)nRegs * 256 * 8;
    // as many workgroups per CU as their LDS leaves room for (register file + the 24 KB partial table), at most 8
    const int64_t perCU = std::max<int64_t>(2, std::min<int64_t>(8, (int64_t)(150u << 10) / (int64_t)(regBytes + GENERIC_LDS_TABLE_WORDS * 8 + 256)));
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(perCU * (int64_t)ctx.numCUs, (nRows + 255) / 256));
    if (regBytes > (40u << 10)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_generic_aggregate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(G_REGS * 256 * 8));
    hipLaunchKernelGGL(k_generic_aggregate, dim3(grid), dim3(256), regBytes, ctx.stream, a);
    RSQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// streaming read bandwidth probe
// ------------------------------------------------------------------------------------------------
typedef i64 ll2 __attribute__((ext_vector_type(2)));

// 16 B per lane, unit stride, non-temporal (the same access form the pipeline kernels use for 8-byte columns)
__global__ void __launch_bounds__(256) k_read_sum(const ll2* __restrict__ p, i64 n2, u64* out) {
    u64 s = 0;
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        ll2 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        ll2 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        s += (u64)a.x + (u64)a.y + (u64)b.x + (u64)b.y + (u64)c.x + (u64)c.y + (u64)d.x + (u64)d.y;
    }
    for (; i < n2; i += stride) { ll2 a = __builtin_nontemporal_load(p + i); s += (u64)a.x + (u64)a.y; }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0 && s != 0x1234567) atomicAdd(out, s);
}

double measureReadBandwidth(Context& ctx, size_t bytes, int iters) {
    if (ctx.device < 0) throw Error(RSQ_ERR_DEVICE, "bandwidth probe needs a device context");
    RSQ_HIP(hipSetDevice(ctx.device));
    bytes &= ~(size_t)1023;
    void* buf = ctx.allocRaw(bytes);      // (gigabytes, once: not through the arena)
    u64* out = (u64*)ctx.alloc(8);
    RSQ_HIP(hipMemsetAsync(buf, 1, bytes, ctx.stream));
    RSQ_HIP(hipMemsetAsync(out, 0, 8, ctx.stream));
    const unsigned grid = (unsigned)(2 * ctx.numCUs);     // the geometry the pipeline kernels use
    hipLaunchKernelGGL(k_read_sum, dim3(grid), dim3(256), 0, ctx.stream, (const ll2*)buf, (i64)(bytes / 16), out);   // warm-up
    RSQ_HIP(hipEventRecord(ctx.ev0, ctx.stream));
    for (int i = 0; i < iters; i++)
        hipLaunchKernelGGL(k_read_sum, dim3(grid), dim3(256), 0, ctx.stream, (const ll2*)buf, (i64)(bytes / 16), out);
    RSQ_HIP(hipEventRecord(ctx.ev1, ctx.stream));
    RSQ_HIP(hipEventSynchronize(ctx.ev1));
    float ms = 0; RSQ_HIP(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1));
    ctx.freeRaw(buf); ctx.free(out);
    return (double)bytes * iters / (ms * 1e-3) / 1e9;
}

}  // namespace rsq
