// engine_devtail.cpp - the tails that stay on the device: a large dense aggregate table, and the group rows of a hash / join-entry
// aggregation with many groups (kernels in devtail.hip; the host's part of the emission order in hostref.cpp).  Called from executeQuery.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <sstream>

#include "engine_internal.h"

namespace rsq {

// ---- the tail of a large dense aggregation on the device (devtail.hip) --------------------------------------------------
// The aggregate table stays in HBM.  Device: groups present (in group-id order) -> ordered by first row (radix sort) -> the
// reference's hash of every group -> [8 bytes per group to the host] -> host: slot order of the reference's table (the cluster-
// parallel replay, hostref.cpp) -> [4 bytes per group back] -> device: packed result tuples in that order -> [the tuples to the
// host].  rsq_config.emission_order = RSQ_EMIT_ANY skips everything between "present" and "tuples".
bool denseDeviceTailWanted(Query& q) {
    const bool off = getenv("RSQ_DEVICE_TAIL") && atoi(getenv("RSQ_DEVICE_TAIL")) == 0;      // (read per execution: tests switch it)
    const int64_t minGroups = getenv("RSQ_DEVICE_TAIL_MIN") ? atoll(getenv("RSQ_DEVICE_TAIL_MIN")) : 65536;
    if (off || q.holdTail || q.aggPad != 1 || q.denseGroups < minGroups || q.denseGroups >= (1ll << 31) || !q.dAgg) return false;
    if (q.devTail < 0) q.devTail = planDenseDeviceTail(q, q.dtKeys, q.dtCols, q.dtTupleSize, q.dtLimitRows) ? 1 : 0;
    return q.devTail == 1;
}

// returns the ms spent behind the first synchronisation (= behind the pipelines' kernels): the tail proper
double runDenseDeviceTail(Query& q) {
    Context& ctx = q.ctx;
    const int64_t D = q.denseGroups;
    const bool trace = getenv("RSQ_TRACE") != nullptr;
    double tPhase = nowMs();
    auto phase = [&](const char* what) {
        if (!trace) return;
        const double t = nowMs();
        fprintf(stderr, "[rsq trace]     device tail: %.3f ms  %s\n", t - tPhase, what);
        tPhase = t;
    };
    if (!q.dtFlags) {
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        // (the work area of the replay on the device is sized for all D groups being present)
        std::vector<std::pair<uint64_t, uint64_t>> lv;
        q.dtReplayBytes = replayLevels((uint64_t)D, opSize(q.agg), lv) ? replayDeviceBytes((uint64_t)D, lv.back().first) : 0;
        q.dtSortTempBytes = radixSortTempBytes(D);
        const size_t sz[] = {up((size_t)(D + 1) * 4), up((size_t)(D + 1) * 8), up(scanTempBytes(D + 1)), up((size_t)D * 8), up((size_t)D * 8), up((size_t)D * 4), up((size_t)D * 4),
                             up(q.dtSortTempBytes), up((size_t)D * 8), up((size_t)D * 4), up((size_t)D * (size_t)q.dtTupleSize)};
        size_t devBytes = up(q.dtReplayBytes); for (size_t b : sz) devBytes += b;
        const size_t psz[] = {up((size_t)D * 8), up((size_t)D * 4), up(std::max<size_t>((size_t)D * (size_t)q.dtTupleSize, 8))};
        size_t pinBytes = 0; for (size_t b : psz) pinBytes += b;
        Context::TailArena& spare = ctx.spareTailArena;
        if (spare.dev && spare.devBytes >= devBytes && spare.pinnedBytes >= pinBytes) { q.dtArena = spare; spare = Context::TailArena(); }
        else {
            q.dtArena.dev = ctx.alloc(devBytes); q.dtArena.devBytes = devBytes;
            q.dtArena.pinned = ctx.allocPinned(pinBytes); q.dtArena.pinnedBytes = pinBytes;
        }
        char* d = (char*)q.dtArena.dev; size_t at = 0; int k = 0;
        auto take = [&]() { void* r = d + at; at += sz[k++]; return r; };
        q.dtFlags = (uint32_t*)take(); q.dtOffs = (uint64_t*)take(); q.dtScanTemp = take();
        q.dtFirst[0] = (uint64_t*)take(); q.dtFirst[1] = (uint64_t*)take(); q.dtGid[0] = (uint32_t*)take(); q.dtGid[1] = (uint32_t*)take();
        q.dtSortTemp = take(); q.dtHashes = (uint64_t*)take(); q.dtOrder = (uint32_t*)take(); q.dtRows = (uint8_t*)take();
        q.dtReplayWork = q.dtReplayBytes ? (void*)(d + at) : nullptr;
        char* h = (char*)q.dtArena.pinned;
        q.hDtHashes = (uint64_t*)h; q.hDtOrder = (uint32_t*)(h + psz[0]); q.resultPinned = (uint8_t*)(h + psz[0] + psz[1]);
    }
    densePresentGroups(ctx, (const int64_t*)(q.dAgg + (size_t)q.accumSlot[0] * (size_t)D), D, q.dtFlags, q.dtOffs, q.dtScanTemp, q.dtFirst[0], q.dtGid[0]);
    uint64_t nPresent = 0;
    RSQ_HIP(hipMemcpyAsync(&nPresent, q.dtOffs + D, 8, hipMemcpyDeviceToHost, ctx.stream));
    RSQ_HIP(hipMemcpyAsync(q.hPinned + q.pinnedWords, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
    waitForStream(ctx);                                    // (this is also where the pipelines' kernels are waited for)
    q.report.num_kernels += 5;
    const double tTail0 = nowMs();
    phase("pipelines done; groups present");
    const int64_t n = (int64_t)nPresent;
    int64_t emit = n;
    if (q.dtLimitRows >= 0) emit = std::min(emit, q.dtLimitRows);
    const uint32_t* dGids = q.dtGid[0];
    const uint32_t* dOrder = nullptr;
    if (ctx.cfg.emission_order != RSQ_EMIT_ANY && n > 1) {
        // first rows are row numbers of the scanned table: the bits they can use
        int64_t maxRow = 1;
        // (over the WHOLE table: the root of a multi-GPU step finalises the merged table, whose first rows come from every shard)
        for (auto& p : q.pipelines) if (p.sink == SinkKind::AGGREGATE) maxRow = std::max<int64_t>(maxRow, std::max(p.src->row0 + p.src->nRows, p.src->totalRows()));
        if (q.firstRowsForeign) maxRow = std::max<int64_t>(maxRow, (int64_t)1 << 40);      // merged from shards this table knows nothing about: all 40 row bits
        int bits = 1; while (bits < 63 && (maxRow >> bits) != 0) bits++;
        const bool inB = radixSortPairs(ctx, q.dtFirst[0], q.dtGid[0], q.dtFirst[1], q.dtGid[1], n, bits, q.dtSortTemp, q.dtSortTempBytes);
        dGids = inB ? q.dtGid[1] : q.dtGid[0];
        denseGroupHashes(ctx, dGids, n, q.dtKeys, q.dtHashes);
        // the replay of the reference's table: on the device too (devtail.hip), unless switched off or out of its range
        std::vector<std::pair<uint64_t, uint64_t>> levels;
        const bool devReplay = !(getenv("RSQ_DEVICE_REPLAY") && atoi(getenv("RSQ_DEVICE_REPLAY")) == 0) && q.dtReplayWork && n >= 4096 &&
                               replayLevels((uint64_t)n, opSize(q.agg), levels) && replayDeviceBytes((uint64_t)n, levels.back().first) <= q.dtReplayBytes;
        if (devReplay) {
            replayEmissionOrderDevice(ctx, q.dtHashes, (uint64_t)n, levels, q.dtReplayWork, q.dtOrder);
            q.report.num_kernels += (uint64_t)((bits + 7) / 8) * 5 + 1 + levels.size() * 12;
            dOrder = q.dtOrder;
            if (trace) { waitForStream(ctx); phase("groups ordered by first row, hashed, the reference's table replayed (device)"); }
        } else {
        RSQ_HIP(hipMemcpyAsync(q.hDtHashes, q.dtHashes, (size_t)n * 8, hipMemcpyDeviceToHost, ctx.stream));
        waitForStream(ctx);
        q.report.num_kernels += (uint64_t)((bits + 7) / 8) * 5 + 1;
        phase("groups ordered by first row, hashed (device), hashes read back");
        std::vector<uint32_t>& order = ctx.replayOrder;
        refEmissionOrderParallel(q.hDtHashes, (size_t)n, opSize(q.agg), order, ctx.replayScratch);
        memcpy(q.hDtOrder, order.data(), (size_t)emit * 4);
        phase("replay of the reference's hash table (host, probe clusters in parallel)");
        RSQ_HIP(hipMemcpyAsync(q.dtOrder, q.hDtOrder, (size_t)emit * 4, hipMemcpyHostToDevice, ctx.stream));
        dOrder = q.dtOrder;
        }
    }
    denseResultRows(ctx, q.dAgg, D, dGids, dOrder, emit, q.dtKeys, q.dtCols, q.dtTupleSize, q.dtRows);
    if (emit > 0) RSQ_HIP(hipMemcpyAsync(q.resultPinned, q.dtRows, (size_t)emit * (size_t)q.dtTupleSize, hipMemcpyDeviceToHost, ctx.stream));
    RSQ_HIP(hipMemcpyAsync(q.hPinned + q.pinnedWords, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
    waitForStream(ctx);
    q.report.num_kernels += 1;
    phase("packed tuples (device), read back");
    q.resultRows = emit;
    q.resultInPinned = true;
    return nowMs() - tTail0;
}

// ---- the tail of a hash / join-entry aggregation with many groups on the device (devtail.hip k_row_*) -------------------------------
// The group rows stay in HBM (q.dGroupRows, n rows of q.groupRowWords words): order by first row (radix sort) -> the reference's hash of
// every group's values -> the replay of the reference's table (device) -> packed tuples in that order -> [the tuples to the host].
// With an ORDER BY above, the host then runs the reference's quicksort over the delivered tuples; without, they are the result.
// The host path this replaces copies all group rows (n x words x 8 bytes over PCIe), decodes them, hashes, replays and builds the
// rows on the worker pool: 5-30 ms per million groups.
bool rowsDeviceTailWanted(Query& q, int64_t n) {
    const bool off = getenv("RSQ_DEVICE_TAIL") && atoi(getenv("RSQ_DEVICE_TAIL")) == 0;
    const int64_t minGroups = getenv("RSQ_DEVICE_TAIL_MIN") ? atoll(getenv("RSQ_DEVICE_TAIL_MIN")) : 65536;
    if (off || q.holdTail || n < minGroups || n >= (1ll << 31) || !q.dGroupRows) return false;
    if (q.aggMode == AggMode::HASH && q.charGroupsNeedMerge) return false;      // groups equal up to trailing spaces: the host merges them first
    if (q.rowTail < 0) q.rowTail = planRowsDeviceTail(q, q.rtKeys, q.rtCols, q.rtTupleSize, q.rtLimitRows, q.rtSorts) ? 1 : 0;
    return q.rowTail == 1;
}

double runRowsDeviceTail(Query& q, int64_t n) {
    Context& ctx = q.ctx;
    const bool trace = getenv("RSQ_TRACE") != nullptr;
    const double t0 = nowMs();
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    if (q.rtCapacity < n) {
        if (q.rtDev) ctx.free(q.rtDev);
        q.rtDev = nullptr; q.rtCapacity = 0;
        const int64_t cap = std::max<int64_t>(n + n / 8, 65536);
        const size_t bytes = up((size_t)cap * 8) * 3 + up((size_t)cap * 4) * 3 + up(radixSortTempBytes(cap)) + up((size_t)cap * (size_t)q.rtTupleSize) +
                             up(replayDeviceBytes((uint64_t)cap, (uint64_t)cap * 4 + 1024));
        q.rtDev = ctx.alloc(bytes);
        q.rtCapacity = cap;
    }
    const int64_t cap = q.rtCapacity;
    char* d = (char*)q.rtDev; size_t at = 0;
    auto take = [&](size_t b) { void* r = d + at; at += up(b); return r; };
    uint64_t* keysA = (uint64_t*)take((size_t)cap * 8); uint64_t* keysB = (uint64_t*)take((size_t)cap * 8); uint64_t* hashes = (uint64_t*)take((size_t)cap * 8);
    uint32_t* idxA = (uint32_t*)take((size_t)cap * 4); uint32_t* idxB = (uint32_t*)take((size_t)cap * 4); uint32_t* order = (uint32_t*)take((size_t)cap * 4);
    void* sortTemp = take(radixSortTempBytes(cap));
    uint8_t* rows = (uint8_t*)take((size_t)cap * (size_t)q.rtTupleSize);
    void* replayWork = d + at;
    const size_t replayBytes = replayDeviceBytes((uint64_t)cap, (uint64_t)cap * 4 + 1024);
    int64_t emit = n;
    const bool limitHere = q.rtLimitRows >= 0 && !q.rtSorts;      // (with an ORDER BY the materialisation's own limit cuts the EMISSION order first, as materialize.h:197-206 does)
    if (q.rtLimitRows >= 0) emit = std::min(emit, q.rtLimitRows);
    (void)limitHere;
    const uint32_t* dIdx = nullptr;
    const uint32_t* dOrder = nullptr;
    const int stride = q.groupRowWords;
    if (ctx.cfg.emission_order != RSQ_EMIT_ANY && n > 1) {
        int64_t maxRow = 1;
        for (auto& p : q.pipelines) if (p.sink == SinkKind::AGGREGATE) maxRow = std::max<int64_t>(maxRow, std::max(p.src->row0 + p.src->nRows, p.src->totalRows()));
        int bits = 1; while (bits < 63 && (maxRow >> bits) != 0) bits++;
        rowTailFirstKeys(ctx, q.dGroupRows, stride, n, keysA, idxA);
        const bool inB = radixSortPairs(ctx, keysA, idxA, keysB, idxB, n, bits, sortTemp, radixSortTempBytes(cap));
        dIdx = inB ? idxB : idxA;
        rowTailHashes(ctx, q.dGroupRows, stride, dIdx, n, q.rtKeys, hashes);
        std::vector<std::pair<uint64_t, uint64_t>> levels;
        if (replayLevels((uint64_t)n, opSize(q.agg), levels) && replayDeviceBytes((uint64_t)n, levels.back().first) <= replayBytes) {
            replayEmissionOrderDevice(ctx, hashes, (uint64_t)n, levels, replayWork, order);
            q.report.num_kernels += (uint64_t)((bits + 7) / 8) * 5 + 2 + levels.size() * 12;
        } else {
            // (a table beyond the device replay's range: the hashes go to the host, the slot order comes back)
            std::vector<uint64_t> hh((size_t)n);
            RSQ_HIP(hipMemcpyAsync(hh.data(), hashes, (size_t)n * 8, hipMemcpyDeviceToHost, ctx.stream));
            waitForStream(ctx);
            std::vector<uint32_t>& ord = ctx.replayOrder;
            refEmissionOrderParallel(hh.data(), (size_t)n, opSize(q.agg), ord, ctx.replayScratch);
            RSQ_HIP(hipMemcpyAsync(order, ord.data(), (size_t)emit * 4, hipMemcpyHostToDevice, ctx.stream));
            RSQ_HIP(hipStreamSynchronize(ctx.stream));      // (ord is the context's scratch: keep it until the copy has read it)
        }
        dOrder = order;
    }
    rowTailResultRows(ctx, q.dGroupRows, stride, dIdx, dOrder, emit, q.rtCols, q.rtTupleSize, rows);
    const size_t outBytes = std::max<size_t>((size_t)emit * (size_t)q.rtTupleSize, 8);
    if (q.rtPinnedBytes < outBytes) {
        if (q.rtPinned) ctx.freePinned(q.rtPinned);
        q.rtPinned = nullptr; q.rtPinnedBytes = 0;
        q.rtPinned = ctx.allocPinned(outBytes + outBytes / 8);
        q.rtPinnedBytes = outBytes + outBytes / 8;
    }
    if (emit > 0) RSQ_HIP(hipMemcpyAsync(q.rtPinned, rows, (size_t)emit * (size_t)q.rtTupleSize, hipMemcpyDeviceToHost, ctx.stream));
    uint32_t err = 0;
    RSQ_HIP(hipMemcpyAsync(&err, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
    waitForStream(ctx);
    q.report.num_kernels += 1;
    if (err) { ctx.errWordClean = false; checkDeviceError(err); }
    if (trace) fprintf(stderr, "[rsq trace]     device tail over %lld group rows: %.3f ms (ordered, hashed, replayed, %lld tuples built and read back)\n", (long long)n, nowMs() - t0, (long long)emit);
    int64_t rowsOut = emit;
    if (q.rtSorts) {
        const double t1 = nowMs();
        runRowsTailSort(q, (uint8_t*)q.rtPinned, rowsOut);
        if (trace) fprintf(stderr, "[rsq trace]     order by over the delivered tuples (the reference's quicksort, host): %.3f ms\n", nowMs() - t1);
    }
    q.resultRows = rowsOut;
    q.resultPinned = (uint8_t*)q.rtPinned;
    q.resultInPinned = true;
    return nowMs() - t0;
}

}  // namespace rsq
