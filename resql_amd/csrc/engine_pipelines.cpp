// engine_pipelines.cpp - one pipeline of a query on the device, by sink: kernel arguments and grids, the launches of an aggregation
// (plain, staged, partitioned), of a materialisation (count / scan / write) and of a hash-table build (sizing pass, rank dictionary or
// hash form), and the same pipelines on the interpreter while their kernels compile.  Called from executeQuery (engine.cpp).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <sstream>

#include "engine_internal.h"

namespace rsq {

// ================================================================================================
// execute
// ================================================================================================
uint64_t argValue(Query& q, const Pipeline& p, const ArgSlot& a, int countOnlyTable) {
    if (a.name == "out") return (uint64_t)(uintptr_t)(q.aggPad > 1 && !q.flatRun ? q.dAggWork : q.dAgg);
    if (a.name == "fin_out") return (uint64_t)(uintptr_t)q.finOut;
    if (a.name == "fin_err") return (uint64_t)(uintptr_t)q.finErr;
    if (a.name == "fin_seq") return q.finOut ? q.finSeq : 0;
    if (a.name == "fin_ticket") return (uint64_t)(uintptr_t)(q.finOut ? q.dFinTicket : nullptr);
    if (a.name == "part_counts") return (uint64_t)(uintptr_t)q.dPartCounts;
    if (a.name == "part_start") return (uint64_t)(uintptr_t)q.dPartStart;
    if (a.name == "tile_step") return (uint64_t)q.partTileStep;
    if (a.name == "rec" || a.name == "sp_rec") return q.dPartRecords.empty() ? 0 : (uint64_t)(uintptr_t)q.dPartRecords[0];
    if (a.name == "dbg") {          // RSQ_DEBUG_TAIL: [workgroup][8] device timestamps (100 MHz), printed by the one-launch step
        if (!q.dDebugStamps) { q.dDebugStamps = (uint64_t*)q.ctx.alloc(4096 * 8 * 8); RSQ_HIP(hipMemset(q.dDebugStamps, 0, 4096 * 8 * 8)); }
        return (uint64_t)(uintptr_t)q.dDebugStamps;
    }
    if (a.name == "sp_base") return (uint64_t)(uintptr_t)q.dStageBase;
    if (a.name == "sp_cap") return (uint64_t)(uintptr_t)q.dStageCap;
    if (a.name == "sp_ctl") return (uint64_t)(uintptr_t)q.dStageCtl;
    if (a.name == "sp_counts") return (uint64_t)(uintptr_t)q.dStageCounts;
    if (a.name == "sp_mode") return (uint64_t)q.stageMode;
    if (a.name == "sp_nwg") return (uint64_t)q.stageWorkgroups;
    (void)p;
    if (a.name == "cq_total") return (uint64_t)(uintptr_t)(q.dPipeStats + (&p - q.pipelines.data()));
    if (a.name == "cnt") return (uint64_t)(uintptr_t)q.dMatCnt;
    if (a.name == "tcnt") return (uint64_t)(uintptr_t)q.dMatTileCnt;
    if (a.name == "toffs") return (uint64_t)(uintptr_t)q.dMatOffs;
    if (a.name == "out_limit") return (uint64_t)q.matLimit;
    if (a.name.size() >= 2 && a.name[0] == 'o' && isdigit((unsigned char)a.name[1])) {
        size_t k = (size_t)atoi(a.name.c_str() + 1);
        return k < q.dMatCols.size() ? (uint64_t)(uintptr_t)q.dMatCols[k] : 0;
    }
    if (a.name.compare(0, 2, "ht") == 0) {
        size_t us = a.name.find('_');
        int id = atoi(a.name.substr(2, us - 2).c_str());
        HashTable& h = *q.hashTables[(size_t)id];
        std::string f = a.name.substr(us + 1);
        if (f == "state") return (uint64_t)(uintptr_t)h.dState;
        if (f == "words") return (uint64_t)(uintptr_t)h.dWords;
        if (f == "cap") return (uint64_t)h.capacity;
        if (f == "count") return (uint64_t)(uintptr_t)h.dCount;
        if (f == "acc") return (uint64_t)(uintptr_t)h.dAcc;
        if (f == "countonly") return id == countOnlyTable ? 1ull : 0ull;
        if (f == "bm") return (uint64_t)(uintptr_t)h.dBitmap;
        if (f == "c_bm") return (uint64_t)(uintptr_t)h.dCompBitmap;
        if (f == "rank") return h.rank ? 1ull : 0ull;
        if (f == "ident") return h.identity ? 1ull : 0ull;
        if (f == "dense") return h.dense ? 1ull : 0ull;
        if (f == "direct") return h.direct ? 1ull : 0ull;
        if (f.compare(0, 3, "src") == 0 && isdigit((unsigned char)f[3])) {      // payload word w of a direct table: the build table's column
            const size_t w = (size_t)atoi(f.c_str() + 3) - h.keys.size();
            return h.direct && h.directSrc && w < h.directCols.size() ? (uint64_t)(uintptr_t)h.directSrc->cols[(size_t)h.directCols[w]].dptr : 0ull;
        }
        if (f == "temp") return (uint64_t)(uintptr_t)h.dTemp;
        if (f == "treg") return (uint64_t)h.tempRegion;
        if (f == "tused") return (uint64_t)(uintptr_t)h.dTempUsed;
    }
    return a.value;
}

// Workgroups of `k` a CU holds at a time.  A grid larger than what is resident runs in ROUNDS: the tiles are dealt to the grid's
// waves up front, so the workgroups of the second round start when the first ones end and the launch takes twice as long as its
// work (TPC-H Q3's lineitem pipeline with four tiles in flight: 8 workgroups per CU asked for, fewer resident - device timestamps
// showed a quarter of the workgroups starting 61 us late and the kernel ending at 107 us with the median workgroup done at 66).
int residentWorkgroupsPerCU(Kernel* k, int blockThreads) {
    if (!k || !k->fn) return 1 << 20;
    auto it = k->residentPerCU.find(blockThreads);
    if (it != k->residentPerCU.end()) return it->second;
    int n = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, k->fn, blockThreads, 0) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 1 << 20; }
    // ... and never more than 24 waves per CU: that is what the boxes of this pool hold of these kernels whatever the query above says
    // (workgroup timestamps, RSQ_DEBUG_TAIL=1: of 1 792 256-thread workgroups - 7 per CU, 32-36 VGPRs, 9-17 KB of LDS - 1 536 start at once
    // and 256 when the first ones leave; TPC-H Q12's orders build then ends at 70 us with its median workgroup done at 41).
    n = std::min(n, std::max(1, 1536 / std::max(64, blockThreads)));
    k->residentPerCU[blockThreads] = n;
    return n;
}

unsigned pipelineGrid(const Query& q, const Pipeline& p, bool lazyForm) {
    const int64_t tiles = p.src->nRows >> 7;
    const int wavesPerBlock = p.blockThreads / 64;
    int64_t want = (tiles + (int64_t)wavesPerBlock * p.unroll - 1) / ((int64_t)wavesPerBlock * p.unroll);
    const int64_t maxGrid = p.maxGrid ? (int64_t)p.maxGrid : (int64_t)(lazyForm ? p.gridPerCULazy : p.gridPerCU) * (int64_t)q.ctx.numCUs;
    int64_t grid = std::min<int64_t>(maxGrid * 256 / p.blockThreads, want);
    const bool clamp = true;
    if (clamp && !p.maxGrid) {
        Kernel* k = lazyForm && p.kernelLazy ? p.kernelLazy : p.kernel;
        grid = std::min<int64_t>(grid, (int64_t)residentWorkgroupsPerCU(k, p.blockThreads) * (int64_t)q.ctx.numCUs);
    }
    return (unsigned)std::max<int64_t>(1, grid);
}

void launchPipelineKernel(Query& q, Pipeline& p, Kernel& k, int countOnlyTable, unsigned grid, unsigned block,
                                 hipEvent_t start, hipEvent_t stop) {
    p.lastGrid = grid ? grid : pipelineGrid(q, p);
    std::vector<uint64_t> args;
    for (auto& a : p.args) args.push_back(argValue(q, p, a, countOnlyTable));
    launch(q.ctx, k, p.lastGrid, block ? block : (unsigned)p.blockThreads, args, start, stop);
    q.report.num_kernels++;
}

// The synchronisation at the end of an execution: the stream is queried in a loop for up to 2 ms before the thread blocks
// (hipStreamSynchronize sleeps on the completion signal's interrupt: 10-20 us of wake-up on a sub-millisecond execution).
void waitForStream(Context& ctx) {
    const bool spin = !(getenv("RSQ_POLL") && atoi(getenv("RSQ_POLL")) == 0);
    if (spin) {
        const double deadline = nowMs() + 2.0;
        for (;;) {
            const hipError_t e = hipStreamQuery(ctx.stream);
            if (e == hipSuccess) { ctx.streamDrained(); return; }
            if (e != hipErrorNotReady) RSQ_HIP(e);
            if (nowMs() > deadline) break;
            __builtin_ia32_pause();
        }
    }
    RSQ_HIP(hipStreamSynchronize(ctx.stream));
    ctx.streamDrained();
}

// RSQ_DEBUG_TAIL=1 (measurement only): the device timestamps a pipeline's workgroups left (codegen.cpp finishPipeline)
void debugStamps(Query& q, Pipeline& p) {
    if (!q.dDebugStamps || p.lastGrid == 0 || p.lastGrid > 4096) return;
    bool has = false;
    for (auto& a : p.args) has = has || a.name == "dbg";
    if (!has) return;
    RSQ_HIP(hipStreamSynchronize(q.ctx.stream));
    std::vector<uint64_t> st((size_t)p.lastGrid * 8);
    RSQ_HIP(hipMemcpy(st.data(), q.dDebugStamps, st.size() * 8, hipMemcpyDeviceToHost));
    RSQ_HIP(hipMemset(q.dDebugStamps, 0, st.size() * 8));
    uint64_t t0 = ~0ull;
    for (unsigned w = 0; w < p.lastGrid; w++) if (st[w * 8]) t0 = std::min(t0, st[w * 8]);
    auto dist = [&](int k) {
        std::vector<double> v;
        for (unsigned w = 0; w < p.lastGrid; w++) if (st[w * 8 + k]) v.push_back((double)(st[w * 8 + k] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        char buf[96];
        if (v.empty()) return std::string("-");
        snprintf(buf, sizeof buf, "%.1f / %.1f / %.1f", v.front(), v[v.size() / 2], v.back());
        return std::string(buf);
    };
    unsigned early = 0;
    for (unsigned w = 0; w < p.lastGrid; w++) if (st[w * 8] && st[w * 8] - t0 < 500) early++;
    if (getenv("RSQ_DEBUG_TAIL") && atoi(getenv("RSQ_DEBUG_TAIL")) >= 2) {
        double byMod[8] = {0}, n8[8] = {0}, byQuarter[4] = {0}, n4[4] = {0};
        for (unsigned w = 0; w < p.lastGrid; w++) {
            if (!st[w * 8 + 1]) continue;
            const double t = (double)(st[w * 8 + 1] - t0) / 100.0;
            byMod[w % 8] += t; n8[w % 8]++; byQuarter[(size_t)w * 4 / p.lastGrid] += t; n4[(size_t)w * 4 / p.lastGrid]++;
        }
        fprintf(stderr, "[rsq tail]   mean 'rows done' by workgroup index mod 8:");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %.1f", byMod[i] / std::max(1.0, n8[i]));
        fprintf(stderr, "; by quarter of the grid:");
        for (int i = 0; i < 4; i++) fprintf(stderr, " %.1f", byQuarter[i] / std::max(1.0, n4[i]));
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "[rsq tail] %s, %u workgroups (%u started within 5 us), us since the first one started (first / median / last workgroup): started %s, rows done %s, drains done %s, end %s\n",
            p.entry.c_str(), p.lastGrid, early, dist(0).c_str(), dist(1).c_str(), dist(2).c_str(), dist(3).c_str());
}

// A hash aggregation that turned out to have a handful of groups (TPC-H Q12: 2, Q5: 5) does not need a front table of 512 or 1024
// slots per workgroup - 28 to 45 KB of LDS that leave the scan two or three workgroups per CU.  The same source with RSQ_LC_SLOTS 64
// (3 KB) is compiled when first wanted and launched with the grid its own occupancy allows.
// Taken only where the large table leaves fewer than three workgroups per CU (Q12 at SF10: 0.566 -> 0.532 ms); where three fit already the
// smaller table gains nothing and twice the workgroups flush twice the tables (Q5: 0.566 -> 0.61 ms, measured).
Kernel* fewGroupsKernel(Query& q, Pipeline& p, const std::string& source, const char* form, Kernel* large) {
    if (p.ldsSlots <= 64 || q.aggMode != AggMode::HASH || p.sink != SinkKind::AGGREGATE || q.aggTable < 0) return nullptr;
    const HashTable& h = *q.hashTables[(size_t)q.aggTable];
    if (h.lastCount == 0 || h.lastCount > 16) return nullptr;
    if (residentWorkgroupsPerCU(large, p.blockThreads) >= 3) return nullptr;
    auto it = p.fewGroupKernels.find(form);
    if (it == p.fewGroupKernels.end()) it = p.fewGroupKernels.emplace(form, &q.ctx.getKernel(tierSource(p, "#define RSQ_LC_SLOTS 64\n" + source, q.quickTier), p.entry)).first;
    return it->second;
}
unsigned fewGroupsGrid(Query& q, Pipeline& p, Kernel* k) {
    const int64_t tiles = p.src->nRows >> 7;
    const int wavesPerBlock = p.blockThreads / 64;
    const int64_t want = std::max<int64_t>(1, (tiles + (int64_t)wavesPerBlock * p.unroll - 1) / ((int64_t)wavesPerBlock * p.unroll));
    return (unsigned)std::min<int64_t>(want, (int64_t)std::min(p.unroll >= 3 ? 6 : 8, residentWorkgroupsPerCU(k, p.blockThreads)) * q.ctx.numCUs);
}

void launchPipeline(Query& q, Pipeline& p, int countOnlyTable, bool pass1) {
    Kernel* k = pass1 ? p.kernelPass1 : (q.flatRun && p.kernelFlat ? p.kernelFlat : p.kernel);
    // late column loads (codegen.cpp compactThen): worth it when the previous execution sent few rows to stage 2
    const int64_t lazyDen = 32;
    if (!pass1 && k == p.kernel && !p.sourceLazy.empty() && p.stage2Rows >= 0 && p.stage2Rows * lazyDen < p.src->nRows) {
        if (!p.kernelLazy) p.kernelLazy = &q.ctx.getKernel(p.sourceLazy, p.entry);
        if (Kernel* few = fewGroupsKernel(q, p, p.sourceLazy, "lazy", p.kernelLazy)) launchPipelineKernel(q, p, *few, countOnlyTable, fewGroupsGrid(q, p, few));
        else
        launchPipelineKernel(q, p, *p.kernelLazy, countOnlyTable, pipelineGrid(q, p, true));
        debugStamps(q, p);
        if (getenv("RSQ_TRACE")) fprintf(stderr, "[rsq trace]     %s: late-load form, %u workgroups (%lld rows reached stage 2 last time)\n", p.entry.c_str(), p.lastGrid, (long long)p.stage2Rows);
        return;
    }
    if (Kernel* few = (!pass1 && k == p.kernel) ? fewGroupsKernel(q, p, p.source, "eager", k) : nullptr) launchPipelineKernel(q, p, *few, countOnlyTable, fewGroupsGrid(q, p, few));
    else
    launchPipelineKernel(q, p, *k, countOnlyTable);
    if (getenv("RSQ_TRACE") && p.compact) fprintf(stderr, "[rsq trace]     %s: %u workgroups (%lld rows reached stage 2 last time)\n", p.entry.c_str(), p.lastGrid, (long long)p.stage2Rows);
    debugStamps(q, p);
}

// the small buffers of a partitioned aggregation (allocated when the query is compiled: an allocation inside an execution
// is a pause of the device the execution's events measure)
void prepareStageBuffers(Query& q, const Pipeline& p) {
    Context& ctx = q.ctx;
    const int P = p.partCount;
    const size_t words = (size_t)ctx.numCUs * (size_t)P;
    if (q.partCountsWords < words) {
        if (q.dPartCounts) ctx.free(q.dPartCounts);
        q.dPartCounts = (uint32_t*)ctx.alloc(words * 4);
        q.partCountsWords = words;
    }
    if (!q.dPartStart) {
        q.dPartStart = (uint32_t*)ctx.alloc(((size_t)P + 1) * 4);
        q.dPartTotals = (uint64_t*)ctx.alloc(((size_t)P + 1) * 8);       // [P] column totals, [P] = grand total
    }
    if (!p.staged) return;
    if (!q.dStageBase) {
        q.dStageBase = (uint64_t*)ctx.alloc((size_t)P * 8);
        q.dStageCap = (uint32_t*)ctx.alloc((size_t)P * 4);
        q.dStageCtl = ctx.alloc(32);
        q.hStageLayout = ctx.allocPinned((size_t)P * 12);     // the layout travels from pinned memory: no wait for the copy
    }
    if (q.stageCountsWords < words) {
        if (q.dStageCounts) ctx.free(q.dStageCounts);
        q.stageCountsWords = words;
        q.dStageCounts = (uint32_t*)ctx.alloc(q.stageCountsWords * 4);
    }
}

// Form 3 of a large dense aggregation (rsq_device.h "staged partitioning"): one pass turns the passing rows into packed
// records, region by region, and one workgroup per partition aggregates them in LDS.  `estimate[p]` is the number of records
// expected in partition p (from the sampled counting pass); every workgroup gets the same share of it plus slack.  A region
// that runs full is reported by the pass; the regions are then sized by counting (the same kernel, tickets only) and the
// pipeline remembers to do so.  Returns false when the exact regions would not be worth their memory (form 2 runs instead).
bool runStagedAggregation(Query& q, Pipeline& p, const std::vector<uint64_t>& estimate, bool tentative) {
    Context& ctx = q.ctx;
    const int P = p.partCount;
    const unsigned block = 1024;
    const int64_t tiles = p.src->nRows >> 7;
    const int64_t tilesPerRound = (int64_t)(block / 64) * (p.stagedRows / 2);
    const int64_t rounds = std::max<int64_t>(1, (tiles + tilesPerRound - 1) / tilesPerRound);
    const unsigned nwg = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx.numCUs, rounds));
    const uint64_t lineRecords = (uint64_t)(16 / p.stagedRecWords);
    const size_t recBytes = 8 * (size_t)p.stagedRecWords;
    const bool trace = getenv("RSQ_TRACE") != nullptr;
    prepareStageBuffers(q, p);
    q.stageWorkgroups = nwg;
    std::vector<uint32_t> cap((size_t)P);
    auto layout = [&]() -> uint64_t {          // region(workgroup, p) = base[p] + workgroup * cap[p]; uploads both, returns the records provided for
        // (every earlier copy out of the pinned buffer has been executed: each pass ends in a synchronisation)
        uint64_t* hb = (uint64_t*)q.hStageLayout; uint32_t* hc = (uint32_t*)(hb + P);
        uint64_t pos = 0;
        for (int i = 0; i < P; i++) { hb[i] = pos; hc[i] = cap[(size_t)i]; pos += (uint64_t)cap[(size_t)i] * nwg; }
        RSQ_HIP(hipMemcpyAsync(q.dStageBase, hb, (size_t)P * 8, hipMemcpyHostToDevice, ctx.stream));
        RSQ_HIP(hipMemcpyAsync(q.dStageCap, hc, (size_t)P * 4, hipMemcpyHostToDevice, ctx.stream));
        const size_t need = (size_t)std::max<uint64_t>(pos, 1) * recBytes;
        if (q.dPartRecords.empty() || q.stageRecBytes < need) {
            for (void* r : q.dPartRecords) ctx.scratchFree(r);
            q.dPartRecords.clear();
            q.partRecordCapacity = 0;                          // (form 2 allocates anew should it run later)
            q.dPartRecords.push_back(ctx.scratchAlloc(need));
            q.stageRecBytes = need;
        }
        return pos;
    };
    auto pass = [&](uint32_t mode) {
        q.stageMode = mode;
        launchPipelineKernel(q, p, *p.kernelStagedScatter, -1, nwg, block);
        q.stageMode = 0;
    };
    auto overflowed = [&]() -> bool {
        uint32_t ctl[8] = {0};
        RSQ_HIP(hipMemcpyAsync(ctl, q.dStageCtl, 32, hipMemcpyDeviceToHost, ctx.stream));
        waitForStream(ctx);
        if (trace) fprintf(stderr, "[rsq trace]     staged pass: %u of %lld groups seen, watermark row %llu%s\n", ctl[4], (long long)q.denseGroups,
                           (unsigned long long)(((uint64_t)ctl[3] << 32) | ctl[2]), ctl[5] ? ", a region ran full" : "");
        return ctl[5] != 0;
    };
    auto aggregate = [&]() {
        std::vector<uint64_t> args;
        for (auto& a : p.argsStagedAgg) args.push_back(argValue(q, p, a, -1));
        launch(ctx, *p.kernelStagedAgg, (unsigned)P, 1024u, args);
        q.report.num_kernels++;
    };
    RSQ_HIP(hipMemsetAsync(q.dStageCtl, 0, 32, ctx.stream));
    if (!p.stagedExact) {
        const bool reuse = estimate.empty();
        if (reuse) cap = p.stagedCaps;
        else for (int i = 0; i < P; i++) {
            const uint64_t per = (uint64_t)((double)estimate[(size_t)i] / (double)nwg * 1.15) + 256;
            cap[(size_t)i] = (uint32_t)std::min<uint64_t>((per + lineRecords - 1) / lineRecords * lineRecords, 0xfffffff0ull);
        }
        const uint64_t provided = layout();
        if (trace) fprintf(stderr, "[rsq trace]     staged partitioning: %u workgroups x %d partitions, %zu-byte records, regions for %llu records (%s)\n",
                           nwg, P, recBytes, (unsigned long long)provided, reuse ? "as in the last execution" : tentative ? "from the column statistics" : "sampled");
        pass(0);
        aggregate();                       // (enqueued before the pass's verdict is read: it never reads beyond a region, and is repeated if one ran full)
        if (!overflowed()) { p.stagedCaps = cap; p.stagedCapsRows = p.src->nRows; return true; }
        p.stagedCaps.clear();
        if (reuse || tentative) return false;           // the data changed under the remembered regions / the statistics misled: the caller samples
        p.stagedExact = true;
        RSQ_HIP(hipMemsetAsync((char*)q.dStageCtl + 20, 0, 4, ctx.stream));      // the overflow flag; the tracker's state stays (its table is final)
    }
    // regions sized by counting: the tickets of the same kernel
    for (int i = 0; i < P; i++) cap[(size_t)i] = 0;
    (void)layout();
    pass(1);
    std::vector<uint32_t> counts((size_t)nwg * (size_t)P);
    RSQ_HIP(hipMemcpyAsync(counts.data(), q.dStageCounts, counts.size() * 4, hipMemcpyDeviceToHost, ctx.stream));
    waitForStream(ctx);
    uint64_t records = 0, provided = 0;
    for (int i = 0; i < P; i++) {
        uint64_t mx = 0;
        for (unsigned w = 0; w < nwg; w++) { const uint64_t c = counts[(size_t)w * (size_t)P + (size_t)i]; records += c; mx = std::max(mx, c); }
        cap[(size_t)i] = (uint32_t)std::min<uint64_t>((mx + lineRecords - 1) / lineRecords * lineRecords, 0xfffffff0ull);
        provided += (uint64_t)cap[(size_t)i] * nwg;
    }
    if (trace) fprintf(stderr, "[rsq trace]     staged partitioning, counted: %llu records, regions for %llu\n", (unsigned long long)records, (unsigned long long)provided);
    if (provided > 2 * records + (64ull << 20)) return false;          // one workgroup's rows differ wildly from the others': exact positions (form 2)
    (void)layout();
    pass(0);
    if (overflowed()) failRuntime("internal error: a counted staging region ran full");
    aggregate();
    return true;
}

// Aggregation into a large dense table (see emitDenseAggregation, DENSE_GLOBAL): pick, per execution, between HBM
// atomics (few rows pass the filter) and count -> scatter -> per-partition LDS aggregation (many rows pass).
// RSQ_PARTITION=0 never partitions (decided at compile time), 2 always does (tests), 1 / unset decides from a sample.
void runLargeDenseAggregation(Query& q, Pipeline& p) {
    Context& ctx = q.ctx;
    const int P = p.partCount;
    // count and scatter: ONE 1024-thread workgroup per CU (codegen.cpp explains why), the same grid for both passes —
    // the scatter positions are the prefix sums of exactly these workgroups' counts
    const unsigned block = 1024;
    const int64_t tiles = p.src->nRows >> 7;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx.numCUs, (tiles + 15) / 16));
    const int64_t rows = p.src->nRows;
    const char* forceEnv = getenv("RSQ_PARTITION");
    const bool force = forceEnv && atoi(forceEnv) == 2;
    const bool trace = getenv("RSQ_TRACE") != nullptr;
    if (!force && rows < (4 << 20)) { launchPipeline(q, p, -1); return; }       // small inputs: the extra passes cost more than they save
    const size_t words = (size_t)grid * (size_t)P;
    prepareStageBuffers(q, p);
    std::vector<uint64_t> hostTotals((size_t)P + 1);        // records per partition, [P] = all of them
    auto countPass = [&](int64_t step) -> uint64_t {
        q.partTileStep = step;
        RSQ_HIP(hipMemsetAsync(q.dPartCounts, 0, words * 4, ctx.stream));
        launchPipelineKernel(q, p, *p.kernelPartCount, -1, grid, block);
        partitionOffsets(ctx, q.dPartCounts, (int)grid, P, q.dPartTotals, q.dPartStart, q.dPartTotals + P);
        q.report.num_kernels += 2;
        RSQ_HIP(hipMemcpyAsync(hostTotals.data(), q.dPartTotals, ((size_t)P + 1) * 8, hipMemcpyDeviceToHost, ctx.stream));
        waitForStream(ctx);
        q.partTileStep = 1;
        return hostTotals[(size_t)P];
    };
    const size_t recBytes = p.staged ? 8 * (size_t)p.stagedRecWords : 8 * (1 + p.partRecordInputs.size());
    // A staged attempt whose region ran full has already aggregated what it did stage: rsq_staged_agg stored partial sums into
    // every accumulator block and the tracker holds first rows.  Whatever form runs next — the atomics form ADDS onto the
    // table — must start from the identity image again, the whole table, not just the tracker block.
    auto afterFailedAttempt = [&] {
        RSQ_HIP(hipMemsetAsync(ctx.dErr, 0, 4, ctx.stream));
        RSQ_HIP(hipMemcpyAsync(q.dAgg, q.dAggInit, q.tableWords * 8, hipMemcpyDeviceToDevice, ctx.stream));
    };
    const int64_t step = rows >= (512 << 20) ? 128 : rows >= (64 << 20) ? 32 : rows >= (8 << 20) ? 8 : 1;       // the sampled pass reads every step-th tile
    // the regions that held the last execution's records hold this one's, unless the table changed (then the pass says so)
    if (p.staged && !p.stagedExact && !p.stagedCaps.empty() && p.stagedCapsRows == rows && q.stageWorkgroups != 0) {
        if (runStagedAggregation(q, p, {})) return;
        afterFailedAttempt();
    }
    // A first execution knows the expected selectivity from the column statistics (the estimate behind the late loads): the
    // regions are laid out from it - every partition its even share - without the sampled pass and its two synchronisations
    // (0.3-0.4 ms of a 10 ms shard).  Skewed keys or a wrong estimate overflow a region and take the counted path below.
    const bool sampleAlways = false;
    if (p.staged && !force && !sampleAlways && !p.stagedExact && p.leadPass >= 0.0) {
        const double passing = (double)rows * p.leadPass;
        const double direct = passing * (double)std::max(1, p.partAtomicsPerRow) / 25e9;
        const double parted = (double)rows * (double)p.bytesPerRow / 6e12 + passing * (double)recBytes * 2.0 / 4e12 + 80e-6;
        if (trace) fprintf(stderr, "[rsq trace]     large dense aggregation: ~%.0f of %lld rows expected to pass (column statistics); atomics %.3f ms vs partitioned %.3f ms\n",
                           passing, (long long)rows, direct * 1e3, parted * 1e3);
        if (direct <= parted) { launchPipeline(q, p, -1); return; }
        std::vector<uint64_t> estimate((size_t)P + 1, (uint64_t)(passing / (double)P) + 1);
        if (runStagedAggregation(q, p, estimate, true)) return;
        afterFailedAttempt();
    }
    if (!force || p.staged) {
        const double passing = (double)countPass(step) * (double)step;
        const double direct = passing * (double)std::max(1, p.partAtomicsPerRow) / 25e9;
        const double parted = p.staged ? (double)rows * (double)p.bytesPerRow / 6e12 + passing * (double)recBytes * 2.0 / 4e12 + 80e-6
                                       : ((double)rows * 16.0 + (double)rows * (double)p.bytesPerRow) / 6e12 + passing * (double)recBytes * 2.0 / 4e12 + 60e-6;
        if (trace) fprintf(stderr, "[rsq trace]     large dense aggregation: ~%.0f of %lld rows pass; atomics %.3f ms vs partitioned %.3f ms\n",
                           passing, (long long)rows, direct * 1e3, parted * 1e3);
        if (!force && direct <= parted) { launchPipeline(q, p, -1); return; }
    }
    if (p.staged) {
        std::vector<uint64_t> estimate = hostTotals;
        for (auto& e : estimate) e *= (uint64_t)step;
        if (runStagedAggregation(q, p, estimate)) return;
        afterFailedAttempt();       // (form 2 overwrites every block; the atomics form below does not)
    }
    const uint64_t total = countPass(1);
    if (total >= 0xffffffffull) { launchPipeline(q, p, -1); return; }          // record positions are 32-bit
    if (q.partRecordCapacity < total || q.dPartRecords.empty()) {
        // the record buffer comes from the context's scratch cache: a 15 GB hipMalloc per query costs more than the passes
        for (void* r : q.dPartRecords) ctx.scratchFree(r);
        q.dPartRecords.clear();
        q.partRecordCapacity = std::max<uint64_t>(total, 1);
        q.dPartRecords.push_back(ctx.scratchAlloc((size_t)q.partRecordCapacity * (size_t)recBytes));
    }
    launchPipelineKernel(q, p, *p.kernelPartScatter, -1, grid, block);
    std::vector<uint64_t> args;
    for (auto& a : p.argsPartAgg) args.push_back(argValue(q, p, a, -1));
    launch(ctx, *p.kernelPartAgg, (unsigned)P, 1024u, args);
    q.report.num_kernels++;
}

// count / scan / write (see consumeMaterialize in codegen.cpp)
// Output columns of a materialisation.  A result of up to 256 KB lives in host-mapped pinned memory: the write pass stores its few rows over
// PCIe and the host has them when the stream reports completion (TPC-H Q19: 1107 rows; the blocking copy afterwards cost ~15 us).
void freeMatCols(Query& q) {
    for (size_t c = 0; c < q.dMatCols.size(); c++) {
        if (c < q.hMatMapped.size() && q.hMatMapped[c]) q.ctx.freePinned(q.hMatMapped[c]);
        else if (q.dMatCols[c]) q.ctx.free(q.dMatCols[c]);
    }
    q.dMatCols.clear(); q.hMatMapped.clear();
}
void allocMatCols(Query& q, int64_t capacity) {
    freeMatCols(q);
    q.matCapacity = capacity;
    size_t total = 0;
    for (auto& a : q.matSchema) total += (size_t)capacity * (size_t)columnWidth(a.type);
    bool mapped = total <= (256u << 10) && q.ctx.device >= 0;
    for (auto& a : q.matSchema) {
        const size_t bytes = std::max<size_t>(8, (size_t)capacity * (size_t)columnWidth(a.type));
        void* h = nullptr; void* d = nullptr;
        if (mapped) {
            h = q.ctx.allocPinned(bytes);
            if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) {
                (void)hipGetLastError();
                q.ctx.freePinned(h);
                h = nullptr; mapped = false;            // (this column and the ones behind it: device memory)
            }
        }
        if (h) { q.dMatCols.push_back(d); q.hMatMapped.push_back(h); }
        else { q.dMatCols.push_back(q.ctx.alloc(bytes)); q.hMatMapped.push_back(nullptr); }
    }
}

void materializePipeline(Query& q, Pipeline& p) {
    Context& ctx = q.ctx;
    const int64_t n = p.src->nRows;
    // lane counts for every tile (and the two pseudo-tiles of the rows behind the last whole one), ONE count per tile for the scan
    const int64_t tiles = (n >> 7) + 2;
    const int64_t slots = tiles * 64;
    if (tiles + 1 > 0x7fffffff) failUnsupported("materialisation over more than 2^38 rows in one table");
    if (q.matSlots < slots) {
        if (q.dMatCnt) ctx.free(q.dMatCnt);
        if (q.dMatOffs) ctx.free(q.dMatOffs);
        if (q.dMatTileCnt) ctx.free(q.dMatTileCnt);
        if (q.dScanTemp) ctx.free(q.dScanTemp);
        q.dMatCnt = (uint32_t*)ctx.alloc((size_t)slots * 4);
        q.dMatTileCnt = (uint32_t*)ctx.alloc((size_t)(tiles + 1) * 4);
        q.dMatOffs = (uint64_t*)ctx.alloc((size_t)(tiles + 1) * 8);
        q.scanTempBytes = scanTempBytes(tiles + 1);
        q.dScanTemp = ctx.alloc(q.scanTempBytes);
        q.matSlots = slots;
    }
    // (the count pass writes every lane count it will read back and every whole tile's total: only the pseudo-tiles and the trailing
    // slot - whose offset is the total - need their zero)
    RSQ_HIP(hipMemsetAsync(q.dMatTileCnt + (tiles - 2), 0, 3 * 4, ctx.stream));
    q.matLimit = 0;
    launchPipeline(q, p, -1, true);
    const bool chainedOk = !(getenv("RSQ_SCAN_CHAINED") && atoi(getenv("RSQ_SCAN_CHAINED")) == 0);
    // (the one-launch scan pays from ~8 M counts on; one count per 128 rows means tables beyond a billion rows)
    const bool chained = chainedOk && !q.scanChainedOff && tiles + 1 >= (8ll << 20);
    if (chained) exclusiveScanCountsChained(ctx, q.dMatTileCnt, q.dMatOffs, tiles + 1, q.dScanTemp, q.scanTempBytes);
    else exclusiveScanCounts(ctx, q.dMatTileCnt, q.dMatOffs, tiles + 1, q.dScanTemp, q.scanTempBytes);
    q.report.num_kernels++;
    // A materialisation that has run before does not wait for the total: the write pass runs with the remembered one, the total arrives
    // with the status words (publishStatusAsync word 3) and executeQuery starts over on this careful path if it differs.
    q.dMatTotal = q.dMatOffs + tiles;
    q.matWarmRun = false;
    {
        const bool publish = q.dPinnedDev && !(getenv("RSQ_PUBLISH_STATUS") && atoi(getenv("RSQ_PUBLISH_STATUS")) == 0);
        uint64_t keepWarm = (uint64_t)std::max<int64_t>(q.matLastTotal, 0);
        if (q.matOp->hasLimit) keepWarm = std::min<uint64_t>(keepWarm, (uint64_t)std::max<int64_t>(q.matOp->limit, 1));
        if (q.matLastTotal >= 0 && publish && !chained && !getenv("RSQ_TRACE") && !q.holdTail && !q.dMatCols.empty() && (int64_t)keepWarm <= q.matCapacity) {
            q.matRows = (int64_t)keepWarm;
            q.matLimit = keepWarm;
            q.matWarmRun = true;
            launchPipeline(q, p, -1, false);
            q.report.bytes_read += p.matSkip ? (uint64_t)(p.bytesPerRow * p.src->nRows) + (uint64_t)tiles * 4 : 2 * (uint64_t)(p.bytesPerRow * p.src->nRows);
            return;
        }
    }
    uint64_t total = 0;
    uint32_t scanErr = 0;
    RSQ_HIP(hipMemcpyAsync(&total, q.dMatOffs + tiles, 8, hipMemcpyDeviceToHost, ctx.stream));
    if (chained) RSQ_HIP(hipMemcpyAsync(&scanErr, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
    waitForStream(ctx);
    if (scanErr & 512u) {
        // a look-back of the one-launch scan gave up: its offsets are void.  The three-launch scan from now on, and now.
        q.scanChainedOff = true;
        scanErr &= ~512u;
        RSQ_HIP(hipMemcpyAsync(ctx.dErr, &scanErr, 4, hipMemcpyHostToDevice, ctx.stream));
        exclusiveScanCounts(ctx, q.dMatTileCnt, q.dMatOffs, tiles + 1, q.dScanTemp, q.scanTempBytes);
        RSQ_HIP(hipMemcpyAsync(&total, q.dMatOffs + tiles, 8, hipMemcpyDeviceToHost, ctx.stream));
        waitForStream(ctx);
    }
    q.matLastTotal = (int64_t)total;
    // MaterializeOp with a LIMIT leaves the pipeline once count >= limit, i.e. after max(limit, 1) tuples (materialize.h:197-206)
    uint64_t keep = total;
    if (q.matOp->hasLimit) keep = std::min<uint64_t>(total, (uint64_t)std::max<int64_t>(q.matOp->limit, 1));
    q.matRows = (int64_t)keep;
    if ((int64_t)keep > q.matCapacity || q.dMatCols.empty()) {
        allocMatCols(q, std::max<int64_t>((int64_t)keep, 1));
    }
    q.matLimit = keep;
    launchPipeline(q, p, -1, false);
    // (bytes the passes ask for: both read every row - unless the write pass skips the tiles that counted nothing; then it is the
    // count pass, the counts, and whatever tiles do hold result rows, which the host cannot know: not counted)
    q.report.bytes_read += p.matSkip ? (uint64_t)(p.bytesPerRow * p.src->nRows) + (uint64_t)tiles * 4 : 2 * (uint64_t)(p.bytesPerRow * p.src->nRows);
}

// Form and buffers of a join table from the number of its build rows and whether two of them share a key: what the sizing pass of a
// query's first execution finds out - or what the context's plan memo remembers of an earlier query with the same pipelines over the
// same table versions (engine.cpp applyPlanMemo).  Either way the build re-checks both (NOTE_BUILD_KEYS_NOT_UNIQUE, table full).
void sizeJoinTable(Query& q, Pipeline& p, HashTable& h, uint32_t n, bool dupKeys) {
    Context& ctx = q.ctx;
    const size_t nWords = h.keys.size() + h.payload.size();
    if (h.hasBitmap) {
        h.bmBlocks = h.bmInterleaved ? (h.bmBits + 223) / 224 : (h.bmBits + 255) / 256;
        if (!h.dBitmap) h.dBitmap = (uint32_t*)ctx.alloc((size_t)h.bmBlocks * 32);
    }
    h.buildRows = (int64_t)n; h.dupKeys = h.dupKeys || dupKeys;      // (a dictionary that met duplicates once stays a hash table)
    dupKeys = h.dupKeys;
    // (a table of a few hundred rows stays a hash table: the dictionary's index and placement are two launches of 3-4 us each for
    // entries that sit in one cache line either way - TPC-H Q5's region and nation tables)
    // ... unless it would be a direct table with dense keys (below): that form costs no launch at all (TPC-H's nation: 25 rows, keys 0..24)
    const bool smallDirect = h.directCapable && h.uniqueKnown && h.identityCapable && !h.setOnly && q.aggTable != h.id && h.directSrc == p.src &&
                             (int64_t)n == p.src->nRows && n > 0 && (int64_t)n == h.bmBits;
    h.rank = h.rankCapable && !dupKeys && (n > 1024 || h.setOnly || smallDirect);
    h.identity = h.rank && h.identityCapable && (int64_t)n == p.src->nRows && n > 0;
    // ... and every value of the key range occurs (as many entries as the range has values, all keys different): rank(key) = key - min
    h.dense = h.identity && !h.setOnly && (int64_t)n == h.bmBits;
    // ... and is the build table itself when the statistics said so beforehand (HashTable::directCapable) - unless the aggregation keeps its
    // accumulators beside the entries and makes its group rows from them
    h.direct = h.identity && h.directCapable && h.uniqueKnown && q.aggTable != h.id && h.directSrc == p.src;
    for (int c : h.directCols) h.direct = h.direct && p.src->cols[(size_t)c].dptr != nullptr;
    if (h.direct && !h.dense && !h.keyIndex && !ctx.planMemoOff && h.directKeyCol >= 0) {
        // the context may hold this column's key index already (another query built it, or this statement's previous query did)
        auto it = ctx.keyIndexes.find(Context::KeyIndexKey{p.src->uid, p.src->version, p.src->nRows, p.src->row0, h.directKeyCol});
        if (it != ctx.keyIndexes.end() && !it->second.retired && it->second.bmBlocks == h.bmBlocks && it->second.bmMin == h.bmMin && it->second.bmBits == h.bmBits) {
            if (h.dBitmap) ctx.free(h.dBitmap);
            h.dBitmap = it->second.dBitmap;
            h.keyIndex = &it->second; it->second.refs++;
            h.keyIndexReady = true; h.keyIndexVersion = p.src->version;
        }
    }
    if (h.direct) {
        const std::string note = "join table ht" + std::to_string(h.id) + ": the build table's own columns (" +
                                 (h.dense ? "dense keys in row order: nothing is built" : "keys in row order: only the key bitmap and its index are built") + ")\n";
        if (q.explainText.find(note) == std::string::npos) q.explainText += note;
    }
    if (h.rank && h.setOnly) {
        h.capacity = std::max<int64_t>(64, ((int64_t)n + 63) & ~(int64_t)63);          // (nothing is allocated: the bitmap is the table)
    } else if (h.rank) {
        // a dictionary that carries aggregates scatters them with a multiplicative bijection: power-of-two capacity
        const int64_t capMul = 1;
        h.capacity = q.aggTable == h.id ? nextPow2(std::max<int64_t>(64, (int64_t)n * capMul)) : std::max<int64_t>(64, ((int64_t)n + 63) & ~(int64_t)63);
        const int64_t nChunks = (h.bmBlocks + RSQ_RANK_CHUNK_BLOCKS - 1) / RSQ_RANK_CHUNK_BLOCKS;
        // arrival-order buffer: one region per wave of the largest grid this pipeline launches, four times the mean
        // number of records per wave of the smallest one
        const int64_t wpb = p.blockThreads / 64;
        if (!p.sourceLazy.empty() && !p.kernelLazy) p.kernelLazy = &ctx.getKernel(p.sourceLazy, p.entry);      // (both forms' grids are final below)
        const int64_t wavesMax = (int64_t)std::max(pipelineGrid(q, p, false), pipelineGrid(q, p, true)) * wpb;
        const int64_t wavesMin = (int64_t)std::min(pipelineGrid(q, p, false), pipelineGrid(q, p, true)) * wpb;
        h.tempWaves = wavesMax;
        h.tempRegion = h.identity ? 64 : ((4 * (int64_t)n / std::max<int64_t>(1, wavesMin) + 64 + 63) / 64) * 64;      // (identity: nothing is appended)
        if (!h.direct) {      // (direct: no entries, no arrival buffer ...
            h.dWords = (int64_t*)ctx.alloc((size_t)h.capacity * 8 * std::max<size_t>(1, nWords));
            h.dTemp = (int64_t*)ctx.alloc((size_t)h.tempWaves * (size_t)h.tempRegion * 8 * std::max<size_t>(1, nWords));
        }
        if (!(h.direct && h.dense)) {      // ... and with dense keys no index either)
            h.dTempUsed = (uint32_t*)ctx.alloc((size_t)h.tempWaves * 4);
            h.dChunkTotal = (uint32_t*)ctx.alloc((size_t)nChunks * 4);
            h.dChunkBase = (uint32_t*)ctx.alloc((size_t)(nChunks + 1) * 4);
        }
    } else {
        h.capacity = nextPow2(std::max<int64_t>(1024, 2 * (int64_t)n));
        if (!h.keyCas) h.dState = (uint32_t*)ctx.alloc((size_t)h.capacity * 4);
        h.dWords = (int64_t*)ctx.alloc((size_t)h.capacity * 8 * std::max<size_t>(1, nWords));
    }
    if (q.aggTable == h.id) h.dAcc = (int64_t*)ctx.alloc((size_t)h.capacity * 8 * (size_t)h.nAccBlocks);
    if (getenv("RSQ_TRACE")) fprintf(stderr, "[rsq trace]     ht%d: %s, %u build rows\n", h.id, h.direct ? (h.dense ? "the build table's own columns (dense keys in row order: nothing is built)" : "the build table's own columns (keys in row order: only the key bitmap and its index are built)") : h.identity ? "bitmap-rank dictionary, entries in row order" : h.rank ? "bitmap-rank dictionary" : "hash table", n);
}

// size (by a counting pass of the same pipeline), allocate and clear a join table, then build it
void buildHashTable(Query& q, Pipeline& p) {
    Context& ctx = q.ctx;
    HashTable& h = *q.hashTables[(size_t)p.buildTable];
    const size_t nWords = h.keys.size() + h.payload.size();
    size_t bmWords = 0;
    if (h.hasBitmap) {
        // whole 32-byte blocks: 256 bits, or [rank word | 224 bits] for a table that may become a rank dictionary
        h.bmBlocks = h.bmInterleaved ? (h.bmBits + 223) / 224 : (h.bmBits + 255) / 256;
        bmWords = (size_t)h.bmBlocks * 8;
        if (!h.dBitmap) h.dBitmap = (uint32_t*)ctx.alloc(bmWords * 4);
    }
    if (h.hasCompBitmap && !h.dCompBitmap) {      // zeroed once: bits are only ever set (HashTable::hasCompBitmap)
        const size_t cbWords = (size_t)((h.cbBits + 31) / 32) + 1;
        h.dCompBitmap = (uint32_t*)ctx.alloc(cbWords * 4);
        RSQ_HIP(hipMemsetAsync(h.dCompBitmap, 0, cbWords * 4, ctx.stream));
    }
    // a bare scan of a table whose key column is known to be unique (strictly ascending, engine-owned): all rows go in, each under its own
    // key - what the sizing pass would find (TPC-H Q12 builds on all 15 M orders: the pass was a second scan of the table and a host round trip)
    if (h.capacity == 0 && h.uniqueKnown && !h.dupKeys) sizeJoinTable(q, p, h, (uint32_t)p.src->nRows, false);
    if (h.capacity == 0) {
        // sizing pass.  For a table that could be a rank dictionary the pass also sets the key bits and notes a bit that was
        // already set (two build rows with one key): only then does the table stay a hash table.
        if (h.rankCapable) prepareTableAsync(ctx, nullptr, 0, 0, nullptr, 0, h.dBitmap, bmWords, h.dCount);
        else RSQ_HIP(hipMemsetAsync(h.dCount, 0, 4, ctx.stream));
        launchPipeline(q, p, h.id);                     // counting pass
        uint32_t n = 0, err = 0;
        RSQ_HIP(hipMemcpyAsync(&n, h.dCount, 4, hipMemcpyDeviceToHost, ctx.stream));
        RSQ_HIP(hipMemcpyAsync(&err, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
        waitForStream(ctx);
        if (err & 64u) { const uint32_t cleared = err & ~64u; RSQ_HIP(hipMemcpy(ctx.dErr, &cleared, 4, hipMemcpyHostToDevice)); }
        sizeJoinTable(q, p, h, n, (err & 64u) != 0);
    }
    // (an execution whose tables are all sized readies them in its first fill launch — prologueFills below — and h.prepared says so)
    const bool prepared = h.prepared;
    h.prepared = false;
    if (h.direct && h.dense) return;      // the build table's columns are the entries and key - min is the row: nothing to ready, build or index
    if (h.direct && h.keyIndexReady) {
        if (h.keyIndexVersion == p.src->version) return;      // ... or the key bits and their index stand as an earlier execution left them
        // (the table changed under a compiled query: its own bitmap again, built by every execution as before)
        if (h.keyIndex) { ctx.releaseKeyIndex(h.keyIndex); h.keyIndex = nullptr; h.dBitmap = (uint32_t*)ctx.alloc(bmWords * 4); }
        h.keyIndexReady = false;
    }
    // (direct with gaps in the key range - TPC-H's o_orderkey -: the build below only sets the key bits, the index follows, no records are kept)
    if (h.rank && h.setOnly) {
        if (!prepared) { prepareTableAsync(ctx, nullptr, 0, 0, nullptr, 0, h.dBitmap, bmWords, h.dCount); q.report.num_kernels++; }
        launchPipeline(q, p, -1);
        q.report.bytes_read += (uint64_t)(p.bytesPerRow * p.src->nRows);
        return;
    }
    if (h.rank) {
        // ONE launch clears the bitmap and both counters' words; the records then arrive in the append buffer, the bitmap
        // becomes the index, the records move to their entries
        if (!prepared) { prepareTableAsync(ctx, nullptr, 0, 0, h.dTempUsed, (size_t)h.tempWaves, h.dBitmap, bmWords, h.dCount); q.report.num_kernels++; }
        launchPipeline(q, p, -1);
        // the index in one launch (chunk totals chained between the workgroups) when the prologue has zeroed the chain words
        const bool chainedOk = !(getenv("RSQ_RANK_CHAINED") && atoi(getenv("RSQ_RANK_CHAINED")) == 0);
        if (prepared && chainedOk && !q.chainedIndexOff) { rankTableIndexChained(ctx, h.dBitmap, h.bmBlocks, h.dChunkTotal, h.dChunkBase); q.report.num_kernels += 1; }
        else { rankTableIndex(ctx, h.dBitmap, h.bmBlocks, h.dChunkTotal, h.dChunkBase); q.report.num_kernels += 2; }
        if (!h.identity) {      // (identity: the build wrote every record to the entry with its row's number)
            rankTablePlace(ctx, h.dTemp, h.dTempUsed, (uint32_t)h.tempWaves, (uint32_t)h.tempRegion, h.dCount, (int)std::max<size_t>(1, nWords), h.dBitmap, h.bmMin,
                           h.bmBits, h.dChunkBase, h.bmBlocks, h.dWords, h.capacity);
            q.report.num_kernels += 1;
        }
        q.report.bytes_read += (uint64_t)(p.bytesPerRow * p.src->nRows);
        return;
    }
    // ONE launch readies the table: every key word = EMPTY (with a slot's words next to each other that is a fill of the whole
    // table; the payload words are overwritten by the inserts) or the state words = 0, the key bitmap and the entry counter = 0
    if (!prepared) {
        if (h.keyCas) prepareTableAsync(ctx, (uint64_t*)h.dWords, (size_t)h.capacity * (h.aos ? std::max<size_t>(1, nWords) : 1), 0x8000000000000000ull,
                                        nullptr, 0, h.dBitmap, bmWords, h.dCount);
        else prepareTableAsync(ctx, nullptr, 0, 0, h.dState, (size_t)h.capacity, h.dBitmap, bmWords, h.dCount);
        q.report.num_kernels++;
    }
    launchPipeline(q, p, -1);
    q.report.bytes_read += (uint64_t)(p.bytesPerRow * p.src->nRows);
}

void checkDeviceError(uint32_t err) {
    err &= ~(32u | 64u | 128u | 256u | 512u);  // NOTE_CHAR_GROUP_ENDS_WITH_SPACE / NOTE_BUILD_KEYS_NOT_UNIQUE / the chained index's time-out are information for the host, not errors
    if (err & 1) failRuntime("Division by zero");
    if (err & 2) failRuntime("Hash table full");
    if (err & 16) failRuntime("internal error: a hash-table slot stayed in the 'being written' state");
    if (err & 8) failRuntime("a key lies outside the column statistics the query was compiled with: the table's data changed after the table was created "
                             "(adopted device columns must stay immutable, see rsq_table_create_device)");
    if (err) failRuntime("device error word " + std::to_string(err));
}

// An asynchronous step cannot start over (its caller has already enqueued the merge behind it), so it only ever runs plans
// without join tables (executeQuery refuses the others) and NOTE_BUILD_KEYS_NOT_UNIQUE — "a rank dictionary dropped build rows,
// go back to the hash form and repeat" — cannot be raised; should it be set all the same, the step fails instead of returning
// an answer with missing matches.
void checkAsyncDeviceError(uint32_t err) {
    if (err & 64u) failRuntime("internal error: an asynchronous step reported build keys that are not unique (no join tables can run asynchronously)");
    checkDeviceError(err);
}

// the dense aggregate table at the start / end of an execution: register-mode kernels work on the padded copy
void enqueueTableInit(Query& q) {
    q.fusedReady = false;          // the plain path leaves the working table as the kernels left it
    if (q.aggPad > 1 && !q.flatRun) RSQ_HIP(hipMemcpyAsync(q.dAggWork, q.dAggWorkInit, q.padWords * 8, hipMemcpyDeviceToDevice, q.ctx.stream));
    else RSQ_HIP(hipMemcpyAsync(q.dAgg, q.dAggInit, q.tableWords * 8, hipMemcpyDeviceToDevice, q.ctx.stream));
}
void enqueueTableReadback(Query& q) {
    if (q.aggPad > 1 && !q.flatRun) RSQ_HIP(hipMemcpyAsync(q.hPinned, q.dAggWork, q.padWords * 8, hipMemcpyDeviceToHost, q.ctx.stream));
    else RSQ_HIP(hipMemcpyAsync(q.hPinned, q.dAgg, q.tableWords * 8, hipMemcpyDeviceToHost, q.ctx.stream));
}
void tableFromPinned(Query& q) {
    q.hAggView = nullptr;
    if (q.aggPad > 1 && !q.flatRun) for (size_t i = 0; i < q.tableWords; i++) q.hAgg[i] = q.hPinned[i * (size_t)q.aggPad];
    else if (q.tableWords >= (1u << 16)) q.hAggView = q.hPinned;      // tens of MB: the tail reads the pinned buffer itself (a copy is 3 ms of one core)
    else memcpy(q.hAgg.data(), q.hPinned, q.tableWords * 8);
}

// ---- the interpreter for whole pipelines (generic2.cpp, generic_kernels.hip) ---------------------------------------------
// Its join / aggregation tables live in the HashTable objects the specialised kernels use, in the interpreter's one layout:
// state[cap], words[cap][keys + payload], acc[block][cap] — which is also what the entry compaction and everything behind it
// read (HashTable::aos, no rank dictionary).  When the specialised kernels take over, the tables are dropped and sized afresh.
void dropTable(Context& ctx, HashTable& h) {
    for (void* p : {(void*)h.dState, (void*)h.dWords, (void*)h.dAcc, (void*)h.dTemp, (void*)h.dTempUsed, (void*)h.dChunkTotal, (void*)h.dChunkBase})
        if (p) ctx.free(p);
    h.dState = nullptr; h.dWords = nullptr; h.dAcc = nullptr; h.dTemp = nullptr; h.dTempUsed = nullptr; h.dChunkTotal = nullptr; h.dChunkBase = nullptr;
    h.capacity = 0; h.lastCount = 0; h.rank = false; h.prepared = false;
}
void leaveGeneric2(Query& q) {
    for (size_t i = 0; i < q.hashTables.size(); i++) {
        dropTable(q.ctx, *q.hashTables[i]);
        if (i < q.savedAos.size()) q.hashTables[i]->aos = q.savedAos[i];
    }
    for (auto& p : q.pipelines) p.stage2Rows = -1;
}

void generic2Launch(Query& q, size_t pi, int matPass) {
    Context& ctx = q.ctx;
    Pipeline& p = q.pipelines[pi];
    GenericProgram2& gp = q.generic2Progs[pi];
    GenericPipelineLaunch L;
    memset(&L, 0, sizeof L);
    L.prog = &gp; L.dCode = gp.dCode; L.dProbes = gp.dProbes; L.dConstPool = gp.dConstPool;
    for (size_t t = 0; t < q.hashTables.size() && t < G2_MAX_TABLES; t++) {
        HashTable& h = *q.hashTables[t];
        L.tables[t] = GenericTableRef{h.dState, h.dWords, h.dAcc, (uint64_t)h.capacity, h.dCount, (int)std::max<size_t>(1, h.keys.size() + h.payload.size())};
    }
    L.nRows = p.src->nRows; L.row0 = p.src->row0;
    L.matCnt = q.dG2Cnt; L.matOffs = q.dG2Offs; L.matLimit = q.matLimit; L.matPass = matPass;
    for (size_t c = 0; c < q.dMatCols.size() && c < G2_MAX_OUT; c++) L.matOut[c] = q.dMatCols[c];
    L.dense = q.dAgg; L.denseGroups = q.denseGroups;
    launchGenericPipeline(ctx, L);
    q.report.num_kernels++;
}

void runGeneric2Pipeline(Query& q, size_t pi) {
    Context& ctx = q.ctx;
    Pipeline& p = q.pipelines[pi];
    const GenericSinkDesc& S = q.generic2Progs[pi].sink;
    auto identityOfBlock = [&](int b) { return b < q.nMinBlocks ? 0x7fffffffffffffffull : b < q.nMinBlocks + q.nMaxBlocks ? 0x8000000000000000ull : 0ull; };
    auto clearAcc = [&](HashTable& h) {
        std::vector<FillItem> f;
        for (int b = 0; b < h.nAccBlocks; b++) f.push_back(FillItem{(uint64_t*)h.dAcc + (size_t)b * (size_t)h.capacity, (size_t)h.capacity * 8, identityOfBlock(b)});
        fillBatchAsync(ctx, f.data(), (int)f.size());
        q.report.num_kernels++;
    };
    switch (S.kind) {
        case G2_SINK_BUILD: {
            HashTable& h = *q.hashTables[(size_t)S.table];
            const size_t NW = std::max<size_t>(1, h.keys.size() + h.payload.size());
            // no sizing pass: twice the scanned rows always hold the entries (a cold first execution may be generous with memory)
            const int64_t cap = nextPow2(std::max<int64_t>(64, 2 * p.src->nRows));
            if (h.capacity != cap || !h.dState || !h.dWords) {
                dropTable(ctx, h);
                h.capacity = cap;
                h.dState = (uint32_t*)ctx.alloc((size_t)cap * 4);
                h.dWords = (int64_t*)ctx.alloc((size_t)cap * 8 * NW);
                if (q.aggTable == h.id && q.aggMode == AggMode::AT_JOIN_ENTRY) h.dAcc = (int64_t*)ctx.alloc((size_t)cap * 8 * (size_t)h.nAccBlocks);
            }
            h.aos = true; h.rank = false;
            RSQ_HIP(hipMemsetAsync(h.dState, 0, (size_t)cap * 4, ctx.stream));
            RSQ_HIP(hipMemsetAsync(h.dCount, 0, 4, ctx.stream));
            generic2Launch(q, pi, 0);
            break;
        }
        case G2_SINK_DENSE:
            generic2Launch(q, pi, 0);              // (executeQuery has put the identity image into q.dAgg: this execution is "flat")
            break;
        case G2_SINK_ENTRY: {
            HashTable& h = *q.hashTables[(size_t)S.table];
            if (!h.dAcc || h.capacity == 0) failRuntime("internal error: the aggregation's join table was not built");
            clearAcc(h);
            generic2Launch(q, pi, 0);
            break;
        }
        case G2_SINK_HASH: {
            HashTable& h = *q.hashTables[(size_t)S.table];
            const size_t NW = std::max<size_t>(1, h.keys.size() + h.payload.size());
            h.aos = true;
            if (h.capacity == 0) h.capacity = nextPow2(std::max<int64_t>(4096, 4 * (int64_t)opSize(q.agg, true)));
            while ((int64_t)h.lastCount * 2 > h.capacity) { if (h.dState) { ctx.free(h.dState); ctx.free(h.dWords); ctx.free(h.dAcc); } h.dState = nullptr; h.dWords = nullptr; h.dAcc = nullptr; h.capacity *= 2; }
            for (;;) {
                if (!h.dState) {
                    h.dState = (uint32_t*)ctx.alloc((size_t)h.capacity * 4);
                    h.dWords = (int64_t*)ctx.alloc((size_t)h.capacity * 8 * NW);
                    h.dAcc = (int64_t*)ctx.alloc((size_t)h.capacity * 8 * (size_t)h.nAccBlocks);
                }
                RSQ_HIP(hipMemsetAsync(h.dState, 0, (size_t)h.capacity * 4, ctx.stream));
                RSQ_HIP(hipMemsetAsync(h.dCount, 0, 4, ctx.stream));
                clearAcc(h);
                generic2Launch(q, pi, 0);
                uint32_t err = 0;
                RSQ_HIP(hipMemcpyAsync(&err, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
                waitForStream(ctx);
                q.charGroupsNeedMerge = (err & 32u) != 0;
                if (!(err & 2)) break;
                if (h.capacity >= ((int64_t)1 << 31)) failRuntime("Hash table full");
                ctx.free(h.dState); ctx.free(h.dWords); ctx.free(h.dAcc);
                h.dState = nullptr; h.dWords = nullptr; h.dAcc = nullptr;
                h.capacity *= 4;
                RSQ_HIP(hipMemsetAsync(ctx.dErr, 0, 4, ctx.stream));
            }
            break;
        }
        case G2_SINK_MATERIALIZE: {
            // count per row / exclusive scan / write: the rows keep scan order (materialize.h:78-220 appends with one thread)
            const int64_t n = p.src->nRows;
            if (n + 1 > 0x7fffffff) failUnsupported("materialisation over more than 2 G rows in the interpreter");
            if (q.g2CntRows < n + 1) {
                if (q.dG2Cnt) { ctx.free(q.dG2Cnt); ctx.free(q.dG2Offs); ctx.free(q.dG2ScanTemp); }
                q.dG2Cnt = (uint32_t*)ctx.alloc((size_t)(n + 1) * 4);
                q.dG2Offs = (uint64_t*)ctx.alloc((size_t)(n + 1) * 8);
                q.dG2ScanTemp = ctx.alloc(scanTempBytes(n + 1));
                q.g2CntRows = n + 1;
            }
            RSQ_HIP(hipMemsetAsync(q.dG2Cnt, 0, (size_t)(n + 1) * 4, ctx.stream));
            q.matLimit = 0;
            generic2Launch(q, pi, 1);
            exclusiveScanCounts(ctx, q.dG2Cnt, q.dG2Offs, n + 1, q.dG2ScanTemp, scanTempBytes(n + 1));
            q.report.num_kernels += 3;
            uint64_t total = 0;
            RSQ_HIP(hipMemcpyAsync(&total, q.dG2Offs + n, 8, hipMemcpyDeviceToHost, ctx.stream));
            waitForStream(ctx);
            uint64_t keep = total;
            if (q.matOp->hasLimit) keep = std::min<uint64_t>(total, (uint64_t)std::max<int64_t>(q.matOp->limit, 1));      // materialize.h:197-206
            q.matRows = (int64_t)keep;
            if ((int64_t)keep > q.matCapacity || q.dMatCols.empty()) {
                allocMatCols(q, std::max<int64_t>((int64_t)keep, 1));
            }
            q.matLimit = keep;
            generic2Launch(q, pi, 2);
            break;
        }
        default: failRuntime("internal error: interpreter sink");
    }
    q.report.bytes_read += (uint64_t)(p.bytesPerRow * p.src->nRows);
}

}  // namespace rsq
