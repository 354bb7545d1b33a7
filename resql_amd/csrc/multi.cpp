// multi.cpp — one host process, N GPUs: the rsq_multi_* part of include/resql_hip.h.
//
// The reference's execute() fans one compiled function out to config.numThreads worker threads that pull morsels from a
// shared iterator, and joins them (reference src/JitContextFlounder.h:459-487); its aggregation state is one hash table
// all workers reach.  Across GPUs the morsels are row-range shards resident in each GPU's HBM, every GPU runs the same
// compiled pipelines to a partial aggregate table, and the one exchange step is the group-by merge of those tables:
// an RCCL reduce over xGMI to the root GPU — ncclReduce per [min | max | sum] segment, grouped into one launch per GPU —
// issued from this process through communicators made by ncclCommInitAll.  librccl is loaded with dlopen when the first
// multi-GPU handle is created, so that hosts with one GPU (and machines without RCCL) never need it.
//
// A second merge path moves the partial tables with peer copies into the root GPU and reduces them with the engine's own
// merge kernel (no RCCL): chosen with rsq_multi_config.merge = RSQ_MERGE_PEER_COPY, and the only one possible when a device
// ordinal is listed twice — which is how a box with ONE GPU exercises N-shard execution end to end (tests).
//
// Plans that do not end in a dense partial table (joins with many groups, hash aggregation, plain materialisation) run
// whole on every shard in parallel host threads; the shards must then be disjoint in the group key (the caller shards the
// probe-side table on a key boundary and replicates the build sides), and the merge is a host-side ordered
// merge of the (LIMIT-ed) result rows.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <set>
#include <thread>

#include "engine.h"

using namespace rsq;

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl() {
    static RcclApi api;
    if (api.handle) return api;
    // a process that already has an RCCL (PyTorch brings its own) keeps using that one: same SONAME, same handle
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) throw Error(RSQ_ERR_DEVICE, std::string("cannot load librccl: ") + dlerror());
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) throw Error(RSQ_ERR_DEVICE, std::string("librccl lacks ") + n); return p; };
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.Reduce = (decltype(api.Reduce))sym("ncclReduce");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.handle = h;
    return api;
}

#define RSQ_NCCL(call)                                                                                          \
    do {                                                                                                        \
        ncclResult_t _r = (call);                                                                               \
        if (_r != ncclSuccess) throw Error(RSQ_ERR_DEVICE, std::string(#call) + ": " + rccl().GetErrorString(_r)); \
    } while (0)

thread_local std::string g_multiCreateError;

}  // namespace

struct rsq_multi {
    std::vector<Context*> ctxs;
    std::vector<int> devices;
    bool sharedDevice = false;          // a device ordinal is listed more than once (tests on one GPU)
    int merge = RSQ_MERGE_RCCL;
    std::vector<ncclComm_t> comms;      // RCCL mode, one per device
    std::string lastError;
    ~rsq_multi() {
        for (ncclComm_t c : comms) if (c) (void)rccl().CommDestroy(c);
        for (Context* c : ctxs) delete c;
    }
};

struct rsq_multi_query {
    rsq_multi* m = nullptr;
    std::vector<Query*> qs;             // one compiled query per device, same plan
    bool dense = false;
    bool async = false;                 // dense and every pipeline can be enqueued without the host in between (no join builds)
    bool orderedFast = false;           // not dense: shards provably disjoint in a group key + ORDER BY ... LIMIT: merge of the shards' ordered rows
    std::string mergeText;              // which merge this query takes, and why
    int64_t nMin = 0, nMax = 0, nSum = 0;
    int64_t* gathered = nullptr;        // peer-copy mode: [n][words] on the root device
    std::vector<hipEvent_t> ready;      // peer-copy mode: shard i's partial table is complete
    rsq_report report{};
    std::vector<double> shardKernelMs;
    hipEvent_t evMerge0 = nullptr, evMerge1 = nullptr;      // on the root's stream, around the group-by merge of the last execution
    double collectiveMs = 0;
    ~rsq_multi_query() {
        if (evMerge0 && m && !m->ctxs.empty()) { (void)hipSetDevice(m->ctxs[0]->device); (void)hipEventDestroy(evMerge0); (void)hipEventDestroy(evMerge1); }
        if (gathered && m && !m->ctxs.empty()) { (void)hipSetDevice(m->ctxs[0]->device); m->ctxs[0]->free(gathered); }
        for (size_t i = 0; i < ready.size(); i++) if (ready[i]) { (void)hipSetDevice(m->ctxs[i]->device); (void)hipEventDestroy(ready[i]); }
        for (Query* q : qs) destroyQuery(q);
    }
};

namespace {

template <typename F>
int guardedM(rsq_multi* m, F&& f) {
    try { f(); return RSQ_OK; }
    catch (const Error& e) { if (m) m->lastError = e.what(); else g_multiCreateError = e.what(); return e.status; }
    catch (const std::bad_alloc&) { if (m) m->lastError = "out of host memory"; return RSQ_ERR_NOMEM; }
    catch (const std::exception& e) { if (m) m->lastError = e.what(); else g_multiCreateError = e.what(); return RSQ_ERR_INVALID; }
}

void enqueueMergeUntimed(rsq_multi_query& mq);
// the group-by merge of one step, enqueued behind every shard's kernels; returns after enqueueing.  An event pair on the root's
// stream brackets it: rsq_multi_query_collective_ms (from the root's last kernel to the merged table, i.e. including the wait for
// the slowest shard)
void enqueueMerge(rsq_multi_query& mq) {
    rsq_multi& m = *mq.m;
    if (m.ctxs.size() == 1 && m.comms.empty()) return;
    Context& root = *m.ctxs[0];
    RSQ_HIP(hipSetDevice(root.device));
    if (!mq.evMerge0) { RSQ_HIP(hipEventCreate(&mq.evMerge0)); RSQ_HIP(hipEventCreate(&mq.evMerge1)); }
    RSQ_HIP(hipEventRecord(mq.evMerge0, root.stream));
    enqueueMergeUntimed(mq);
    RSQ_HIP(hipSetDevice(root.device));
    RSQ_HIP(hipEventRecord(mq.evMerge1, root.stream));
}
void enqueueMergeUntimed(rsq_multi_query& mq) {
    rsq_multi& m = *mq.m;
    const int n = (int)m.ctxs.size();
    const int64_t words = mq.nMin + mq.nMax + mq.nSum;
    if (n == 1 && m.comms.empty()) return;          // (a one-GPU handle made with RSQ_MERGE_RCCL still runs its collective)
    std::vector<int64_t*> part((size_t)n);
    for (int i = 0; i < n; i++) { int64_t a, b, c; void* p; queryDenseLayout(*mq.qs[(size_t)i], &a, &b, &c, &p); part[(size_t)i] = (int64_t*)p; }
    if (m.merge == RSQ_MERGE_RCCL) {
        // one reduce per non-empty segment to the root (device 0, in place), all of them for all GPUs in ONE group:
        // RCCL launches a single kernel per GPU for the group.  Integer min / max / sum: bit-exact in any order.
        RcclApi& R = rccl();
        const int64_t off[3] = {0, mq.nMin, mq.nMin + mq.nMax};
        const int64_t cnt[3] = {mq.nMin, mq.nMax, mq.nSum};
        const ncclRedOp_t op[3] = {ncclMin, ncclMax, ncclSum};
        RSQ_NCCL(R.GroupStart());
        for (int i = 0; i < n; i++)
            for (int s = 0; s < 3; s++)
                if (cnt[s] > 0)
                    RSQ_NCCL(R.Reduce(part[(size_t)i] + off[s], part[(size_t)i] + off[s], (size_t)cnt[s], ncclInt64, op[s], 0, m.comms[(size_t)i],
                                      m.ctxs[(size_t)i]->stream));
        RSQ_NCCL(R.GroupEnd());
        return;
    }
    // peer copies: every shard's table -> root's gather buffer (ordered behind the shard's kernels by an event), then the
    // engine's merge kernel on the root's stream
    Context& root = *m.ctxs[0];
    for (int i = 0; i < n; i++) {
        Context& c = *m.ctxs[(size_t)i];
        if (i > 0) {
            RSQ_HIP(hipSetDevice(c.device));
            RSQ_HIP(hipEventRecord(mq.ready[(size_t)i], c.stream));
            RSQ_HIP(hipSetDevice(root.device));
            RSQ_HIP(hipStreamWaitEvent(root.stream, mq.ready[(size_t)i], 0));
        }
        RSQ_HIP(hipSetDevice(root.device));
        if (c.device == root.device)
            RSQ_HIP(hipMemcpyAsync(mq.gathered + (size_t)i * (size_t)words, part[(size_t)i], (size_t)words * 8, hipMemcpyDeviceToDevice, root.stream));
        else
            RSQ_HIP(hipMemcpyPeerAsync(mq.gathered + (size_t)i * (size_t)words, root.device, part[(size_t)i], c.device, (size_t)words * 8, root.stream));
    }
    mergePartialsAsync(root, mq.gathered, n, words, mq.nMin, mq.nMax, mq.nSum, part[0]);
}

}  // namespace

extern "C" {

int rsq_multi_create(const rsq_multi_config* hostCfg, rsq_multi** out) {
    if (!out || !hostCfg) return RSQ_ERR_INVALID;
    *out = nullptr;
    return guardedM(nullptr, [&] {
        // the host's struct, struct_size bytes of it (every later field reads as 0), like rsq_config (api.cpp readConfig)
        rsq_multi_config mc{};
        const uint32_t have = hostCfg->struct_size;
        if (have < offsetof(rsq_multi_config, merge) || have > 4096)
            failInvalid("rsq_multi_config.struct_size is " + std::to_string(have) + ": set it to sizeof(rsq_multi_config) (" + std::to_string(sizeof(rsq_multi_config)) + " in this library)");
        memcpy(&mc, hostCfg, std::min<size_t>(have, sizeof mc));
        const rsq_multi_config* cfg = &mc;
        if (cfg->n_devices < 1 || !cfg->devices) failInvalid("rsq_multi_create needs at least one device");
        std::unique_ptr<rsq_multi> m(new rsq_multi());
        std::set<int> distinct;
        for (int i = 0; i < cfg->n_devices; i++) {
            if (cfg->devices[i] < 0) failInvalid("rsq_multi_create needs device ordinals (a compile-only context has no shards to run)");
            m->devices.push_back(cfg->devices[i]);
            distinct.insert(cfg->devices[i]);
        }
        m->sharedDevice = (int)distinct.size() != cfg->n_devices;
        m->merge = cfg->merge == RSQ_MERGE_AUTO ? (m->sharedDevice ? RSQ_MERGE_PEER_COPY : RSQ_MERGE_RCCL) : cfg->merge;
        if (m->merge != RSQ_MERGE_RCCL && m->merge != RSQ_MERGE_PEER_COPY) failInvalid("unknown merge mode");
        if (m->merge == RSQ_MERGE_RCCL && m->sharedDevice)
            failInvalid("an RCCL communicator cannot hold the same device twice: use RSQ_MERGE_PEER_COPY for shards that share a GPU");
        for (int i = 0; i < cfg->n_devices; i++) {
            rsq_config c = readConfig(cfg->base);
            c.device = cfg->devices[i];
            m->ctxs.push_back(new Context(c));
        }
        if (m->merge == RSQ_MERGE_PEER_COPY && !m->sharedDevice)
            for (int i = 1; i < cfg->n_devices; i++) {
                int can = 0;
                RSQ_HIP(hipDeviceCanAccessPeer(&can, m->devices[0], m->devices[(size_t)i]));
                if (can) { RSQ_HIP(hipSetDevice(m->devices[0])); hipError_t e = hipDeviceEnablePeerAccess(m->devices[(size_t)i], 0); if (e != hipSuccess) (void)hipGetLastError(); }
            }
        if (m->merge == RSQ_MERGE_RCCL && (cfg->n_devices > 1 || cfg->merge == RSQ_MERGE_RCCL)) {
            m->comms.assign((size_t)cfg->n_devices, nullptr);
            RSQ_NCCL(rccl().CommInitAll(m->comms.data(), cfg->n_devices, m->devices.data()));
        }
        *out = m.release();
    });
}

void rsq_multi_destroy(rsq_multi* m) { delete m; }

const char* rsq_multi_last_error(const rsq_multi* m) { return m ? m->lastError.c_str() : g_multiCreateError.c_str(); }

int32_t rsq_multi_devices(const rsq_multi* m) { return m ? (int32_t)m->ctxs.size() : 0; }

rsq_ctx* rsq_multi_ctx(rsq_multi* m, int32_t shard) {
    if (!m || shard < 0 || shard >= (int32_t)m->ctxs.size()) return nullptr;
    return reinterpret_cast<rsq_ctx*>(m->ctxs[(size_t)shard]);
}

const char* rsq_multi_merge_name(const rsq_multi* m) {
    if (!m) return "";
    if (m->ctxs.size() == 1) return "single GPU (no exchange)";
    return m->merge == RSQ_MERGE_RCCL ? "RCCL reduce to the root GPU, one grouped launch (min | max | sum segments)"
                                      : "peer copies to the root GPU + merge kernel";
}

void rsq_multi_shard_rows(int64_t n_total, int32_t n_shards, int32_t shard, int64_t* row0, int64_t* n_rows) {
    // equal shards on 128-row tile boundaries (the scan's vector loads want 16-byte aligned column offsets); the last
    // shard takes the remainder — resql_amd/dist.py shard_rows, same numbers
    const int64_t tile = 128;
    const int64_t per = n_shards > 0 ? (n_total / (tile * n_shards)) * tile : 0;
    if (row0) *row0 = (int64_t)shard * per;
    if (n_rows) *n_rows = shard < n_shards - 1 ? per : n_total - per * (n_shards - 1);
}

int rsq_multi_table_generate(rsq_multi* m, int32_t kind, int64_t n_rows_total, double scale_factor, int64_t param, uint64_t seed,
                             rsq_table** out_tables) {
    if (!m || !out_tables || n_rows_total < 0) return RSQ_ERR_INVALID;
    const int n = (int)m->ctxs.size();
    for (int i = 0; i < n; i++) out_tables[i] = nullptr;
    return guardedM(m, [&] {
        for (int i = 0; i < n; i++) {
            int64_t r0, nr;
            rsq_multi_shard_rows(n_rows_total, n, i, &r0, &nr);
            std::unique_ptr<Table> t(new Table());
            generateTable(*m->ctxs[(size_t)i], *t, kind, r0, nr, scale_factor, param, seed);
            out_tables[i] = reinterpret_cast<rsq_table*>(t.release());
        }
    });
}

// Row-range shards whose boundaries fall where `key_column` changes: no value of the clustering key spans two shards, so a
// plan grouped by it (TPC-H Q3: l_orderkey) keeps every group on one GPU.  Every tile boundary of
// rsq_multi_shard_rows is moved forward to the first row whose key differs from the row before it; the rows around a boundary
// are generated (the generator is a function of seed and row number) on the root GPU and read back, a window at a time.
int rsq_multi_table_generate_on_key(rsq_multi* m, int32_t kind, int64_t n_rows_total, double scale_factor, int64_t param, uint64_t seed,
                                    const char* key_column, rsq_table** out_tables) {
    if (!m || !out_tables || n_rows_total < 0 || !key_column) return RSQ_ERR_INVALID;
    const int n = (int)m->ctxs.size();
    for (int i = 0; i < n; i++) out_tables[i] = nullptr;
    return guardedM(m, [&] {
        Context& root = *m->ctxs[0];
        auto snap = [&](int64_t b) -> int64_t {
            if (b <= 0 || b >= n_rows_total) return std::max<int64_t>(0, std::min(b, n_rows_total));
            const int64_t window = 4096;
            int64_t prevKey = 0; bool havePrev = false;
            for (int64_t at = b - 1; at < n_rows_total; at += window) {
                const int64_t cnt = std::min(window, n_rows_total - at);
                Table t;
                generateTable(root, t, kind, at, cnt, scale_factor, param, seed);
                const int c = t.findCol(key_column);
                if (c < 0 || !t.cols[(size_t)c].dptr) failInvalid(std::string("table has no generated column ") + key_column);
                const Type& ty = t.cols[(size_t)c].type;
                if (ty.isString() || (columnWidth(ty) != 4 && columnWidth(ty) != 8)) failUnsupported("the shard key must be an INT / DATE / BIGINT / DECIMAL column");
                std::vector<int64_t> keys((size_t)cnt);
                if (columnWidth(ty) == 4) {
                    std::vector<int32_t> k4((size_t)cnt);
                    RSQ_HIP(hipMemcpy(k4.data(), t.cols[(size_t)c].dptr, (size_t)cnt * 4, hipMemcpyDeviceToHost));
                    for (int64_t j = 0; j < cnt; j++) keys[(size_t)j] = k4[(size_t)j];
                } else RSQ_HIP(hipMemcpy(keys.data(), t.cols[(size_t)c].dptr, (size_t)cnt * 8, hipMemcpyDeviceToHost));
                for (int64_t j = 0; j < cnt; j++) {
                    if (havePrev && keys[(size_t)j] != prevKey) return at + j;
                    prevKey = keys[(size_t)j]; havePrev = true;
                }
            }
            return n_rows_total;
        };
        std::vector<int64_t> bound((size_t)n + 1, n_rows_total);
        bound[0] = 0;
        for (int i = 1; i < n; i++) {
            int64_t r0, nr;
            rsq_multi_shard_rows(n_rows_total, n, i, &r0, &nr);
            bound[(size_t)i] = std::max(bound[(size_t)i - 1], snap(r0));
        }
        for (int i = 0; i < n; i++) {
            std::unique_ptr<Table> t(new Table());
            generateTable(*m->ctxs[(size_t)i], *t, kind, bound[(size_t)i], bound[(size_t)i + 1] - bound[(size_t)i], scale_factor, param, seed);
            out_tables[i] = reinterpret_cast<rsq_table*>(t.release());
        }
    });
}

int rsq_multi_query_compile(rsq_multi* m, const rsq_plan_desc* plan, rsq_table* const* tables, int32_t n_tables, rsq_multi_query** out) {
    if (!m || !plan || !out || n_tables < 0 || (n_tables > 0 && !tables)) return RSQ_ERR_INVALID;
    *out = nullptr;
    return guardedM(m, [&] {
        const int n = (int)m->ctxs.size();
        std::unique_ptr<rsq_multi_query> mq(new rsq_multi_query());
        mq->m = m;
        for (int i = 0; i < n; i++)
            for (int t = 0; t < n_tables; t++) {
                const Table* tb = reinterpret_cast<const Table*>(tables[(size_t)i * (size_t)n_tables + (size_t)t]);
                if (!tb) failInvalid("null table");
                if (tb->ctx != m->ctxs[(size_t)i]) failInvalid("table " + tb->name + " of shard " + std::to_string(i) + " does not live on that shard's context");
            }
        // The reference has ONE relation and ONE hash table all workers reach (aggregation.h:240-295, JitContextFlounder.h:459-487): any
        // data works.  Here every shard plans from column statistics, so the shards of a table first receive the statistics of the WHOLE
        // table (union of the byte-value sets, min / max over the shards, summed row count): all of them then derive the same dense
        // group layout whatever their own rows hold, or all of them the hash aggregation where the union is not dense.  A table whose
        // instances are the same rows on every shard (a replicated build side: equal row range and statistics) is left as it is.
        // (two passes: every table's blobs are made and checked first, and only then is any table changed - a schema that differs on one
        // shard must not leave some tables unified and others not)
        if (n > 1) {
            auto tab = [&](int i, int t) { return reinterpret_cast<Table*>(tables[(size_t)i * (size_t)n_tables + (size_t)t]); };
            std::vector<std::vector<char>> allBlobs((size_t)n_tables);
            std::vector<size_t> blobBytes((size_t)n_tables, 0);
            for (int t = 0; t < n_tables; t++) {
                const size_t bb = tableStatsBytes(*tab(0, t));
                std::vector<char>& blobs = allBlobs[(size_t)t];
                blobs.resize((size_t)n * bb);
                bool replicated = true;
                for (int i = 0; i < n; i++) {
                    if (tableStatsBytes(*tab(i, t)) != bb) failInvalid("table " + tab(i, t)->name + " has another schema on shard " + std::to_string(i));
                    exportTableStats(*tab(i, t), blobs.data() + (size_t)i * bb, bb);
                    if (memcmp(blobs.data(), blobs.data() + (size_t)i * bb, bb) != 0) replicated = false;
                }
                blobBytes[(size_t)t] = replicated ? 0 : bb;
            }
            for (int t = 0; t < n_tables; t++)
                if (blobBytes[(size_t)t])
                    for (int i = 0; i < n; i++) unifyShardStats(*tab(i, t), allBlobs[(size_t)t].data(), n, blobBytes[(size_t)t]);
        }
        for (int i = 0; i < n; i++) mq->qs.push_back(compileQuery(*m->ctxs[(size_t)i], *plan, tables + (size_t)i * (size_t)n_tables, n_tables));
        mq->dense = queryIsDense(*mq->qs[0]);
        if (mq->dense) {
            // (with unified statistics every shard derives the same layout; anything else is a defect of this library, not of the data)
            const std::string l0 = queryPartialLayoutText(*mq->qs[0]);
            for (int i = 1; i < n; i++)
                if (!queryIsDense(*mq->qs[(size_t)i]) || queryPartialLayoutText(*mq->qs[(size_t)i]) != l0)
                    throw Error(RSQ_ERR_RUNTIME, "internal: shards planned from the same statistics disagree on the partial aggregate table layout: shard 0 has \"" + l0 +
                                                 "\", shard " + std::to_string(i) + " has \"" + queryPartialLayoutText(*mq->qs[(size_t)i]) + "\"");
            void* p;
            queryDenseLayout(*mq->qs[0], &mq->nMin, &mq->nMax, &mq->nSum, &p);
            if (n > 1 && m->merge == RSQ_MERGE_PEER_COPY) {
                Context& root = *m->ctxs[0];
                mq->gathered = (int64_t*)root.alloc((size_t)n * (size_t)(mq->nMin + mq->nMax + mq->nSum) * 8);
                mq->ready.assign((size_t)n, nullptr);
                for (int i = 1; i < n; i++) { RSQ_HIP(hipSetDevice(m->ctxs[(size_t)i]->device)); RSQ_HIP(hipEventCreateWithFlags(&mq->ready[(size_t)i], hipEventDisableTiming)); }
            }
            mq->async = true;
            for (Query* q : mq->qs) mq->async = mq->async && queryAsyncCapable(*q);
            mq->mergeText = std::string("dense partial tables: ") + rsq_multi_merge_name(m) + (mq->async ? "" : " (join builds: the shards run on host threads)");
        } else {
            for (int i = 1; i < n; i++) if (queryIsDense(*mq->qs[(size_t)i])) throw Error(RSQ_ERR_RUNTIME, "internal: shards planned from the same statistics disagree on the aggregation strategy");
            // Groups may straddle shard boundaries (the reference has ONE hash table all workers reach, aggregation.h:240-295):
            // the general merge reads every shard's group rows back and re-aggregates them by key before the root's tail runs.
            // Only when the column statistics PROVE that no group lives in two shards (the caller sharded on a boundary of a
            // group key) and the plan ends in ORDER BY ... LIMIT k is the short way taken: every shard's own top k, merged by the
            // sort keys.  RSQ_MULTI_GENERAL_MERGE=1 forces the general merge (tests).
            std::string why;
            const bool disjoint = n > 1 && shardGroupsDisjoint(mq->qs, why);
            const bool forceGeneral = getenv("RSQ_MULTI_GENERAL_MERGE") && atoi(getenv("RSQ_MULTI_GENERAL_MERGE")) != 0;
            mq->orderedFast = disjoint && queryOrderedWithLimit(*mq->qs[0]) && !forceGeneral;
            if (n == 1) mq->mergeText = "single shard";
            else if (mq->orderedFast) mq->mergeText = "ordered merge of the shards' LIMIT-ed rows (" + why + ")";
            else mq->mergeText = "general merge: all shards' group rows re-aggregated by key on the host, then one tail (" +
                                 (forceGeneral ? std::string("forced") : disjoint ? std::string("no ORDER BY ... LIMIT to shorten") : why) + ")";
            if (n > 1 && !mq->orderedFast) for (Query* q : mq->qs) setHoldTail(*q, true);
        }
        mq->shardKernelMs.assign((size_t)n, 0.0);
        *out = mq.release();
    });
}

int rsq_multi_query_execute(rsq_multi_query* mq) {
    if (!mq) return RSQ_ERR_INVALID;
    rsq_multi* m = mq->m;
    return guardedM(m, [&] {
        const int n = (int)m->ctxs.size();
        const double t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        // every shard's execution on its own host thread (plans whose pipelines need the host between kernels: join builds size
        // their tables, hash aggregations grow theirs)
        auto onThreads = [&](bool partialOnly) {
            std::vector<std::string> errs((size_t)n);
            std::vector<int> status((size_t)n, RSQ_OK);
            std::vector<std::thread> th;
            for (int i = 0; i < n; i++)
                th.emplace_back([&, i, partialOnly] {
                    try { executeQuery(*mq->qs[(size_t)i], partialOnly); }
                    catch (const Error& e) { status[(size_t)i] = e.status; errs[(size_t)i] = e.what(); }
                    catch (const std::exception& e) { status[(size_t)i] = RSQ_ERR_RUNTIME; errs[(size_t)i] = e.what(); }
                });
            for (auto& t : th) t.join();
            for (int i = 0; i < n; i++) if (status[(size_t)i] != RSQ_OK) throw Error(status[(size_t)i], "shard " + std::to_string(i) + ": " + errs[(size_t)i]);
        };
        if (mq->dense && mq->async) {
            // fan out: enqueue every shard's pipelines (no host synchronisation), then the merge behind them, then ONE
            // synchronising read-back on the root
            for (int i = 0; i < n; i++) executeQuery(*mq->qs[(size_t)i], true, true);
            enqueueMerge(*mq);
            finalizeQuery(*mq->qs[0]);
            for (int i = 1; i < n; i++) settleAsync(*mq->qs[(size_t)i]);
        } else if (mq->dense) {
            // join builds in front of the dense aggregation (TPC-H Q14-like): the shards run to their partial tables with the
            // host's help, the merge and the root's finalisation follow
            onThreads(true);
            enqueueMerge(*mq);
            finalizeQuery(*mq->qs[0]);
        } else {
            onThreads(false);
            double t1 = now();
            if (n > 1) { if (mq->orderedFast) mergeShardResults(*mq->qs[0], mq->qs); else runTailMerged(*mq->qs[0], mq->qs); }
            mq->report.finalize_time_ms = now() - t1;
        }
        // report: the slowest shard's kernel time (the shards run concurrently), bytes of all shards
        rsq_report r0; queryReport(*mq->qs[0], &r0);
        rsq_report rep = r0;
        rep.bytes_read = 0; rep.num_kernels = 0; rep.kernel_time_ms = 0;
        for (int i = 0; i < n; i++) {
            rsq_report r; queryReport(*mq->qs[(size_t)i], &r);
            rep.bytes_read += r.bytes_read; rep.num_kernels += r.num_kernels;
            rep.kernel_time_ms = std::max(rep.kernel_time_ms, r.kernel_time_ms);
            rep.compilation_time_ms = std::max(rep.compilation_time_ms, r.compilation_time_ms);
            mq->shardKernelMs[(size_t)i] = r.kernel_time_ms;
        }
        if (!mq->dense && n > 1) rep.finalize_time_ms = r0.finalize_time_ms + mq->report.finalize_time_ms;
        mq->collectiveMs = 0;
        if (mq->dense && mq->evMerge0) {
            float ms = 0;
            RSQ_HIP(hipSetDevice(m->ctxs[0]->device));
            if (hipEventSynchronize(mq->evMerge1) == hipSuccess && hipEventElapsedTime(&ms, mq->evMerge0, mq->evMerge1) == hipSuccess) mq->collectiveMs = ms;
        }
        rep.execution_time_ms = now() - t0;
        rep.hbm_gbps = rep.kernel_time_ms > 0 ? (double)rep.bytes_read / (rep.kernel_time_ms * 1e-3) / 1e9 : 0;
        mq->report = rep;
    });
}

int rsq_multi_query_result(rsq_multi_query* mq, rsq_result_view* out) {
    if (!mq || !out) return RSQ_ERR_INVALID;
    return guardedM(mq->m, [&] { queryResult(*mq->qs[0], out); });
}

int rsq_multi_query_report(const rsq_multi_query* mq, rsq_report* out, double* shard_kernel_ms) {
    if (!mq || !out) return RSQ_ERR_INVALID;
    *out = mq->report;
    if (shard_kernel_ms) for (size_t i = 0; i < mq->shardKernelMs.size(); i++) shard_kernel_ms[i] = mq->shardKernelMs[i];
    return RSQ_OK;
}

double rsq_multi_query_collective_ms(const rsq_multi_query* mq) { return mq ? mq->collectiveMs : 0; }

const char* rsq_multi_query_merge_name(const rsq_multi_query* mq) { return mq ? mq->mergeText.c_str() : ""; }

void rsq_multi_query_destroy(rsq_multi_query* mq) { delete mq; }

}  // extern "C"
