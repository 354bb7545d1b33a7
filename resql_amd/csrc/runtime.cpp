// runtime.cpp — engine context: device, stream, code-object cache (hiprtc JIT), device tables.
//
// "Compilation" in the reference is Flounder IR -> x86 via asmjit, in process
// (reference src/JitContextFlounder.h:410-456).  Here it is pipeline -> HIP source -> gfx950 code
// object via hiprtc, in process, with an on-disk cache so that a pipeline shape is compiled once
// (the cache is pre-populated at build time, __graft_entry__.build()).
#include "engine.h"
#include "kernel_compile_options.h"

#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <hip/hiprtc.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <fstream>
#include <sstream>
#include <thread>

extern char** environ;

namespace rsq {

static std::string libraryDir() {
    Dl_info info;
    if (dladdr((void*)&libraryDir, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t s = p.find_last_of('/');
        return s == std::string::npos ? "." : p.substr(0, s);
    }
    return ".";
}

static uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

Context::Context(const rsq_config& c) : cfg(c), device(c.device) {
    std::string lib = libraryDir();
    includeDir = lib + "/csrc/kernels";
    struct stat st;
    if (stat((includeDir + "/rsq_device.h").c_str(), &st) != 0) includeDir = lib + "/kernels";
    cacheDir = c.kernel_cache_dir ? std::string(c.kernel_cache_dir) : lib + "/_kcache";
    mkdir(cacheDir.c_str(), 0755);
    if (device >= 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n == 0)
            throw Error(RSQ_ERR_DEVICE, "no HIP device available (the engine has no CPU fallback)");
        if (device >= n) throw Error(RSQ_ERR_DEVICE, "device ordinal out of range");
        RSQ_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        RSQ_HIP(hipGetDeviceProperties(&prop, device));
        numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        RSQ_HIP(hipStreamCreateWithFlags(&ownStream, hipStreamNonBlocking));
        stream = ownStream;
        RSQ_HIP(hipMalloc((void**)&dErr, sizeof(uint32_t)));
        RSQ_HIP(hipMemset(dErr, 0, sizeof(uint32_t)));
        RSQ_HIP(hipEventCreate(&ev0));
        RSQ_HIP(hipEventCreate(&ev1));
        driverAlloc = (c.engine_flags & RSQ_ENGINE_DRIVER_ALLOC) != 0;
        planMemoOff = (c.engine_flags & RSQ_ENGINE_NO_PLAN_MEMO) != 0;
        if (!driverAlloc) {
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); totalB = (size_t)64 << 30; freeB = totalB; }
            arenaKeepBytes = c.arena_keep_bytes > 0 ? (size_t)c.arena_keep_bytes : totalB / 8;
            const int dev = device;
            devArena.reset(new Arena([dev](size_t b) -> void* { void* p = nullptr; (void)hipSetDevice(dev); if (hipMalloc(&p, b) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; },
                                     [](void* p) { (void)hipFree(p); }, (size_t)256 << 20, 256));
            pinArena.reset(new Arena([](size_t b) -> void* { void* p = nullptr; if (hipHostMalloc(&p, b, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; },
                                     [](void* p) { (void)hipHostFree(p); }, (size_t)8 << 20, 256));
            pinNcArena.reset(new Arena([](size_t b) -> void* { void* p = nullptr; if (hipHostMalloc(&p, b, hipHostMallocNonCoherent) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; },
                                       [](void* p) { (void)hipHostFree(p); }, (size_t)32 << 20, 256));
            // the first slabs now: a query's first execution on a fresh context must not wait for the driver either
            const size_t want = c.arena_reserve_bytes < 0 ? 0 : c.arena_reserve_bytes > 0 ? (size_t)c.arena_reserve_bytes : std::min<size_t>((size_t)2 << 30, freeB / 4);
            devArena->reserve(want);
            if (c.arena_reserve_bytes >= 0) { pinArena->reserve((size_t)8 << 20); pinNcArena->reserve((size_t)32 << 20); }
        }
    }
}

Context::~Context() {
    if (device >= 0) {
        (void)hipSetDevice(device);
        for (auto& kv : kernels) if (kv.second.module) (void)hipModuleUnload(kv.second.module);
        (void)hipDeviceSynchronize();
        if (dCompactChain) free(dCompactChain);
        if (spareTailArena.dev) free(spareTailArena.dev);
        if (spareTailArena.pinned) freePinned(spareTailArena.pinned);
        for (auto& e : scratchFreeList) free(e.first);
        for (auto& e : scratchLive) free(e.first);
        for (auto& kv : keyIndexes) if (kv.second.dBitmap) free(kv.second.dBitmap);
        keyIndexes.clear();
        if (dErr) (void)hipFree(dErr);
        devArena.reset(); pinArena.reset(); pinNcArena.reset();
        for (hipEvent_t e : eventPool) (void)hipEventDestroy(e);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ownStream) (void)hipStreamDestroy(ownStream);
    }
}

void* Context::scratchAlloc(size_t bytes) {
    size_t best = scratchFreeList.size();
    for (size_t i = 0; i < scratchFreeList.size(); i++)
        if (scratchFreeList[i].second >= bytes && (best == scratchFreeList.size() || scratchFreeList[i].second < scratchFreeList[best].second)) best = i;
    if (best < scratchFreeList.size()) {
        auto e = scratchFreeList[best];
        scratchFreeList.erase(scratchFreeList.begin() + (long)best);
        scratchLive.push_back(e);
        return e.first;
    }
    // nothing cached is large enough: drop the cache first so that the new buffer finds room
    for (auto& e : scratchFreeList) free(e.first);
    scratchFreeList.clear();
    void* p = alloc(bytes);
    scratchLive.push_back({p, bytes});
    return p;
}
void Context::scratchFree(void* p) {
    for (size_t i = 0; i < scratchLive.size(); i++)
        if (scratchLive[i].first == p) {
            scratchFreeList.push_back(scratchLive[i]);
            scratchLive.erase(scratchLive.begin() + (long)i);
            if (scratchFreeList.size() > 2) { free(scratchFreeList.front().first); scratchFreeList.erase(scratchFreeList.begin()); }
            return;
        }
    free(p);
}

void Context::setStream(hipStream_t s, bool callers) {
    if (device < 0) throw Error(RSQ_ERR_DEVICE, "this context has no device (compile-only)");
    RSQ_HIP(hipSetDevice(device));
    RSQ_HIP(hipStreamSynchronize(stream));     // nothing of ours may still be in flight on the stream we leave
    stream = callers ? s : ownStream;          // a caller's stream may be the null stream (0)
}

namespace {
struct StopWatch {
    double& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit StopWatch(double& a) : acc(a) {}
    ~StopWatch() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

void* Context::allocRaw(size_t bytes) {
    allocStats.rawCalls++;
    StopWatch sw(allocStats.rawMs);
    void* p = nullptr;
    RSQ_HIP(hipSetDevice(device));
    hipError_t e = hipMalloc(&p, bytes ? bytes : 256);
    if (e != hipSuccess && devArena) {
        // the arena's free slabs go back to the driver first (what is pending becomes free once the device is idle)
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        devArena->promote(); devArena->trim(0);
        e = hipMalloc(&p, bytes ? bytes : 256);
    }
    RSQ_HIP(e);
    return p;
}
void Context::freeRaw(void* p) { if (p) { allocStats.rawCalls++; StopWatch sw(allocStats.rawMs); (void)hipFree(p); } }

hipEvent_t Context::takeEvent() {
    if (!eventPool.empty()) { hipEvent_t e = eventPool.back(); eventPool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    RSQ_HIP(hipSetDevice(device));
    RSQ_HIP(hipEventCreate(&e));
    return e;
}

void Context::streamDrained() {
    if (devArena && devArena->hasPending()) devArena->promote();
    if (pinArena && pinArena->hasPending()) pinArena->promote();
    if (pinNcArena && pinNcArena->hasPending()) pinNcArena->promote();
}

static void* arenaAlloc(Context& ctx, Arena& a, size_t bytes, const char* what) {
    // ranges freed while kernels were still enqueued become reusable when the stream is idle: ask (1-2 us) before growing
    if (a.hasPending() && hipStreamQuery(ctx.stream) == hipSuccess) ctx.streamDrained();
    if (!a.fitsWithoutGrowing(bytes) && a.hasPending()) {      // a new slab costs more than waiting for what is in flight
        RSQ_HIP(hipStreamSynchronize(ctx.stream));
        ctx.streamDrained();
    }
    void* p = a.alloc(bytes);
    if (!p) {
        // the driver refused a new slab: give back what is wholly free (here and in the device arena), then once more
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        ctx.streamDrained();
        a.trim(0);
        if (ctx.devArena && &a != ctx.devArena.get()) ctx.devArena->trim(0);
        p = a.alloc(bytes);
    }
    if (!p) throw Error(RSQ_ERR_NOMEM, std::string("out of ") + what + " memory: " + std::to_string(bytes) + " bytes asked for, " +
                                       std::to_string(a.slabBytes()) + " held by the context's arena (" + std::to_string(a.usedBytes()) + " in use)");
    return p;
}

void* Context::alloc(size_t bytes) {
    if (!devArena) return allocRaw(bytes);
    allocStats.devCalls++;
    StopWatch sw(allocStats.devMs);
    RSQ_HIP(hipSetDevice(device));
    return arenaAlloc(*this, *devArena, bytes ? bytes : 256, "device");
}
void Context::free(void* p) {
    if (!p) return;
    if (devArena && devArena->free(p)) {
        if (devArena->freeBytes() > arenaKeepBytes + ((size_t)1 << 30)) devArena->trim(arenaKeepBytes);
        return;
    }
    freeRaw(p);
}
void* Context::allocPinned(size_t bytes, bool nonCoherent) {
    allocStats.pinCalls++;
    StopWatch sw(allocStats.pinMs);
    Arena* a = nonCoherent ? pinNcArena.get() : pinArena.get();
    if (!a) {
        void* p = nullptr;
        allocStats.rawCalls++;
        StopWatch raw(allocStats.rawMs);
        RSQ_HIP(hipHostMalloc(&p, bytes ? bytes : 8, nonCoherent ? hipHostMallocNonCoherent : hipHostMallocDefault));
        return p;
    }
    return arenaAlloc(*this, *a, bytes ? bytes : 8, "pinned host");
}
void Context::freePinned(void* p) {
    if (!p) return;
    if (pinArena && pinArena->free(p)) { if (pinArena->freeBytes() > ((size_t)1 << 30)) pinArena->trim((size_t)256 << 20); return; }
    if (pinNcArena && pinNcArena->free(p)) { if (pinNcArena->freeBytes() > ((size_t)1 << 30)) pinNcArena->trim((size_t)256 << 20); return; }
    allocStats.rawCalls++;
    StopWatch raw(allocStats.rawMs);
    (void)hipHostFree(p);
}

void Table::bumpVersion() {
    version++;
    if (ctx) ctx->retireKeyIndexes(uid);
}
void Context::retireKeyIndexes(uint64_t uid) {
    for (auto it = keyIndexes.begin(); it != keyIndexes.end();) {
        if (it->second.uid != uid) { ++it; continue; }
        it->second.retired = true;
        if (it->second.refs <= 0) { if (it->second.dBitmap) free(it->second.dBitmap); it = keyIndexes.erase(it); } else ++it;
    }
}
void Context::releaseKeyIndex(KeyIndex* k) {
    if (!k || --k->refs > 0 || !k->retired) return;
    for (auto it = keyIndexes.begin(); it != keyIndexes.end(); ++it)
        if (&it->second == k) { if (k->dBitmap) free(k->dBitmap); keyIndexes.erase(it); return; }
}

Table::~Table() {
    if (ctx) ctx->retireKeyIndexes(uid);
    if (ctx && ctx->device >= 0) {
        for (auto& c : cols) if (c.owned && c.dptr) ctx->freeRaw(c.dptr);
    } else {
        for (auto& c : cols) if (c.owned && c.dptr) ::free(c.dptr);
    }
}

static bool readFile(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return false;
    std::ostringstream ss; ss << f.rdbuf();
    out = ss.str();
    return true;
}

static const char* const kHiprtcArch = RSQ_HIPRTC_ARCH;
static const char* const kHiprtcOpt = RSQ_HIPRTC_OPT;
static const char* const kHiprtcStd = RSQ_HIPRTC_STD;

static std::string compileWithHiprtc(Context& ctx, const std::string& source) {
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, source.c_str(), "rsq_pipeline.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        throw Error(RSQ_ERR_DEVICE, "hiprtcCreateProgram failed");
    std::string inc = "-I" + ctx.includeDir;
    const char* opts[] = {kHiprtcArch, kHiprtcOpt, kHiprtcStd, inc.c_str()};
    hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0; hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        throw Error(RSQ_ERR_DEVICE, "hiprtc compilation failed:\n" + log);
    }
    size_t n = 0; hiprtcGetCodeSize(prog, &n);
    std::string code(n, '\0');
    hiprtcGetCode(prog, &code[0]);
    hiprtcDestroyProgram(&prog);
    return code;
}

std::string Context::cacheKey(const std::string& source) {
    // the key covers the generated source AND the hand-written header it includes
    if (headerText.empty() && !readFile(includeDir + "/rsq_device.h", headerText))
        throw Error(RSQ_ERR_DEVICE, "cannot read " + includeDir + "/rsq_device.h");
    // ... and what turns them into a code object: compiler version, target, options (a cache that survives a toolchain upgrade
    // must not hand back the old compiler's code)
    static const std::string toolchain = [] {
        int major = 0, minor = 0;
        (void)hiprtcVersion(&major, &minor);
        return std::string("hiprtc ") + std::to_string(major) + "." + std::to_string(minor) + " " + kHiprtcArch + " " + kHiprtcOpt + " " + kHiprtcStd;
    }();
    char hex[32]; snprintf(hex, sizeof hex, "%016llx", (unsigned long long)(fnv1a(source) ^ (fnv1a(headerText) * 0x9E3779B97F4A7C15ull) ^
                                                                               (fnv1a(toolchain) * 0xBF58476D1CE4E5B9ull)));
    return std::string(hex);
}

bool Context::kernelCachedOnDisk(const std::string& source) {
    const std::string key = cacheKey(source);
    std::string code, stored;
    return readFile(cacheDir + "/" + key + ".hsaco", code) && !code.empty() && (!readFile(cacheDir + "/" + key + ".hip", stored) || stored == source);
}
bool Context::kernelCached(const std::string& source) { return kernels.count(cacheKey(source)) || kernelCachedOnDisk(source); }

static void writeCacheEntry(const std::string& cacheDir, const std::string& key, const std::string& code, const std::string& source) {
    // every writer gets its own temporary names (process id + a counter): two compiler threads of one process may build the same
    // key at once (two queries with one plan on a cold cache), and a rename must never publish a file another writer still fills
    static std::atomic<unsigned> writer{0};
    const std::string tag = ".tmp" + std::to_string((long)getpid()) + "_" + std::to_string(writer.fetch_add(1));
    auto publish = [&](const std::string& path, const std::string& bytes) {
        const std::string tmp = path + tag;
        std::ofstream f(tmp, std::ios::binary);
        if (!f.is_open()) return;
        f.write(bytes.data(), (std::streamsize)bytes.size());
        f.close();
        if (!f || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
    };
    publish(cacheDir + "/" + key + ".hip", source);          // (the source first: a code object is only trusted beside its own source)
    publish(cacheDir + "/" + key + ".hsaco", code);
}

// Several kernels at once: one helper process per kernel (kernel_compiler.cpp), because hiprtc serialises inside a process.  The
// helpers are fresh processes started with posix_spawn - children of this one, never a replacement of it -, load hiprtc only and
// touch no GPU; their results arrive in the on-disk cache.  Whatever a helper did not deliver (no helper binary, a spawn that
// failed, a crash) is compiled in process afterwards, so the function always ends with every source cached or an exception.
void Context::compileManyToCache(const std::vector<std::string>& sources) {
    (void)cacheKey("");
    std::vector<std::pair<std::string, const std::string*>> todo;      // (key, source) of what is not cached yet, each key once
    for (const std::string& src : sources) {
        if (kernelCachedOnDisk(src)) continue;
        const std::string key = cacheKey(src);
        bool seen = false;
        for (auto& t : todo) seen = seen || t.first == key;
        if (!seen) todo.emplace_back(key, &src);
    }
    const bool helpersOff = getenv("RSQ_COMPILE_HELPERS") && atoi(getenv("RSQ_COMPILE_HELPERS")) == 0;
    const std::string helper = libraryDir() + "/rsq_kernel_compiler";
    if (todo.size() > 1 && !helpersOff && access(helper.c_str(), X_OK) == 0) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t nProc = std::min<size_t>(todo.size(), std::max<size_t>(1, std::min<size_t>(hw, 32)));
        std::vector<std::vector<std::string>> share(nProc);
        for (size_t i = 0; i < todo.size(); i++) {
            // the source goes first (a code object is only trusted beside its own source), and is what the helper reads
            const std::string path = cacheDir + "/" + todo[i].first + ".hip";
            const std::string tmp = path + ".tmp" + std::to_string((long)getpid()) + "_" + std::to_string(i);
            std::ofstream f(tmp, std::ios::binary);
            f.write(todo[i].second->data(), (std::streamsize)todo[i].second->size());
            f.close();
            if (!f || rename(tmp.c_str(), path.c_str()) != 0) { (void)remove(tmp.c_str()); continue; }
            (void)remove((cacheDir + "/" + todo[i].first + ".err").c_str());
            share[i % nProc].push_back(todo[i].first);
        }
        std::vector<pid_t> pids;
        for (auto& keys : share) {
            if (keys.empty()) continue;
            std::vector<char*> argv;
            argv.push_back(const_cast<char*>(helper.c_str()));
            argv.push_back(const_cast<char*>(includeDir.c_str()));
            argv.push_back(const_cast<char*>(cacheDir.c_str()));
            argv.push_back(const_cast<char*>(kHiprtcArch)); argv.push_back(const_cast<char*>(kHiprtcOpt)); argv.push_back(const_cast<char*>(kHiprtcStd));
            for (auto& k : keys) argv.push_back(const_cast<char*>(k.c_str()));
            argv.push_back(nullptr);
            pid_t pid = 0;
            if (posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv.data(), environ) == 0) pids.push_back(pid);
        }
        // the helpers are waited for with a deadline (hiprtc takes 0.2-2 s per kernel; a helper that hangs must not hold a compile - and the
        // destructor of a query whose compiler thread sits here - for ever): at the deadline they are killed and what they did not deliver is
        // compiled in process below
        {
            const double deadlineS = 120.0 + 10.0 * (double)todo.size();
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<bool> done(pids.size(), false);
            size_t left = pids.size();
            while (left) {
                for (size_t i = 0; i < pids.size(); i++) {
                    if (done[i]) continue;
                    int st = 0;
                    const pid_t r = waitpid(pids[i], &st, WNOHANG);
                    if (r == pids[i] || (r < 0 && errno != EINTR)) { done[i] = true; left--; }
                }
                if (!left) break;
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadlineS) {
                    for (size_t i = 0; i < pids.size(); i++) if (!done[i]) { (void)kill(pids[i], SIGKILL); int st = 0; (void)waitpid(pids[i], &st, 0); }
                    break;
                }
                usleep(2000);
            }
        }
        for (auto& t : todo) {          // a compile error is an error of the kernel text, the same in any process: report it, do not repeat it
            std::string log;
            if (readFile(cacheDir + "/" + t.first + ".err", log)) { (void)remove((cacheDir + "/" + t.first + ".err").c_str()); throw Error(RSQ_ERR_DEVICE, "hiprtc compilation failed:\n" + log); }
        }
    }
    for (auto& t : todo) if (!kernelCachedOnDisk(*t.second)) compileToCache(*t.second);
    { std::lock_guard<std::mutex> g(freshMutex); for (auto& t : todo) freshlyCompiled.insert(t.first); }      // (the report counts them as compiles, not as cache hits)
}

void Context::compileToCache(const std::string& source) {
    // (cacheKey reads headerText, which the calling thread has filled before it started this one)
    const std::string key = cacheKey(source);
    writeCacheEntry(cacheDir, key, compileWithHiprtc(*this, source), source);
}

Kernel& Context::getKernel(const std::string& source, const std::string& entry) {
    const std::string key = cacheKey(source);
    // (the build lists the keys it resolves so that it can remove the code objects of older kernel texts afterwards: __graft_entry__.py)
    if (const char* usedLog = getenv("RSQ_KCACHE_USED_LOG")) {
        if (FILE* f = fopen(usedLog, "a")) { fprintf(f, "%s\n", key.c_str()); fclose(f); }
    }
    auto it = kernels.find(key);
    if (it != kernels.end()) { jitCacheHits++; return it->second; }      // (loaded by an earlier query of this context: a cache hit in the report, like one from disk)

    std::string path = cacheDir + "/" + key + ".hsaco";
    std::string code;
    Kernel k;
    std::string storedSource;
    if (readFile(path, code) && !code.empty() && (!readFile(cacheDir + "/" + key + ".hip", storedSource) || storedSource == source)) {
        bool fresh;
        { std::lock_guard<std::mutex> g(freshMutex); fresh = freshlyCompiled.erase(key) != 0; }
        k.fromCache = !fresh;            // (the source stored beside the code object is compared: a key collision compiles afresh)
        if (fresh) jitCompiles++; else jitCacheHits++;
    } else {
        code = compileWithHiprtc(*this, source);
        jitCompiles++;
        writeCacheEntry(cacheDir, key, code, source);
    }
    if (device >= 0) {
        RSQ_HIP(hipSetDevice(device));
        RSQ_HIP(hipModuleLoadData(&k.module, code.data()));
        RSQ_HIP(hipModuleGetFunction(&k.fn, k.module, entry.c_str()));
    }
    return kernels.emplace(key, k).first->second;
}

void launch(Context& ctx, Kernel& k, unsigned grid, unsigned block, const std::vector<uint64_t>& args, hipEvent_t start, hipEvent_t stop) {
    if (!k.fn) throw Error(RSQ_ERR_DEVICE, "kernel not loaded (context without device)");
    size_t size = args.size() * sizeof(uint64_t);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, (void*)args.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &size,
                      HIP_LAUNCH_PARAM_END};
    if (start || stop) {
        // the events take the timestamps of this dispatch itself: no marker packets around the kernel
        // (the extended launch counts the GLOBAL size in work-items, not workgroups)
        RSQ_HIP(hipExtModuleLaunchKernel(k.fn, grid * block, 1, 1, block, 1, 1, 0, ctx.stream, nullptr, config, start, stop, 0));
        return;
    }
    RSQ_HIP(hipModuleLaunchKernel(k.fn, grid, 1, 1, block, 1, 1, 0, ctx.stream, nullptr, config));
}

}  // namespace rsq
