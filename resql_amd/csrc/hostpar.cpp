// hostpar.cpp — see hostpar.h.
#include "hostpar.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>

#include <pthread.h>

namespace rsq {

namespace {

struct Pool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cvWork, cvDone;
    uint64_t generation = 0;
    const std::function<void(int)>* fn = nullptr;
    int parts = 0;
    std::atomic<int> next{0};
    int pending = 0;                    // workers that have not finished the current region yet
    std::exception_ptr error;
    std::atomic<bool> busy{false};      // one parallel region at a time; whoever finds it taken (another thread's region, or a
                                        // nested call from inside this thread's own region) runs its parts itself
    bool stop = false;

    explicit Pool(int nWorkers) {
        for (int i = 0; i < nWorkers; i++) workers.emplace_back([this] { work(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m); stop = true; generation++; }
        cvWork.notify_all();
        for (auto& t : workers) t.join();
    }
    void drain() {
        for (;;) {
            const int p = next.fetch_add(1, std::memory_order_relaxed);
            if (p >= parts) return;
            try { (*fn)(p); }
            catch (...) { std::lock_guard<std::mutex> g(m); if (!error) error = std::current_exception(); }
        }
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m);
                cvWork.wait(l, [&] { return generation != seen; });
                seen = generation;
                if (stop) return;
            }
            drain();
            bool last;
            { std::lock_guard<std::mutex> g(m); last = --pending == 0; }
            if (last) cvDone.notify_one();
        }
    }
    void run(int nParts, const std::function<void(int)>& f) {
        {
            std::lock_guard<std::mutex> g(m);
            fn = &f; parts = nParts; next.store(0); pending = (int)workers.size(); error = nullptr;
            generation++;
        }
        cvWork.notify_all();
        drain();
        std::unique_lock<std::mutex> l(m);
        cvDone.wait(l, [&] { return pending == 0; });
        fn = nullptr;
        if (error) { std::exception_ptr e = error; error = nullptr; l.unlock(); std::rethrow_exception(e); }
    }
};

int configuredThreads() {
    static const int n = [] {
        if (const char* e = getenv("RSQ_TAIL_THREADS")) return std::max(1, std::min(64, atoi(e)));
        unsigned hw = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(16u, hw ? hw : 1u));
    }();
    return n;
}

// (kept until the process ends: no shutdown-order games with a library that Python unloads late)
std::atomic<Pool*> g_pool{nullptr};
std::once_flag g_poolOnce;
// worker threads do not survive fork(): the child forgets the pool (its threads, mutexes and condition variables belong to the
// parent) and runs serially — a region started in a forked child would otherwise wait for workers that do not exist
std::atomic<bool> g_forked{false};

Pool* pool() {
    if (g_forked.load(std::memory_order_relaxed)) return nullptr;
    std::call_once(g_poolOnce, [] {
        if (configuredThreads() > 1) g_pool.store(new Pool(configuredThreads() - 1));
        pthread_atfork(nullptr, nullptr, [] { g_forked.store(true); });
    });
    return g_pool.load();
}

}  // namespace

int hostThreads() { return configuredThreads(); }

int partsFor(size_t n) {
    if (n < 32768) return 1;
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)hostThreads(), n / 8192));
}

void parallelRun(int parts, const std::function<void(int)>& fn) {
    if (parts <= 0) return;
    Pool* p = parts > 1 ? pool() : nullptr;
    bool expected = false;
    if (!p || !p->busy.compare_exchange_strong(expected, true, std::memory_order_acquire)) { for (int i = 0; i < parts; i++) fn(i); return; }
    struct Release { Pool* p; ~Release() { p->busy.store(false, std::memory_order_release); } } release{p};
    p->run(parts, fn);
}

void parallelRanges(size_t n, int parts, const std::function<void(size_t, size_t, int)>& fn) {
    if (parts <= 1 || n == 0) { fn(0, n, 0); return; }
    const size_t per = (n + (size_t)parts - 1) / (size_t)parts;
    parallelRun(parts, [&](int p) {
        const size_t b = std::min(n, per * (size_t)p), e = std::min(n, b + per);
        fn(b, e, p);
    });
}

void parallelSortIndex(const uint64_t* keys, size_t n, std::vector<uint32_t>& idx, SortScratch& s) {
    idx.resize(n);
    if (n == 0) return;
    const int parts = partsFor(n);
    if (n < 4096) {
        for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
        return;
    }
    if (s.k0.size() < n) { s.k0.resize(n); s.k1.resize(n); s.i0.resize(n); s.i1.resize(n); }
    uint64_t* ka = s.k0.data(); uint64_t* kb = s.k1.data();
    uint32_t* ia = s.i0.data(); uint32_t* ib = s.i1.data();
    std::vector<uint64_t> maxOf((size_t)parts, 0);
    parallelRanges(n, parts, [&](size_t b, size_t e, int p) {
        uint64_t mx = 0;
        for (size_t i = b; i < e; i++) { ka[i] = keys[i]; ia[i] = (uint32_t)i; mx = std::max(mx, keys[i]); }
        maxOf[(size_t)p] = mx;
    });
    uint64_t maxKey = 0;
    for (uint64_t m : maxOf) maxKey = std::max(maxKey, m);
    constexpr int BITS = 11, R = 1 << BITS;
    s.hist.assign((size_t)parts * R, 0);
    for (int shift = 0; shift < 64 && (maxKey >> shift) != 0; shift += BITS) {
        size_t* hist = s.hist.data();
        // per-part digit counts, then positions: digit-major, part-minor — part p's keys of digit d go behind those of the parts before it
        parallelRanges(n, parts, [&](size_t b, size_t e, int p) {
            size_t* h = hist + (size_t)p * R;
            memset(h, 0, R * sizeof(size_t));
            for (size_t i = b; i < e; i++) h[(ka[i] >> shift) & (R - 1)]++;
        });
        size_t pos = 0;
        for (int d = 0; d < R; d++)
            for (int p = 0; p < parts; p++) { size_t c = hist[(size_t)p * R + (size_t)d]; hist[(size_t)p * R + (size_t)d] = pos; pos += c; }
        parallelRanges(n, parts, [&](size_t b, size_t e, int p) {
            size_t* h = hist + (size_t)p * R;
            for (size_t i = b; i < e; i++) { const size_t o = h[(ka[i] >> shift) & (R - 1)]++; kb[o] = ka[i]; ib[o] = ia[i]; }
        });
        std::swap(ka, kb); std::swap(ia, ib);
    }
    parallelRanges(n, parts, [&](size_t b, size_t e, int) { memcpy(idx.data() + b, ia + b, (e - b) * sizeof(uint32_t)); });
}

}  // namespace rsq
