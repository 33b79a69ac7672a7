// host_parallel.hpp — a few host threads for the packing / unpacking loops of the C-ABI layer.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <unistd.h>
#include <sched.h>
#include <cstdio>
#include <cstdint>
#include <immintrin.h>

namespace vgk {

constexpr unsigned MAX_THREADS = 128;

// CPUs this process may actually use: the affinity mask, cut by the cgroup's CPU quota when there is one (a container on a
// 256-thread host is often given 16 CPUs' worth of time; starting 48 threads there only adds context switches)
inline unsigned usable_cpus() {
    static const unsigned n = [] {
        unsigned hw = std::thread::hardware_concurrency();
        if (!hw) hw = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) hw = std::min<unsigned>(hw, (unsigned)c); }
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota|max> <period>"
            char q[32] = {0}; long long period = 0;
            if (std::fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm' && period > 0) {
                const long long quota = std::atoll(q);
                if (quota > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
            }
            std::fclose(f);
        } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
            long long quota = -1, period = 0;
            if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
            std::fclose(g);
            if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(h, "%lld", &period) != 1) period = 0; std::fclose(h); }
            if (quota > 0 && period > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
        return hw;
    }();
    return n;
}

// Worker threads that stay around between calls: a pack / unpack loop over a thousand problems takes less time than starting
// 32 threads does.  One job at a time; a caller that finds the pool busy (several callers packing at once) starts its own
// threads instead, and a process that was forked after the pool came up gets a new pool.
class WorkerPool {
public:
    // two pools: a caller that packs the next batch while another thread fetches the previous one finds the second one free
    static WorkerPool& get(int which = 0) {
        static std::mutex guard;
        static std::unique_ptr<WorkerPool> pools[2];
        std::unique_ptr<WorkerPool>& pool = pools[which & 1];
        std::lock_guard<std::mutex> lock(guard);
        if (!pool || pool->owner != getpid()) { if (pool) (void)pool.release(); pool.reset(new WorkerPool()); }    // a forked child cannot use (or join) the parent's threads
        return *pool;
    }
    static bool run_on_any(unsigned T, const std::function<void(unsigned)>& job) { return get(0).try_run(T, job) || get(1).try_run(T, job); }
    // runs job(t) for t in [0, T) — t = 0 on the calling thread — and returns when all are done; false = pool busy, nothing ran
    // A loop body that itself calls parallel_for finds inside_job() set and runs its inner loop on its own thread: the pool's submit
    // lock is not re-entrant.  A body that throws (bad_alloc from a growing vector, say) is caught on whichever thread it ran; the
    // call still waits for every worker before the first exception is rethrown on the caller, so no worker is left running a job
    // whose captures are gone.
    static bool& inside_job() { static thread_local bool in = false; return in; }
    bool try_run(unsigned T, const std::function<void(unsigned)>& job) {
        if (inside_job()) return false;
        std::unique_lock<std::mutex> one(submit, std::try_to_lock);
        if (!one.owns_lock()) return false;
        {
            std::lock_guard<std::mutex> lock(m);
            while (threads.size() + 1 < T) { const unsigned id = (unsigned)threads.size() + 1; threads.emplace_back([this, id] { work(id); }); }
            current = &job; want = T; pending = T - 1; failed = nullptr; ++epoch;
        }
        cv_start.notify_all();
        std::exception_ptr mine;
        inside_job() = true;
        try { job(0); } catch (...) { mine = std::current_exception(); }
        inside_job() = false;
        std::exception_ptr first;
        {
            std::unique_lock<std::mutex> lock(m);
            cv_done.wait(lock, [this] { return pending == 0; });
            current = nullptr;
            first = mine ? mine : failed; failed = nullptr;
        }
        if (first) std::rethrow_exception(first);
        return true;
    }
    ~WorkerPool() {
        if (owner != getpid()) { for (auto& t : threads) t.detach(); return; }
        { std::lock_guard<std::mutex> lock(m); stop = true; ++epoch; }
        cv_start.notify_all();
        for (auto& t : threads) t.join();
    }
private:
    WorkerPool() : owner(getpid()) {}
    void work(unsigned id) {
        uint64_t seen;
        { std::lock_guard<std::mutex> lock(m); seen = epoch - 1; }       // the job that created this thread is its first
        for (;;) {
            const std::function<void(unsigned)>* job = nullptr;
            {
                std::unique_lock<std::mutex> lock(m);
                cv_start.wait(lock, [&] { return epoch != seen; });
                seen = epoch;
                if (stop) return;
                if (id < want) job = current;
            }
            if (job) {
                std::exception_ptr err;
                inside_job() = true;
                try { (*job)(id); } catch (...) { err = std::current_exception(); }
                inside_job() = false;
                std::lock_guard<std::mutex> lock(m);
                if (err && !failed) failed = err;
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    pid_t owner;
    std::mutex submit, m;
    std::condition_variable cv_start, cv_done;
    std::vector<std::thread> threads;
    const std::function<void(unsigned)>* current = nullptr;
    uint64_t epoch = 0; unsigned want = 0, pending = 0; bool stop = false;
    std::exception_ptr failed;                     // the first exception a worker's share of the current job threw
};

// threads of the call's own (the pools are busy): the same contract as WorkerPool::try_run — every thread is joined before the first
// exception any of them threw is rethrown on the caller
inline void run_on_own_threads(unsigned T, const std::function<void(unsigned)>& body) {
    std::mutex em; std::exception_ptr first;
    auto guarded = [&](unsigned t) {
        WorkerPool::inside_job() = true;
        try { body(t); } catch (...) { std::lock_guard<std::mutex> lock(em); if (!first) first = std::current_exception(); }
        WorkerPool::inside_job() = false;
    };
    std::vector<std::thread> ts;
    for (unsigned t = 1; t < T; ++t) ts.emplace_back(guarded, t);
    guarded(0);
    for (auto& t : ts) t.join();
    if (first) std::rethrow_exception(first);
}

// run f(i, thread) for i in [0, n) on a few host threads
template <class F> inline void parallel_for(uint32_t n, F f) {
    unsigned hw = usable_cpus();
    unsigned T = std::min<unsigned>(hw ? hw : 1, 48u);       // never more than 48: beyond that the loops are bound by memory, not by threads
    if (const char* e = std::getenv("VGAMD_HOST_THREADS")) T = (unsigned)std::min<int>(MAX_THREADS, std::max(1, std::atoi(e)));
    if (n < 256 || T <= 1 || WorkerPool::inside_job()) { for (uint32_t i = 0; i < n; ++i) f(i, 0u); return; }      // (a nested loop runs on the thread that reached it)
    T = std::min<unsigned>(T, (n + 63) / 64);                      // no more threads than blocks of work
    std::atomic<uint32_t> next{0};
    const std::function<void(unsigned)> body = [&](unsigned t) {
        for (;;) { const uint32_t b = next.fetch_add(64); if (b >= n) break; for (uint32_t i = b; i < std::min(n, b + 64); ++i) f(i, t); }
    };
    if (T <= 1) { body(0); return; }
    if (WorkerPool::run_on_any(T, body)) return;
    run_on_own_threads(T, body);                                     // the pool is busy with another caller's loop
}

// a few coarse tasks (slices of a sort, say) on the same threads: task(i) for i in [0, count), whatever the count
template <class F> inline void parallel_tasks(uint32_t count, F task) {
    unsigned hw = usable_cpus();
    unsigned T = std::min<unsigned>(hw ? hw : 1, 48u);
    if (const char* e = std::getenv("VGAMD_HOST_THREADS")) T = (unsigned)std::min<int>(MAX_THREADS, std::max(1, std::atoi(e)));
    T = std::min<unsigned>(T, count);
    if (T <= 1 || WorkerPool::inside_job()) { for (uint32_t i = 0; i < count; ++i) task(i); return; }
    std::atomic<uint32_t> next{0};
    const std::function<void(unsigned)> body = [&](unsigned) { for (;;) { const uint32_t i = next.fetch_add(1); if (i >= count) break; task(i); } };
    if (WorkerPool::run_on_any(T, body)) return;
    run_on_own_threads(T, body);
}

// the same over fixed chunks of the index range: f(lo, hi, chunk) — for two-level prefix sums and reductions
constexpr uint32_t CHUNK = 512;
inline uint32_t chunk_count(uint32_t n) { return (n + CHUNK - 1) / CHUNK; }
template <class F> inline void parallel_chunks(uint32_t n, F f) {      // (coarse tasks: parallel_for would run fewer than 256 of them on the calling thread)
    parallel_tasks(chunk_count(n), [&](uint32_t c) { const uint32_t lo = c * CHUNK; f(lo, std::min<uint32_t>(n, lo + CHUNK), c); });
}

// base -> code (A C G T = 0 1 2 3, everything else 4), sixteen bases per step with SSE2 — part of every x86-64 — or thirty-two with AVX2
// where the processor has it.  A switch or a table lookup per base is most of a nanosecond, and the packers code hundreds of megabytes per
// batch; the last, partial vector goes through a 16-byte buffer instead of a byte loop (a tail of 13 bases cost as much as 60 whole ones).
// FOLD: case-insensitive (gssw_create_nt_table, for reads); graph bases are taken as they are (after nonATGCNtoN, src/aligner.cpp:39:
// upper-case ACGT only).
template <bool FOLD> inline __m128i code16(__m128i b) {
    if (FOLD) b = _mm_and_si128(b, _mm_set1_epi8((char)0xdf));
    __m128i r = _mm_set1_epi8(4);                                  // 4, minus (4 - code) where a base matches
    r = _mm_sub_epi8(r, _mm_and_si128(_mm_cmpeq_epi8(b, _mm_set1_epi8('A')), _mm_set1_epi8(4)));
    r = _mm_sub_epi8(r, _mm_and_si128(_mm_cmpeq_epi8(b, _mm_set1_epi8('C')), _mm_set1_epi8(3)));
    r = _mm_sub_epi8(r, _mm_and_si128(_mm_cmpeq_epi8(b, _mm_set1_epi8('G')), _mm_set1_epi8(2)));
    r = _mm_sub_epi8(r, _mm_and_si128(_mm_cmpeq_epi8(b, _mm_set1_epi8('T')), _mm_set1_epi8(1)));
    return r;
}
template <bool FOLD> inline void code_bases_sse2(uint8_t* __restrict dst, const char* __restrict src, size_t n) {
    size_t k = 0;
    for (; k + 16 <= n; k += 16) _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + k), code16<FOLD>(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + k))));
    if (k < n) {
        alignas(16) char in[16] = {0}; alignas(16) uint8_t out[16];
        __builtin_memcpy(in, src + k, n - k);
        _mm_store_si128(reinterpret_cast<__m128i*>(out), code16<FOLD>(_mm_load_si128(reinterpret_cast<const __m128i*>(in))));
        __builtin_memcpy(dst + k, out, n - k);
    }
}
template <bool FOLD> __attribute__((target("avx2"))) inline void code_bases_avx2(uint8_t* __restrict dst, const char* __restrict src, size_t n) {
    size_t k = 0;
    const __m256i four = _mm256_set1_epi8(4), fold = _mm256_set1_epi8((char)0xdf);
    const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
    const __m256i dC = _mm256_set1_epi8(3), dG = _mm256_set1_epi8(2), dT = _mm256_set1_epi8(1);
    for (; k + 32 <= n; k += 32) {
        __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + k));
        if (FOLD) b = _mm256_and_si256(b, fold);
        __m256i r = four;
        r = _mm256_sub_epi8(r, _mm256_and_si256(_mm256_cmpeq_epi8(b, cA), four)); r = _mm256_sub_epi8(r, _mm256_and_si256(_mm256_cmpeq_epi8(b, cC), dC));
        r = _mm256_sub_epi8(r, _mm256_and_si256(_mm256_cmpeq_epi8(b, cG), dG)); r = _mm256_sub_epi8(r, _mm256_and_si256(_mm256_cmpeq_epi8(b, cT), dT));
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + k), r);
    }
    code_bases_sse2<FOLD>(dst + k, src + k, n - k);
}
template <bool FOLD> inline void code_bases(uint8_t* __restrict dst, const char* __restrict src, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 32) code_bases_avx2<FOLD>(dst, src, n); else code_bases_sse2<FOLD>(dst, src, n);
}

// uninitialised array: the threads that fill it also fault its pages in (a std::vector would zero it on one thread first)
template <class T> struct RawArray {
    T* p = nullptr; size_t n = 0;
    bool alloc(size_t count) { std::free(p); p = (T*)std::malloc((count ? count : 1) * sizeof(T)); n = p ? count : 0; return p != nullptr; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T* data() { return p; } const T* data() const { return p; }
    size_t size() const { return n; }
    T* begin() { return p; } T* end() { return p + n; }
    const T* begin() const { return p; } const T* end() const { return p + n; }
    RawArray() = default; RawArray(const RawArray&) = delete; RawArray& operator=(const RawArray&) = delete;
    ~RawArray() { std::free(p); }
};

// host staging kept on the context between calls: uninitialised storage, so a warm call neither zero-fills nor page-faults
template <class T> struct RawBuf {
    T* p = nullptr; size_t cap = 0;
    T* get(size_t n) { if (n > cap) { std::free(p); cap = n + n / 4 + 64; p = (T*)std::malloc(cap * sizeof(T)); } return p; }
    ~RawBuf() { std::free(p); }
};

}  // namespace vgk
