// host_parallel.hpp — a few host threads for the packing / unpacking loops of the C-ABI layer.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

namespace vgk {

constexpr unsigned MAX_THREADS = 128;
// run f(i, thread) for i in [0, n) on a few host threads
template <class F> inline void parallel_for(uint32_t n, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned T = std::min<unsigned>(hw ? hw : 1, 32u);
    if (const char* e = std::getenv("VGAMD_HOST_THREADS")) T = (unsigned)std::min<int>(MAX_THREADS, std::max(1, std::atoi(e)));
    if (n < 256 || T <= 1) { for (uint32_t i = 0; i < n; ++i) f(i, 0u); return; }
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> ts;
    for (unsigned t = 0; t < T; ++t) ts.emplace_back([&, t]() {
        for (;;) { const uint32_t b = next.fetch_add(64); if (b >= n) break; for (uint32_t i = b; i < std::min(n, b + 64); ++i) f(i, t); }
    });
    for (auto& t : ts) t.join();
}

// host staging kept on the context between calls: uninitialised storage, so a warm call neither zero-fills nor page-faults
template <class T> struct RawBuf {
    T* p = nullptr; size_t cap = 0;
    T* get(size_t n) { if (n > cap) { std::free(p); cap = n + n / 4 + 64; p = (T*)std::malloc(cap * sizeof(T)); } return p; }
    ~RawBuf() { std::free(p); }
};

}  // namespace vgk
