// How fast do host threads WRITE page-locked staging memory?  (tools/gpu_r04_w.sh)  16 threads fill 256 MB with 16-byte stores, and with the
// packers' pattern (200-byte pieces, a 16-byte header in between), into: malloc'd memory, hipHostMalloc default, hipHostMallocNonCoherent.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <emmintrin.h>
static double run(char* dst, const char* src, size_t bytes, int threads, int pattern) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; ++t) ts.emplace_back([=]() {
        const size_t per = bytes / threads; char* d = dst + per * t; const char* s = src + per * t;
        if (pattern == 0) { for (size_t k = 0; k + 16 <= per; k += 16) _mm_storeu_si128((__m128i*)(d + k), _mm_loadu_si128((const __m128i*)(s + k))); }
        else { for (size_t k = 0; k + 216 <= per; k += 216) { for (int b = 0; b < 200; b += 8) std::memcpy(d + k + b, s + k + b, 8); uint32_t h[4] = {(uint32_t)k, 1, 2, 3}; std::memcpy(d + k + 200, h, 16); } }
    });
    for (auto& t : ts) t.join();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    const size_t bytes = 256u << 20; const int T = 16;
    char* src = (char*)malloc(bytes); memset(src, 1, bytes);
    char* a = (char*)malloc(bytes); memset(a, 0, bytes);
    char* b = nullptr; char* c = nullptr;
    if (hipHostMalloc((void**)&b, bytes, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void**)&c, bytes, hipHostMallocNonCoherent) != hipSuccess) { printf("alloc failed\n"); return 1; }
    memset(b, 0, bytes); memset(c, 0, bytes);
    const char* names[3] = {"malloc", "hipHostMalloc default", "hipHostMalloc non-coherent"}; char* bufs[3] = {a, b, c};
    for (int rep = 0; rep < 2; ++rep) for (int p = 0; p < 2; ++p) for (int k = 0; k < 3; ++k)
        printf("rep %d pattern %s  %-28s %7.2f ms for 256 MB with %d threads\n", rep, p ? "pieces" : "stream", names[k], run(bufs[k], src, bytes, T, p), T);
    // and a device copy from each, to see what the link makes of them
    void* d = nullptr; hipMalloc(&d, bytes);
    for (int k = 0; k < 3; ++k) { hipDeviceSynchronize(); auto t0 = std::chrono::steady_clock::now(); hipMemcpy(d, bufs[k], bytes, hipMemcpyHostToDevice); hipDeviceSynchronize();
        printf("H2D from %-28s %7.2f ms\n", names[k], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
    return 0;
}
