// h2d_probe — what the host link and the host cores give for the PCIe-inclusive leg (bench.py `pcie_inclusive`):
//   (1) one pinned H2D copy of 64 scans of 32-byte PCL structs (236 MB), (2) packing them to 16-byte records with T host threads,
//   (3) H2D of the packed records.    hipcc -O3 -o /tmp/h2d_probe tests/probes/h2d_probe.hip -lpthread && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <cstdint>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 64ull * 115200ull;
    unsigned char *h32, *h16; void *d;
    hipHostMalloc((void**)&h32, n * 32, hipHostMallocDefault); hipHostMalloc((void**)&h16, n * 16, hipHostMallocDefault);
    hipMalloc(&d, n * 32);
    memset(h32, 1, n * 32); memset(h16, 1, n * 16);
    hipStream_t st; hipStreamCreate(&st);
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int k = 0; k < 5; ++k) hipMemcpyAsync(d, h32, n * 32, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        double dt = (now() - t0) / 5;
        printf("H2D %zu MB pinned, one copy: %.3f ms = %.1f GB/s\n", n * 32 >> 20, dt * 1e3, n * 32 / dt * 1e-9);
        t0 = now();
        for (int k = 0; k < 5; ++k) for (int c = 0; c < 128; ++c) hipMemcpyAsync((char*)d + c * (n * 32 / 128), h32 + c * (n * 32 / 128), n * 32 / 128, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        dt = (now() - t0) / 5;
        printf("H2D the same in 128 copies: %.3f ms = %.1f GB/s\n", dt * 1e3, n * 32 / dt * 1e-9);
    }
    for (int T : { 1, 2, 4, 8, 16, 32 }) {
        if (T > (int)std::thread::hardware_concurrency()) break;
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([=] {
                const size_t a = n * t / T, b = n * (t + 1) / T;
                for (size_t i = a; i < b; ++i) {
                    const unsigned char* s = h32 + i * 32; unsigned char* o = h16 + i * 16;
                    uint64_t xy; uint32_t z; uint16_t lab; memcpy(&xy, s, 8); memcpy(&z, s + 8, 4); memcpy(&lab, s + 20, 2);
                    const uint32_t w = lab; memcpy(o, &xy, 8); memcpy(o + 8, &z, 4); memcpy(o + 12, &w, 4);
                }
            });
            for (auto& x : th) x.join();
            best = std::min(best, now() - t0);
        }
        printf("pack 32 -> 16 B, %2d threads: %.3f ms (%.1f GB/s read + write)\n", T, best * 1e3, n * 48 / best * 1e-9);
    }
    double t0 = now();
    for (int k = 0; k < 5; ++k) hipMemcpyAsync(d, h16, n * 16, hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);
    double dt = (now() - t0) / 5;
    printf("H2D %zu MB packed: %.3f ms = %.1f GB/s; hardware threads %u\n", n * 16 >> 20, dt * 1e3, n * 16 / dt * 1e-9, std::thread::hardware_concurrency());
    return 0;
}
