#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_atom(int* tab, uint32_t mask_n, int n, uint32_t* out, int spread)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    uint32_t idx = spread ? (h % mask_n) : ((uint32_t)i % mask_n);
    out[i] = (uint32_t)atomicAdd(&tab[idx], 1);
}
__global__ void k_lds(const float4* pts, int n, int* tab, uint32_t nbins, uint32_t* out)
{
    __shared__ int h[4096];
    for (int k = threadIdx.x; k < 4096; k += 256) h[k] = 0;
    __syncthreads();
    int base = blockIdx.x * 4096;
    uint32_t r[16];
    for (int k = 0; k < 16; ++k) {
        int i = base + k * 256 + threadIdx.x;
        uint32_t hh = (uint32_t)i * 2654435761u; hh ^= hh >> 15; hh *= 2246822519u; hh ^= hh >> 13;
        r[k] = atomicAdd(&h[hh & 4095], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 4096; k += 256) { int c = h[k]; if (c) h[k] = atomicAdd(&tab[(blockIdx.x % 32) * 4096 + k], c); }
    __syncthreads();
    for (int k = 0; k < 16; ++k) {
        int i = base + k * 256 + threadIdx.x;
        uint32_t hh = (uint32_t)i * 2654435761u; hh ^= hh >> 15; hh *= 2246822519u; hh ^= hh >> 13;
        out[i] = r[k] + h[hh & 4095];
    }
}
int main()
{
    const int n = 12800000;
    int* tab; uint32_t* out;
    hipMalloc(&tab, 200u << 20); hipMalloc(&out, sizeof(uint32_t) * n);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double mbs[] = { 0.4, 3, 13, 26, 50, 100, 172 };
    for (int spread = 1; spread >= 0; --spread)
    for (double mb : mbs) {
        uint32_t cells = (uint32_t)(mb * 1e6 / 4);
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemsetAsync(tab, 0, (size_t)cells * 4, 0);
            hipEventRecord(a, 0);
            k_atom<<<(n + 255) / 256, 256>>>(tab, cells, n, out, spread);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%s table %.1f MB: %.3f ms for %d atomics\n", spread ? "random" : "linear", mb, best, n);
    }
    { float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipMemsetAsync(tab, 0, 32 * 4096 * 4, 0);
        hipEventRecord(a, 0);
        k_lds<<<n / 4096, 256>>>(nullptr, n, tab, 4096, out);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
      printf("LDS histogram 4096 bins per 4096-point chunk + merge: %.3f ms\n", best); }
    return 0;
}
