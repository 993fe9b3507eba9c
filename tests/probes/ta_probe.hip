// ta_probe.hip — what a vector memory instruction costs on gfx950 as a function of its width and of the number of distinct cache lines its
// 64 lanes touch (the correspondence kernel's loads are gathers out of L1 / L2-resident tables), and what a vector ALU instruction costs
// plain vs packed.  Throughput at full occupancy: every CU runs 8 waves per SIMD, each wave issues N independent loads (or N ALU ops).
//   hipcc --offload-arch=gfx950 -O3 -o ta_probe ta_probe.hip && ./ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v2f __attribute__((ext_vector_type(2)));

// lane address = base + (wave offset + (lane / share) * stride) bytes; every repetition moves all lanes elsewhere (same pattern)
template <int W>
__global__ __launch_bounds__(256) void k_loads(const char* __restrict__ base, size_t bytes, int share, int stride, int reps, float* out)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    size_t off = ((size_t)wave * 8192 + (size_t)(lane / share) * (size_t)stride) % bytes;
    off &= ~(size_t)15;
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < reps; ++r) {
        const char* p = base + off;
        if (W == 4) { const v4f v = *(const v4f*)p; acc += v.x + v.w; }
        else if (W == 3) { const v3f v = *(const v3f*)p; acc += v.x + v.z; }
        else if (W == 2) { const v2f v = *(const v2f*)p; acc += v.x + v.y; }
        else { acc += *(const float*)p; }
        off = (off + 64 * 1024 + 16) % bytes;
        off &= ~(size_t)15;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// L1-resident: every wave of the device walks the same 16 KB (64 rows of 256 B: each lane its own row, rotating through four 64-byte
// pieces) — after the first touch everything is a hit in the CU's vector L1
template <int W>
__global__ __launch_bounds__(256) void k_loads_hot(const char* __restrict__ base, int share, int reps, float* out)
{
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)(lane / share) * 256;
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < reps; ++r) {
        const char* p = base + row + (size_t)((r & 3) * 64) + (size_t)(((r >> 2) & 3) * 16);
        if (W == 4) { const v4f v = *(const v4f*)p; acc += v.x + v.w; }
        else { acc += *(const float*)p; }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_minmax(int reps, float* out, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int r = 0; r < reps; ++r) {
        const float l0 = fminf(a0, a1), h0 = fmaxf(a0, a1), l1 = fminf(a2, a3), h1 = fmaxf(a2, a3);
        const float l2 = fminf(a4, a5), h2 = fmaxf(a4, a5), l3 = fminf(a6, a7), h3 = fmaxf(a6, a7);
        a0 = l0 + s; a1 = h1; a2 = l1 + s; a3 = h2; a4 = l2 + s; a5 = h3; a6 = l3 + s; a7 = h0;
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = a0;
}
__global__ __launch_bounds__(256) void k_cndonly(int reps, float* out, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const bool c = threadIdx.x & 1;
    for (int r = 0; r < reps; ++r) {
        // one compare, eight selects on it
        const bool d = (a0 < s) != c;
        const float t0 = d ? a1 : a0, t1 = d ? a2 : a1, t2 = d ? a3 : a2, t3 = d ? a4 : a3, t4 = d ? a5 : a4, t5 = d ? a6 : a5, t6 = d ? a7 : a6, t7 = d ? a0 : a7;
        a0 = t0; a1 = t1; a2 = t2; a3 = t3; a4 = t4; a5 = t5; a6 = t6; a7 = t7 + 1.f;
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = a0;
}

__global__ __launch_bounds__(256) void k_fma(int reps, float* out, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int r = 0; r < reps; ++r) {
        a0 = __builtin_fmaf(a0, s, 1.f); a1 = __builtin_fmaf(a1, s, 1.f); a2 = __builtin_fmaf(a2, s, 1.f); a3 = __builtin_fmaf(a3, s, 1.f);
        a4 = __builtin_fmaf(a4, s, 1.f); a5 = __builtin_fmaf(a5, s, 1.f); a6 = __builtin_fmaf(a6, s, 1.f); a7 = __builtin_fmaf(a7, s, 1.f);
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = a0;
}
__global__ __launch_bounds__(256) void k_pkfma(int reps, float* out, float s)
{
    v2f a0 = { (float)threadIdx.x, 1.f }, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const v2f ss = { s, s }, one = { 1.f, 1.f };
    for (int r = 0; r < reps; ++r) {
        a0 = __builtin_elementwise_fma(a0, ss, one); a1 = __builtin_elementwise_fma(a1, ss, one); a2 = __builtin_elementwise_fma(a2, ss, one); a3 = __builtin_elementwise_fma(a3, ss, one);
        a4 = __builtin_elementwise_fma(a4, ss, one); a5 = __builtin_elementwise_fma(a5, ss, one); a6 = __builtin_elementwise_fma(a6, ss, one); a7 = __builtin_elementwise_fma(a7, ss, one);
    }
    const v2f t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (t.x + t.y == 1.2345f) out[0] = t.x;
}
__global__ __launch_bounds__(256) void k_cndmask(int reps, float* out, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int r = 0; r < reps; ++r) {
        a0 = a0 < s ? a1 : a0 + 1.f; a1 = a1 < s ? a2 : a1 + 1.f; a2 = a2 < s ? a3 : a2 + 1.f; a3 = a3 < s ? a4 : a3 + 1.f;
        a4 = a4 < s ? a5 : a4 + 1.f; a5 = a5 < s ? a6 : a5 + 1.f; a6 = a6 < s ? a7 : a6 + 1.f; a7 = a7 < s ? a0 : a7 + 1.f;
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f) out[0] = a0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const double ghz = pr.clockRate * 1e-6;
    printf("device %s, %d CUs, clock %.2f GHz (nominal)\n", pr.name, cus, ghz);
    const size_t bytes = (size_t)24 << 20;            // 24 MB: L2 (4 MB per XCD) + Infinity Cache resident
    char* buf; float* out;
    CK(hipMalloc(&buf, bytes + 4096)); CK(hipMemset(buf, 0, bytes + 4096)); CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = cus * 8 * 4;                   // 8 workgroups of 4 waves per CU resident, 4 generations
    const int reps = 256;
    auto run = [&](const char* name, auto launch, double per_wave_instr) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        const double waves_per_cu = (double)blocks * 4 / cus;
        const double ns_per_instr_cu = ms * 1e6 / (waves_per_cu * per_wave_instr);
        printf("%-58s %8.3f ms  %7.2f ns per wave-instruction per CU (%.1f cycles at %.2f GHz)\n", name, ms, ns_per_instr_cu, ns_per_instr_cu * ghz, ghz);
        return 0;
    };
    struct Pat { const char* name; int share, stride; };
    const Pat pats[] = { { "64 lines (stride 1 KB)", 1, 1024 }, { "32 lines (2 lanes share, stride 1 KB)", 2, 1024 }, { "16 lines (4 share)", 4, 1024 },
                         { "8 lines (8 share)", 8, 1024 }, { "coalesced 16 B (8 lines)", 1, 16 }, { "64 lanes x 64-B stride (32 lines)", 1, 64 } };
    for (const Pat& p : pats) {
        char nm[128];
        snprintf(nm, sizeof nm, "dwordx4 %s", p.name); run(nm, [&] { k_loads<4><<<blocks, 256>>>(buf, bytes, p.share, p.stride, reps, out); }, reps);
        snprintf(nm, sizeof nm, "dwordx3 %s", p.name); run(nm, [&] { k_loads<3><<<blocks, 256>>>(buf, bytes, p.share, p.stride, reps, out); }, reps);
        snprintf(nm, sizeof nm, "dwordx2 %s", p.name); run(nm, [&] { k_loads<2><<<blocks, 256>>>(buf, bytes, p.share, p.stride, reps, out); }, reps);
        snprintf(nm, sizeof nm, "dword   %s", p.name); run(nm, [&] { k_loads<1><<<blocks, 256>>>(buf, bytes, p.share, p.stride, reps, out); }, reps);
    }
    for (int share : { 1, 4, 16 }) {
        char nm[128];
        snprintf(nm, sizeof nm, "dwordx4 L1-resident, %d distinct 64-B pieces", 64 / share); run(nm, [&] { k_loads_hot<4><<<blocks, 256>>>(buf, share, reps, out); }, reps);
        snprintf(nm, sizeof nm, "dword   L1-resident, %d distinct 64-B pieces", 64 / share); run(nm, [&] { k_loads_hot<1><<<blocks, 256>>>(buf, share, reps, out); }, reps);
    }
    run("v_min + v_max + v_add (12 per trip)", [&] { k_minmax<<<blocks, 256>>>(4096, out, 0.5f); }, 4096.0 * 12 / 4);
    run("1 v_cmp + 8 v_cndmask + 1 add (10 per trip)", [&] { k_cndonly<<<blocks, 256>>>(4096, out, 0.5f); }, 4096.0 * 10 / 4);
    run("v_fma_f32 x8 chains", [&] { k_fma<<<blocks, 256>>>(4096, out, 0.999f); }, 4096.0 * 8 / 4);          // per SIMD: a CU issues on 4 SIMDs
    run("v_pk_fma_f32 x8 chains", [&] { k_pkfma<<<blocks, 256>>>(4096, out, 0.999f); }, 4096.0 * 8 / 4);
    run("v_cmp + v_cndmask + v_add x8", [&] { k_cndmask<<<blocks, 256>>>(4096, out, 0.5f); }, 4096.0 * 8 * 3 / 4);
    printf("(ALU rows: ns per wave-instruction per SIMD — a CU's four SIMDs issue in parallel)\n");
    return 0;
}
