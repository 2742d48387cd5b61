// Micro-benchmark: what does a wave-level 16-byte-per-lane load cost the vector-memory return
// path when only some lanes are active?  Every wave issues dependent-free dwordx4 loads from a
// small, L1-resident table; the lane mask is a launch parameter.
//   hipcc -O3 --offload-arch=gfx950 -o td_cost tools/td_cost.hip && ./td_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <uint32_t RECS>
__global__ __launch_bounds__(256) void k(const float4* table, uint32_t mask_lo, uint32_t mask_hi, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long mask = ((unsigned long long)mask_hi << 32) | mask_lo;
    float acc = 0.0f;
    if ((mask >> lane) & 1ull) {
        uint32_t idx = (threadIdx.x * 7u + blockIdx.x * 13u) & (RECS - 1u);
        for (int i = 0; i < iters; i++) {
            // four independent loads per iteration, 64-byte records scattered over a 64 KiB table
            const float4 a = table[idx * 4u + 0u], b = table[idx * 4u + 1u], c = table[idx * 4u + 2u], d = table[idx * 4u + 3u];
            acc += a.x + b.y + c.z + d.w;
            idx = (idx * 5u + 1u + (uint32_t)__float_as_uint(a.x)) & (RECS - 1u);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// one 4-byte load per lane and iteration x 4 (same addresses as the 16-byte case)
template <uint32_t RECS>
__global__ __launch_bounds__(256) void k_dword(const float* table, int iters, float* out) {
    float acc = 0.0f;
    uint32_t idx = (threadIdx.x * 7u + blockIdx.x * 13u) & (RECS - 1u);
    for (int i = 0; i < iters; i++) {
        const float a = table[idx * 16u], b = table[idx * 16u + 4u], c = table[idx * 16u + 8u], d = table[idx * 16u + 12u];
        acc += a + b + c + d;
        idx = (idx * 5u + 1u + (uint32_t)__float_as_uint(a)) & (RECS - 1u);
    }
    if (acc == 12345.678f) out[0] = acc;
}

// The same 64 records per wave and iteration, but every instruction fetches 16 whole records: the
// four lanes of a quad read the four 16-byte pieces of one record (one 64-byte access).
__global__ __launch_bounds__(256) void k_quad(const float4* table, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63u;
    float acc = 0.0f;
    uint32_t idx = (threadIdx.x * 7u + blockIdx.x * 13u) & 1023u;
    for (int i = 0; i < iters; i++) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // the record of lane (quad base + j), as every lane of the quad sees it
            const uint32_t rec = __shfl(idx, (int)((lane & ~3u) + j), 64);
            v[j] = table[rec * 4u + (lane & 3u)];
        }
        acc += v[0].x + v[1].y + v[2].z + v[3].w;
        idx = (idx * 5u + 1u + (uint32_t)__float_as_uint(v[0].x)) & 1023u;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    float4* table;
    float* out;
    hipMalloc(&table, 1024 * 4 * sizeof(float4));
    hipMemset(table, 0, 1024 * 4 * sizeof(float4));
    hipMalloc(&out, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 5, iters = 4000;
    struct Case { const char* name; uint32_t lo, hi; } cases[] = {
        {"all 64 lanes", 0xFFFFFFFFu, 0xFFFFFFFFu}, {"low 32 lanes", 0xFFFFFFFFu, 0u}, {"low 16 lanes", 0xFFFFu, 0u},
        {"every 2nd lane (32)", 0x55555555u, 0x55555555u}, {"every 4th lane (16)", 0x11111111u, 0x11111111u},
        {"one lane per 16 (4)", 0x00010001u, 0x00010001u}, {"lane 0 only", 1u, 0u}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int small = 0; small < 2; small++) {
        hipLaunchKernelGGL(small ? k_dword<128> : k_dword<1024>, dim3(blocks), dim3(256), 0, 0, (const float*)table, 100, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(small ? k_dword<128> : k_dword<1024>, dim3(blocks), dim3(256), 0, 0, (const float*)table, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-24s %8.3f ms  %6.1f ns per wave-level load per CU (4-byte loads, %s table)\n", "all 64 lanes", ms, ms * 1e6 / (5.0 * 4.0 * iters * 4.0), small ? "8 KiB" : "64 KiB");
    }
    for (int small = 0; small < 2; small++)
    for (const Case& c : cases) {
        if (small) printf("(8 KiB table) ");
        hipLaunchKernelGGL(small ? k<128> : k<1024>, dim3(blocks), dim3(256), 0, 0, table, c.lo, c.hi, 100, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(small ? k<128> : k<1024>, dim3(blocks), dim3(256), 0, 0, table, c.lo, c.hi, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // wave-level load instructions per CU: 5 blocks x 4 waves x iters x 4
        const double loads_per_cu = 5.0 * 4.0 * iters * 4.0;
        printf("%-24s %8.3f ms  %6.1f ns per wave-level load per CU\n", c.name, ms, ms * 1e6 / loads_per_cu);
    }
    hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(256), 0, 0, table, 100, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(256), 0, 0, table, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-24s %8.3f ms  %6.1f ns per wave-level load per CU (64 records per 4 loads, as above)\n", "quad-cooperative", ms, ms * 1e6 / (5.0 * 4.0 * iters * 4.0));
    return 0;
}
