// Micro-benchmark: issue rate of wave-level vector loads from an L1-resident table as a function
// of the lane mask and the width per lane.  Unlike td_cost.hip the loads of an iteration do not
// depend on each other's data, and the occupancy is eight waves per SIMD, so that the number
// measured is the throughput of the address / tag / return pipeline, not a latency.
//   hipcc -O3 --offload-arch=gfx950 -o ta_rate tools/ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// W = dwords per lane and load (1, 2, 3, 4); SAME_LINE: the four loads of an iteration read one
// 64-byte record (like a node pair) instead of four records
template <int W, bool SAME_LINE>
__global__ __launch_bounds__(256, 8) void k(const float* table, uint32_t mask_lo, uint32_t mask_hi, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long mask = ((unsigned long long)mask_hi << 32) | mask_lo;
    float acc = 0.0f;
    if ((mask >> lane) & 1ull) {
        uint32_t idx = (threadIdx.x * 7u + blockIdx.x * 13u) & 255u; // 256 records of 64 bytes = 16 KiB
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t rec = SAME_LINE ? idx : ((idx + 67u * j) & 255u);
                const float* p = table + rec * 16u + (SAME_LINE ? 4u * j : 0u);
                if (W == 1) { acc += *p; }
                if (W == 2) { const float2 v = *reinterpret_cast<const float2*>(p); acc += v.x + v.y; }
                if (W == 3) { typedef float f3_t __attribute__((ext_vector_type(3))); const f3_t v = *reinterpret_cast<const f3_t*>(p); acc += v.x + v.z; }
                if (W == 4) { const float4 v = *reinterpret_cast<const float4*>(p); acc += v.x + v.w; }
            }
            idx = (idx * 5u + 1u) & 255u;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// Quad-cooperative: the four lanes of a quad read the four 16-byte pieces of ONE 64-byte record per
// load, so a wave-level load fetches 16 whole records (the four loads of an iteration: the records
// of the quad's four lanes in turn).  owner_mask: which lanes "own" a record that has to be fetched;
// load j is executed by the quads whose lane j is in the mask.
__global__ __launch_bounds__(256, 8) void k_quad(const float* table, uint32_t mask_lo, uint32_t mask_hi, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long mask = ((unsigned long long)mask_hi << 32) | mask_lo;
    float acc = 0.0f;
    uint32_t idx = (threadIdx.x * 7u + blockIdx.x * 13u) & 255u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t owner = (lane & ~3u) + j;
            const uint32_t rec = __shfl(idx, (int)owner, 64);
            if ((mask >> owner) & 1ull) {
                const float4 v = *reinterpret_cast<const float4*>(table + rec * 16u + 4u * (lane & 3u));
                acc += v.x + v.w;
            }
        }
        idx = (idx * 5u + 1u) & 255u;
    }
    if (acc == 12345.678f) out[0] = acc;
}

static void run_quad(const float* table, float* out, int blocks, double mhz) {
    struct Case { const char* name; uint32_t lo, hi; } cases[] = {
        {"all 64", 0xFFFFFFFFu, 0xFFFFFFFFu}, {"low 32", 0xFFFFFFFFu, 0u}, {"every 2nd (32)", 0x55555555u, 0x55555555u},
        {"low 16", 0xFFFFu, 0u}, {"every 4th (16)", 0x11111111u, 0x11111111u}, {"random 29", 0x5A3C96E1u, 0x0F6931A6u}};
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Case& c : cases) {
        hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(256), 0, 0, table, c.lo, c.hi, 50, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(256), 0, 0, table, c.lo, c.hi, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double iters_per_cu = 8.0 * 4.0 * iters; // one iteration = the node pairs of up to 64 owners
        printf("%-34s %-16s %7.3f ms  %6.1f cycles per wave iteration (4 loads) per CU\n", "quad-cooperative 64-byte records", c.name, ms, ms * 1e-3 * mhz * 1e6 / iters_per_cu);
    }
}

template <int W, bool SAME_LINE>
static void run(const char* what, const float* table, float* out, int blocks, double mhz) {
    struct Case { const char* name; uint32_t lo, hi; } cases[] = {
        {"all 64", 0xFFFFFFFFu, 0xFFFFFFFFu}, {"low 32", 0xFFFFFFFFu, 0u}, {"every 2nd (32)", 0x55555555u, 0x55555555u},
        {"low 16", 0xFFFFu, 0u}, {"every 4th (16)", 0x11111111u, 0x11111111u}, {"random 29", 0x5A3C96E1u, 0x0F6931A6u}};
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Case& c : cases) {
        hipLaunchKernelGGL((k<W, SAME_LINE>), dim3(blocks), dim3(256), 0, 0, table, c.lo, c.hi, 50, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<W, SAME_LINE>), dim3(blocks), dim3(256), 0, 0, table, c.lo, c.hi, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double loads_per_cu = 8.0 * 4.0 * iters * 4.0; // 8 blocks x 4 waves x iters x 4 loads
        printf("%-34s %-16s %7.3f ms  %6.1f cycles per wave-level load per CU\n", what, c.name, ms, ms * 1e-3 * mhz * 1e6 / loads_per_cu);
    }
}

int main() {
    float* table;
    float* out;
    hipMalloc(&table, 256 * 64);
    hipMemset(table, 0, 256 * 64);
    hipMalloc(&out, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8;
    const double mhz = p.clockRate / 1000.0;
    printf("%d CUs, %.0f MHz\n", p.multiProcessorCount, mhz);
    run<4, false>("16 B/lane, 4 records", table, out, blocks, mhz);
    run<4, true>("16 B/lane, one 64-byte record", table, out, blocks, mhz);
    run_quad(table, out, blocks, mhz);
    run<3, true>("12 B/lane, one 64-byte record", table, out, blocks, mhz);
    run<2, true>("8 B/lane, one 64-byte record", table, out, blocks, mhz);
    run<1, true>("4 B/lane, one 64-byte record", table, out, blocks, mhz);
    return 0;
}
