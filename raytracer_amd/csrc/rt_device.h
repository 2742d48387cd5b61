// Device-side records, vector helpers and the integer RNG of the path-tracing kernels.
// gfx950 only; wave size is 64 everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_WAVE 64

// ------------------------------------------------------------------ records in HBM

// Extend-stage view of a primitive.  The first 80 bytes are what a lane needs when it enters the
// primitive (five independent 16-byte loads): inv holds rows 0..2 of the four columns of the
// uploaded inv_transform (scene.rs:185); root_box, root_first and root_word are a copy of the BVH
// root's record, so that the root test does not wait for a second, dependent fetch, with the
// primitive's alpha mode in the spare bits of the count.
struct PrimX {
    // the 80 bytes the extend stage reads when a ray enters the primitive
    float    inv[12];
    float    root_box[6];     // the root node: min.xyz, max.xyz,
    uint32_t root_first;      // first (in device numbering) and
    uint32_t root_word;       // count in the low 28 bits, RT_PRIMX_* flags above
    // the rest (alpha test, shadow rays)
    uint32_t root;            // device node index of the BVH root
    uint32_t alpha_settings;  // scene.rs:164, 170-177
    uint32_t albedo_offset, albedo_size;
    uint32_t albedo_texcoord;
    uint32_t pad[3];
};
static_assert(sizeof(PrimX) == 112, "PrimX");
#define RT_PRIMX_COUNT_MASK 0x0FFFFFFFu
#define RT_PRIMX_MODE_SHIFT 28   // two bits: alpha mode (alpha_settings & 3)
// the root is an inner node and both child boxes lie inside the root box (true for every tree the
// reference's builder makes; checked at upload because any tree is accepted)
#define RT_PRIMX_NESTED_ROOT 0x40000000u

// Uploaded records, read as-is by the shade stage.
struct TexRefD { uint32_t offset, size; };
struct MaterialD {
    TexRefD  albedo, emissive, normal, metallic_roughness;
    float    albedo_factor[4];
    float    emissive_factor[3];
    float    normal_scale;
    uint32_t albedo_texcoord, emissive_texcoord, normal_texcoord, metal_r_texcoord;
    float    metallic_factor, roughness_factor;
    uint32_t id, alpha_settings;
};
struct PrimitiveD {
    float     transform[16];
    float     inv_transform[16];
    MaterialD material;
    uint32_t  bvh_idx;
    uint32_t  pad[3];
};
static_assert(sizeof(PrimitiveD) == 240, "PrimitiveD");

struct SceneDev {
    const float4*     nodes;    // 2 float4 per 32-byte node: (min.xyz,max.x)(max.y,max.z,first,count)
    // what an inner-node step reads: per sibling pair (children first, first + 1 of an inner node)
    // 56 bytes at byte offset 32 * first: the two boxes (min.xyz, max.xyz each) and one descriptor
    // dword per child in the format of a traversal stack entry
    const float4*     pairs;
    const float4*     tris;     // 3 float4 per triangle: (p0,0)(e1,0)(e2,0)
    const float4*     exts;     // 6 float4 per triangle: 3 x {uv0, uv1 | oct normal, colour, pad}
    const PrimX*      primx;
    // padded world-space bounds of the primitives' root boxes for the culling scan, in groups of
    // eight primitives: [group][min x, min y, min z, max x, max y, max z][8] (192 bytes a group,
    // read through the scalar cache); the slots past the last primitive are padding whose
    // result the scan discards
    const float*      wbounds;
    // union bounds (min.xyz, max.xyz, 0, 0: 32 bytes each) of every group of eight primitives
    // and of every chunk of 128, for scenes of more than one chunk
    const float*      gbounds;
    const float*      cbounds;
    // the same padded bounds, one 32-byte record per primitive (min.xyz, max.xyz, 0, 0), for the
    // camera-ray cull, which looks at single primitives
    const float*      pbounds;
    const PrimitiveD* prims;
    const uint32_t*   texels;
    const float4*     env;
    uint64_t          n_texels;
    uint32_t          env_w, env_h;
    uint32_t          n_prims;
    uint32_t          n_prim_groups;  // ceil(n_prims / 8)
    uint32_t          n_chunks;       // ceil(n_prims / 128)
    // scenes of more than 64 chunks: 2^chunk_shift consecutive chunks form a super-chunk, so that
    // there are at most 64 of them (n_super), with union bounds of their own (sbounds; = cbounds
    // when chunk_shift is 0)
    uint32_t          chunk_shift;
    uint32_t          n_super;
    uint32_t          pad;
    const float*      sbounds;
};

// Path state: four float4 streams per queue slot, so every access is a 16-byte-per-lane,
// 1-KiB-per-wave coalesced transaction.
//   s0 = (origin.xyz, dir.x)   s1 = (dir.y, dir.z, seed, pixel)
//   s2 = (throughput.rgb, -)   s3 = (lighting.rgb, -)
struct PathQueue {
    float4* s0;
    float4* s1;
    float4* s2;
    float4* s3;
};

// Hit record written by extend: (t, triangle, primitive | overflow << 31, seed after traversal)
// as the four lanes of one float4 (bit casts).

struct Counters {
    unsigned long long v[24];
};
enum {
    C_RAYS = 0, C_PRIMS, C_INNER, C_LEAF, C_TRI, C_ALPHA, C_TEXEL, C_SHADE, C_MISS, C_KILL,
    C_OVERFLOW, C_PATHS,
    C_ROUNDS_PRIM, C_ROUNDS_INNER, C_ROUNDS_LEAF, C_BATCHES,  // wave-level: executions of each step kind
    C_SHADOW                                                   // sun option: shadow rays traced
};

// ------------------------------------------------------------------ float3 helpers
// Evaluation order is fixed (left to right) and nothing is contracted (-ffp-contract=off), so
// these produce the same bits as a scalar CPU evaluation of the same expressions.

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ f3 neg(f3 a) { return F3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float length(f3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ f3 normalize(f3 a) { return a / length(a); }
__device__ __forceinline__ f3 rcp3(f3 a) { return F3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ float fractf(float x) { return x - floorf(x); }
__device__ __forceinline__ float sign11(float x) { return x < 0.0f ? -1.0f : 1.0f; } // shader.wgsl:366-372
// f32 -> u32 with the saturating semantics of v_cvt_u32_f32 (negative and NaN -> 0)
__device__ __forceinline__ uint32_t f2u(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// ------------------------------------------------------------------ integer RNG (bit-exact)

// shader.wgsl:641-647
__device__ __forceinline__ uint32_t hash21(uint32_t x, uint32_t y) {
    x *= 73333u;
    y *= 7777u;
    x ^= 3333777777u >> (x >> 28);
    y ^= 3333777777u >> (y >> 28);
    const uint32_t n = x * y;
    return n ^ (n >> 15);
}
// shader.wgsl:650-659; 1/f32(0xffffffff) is exactly 2^-32
__device__ __forceinline__ float rnd(uint32_t& seed) {
    const uint32_t old = seed;
    seed = hash21(old, old ^ 39213742u);
    return (float)hash21(old, seed) * (1.0f / 4294967296.0f);
}

// ------------------------------------------------------------------ wave helpers

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (RT_WAVE - 1); }

// Order-preserving compaction of the threads of a 256-thread block (4 waves) with pred set:
// returns this thread's slot in the output queue (valid only where pred).  One atomic per block.  `scratch` is 8 dwords of LDS
// owned by the caller.  Must be reached by every thread of the block.
__device__ __forceinline__ uint32_t block_append(bool pred, uint32_t* counter, uint32_t* scratch) {
    const unsigned long long m = __ballot(pred);
    const uint32_t n = (uint32_t)__popcll(m);
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const uint32_t w = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 0) scratch[w] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
        scratch[4] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = scratch[4];
    for (uint32_t k = 0; k < w; k++) base += scratch[k];
    return base + prefix;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RT_WAVE);
    return v;
}
