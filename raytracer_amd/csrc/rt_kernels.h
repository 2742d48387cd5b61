// The stages of the path-tracing hot path as HIP kernels for gfx950 (wave64):
//   generate    camera rays of a group of frames  (shader.wgsl:895-897, 741-779)
//   cull        candidate primitives of each ray  (shader.wgsl:603-615; scenes of up to 128 primitives)
//   extend      closest-hit traversal             (shader.wgsl:598-627, 442-525, 339-438)
//   shade       miss / kill / BRDF                (shader.wgsl:898-914, 803-885, 941-946)
//   shadow      sun option: extend's any-hit form + the path's end (shader.wgsl:916-937, 528-575, 630-638)
//   accumulate  frame-ordered sum into screen     (shader.wgsl:947-951)
//   resolve     mean + tonemap + sRGB             (shader.wgsl:969-1012; main.rs:116)
// One launch of extend + shade per bounce for all frames of the group; paths that survive a bounce are compacted into the
// next queue with a ballot + prefix count and one atomic per 256-thread block.
#pragma once

#include "rt_device.h"

#define RT_PI 3.141592654f  // shader.wgsl:11

// ------------------------------------------------------------------ intersection primitives

// intersect, shader.wgsl:339-364, on a (p0, e1, e2) record.  Returns t or -1; u, v are the
// barycentrics intersect_full (shader.wgsl:376-412) would recompute from the same operands.
__device__ __forceinline__ float hit_tri(f3 o, f3 d, f3 p0, f3 e1, f3 e2, float& u, float& v) {
    const f3 h = cross(d, e2);
    const float a = dot(e1, h);
    if (a > -0.000002f && a < 0.000002f) return -1.0f;
    const float f = 1.0f / a;
    const f3 s = o - p0;
    u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const f3 q = cross(s, e1);
    v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    const float t = f * dot(e2, q);
    return t > 0.000002f ? t : -1.0f;
}

// shader.wgsl:577-595 on the 3x4 part of inv_transform (column-major, rows 0..2)
__device__ __forceinline__ f3 xform_dir(f3 x, const float* m) {
    return F3((m[0] * x.x + m[3] * x.y) + m[6] * x.z, (m[1] * x.x + m[4] * x.y) + m[7] * x.z,
              (m[2] * x.x + m[5] * x.y) + m[8] * x.z);
}
__device__ __forceinline__ f3 xform_pos(f3 x, const float* m) {
    return F3(((m[0] * x.x + m[3] * x.y) + m[6] * x.z) + m[9], ((m[1] * x.x + m[4] * x.y) + m[7] * x.z) + m[10],
              ((m[2] * x.x + m[5] * x.y) + m[8] * x.z) + m[11]);
}

// unpack_rgba8 + sample_texture, shader.wgsl:144-160, 175-182.  The flat index follows the
// reference arithmetic and is clamped to the texel array.
__device__ __forceinline__ uint64_t texel_index(const SceneDev& sc, uint32_t offset, uint32_t size, float tcx, float tcy) {
    const uint32_t w = size >> 16, h = size & 0xFFFFu;
    const uint32_t px = f2u(fractf(tcx) * (float)w);
    const uint32_t py = f2u(fractf(tcy) * (float)h);
    uint64_t idx = (uint64_t)(uint32_t)(offset + py * w + px);
    if (idx >= sc.n_texels) idx = sc.n_texels ? sc.n_texels - 1 : 0;
    return idx;
}
__device__ __forceinline__ uint32_t fetch_texel(const SceneDev& sc, uint32_t offset, uint32_t size, float tcx, float tcy) {
    const uint64_t idx = texel_index(sc, offset, size, tcx, tcy);
    return sc.n_texels ? sc.texels[idx] : 0u;
}

// ------------------------------------------------------------------ generate

struct GenArgs {
    PathQueue out;
    uint32_t* count_out;
    float     origin[3];
    float     forward[3];
    float     right[3];
    float     up[3];
    float     fov_factor, aspect, focus, aperture_radius;
    uint32_t  res_x, res_y, tiles_x, n_tiles;
    uint32_t  first_frame, frame_stride, n_frames; // frames first, first + stride, ... share this launch
    uint32_t  n_pixels;
    uint32_t  spread_shift;       // 0: a run of 64 slots is one 8x8 tile; 2: four 4x4 blocks of four far-apart tiles; 4: sixteen 2x2 blocks
    const uint32_t* tile_order;   // not null (staged pipeline, spread_shift 0): the tiles of the queue in this order (a permutation of the tiles)
    unsigned long long* counters; // may be null
};

// One lane per pixel and frame, one wave per 8x8 pixel tile (the reference's workgroup shape,
// shader.wgsl:887), so a wave's camera rays are spatially coherent.  Several frames (samples
// per pixel) are generated into one queue so that the later, thinner bounces still fill the
// chip.  Queue slot = (tile, frame slot, lane); lanes of edge tiles that fall outside the image
// leave a slot whose pixel word is RT_NO_PIXEL, which the later stages skip, so no compaction
// (and no atomic) is needed here.  The pixel word of a path is frame_slot * n_pixels + pixel:
// the index of its private accumulation slot.
#define RT_NO_PIXEL 0xFFFFFFFFu

// Which pixel (px, py) of which frame slot fs queue slot gid renders; false for a lane of an edge
// tile that falls outside the image.
__device__ __forceinline__ bool slot_pixel(const GenArgs& a, uint32_t gid, uint32_t& px, uint32_t& py, uint32_t& fs) {
    // queue order: tile-major, the frames of a tile next to each other, so that neighbouring
    // runs of 64 slots hold rays through the same pixels (and stay neighbours through the
    // order-preserving compaction of the later bounces)
    const uint32_t run = gid >> 6, l = gid & 63u;
    uint32_t tile = run / a.n_frames;
    fs = run - tile * a.n_frames;
    if (a.tile_order) tile = a.tile_order[tile];
    // Which pixel of which tile lane l of the run renders.  A whole tile per run keeps a wave's
    // rays together (they walk the same nodes), and with it their cost: the rays of a tile full of
    // foliage all take several times the average, and the wave that holds them ends long after the
    // others.  Calls of a few frames are too short to average that out (the launch waits for its
    // slowest wave), so they build a run from 4 or 16 square blocks of far-apart tiles instead.
    uint32_t t2 = tile, lx = l & 7u, ly = l >> 3;
    if (a.spread_shift) {
        const uint32_t sh = a.spread_shift;            // 2 or 4
        const uint32_t q = l >> (6u - sh);             // block of the run
        const uint32_t j = l & ((64u >> sh) - 1u);     // pixel of the block
        const uint32_t side = 8u >> (sh >> 1);         // 4 or 2 pixels
        const uint32_t per_row = 8u / side;
        lx = (q % per_row) * side + (j % side);
        ly = (q / per_row) * side + (j / side);
        t2 = (tile + q * (a.n_tiles >> sh)) % a.n_tiles; // for every q a bijection of the tiles
    }
    px = (t2 % a.tiles_x) * 8u + lx;
    py = (t2 / a.tiles_x) * 8u + ly;
    return px < a.res_x && py < a.res_y;
}

// The camera path of queue slot gid: which pixel of which frame it renders, its random state after
// the four camera draws, and its ray (shader.wgsl:895-897, 741-779).  False for a lane of an edge
// tile that falls outside the image.
__device__ __forceinline__ bool camera_path(const GenArgs& a, uint32_t gid, f3& origin, f3& dir, uint32_t& seed, uint32_t& pix) {
    uint32_t px, py, fs;
    if (!slot_pixel(a, gid, px, py, fs)) return false;
    seed = hash21(hash21(px, py), a.first_frame + fs * a.frame_stride); // shader.wgsl:895
    // camera_ray, shader.wgsl:741-779 (basis vectors and fov factor are per-frame constants)
    const f3 origin0 = F3(a.origin[0], a.origin[1], a.origin[2]);
    const f3 forward = F3(a.forward[0], a.forward[1], a.forward[2]);
    const f3 right = F3(a.right[0], a.right[1], a.right[2]);
    const f3 up = F3(a.up[0], a.up[1], a.up[2]);
    f3 pixel_pos = origin0 + forward;
    const float rx = rnd(seed);
    const float ry = rnd(seed);
    const float aax = (float)px + rx, aay = (float)py + ry;
    const float resx = (float)a.res_x, resy = (float)a.res_y;
    pixel_pos = pixel_pos + ((right * (aax / resx - 0.5f)) * a.fov_factor) * a.aspect;
    pixel_pos = pixel_pos + (up * (0.5f - aay / resy)) * a.fov_factor;
    dir = normalize(pixel_pos - origin0);
    const float theta = rnd(seed) * 2.0f * RT_PI; // rand_disk, shader.wgsl:669-673
    const float radius = sqrtf(rnd(seed));
    float sn, cs;
    sincosf(theta, &sn, &cs);
    const float ax = a.aperture_radius * (radius * cs);
    const float ay = a.aperture_radius * (radius * sn);
    origin = origin0 + right * ax;
    origin = origin + up * ay;
    pixel_pos = pixel_pos + dir * (a.focus - 1.0f);
    dir = normalize(pixel_pos - origin);
    pix = fs * a.n_pixels + px + a.res_x * py;
    return true;
}

__global__ __launch_bounds__(256) void k_generate(GenArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per_frame = a.n_tiles * 64u;
    if ((gid >> 6) / a.n_frames >= a.n_tiles) return;
    if (gid == 0) *a.count_out = per_frame * a.n_frames;

    float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, __uint_as_float(RT_NO_PIXEL));
    f3 origin, dir;
    uint32_t seed, pix;
    const bool valid = camera_path(a, gid, origin, dir, seed, pix);
    if (valid) {
        s0 = make_float4(origin.x, origin.y, origin.z, dir.x);
        s1 = make_float4(dir.y, dir.z, __uint_as_float(seed), __uint_as_float(pix));
    }
    a.out.s0[gid] = s0;
    a.out.s1[gid] = s1; // throughput (1) and lighting (0) are implied for camera paths
    if (a.counters) {
        const unsigned long long n = wave_sum(valid ? 1ull : 0ull);
        if (lane_id() == 0 && n) atomicAdd(&a.counters[C_PATHS], n);
    }
}

// ------------------------------------------------------------------ extend

struct ExtArgs {
    SceneDev  sc;
    PathQueue in;
    float4*   hits;
    uint32_t* spill;          // deep-stack levels beyond the LDS part (DEEP instantiation only)
    const uint32_t* count_in;
    uint32_t* cursor;
    uint32_t  stack_depth;
    uint32_t  tap;            // 1: record primary hits per pixel (paths of frame slot 0 only)
    uint32_t  n_pixels;
    uint32_t  no_cull;        // 1: no candidate masks were computed (a handful of primitives): every primitive is a candidate
    int32_t*  tap_tri;
    int32_t*  tap_prim;
    float*    tap_t;
    unsigned long long* counters;
    // what a ray with a NaN direction does, computed once on the host: it visits every node of
    // every primitive and hits nothing
    unsigned long long nan_inner, nan_leaf, nan_tri, nan_dropped;
    uint32_t  nan_overflow;
    uint32_t  pad2;
    // TREE instantiation only (long primitive lists): the pyramid of union bounds
    // over the primitives in index order, and its directory
    const float4*   tlas;      // records of 192 bytes: the padded world bounds of eight consecutive entities of one level, [6 planes][8]
    const uint32_t* tlas_dir;  // [0] = top level T; [1 + L] = first record of level L; [1 + RT_TLAS_MAX_LEVELS + L] = entities at level L
    // LIST instantiation only: per queue slot the candidate list of k_cull_list, and its length in entries
    const uint32_t* lists;
    uint32_t        list_len;
#ifdef RT_PROBE
    unsigned long long* probe; // diagnostic build: per wave (start, queue seen empty, end) in 100 MHz ticks and the most steps one of its rays took
#endif
};

// Conservative slab overlap of a world-space ray with a padded world-space box: a ray that
// fails it cannot touch anything inside the primitive, so the primitive is skipped before its
// ray transform.  Primitives that pass go through the reference's exact local-space root test.
// The box is padded far beyond f32 rounding, so the fused form (b * idir - o * idir) is safe
// provided idir is finite (see the caller).
__device__ __forceinline__ bool overlaps_world_box(f3 oi, f3 idir, const float (&b)[6]) {
    const float ax = __builtin_fmaf(b[0], idir.x, -oi.x), bx = __builtin_fmaf(b[3], idir.x, -oi.x);
    const float ay = __builtin_fmaf(b[1], idir.y, -oi.y), by = __builtin_fmaf(b[4], idir.y, -oi.y);
    const float az = __builtin_fmaf(b[2], idir.z, -oi.z), bz = __builtin_fmaf(b[5], idir.z, -oi.z);
    const float t0 = fmaxf(fminf(ax, bx), fmaxf(fminf(ay, by), fminf(az, bz)));
    const float t1 = fminf(fmaxf(ax, bx), fminf(fmaxf(ay, by), fmaxf(az, bz)));
    return !(t0 > t1 || t1 < 0.0f);
}

// Lane states of the extend stage.
#define ST_PRIM  0u   // between primitives, with at least one candidate left in its mask
#define ST_INNER 1u   // at an inner node: test the two children (shader.wgsl:496-521)
#define ST_LEAF  2u   // inside a leaf: test one triangle per step (shader.wgsl:458-495)
#define ST_SCAN  3u   // has a ray, needs the candidate mask of primitives [cb, cb + 128)
#define ST_TLAS  6u   // TREE: stands at a record of the bounds pyramid that it has not tested yet
#define ST_DONE  4u   // its ray is finished and written back: wants a new one
#define ST_IDLE  5u   // the queue is drained: nothing more to do

// Inner-node steps a wave runs back to back once the vote picked them (lanes that leave the
// inner state sit the remaining ones out); amortises the vote's scalar work.
#ifndef RT_INNER_BURST
#define RT_INNER_BURST 3
#endif
// Lanes that must be waiting before the wave spends a refill (queue fetch) or a candidate scan
// on them while other lanes still have traversal work.
// Queue slots a wave reserves per atomic: two runs of 64, i.e. (for camera rays) two frames of one
// pixel tile, whose rays walk the same nodes and triangles while those are still in the L1
// (64: 1.03 ms per frame of the atrium, 128: 1.00, 256: 1.01, 512: 1.05 -- the tail grows).
#ifndef RT_RUN_SLOTS
#define RT_RUN_SLOTS 128u
#endif
#ifndef RT_RUN_MIN
#define RT_RUN_MIN 32u
#endif
#ifndef RT_RUN_DIV
#define RT_RUN_DIV 4u
#endif
#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 32
#endif
#ifndef RT_SCAN_LANES
#define RT_SCAN_LANES 16
#endif

// Loads through the scalar cache: the address is wave-uniform and the data is never written by
// a kernel, which the constant address space tells the compiler.
typedef float rt_float8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) rt_float8* rt_cfloat8_p;

typedef float rt_float3 __attribute__((ext_vector_type(3)));
// the first 12 bytes of a 16-byte row
__device__ __forceinline__ rt_float3 ld12(const float4* row) { return *reinterpret_cast<const rt_float3*>(row); }
typedef float rt_float4 __attribute__((ext_vector_type(4)));
typedef float rt_float2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rt_float4 ld16(const float4* row) { return *reinterpret_cast<const rt_float4*>(row); }
__device__ __forceinline__ rt_float2 ld8(const float4* row) { return *reinterpret_cast<const rt_float2*>(row); }

// Address of a record given its position in 16-byte units (one 64-bit shift-add; the record's
// further 16-byte pieces are immediate offsets of the loads).  32 bits of units span 64 GiB.
__device__ __forceinline__ const float4* rec16(const void* base, uint32_t unit) {
    return reinterpret_cast<const float4*>(static_cast<const char*>(base) + ((uint64_t)unit << 4));
}

// intersect_aabb (shader.wgsl:415-438) on a node record q0 = (min.xyz, max.x), q1 = (max.y,
// max.z, first, count), with the origin-inside test taken from the signs of the
// same differences the slab test needs: o >= min  <=>  min - o <= 0, o < max  <=>  max - o > 0
// (f32 subtraction never flips or drops the sign of an exact non-zero difference).
__device__ __forceinline__ float hit_aabb_s(f3 o, f3 idir, float4 q0, float4 q1) {
    const float d0x = q0.x - o.x, d0y = q0.y - o.y, d0z = q0.z - o.z;
    const float d1x = q0.w - o.x, d1y = q1.x - o.y, d1z = q1.y - o.z;
    const bool inside = fmaxf(d0x, fmaxf(d0y, d0z)) <= 0.0f && fminf(d1x, fminf(d1y, d1z)) > 0.0f;
    const float ax = d0x * idir.x, ay = d0y * idir.y, az = d0z * idir.z;
    const float bx = d1x * idir.x, by = d1y * idir.y, bz = d1z * idir.z;
    const float t0 = fmaxf(fminf(ax, bx), fmaxf(fminf(ay, by), fminf(az, bz)));
    const float t1 = fminf(fmaxf(ax, bx), fminf(fmaxf(ay, by), fmaxf(az, bz)));
    const float r = (t0 > t1 || t0 < 0.0f) ? -1.0f : t0;
    return inside ? 0.0f : r;
}

// The clamped reciprocal direction and origin * reciprocal the world-bound test works with.  A
// zero direction component makes 1/d infinite and b * inf - o * inf a NaN that would silently
// drop that axis' bound on one side only; a huge finite reciprocal keeps the sign logic of the
// slab test intact (the reference's own tests keep the true 1/d).
__device__ __forceinline__ void world_test_ray(f3 o, f3 d, f3& oi, f3& idir) {
    const f3 r = rcp3(d);
    idir = F3(fminf(fmaxf(r.x, -1e30f), 1e30f), fminf(fmaxf(r.y, -1e30f), 1e30f), fminf(fmaxf(r.z, -1e30f), 1e30f));
    oi = F3(o.x * idir.x, o.y * idir.y, o.z * idir.z);
}

// ------------------------------------------------------------------ cull

// Scenes of at most this many primitives skip the stage: every primitive is a candidate.
#define RT_CULL_MIN_PRIMS 4u

struct CullArgs {
    SceneDev  sc;
    PathQueue in;
    uint4*    masks;          // per queue slot: bit p set = the ray meets the padded world bound of primitive p
    const uint32_t* count_in;
};

// Candidate primitives of every ray of a queue, for scenes of at most 128 primitives: the
// per-primitive loop of trace() (shader.wgsl:603-615) visits only primitives whose padded
// world-space bound the ray meets, and skipping the others changes nothing the reference would
// compute.  One lane per ray, all lanes busy, bounds through the scalar cache eight primitives
// at a time: the same test inside the extend stage ran for the few lanes that had just fetched a
// ray and cost a fifth of that stage.
__global__ __launch_bounds__(256) void k_cull(CullArgs a) {
    const uint32_t n = *a.count_in;
    const SceneDev& sc = a.sc;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t slot = base + threadIdx.x;
        f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
        bool live = false;
        if (slot < n) {
            const float4 s0 = a.in.s0[slot], s1 = a.in.s1[slot];
            live = __float_as_uint(s1.w) != RT_NO_PIXEL;
            o = F3(s0.x, s0.y, s0.z);
            d = F3(s0.w, s1.x, s1.y);
        }
        f3 oi, idir;
        world_test_ray(o, d, oi, idir);
        uint32_t m[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t mw = 0;
#pragma unroll 1
            for (uint32_t gg = 0; gg < 4u; gg++) {
                const uint32_t g = 4u * w + gg;
                if (g >= sc.n_prim_groups) break;
                // a group whose union bound none of the wave's rays meets is skipped as a whole
                // (neighbouring queue slots hold neighbouring rays)
                const rt_float8 u = *(rt_cfloat8_p)(sc.gbounds + (size_t)g * 8u);
                const float ub[6] = {u[0], u[1], u[2], u[3], u[4], u[5]};
                if (__ballot(live && overlaps_world_box(oi, idir, ub)) == 0ull) continue;
                const rt_cfloat8_p bp = (rt_cfloat8_p)(sc.wbounds + (size_t)g * 48u);
                const rt_float8 lx = bp[0], ly = bp[1], lz = bp[2], hx = bp[3], hy = bp[4], hz = bp[5];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float b[6] = {lx[k], ly[k], lz[k], hx[k], hy[k], hz[k]};
                    mw |= (overlaps_world_box(oi, idir, b) ? 1u : 0u) << (gg * 8u + k);
                }
            }
            // the last group is padded to eight: drop the bits of primitives that do not exist
            const uint32_t first = 32u * w;
            const uint32_t have = sc.n_prims > first ? min(sc.n_prims - first, 32u) : 0u;
            m[w] = have >= 32u ? mw : (mw & ((1u << have) - 1u));
        }
        if (live) a.masks[slot] = make_uint4(m[0], m[1], m[2], m[3]);
    }
}

__device__ __forceinline__ float wave_min_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// The cull stage for camera rays.  A wave's 64 queue slots are one 8x8 pixel tile of one frame:
// rays from nearly one point into a narrow pyramid.  The wave first bounds its rays (origin box,
// direction box) and asks, one primitive per lane, which primitives that set of rays can meet at
// all -- the slab test in interval arithmetic: per axis a lower bound L of every ray's entry
// distance and an upper bound U of its exit distance; no ray meets the box if some axis' L is
// above another's U, or an exit lies behind the origin -- and then runs the exact per-ray test of
// k_cull for those primitives only.  The interval test is
// conservative: its bounds are widened beyond f32 rounding, and an axis on which the directions
// straddle (or come close to) zero puts no constraint.  The masks are those k_cull computes.
// GENERATE: the rays are not read from the queue but made here, as k_generate makes them (g says
// how), and written to it: one launch and one pass over the queue less.
template <bool GENERATE>
__global__ __launch_bounds__(256) void k_cull_camera(CullArgs a, GenArgs g) {
    const uint32_t n = GENERATE ? g.n_tiles * 64u * g.n_frames : *a.count_in;
    const SceneDev& sc = a.sc;
    const uint32_t lane = lane_id();
    const float inf = __builtin_inff();
    if (GENERATE && blockIdx.x == 0 && threadIdx.x == 0) *g.count_out = n;
    for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t slot = base + lane;
        f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
        bool live = false;
        if (GENERATE) { // whole runs of 64: slot < n
            uint32_t seed = 0, pix = RT_NO_PIXEL;
            f3 co, cd;
            live = camera_path(g, slot, co, cd, seed, pix);
            float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, __uint_as_float(RT_NO_PIXEL));
            if (live) {
                o = co;
                d = cd;
                s0 = make_float4(o.x, o.y, o.z, d.x);
                s1 = make_float4(d.y, d.z, __uint_as_float(seed), __uint_as_float(pix));
            }
            g.out.s0[slot] = s0;
            g.out.s1[slot] = s1;
            if (g.counters) {
                const unsigned long long np = wave_sum(live ? 1ull : 0ull);
                if (lane == 0 && np) atomicAdd(&g.counters[C_PATHS], np);
            }
        } else if (slot < n) {
            const float4 s0 = a.in.s0[slot], s1 = a.in.s1[slot];
            live = __float_as_uint(s1.w) != RT_NO_PIXEL;
            o = F3(s0.x, s0.y, s0.z);
            d = F3(s0.w, s1.x, s1.y);
        }
        f3 oi, idir;
        world_test_ray(o, d, oi, idir);
        if (__ballot(live) == 0ull) continue;
        // the box of the wave's origins and of its directions
        const float on[3] = {wave_min_f(live ? o.x : inf), wave_min_f(live ? o.y : inf), wave_min_f(live ? o.z : inf)};
        const float ox[3] = {wave_max_f(live ? o.x : -inf), wave_max_f(live ? o.y : -inf), wave_max_f(live ? o.z : -inf)};
        const float dn[3] = {wave_min_f(live ? d.x : inf), wave_min_f(live ? d.y : inf), wave_min_f(live ? d.z : inf)};
        const float dx[3] = {wave_max_f(live ? d.x : -inf), wave_max_f(live ? d.y : -inf), wave_max_f(live ? d.z : -inf)};
        unsigned long long cand[2];
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const uint32_t q = lane + 64u * pass;
            const bool valid = q < sc.n_prims;
            float b[6] = {0, 0, 0, 0, 0, 0};
            if (valid) {
                const float4 b0 = reinterpret_cast<const float4*>(sc.pbounds)[2u * q], b1 = reinterpret_cast<const float4*>(sc.pbounds)[2u * q + 1u];
                b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y;
            }
            float l_max = -inf, u_min = inf;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (dn[k] > 1e-30f || dx[k] < -1e-30f) { // wave-uniform
                    const float i0 = 1.0f / dx[k], i1 = 1.0f / dn[k];
                    const float n0 = b[k] - ox[k], n1 = b[k] - on[k], n2 = b[3 + k] - ox[k], n3 = b[3 + k] - on[k];
                    const float p0 = n0 * i0, p1 = n0 * i1, p2 = n1 * i0, p3 = n1 * i1;
                    const float p4 = n2 * i0, p5 = n2 * i1, p6 = n3 * i0, p7 = n3 * i1;
                    float lo = fminf(fminf(fminf(p0, p1), fminf(p2, p3)), fminf(fminf(p4, p5), fminf(p6, p7)));
                    float hi = fmaxf(fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)), fmaxf(fmaxf(p4, p5), fmaxf(p6, p7)));
                    lo -= fabsf(lo) * 1e-5f + 1e-6f;
                    hi += fabsf(hi) * 1e-5f + 1e-6f;
                    l_max = fmaxf(l_max, lo);
                    u_min = fminf(u_min, hi);
                }
            }
            const bool no_ray_meets = l_max > u_min || u_min < 0.0f;
            cand[pass] = __ballot(valid && !no_ray_meets);
        }
        // the exact test, per ray, for the primitives that are left
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            unsigned long long bits = cand[pass];
            while (bits) {
                const uint32_t p = (uint32_t)__builtin_ctzll(bits) + 64u * pass;
                bits &= bits - 1ull;
                const rt_float8 u = *(rt_cfloat8_p)(sc.pbounds + (size_t)p * 8u);
                const float pb[6] = {u[0], u[1], u[2], u[3], u[4], u[5]};
                const uint32_t bit = overlaps_world_box(oi, idir, pb) ? (1u << (p & 31u)) : 0u;
                const uint32_t w = (p >> 5) & 1u;
                if (pass == 0) { if (w == 0u) m0 |= bit; else m1 |= bit; }
                else { if (w == 0u) m2 |= bit; else m3 |= bit; }
            }
        }
        if (live) a.masks[slot] = make_uint4(m0, m1, m2, m3);
    }
}

// ------------------------------------------------------------------ candidate lists
//
// Long primitive lists whose order says nothing about space (neither the chunk scan nor the bounds
// pyramid can skip anything then): a 4-ary tree over the primitives' padded world bounds, built
// at upload by median splits, is walked per ray in a kernel of its own, and the primitives whose
// bound the ray meets -- the candidates of trace(), shader.wgsl:603-615 -- are written out SORTED
// by index, so that the extend stage still visits them in the reference's order.  A list has 16
// (or 32) entries: indices ascending, then RT_LIST_END; a ray with more candidates keeps all but
// one entry's worth of its smallest, then RT_LIST_MORE, and the extend stage finds the rest by the
// chunk scan from the last listed index on.  The walk has no distance bound and no order (every
// subtree the ray meets is visited).  The tree is stored level by level (node j of level d has the
// children 4j .. 4j + 3 of level d + 1; level d starts at record (4^d - 1) / 3) and walked without
// a stack: four bits per level remember which siblings of the path's node at that level the ray
// meets and has not visited yet, and the path's node at any level is the current index shifted
// right.  The list under construction is an LDS column.
#define RT_LIST_LEN 16u               // entries of a list; RT_LIST_LEN_LONG for the longest primitive lists (rthip_upload_scene picks)
#define RT_LIST_LEN_LONG 32u
#define RT_LIST_END 0xFFFFFFFFu
#define RT_LIST_MORE 0xFFFFFFFEu
#define RT_LIST_LEAF 0x80000000u      // descriptor of a tree child that is a primitive
#define RT_LIST_INNER 0u              // ... that is a node (its index follows from its parent's)
#define RT_LIST_NONE 0xFFFFFFFFu      // ... that does not exist
#define RT_LIST_MAX_DEPTH 12u         // inner levels 0 .. 11: 4^12 = 16.7 M primitives

// first record of level d: (4^d - 1) / 3 = 0b0101..01 with d ones
__host__ __device__ constexpr uint32_t rt_list_level_base(uint32_t d) { return 0x55555555u & ((1u << (2u * d)) - 1u); }

struct CullListArgs {
    const float4*   tree;   // 128-byte records: the bounds of a node's four children as six rows of four floats (min x, min y, min z, max x, max y, max z) + a descriptor each
    PathQueue       in;
    uint32_t*       lists;  // [slot][LEN]
    const uint32_t* count_in;
};

template <uint32_t LEN>
__global__ __launch_bounds__(256) void k_cull_list(CullListArgs a) {
    __shared__ uint32_t lds_cand[LEN * 256];
    uint32_t* const cand = lds_cand + threadIdx.x;
    const uint32_t n = *a.count_in;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t slot = base + threadIdx.x;
        if (slot >= n) continue;
        const float4 s0 = a.in.s0[slot], s1 = a.in.s1[slot];
        if (__float_as_uint(s1.w) == RT_NO_PIXEL) continue;
        f3 oi, idir;
        world_test_ray(F3(s0.x, s0.y, s0.z), F3(s0.w, s1.x, s1.y), oi, idir);
        uint32_t cnt = 0, node = 0, depth = 0;
        unsigned long long pending = 0; // nibble d: children of the path's node at level d - 1 still to visit
        bool dropped = false;
        for (;;) {
            const float4* np = rec16(a.tree, (rt_list_level_base(depth) + node) * 8u);
            const float4 lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5];
            const uint4 dsc = *reinterpret_cast<const uint4*>(np + 6);
            const float b0[6] = {lx.x, ly.x, lz.x, hx.x, hy.x, hz.x}, b1[6] = {lx.y, ly.y, lz.y, hx.y, hy.y, hz.y};
            const float b2[6] = {lx.z, ly.z, lz.z, hx.z, hy.z, hz.z}, b3[6] = {lx.w, ly.w, lz.w, hx.w, hy.w, hz.w};
            const uint32_t child[4] = {dsc.x, dsc.y, dsc.z, dsc.w};
            const bool hit[4] = {overlaps_world_box(oi, idir, b0), overlaps_world_box(oi, idir, b1),
                                 overlaps_world_box(oi, idir, b2), overlaps_world_box(oi, idir, b3)};
            uint32_t down = 0; // the inner children the ray meets
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!hit[k] || child[k] == RT_LIST_NONE) continue;
                if (child[k] == RT_LIST_INNER) { down |= 1u << k; continue; }
                // a primitive: insert into the ascending list; one candidate too many pushes the largest out
                const uint32_t p = child[k] & ~RT_LIST_LEAF;
                if (cnt == LEN) {
                    dropped = true;
                    if (p > cand[(LEN - 1u) * 256u]) continue;
                    cnt = LEN - 1u;
                }
                uint32_t i = cnt;
                while (i > 0u && cand[(i - 1u) * 256u] > p) { cand[i * 256u] = cand[(i - 1u) * 256u]; i--; }
                cand[i * 256u] = p;
                cnt++;
            }
            if (down != 0u) {
                const uint32_t k = (uint32_t)__builtin_ctz(down);
                depth++;
                pending |= (unsigned long long)(down & (down - 1u)) << (4u * depth); // the others, at the new depth
                node = 4u * node + k;
            } else {
                if (pending == 0ull) break;
                const uint32_t level = (63u - (uint32_t)__builtin_clzll(pending)) >> 2; // deepest level with a sibling to visit
                const uint32_t sib = (uint32_t)(pending >> (4u * level)) & 15u;
                const uint32_t k = (uint32_t)__builtin_ctz(sib);
                pending &= ~(1ull << (4u * level + k));
                node = ((node >> (2u * (depth - level))) & ~3u) | k; // the path's node at that level, its sibling k
                depth = level;
            }
        }
        if (dropped && cnt == LEN) cnt = LEN - 1u;
        uint32_t out[LEN];
#pragma unroll
        for (uint32_t k = 0; k < LEN; k++) out[k] = k < cnt ? cand[k * 256u] : ((k == cnt && dropped) ? RT_LIST_MORE : RT_LIST_END);
        uint4* dst = reinterpret_cast<uint4*>(a.lists + (size_t)slot * LEN);
#pragma unroll
        for (uint32_t k = 0; k < LEN / 4u; k++) dst[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
    }
}

// intersect_aabb on six separate bounds (the sibling-pair records), same arithmetic as hit_aabb_s
__device__ __forceinline__ float hit_aabb6(f3 o, f3 idir, float nx, float ny, float nz, float xx, float xy, float xz) {
    const float d0x = nx - o.x, d0y = ny - o.y, d0z = nz - o.z;
    const float d1x = xx - o.x, d1y = xy - o.y, d1z = xz - o.z;
    const bool inside = fmaxf(d0x, fmaxf(d0y, d0z)) <= 0.0f && fminf(d1x, fminf(d1y, d1z)) > 0.0f;
    const float ax = d0x * idir.x, ay = d0y * idir.y, az = d0z * idir.z;
    const float bx = d1x * idir.x, by = d1y * idir.y, bz = d1z * idir.z;
    const float t0 = fmaxf(fminf(ax, bx), fmaxf(fminf(ay, by), fminf(az, bz)));
    const float t1 = fminf(fmaxf(ax, bx), fminf(fmaxf(ay, by), fmaxf(az, bz)));
    const float r = (t0 > t1 || t0 < 0.0f) ? -1.0f : t0;
    return inside ? 0.0f : r;
}

// A traversal stack entry is the (first, count) pair of the postponed node packed into one
// dword when it fits (count < 16, first < 2^27), so that a pop needs no memory access; otherwise
// bit 31 is set and the low bits are the node's index, whose record the pop re-reads.
#define RT_STACK_RELOAD 0x80000000u
// the packed per-lane word (primitive, overflow flag, alpha mode): primitives are limited to 2^29
#define RT_PM_PRIM 0x1FFFFFFFu
#define RT_PM_OVERFLOW 0x20000000u
#define RT_PM_MODE_SHIFT 30
// Eight waves per SIMD: 32 waves of a CU share its 160 KB of LDS, 20 dwords per lane: six hold the
// ray's origin and direction, 13 the first levels of the traversal stack (deeper levels are rare:
// 13 levels run as fast as 19); each wave gets 64 VGPRs.  (Measured per frame of the atrium, with
// the candidate scan still inside this kernel: 6 waves with 24 levels 1.38 ms, 7 with 22 1.34, 8
// with 20 1.39; with the scan in its own kernel the lane state fits 64 registers: 7 waves 1.12 ms,
// 8 waves 1.10, with the ray in LDS instead of re-read from the queue 1.08.)
// The MULTI instantiation carries more per-lane state (chunk masks) and stays at seven waves with
// 16 levels.
#ifndef RT_LDS_LEVELS_SINGLE
#define RT_LDS_LEVELS_SINGLE 13u
#endif
#ifndef RT_LDS_LEVELS_MULTI
#define RT_LDS_LEVELS_MULTI 16u
#endif
#ifndef RT_EXTEND_WAVES_SINGLE
#define RT_EXTEND_WAVES_SINGLE 8
#endif
#ifndef RT_EXTEND_WAVES_MULTI
#define RT_EXTEND_WAVES_MULTI 7
#endif
#ifndef RT_EXTEND_WAVES_TREE
#define RT_EXTEND_WAVES_TREE 7
#endif
// The TREE instantiation keeps its walk state (the lane's position in the pyramid and a byte of
// pending children per level: three dwords per lane) in LDS as well, which is what keeps it out of
// scratch at 72 registers; three stack levels make room for it.
#ifndef RT_LDS_LEVELS_TREE
#define RT_LDS_LEVELS_TREE 13u
#endif
__host__ __device__ constexpr uint32_t rt_lds_levels(bool multi, bool tree = false) { return tree ? RT_LDS_LEVELS_TREE : multi ? RT_LDS_LEVELS_MULTI : RT_LDS_LEVELS_SINGLE; }
#define RT_MAX_STACK 64u

// trace(), shader.wgsl:598-627, with trace_bvh (442-525) inlined as a per-lane state machine.
//
// Every lane runs exactly the reference's sequence of operations for its own ray -- primitives
// in index order, children near-first, first-found-wins ties, a 'stack_depth'-entry stack whose
// overflowing push is dropped -- but the 64 lanes of a wave do not have to be at the same point
// of that sequence, nor on rays that were fetched together.  Per round the wave votes (ballot +
// popcount) for the state most lanes are in and executes only that step, so inner-node steps of
// lanes that are in different primitives, at different depths or on different rays share the
// same instructions.
//
// Persistent waves with lane refill: a wave reserves RT_RUN_SLOTS consecutive queue slots per atomic on
// a global cursor; a lane whose ray is finished writes its hit record and, once enough lanes
// are waiting, the waiting lanes take the next reserved slots.  No lane idles until the
// slowest ray of a batch is done.
//
// Candidate primitives: a per-lane bit mask of the primitives whose padded world-space bound
// the ray meets; skipping the others changes nothing the reference would compute.  For scenes of
// up to 128 primitives the mask was computed by k_cull and waits in the ray's hit record.
//
// MULTI instantiation (scenes of more than 128 primitives): the mask describes one chunk of 128
// primitives at a time and comes from a wave-uniform scan inside this kernel (scalar loads of
// eight primitives' bounds at a time) for the lanes that wait for one.  Every chunk and every
// group of eight primitives also has a union bound; a new ray first learns which chunks it can
// touch at all (cmask), and the scan of a chunk skips the groups that no waiting lane's ray meets.
//
// The traversal stack is one LDS dword column per lane (stack[level][lane]: conflict-free) for
// the first rt_lds_levels() entries; the DEEP instantiation (stack_depth above that, for trees of
// tens of millions of triangles) continues in a global-memory column per thread, so that a deep
// stack does not cost occupancy.  The reference does not re-test the box of a popped node either.
//
// ANYHIT instantiation: trace_shadow / trace_bvh_shadow (shader.wgsl:528-575, 630-638) for the
// shadow rays of the sun option.  The same walk -- primitives in index order, near child first,
// an overflowing push dropped -- without a distance bound (a box is culled only when the ray misses
// it), without the alpha test, and over at the first triangle with t >= 0: the ray retires with
// that triangle in its hit record (-1 = nothing lies along the ray).
//
// TREE instantiation (with MULTI; long primitive lists whose neighbours are neighbours in space,
// where the chunk scan grows linearly): the candidates come from a walk of an implicit 8-ary pyramid over the primitives IN
// INDEX ORDER instead.  Level 0 are the primitives, an entity of level L + 1 is the union bound of
// eight consecutive entities of level L, the top level has at most eight.  A lane walks it depth
// first, children in ascending order, so the primitives whose padded bounds its ray meets come out
// in the reference's index order; its whole walk state is one dword (level, record) and one byte
// of pending children per level.  One step tests the eight bounds of a record (twelve 16-byte
// loads); steps are a fourth kind of round in the vote.
#define RT_TLAS_MAX_LEVELS 8u   // 8^8 = 16.7 M primitives; a scene of more keeps the chunk scan
//
// LIST instantiation (with MULTI; long primitive lists in an order that says nothing about space):
// the first candidates come, in index order, from the ray's list (k_cull_list), one dword load per
// candidate; a list that ends in RT_LIST_MORE hands over to the chunk scan at the index after its
// last entry.
#define RT_LIST_SCANNING 0xFFu
template <bool COUNT, bool DEEP, bool MULTI, bool ANYHIT = false, bool TREE = false, bool LIST = false>
__global__ __launch_bounds__(256, (TREE && ANYHIT && DEEP) ? RT_EXTEND_WAVES_TREE - 1 : TREE ? RT_EXTEND_WAVES_TREE : MULTI ? RT_EXTEND_WAVES_MULTI : RT_EXTEND_WAVES_SINGLE) void k_extend(ExtArgs a) {
    // (shadow rays through the pyramid with a deep stack: the one instantiation that does not fit 72
    // registers without scratch; one wave per SIMD fewer gives it 80)
    static_assert(!TREE || MULTI, "TREE is a variant of MULTI");
    static_assert(!LIST || (MULTI && !TREE), "LIST is a variant of the scan");
    constexpr bool SCAN = MULTI && !TREE;
    constexpr uint32_t RT_LDS_LEVELS = rt_lds_levels(MULTI, TREE);
    __shared__ uint32_t lds_tlas_dir[1 + 2 * RT_TLAS_MAX_LEVELS];
    if (TREE) {
        if (threadIdx.x < 1u + 2u * RT_TLAS_MAX_LEVELS) lds_tlas_dir[threadIdx.x] = a.tlas_dir[threadIdx.x];
        __syncthreads();
    }
    __shared__ uint32_t lds_stack[4 * RT_LDS_LEVELS * RT_WAVE];
    // TREE: byte L of a lane's two dwords = children of the lane's record at pyramid level L still to visit
    __shared__ uint32_t lds_tl_pending[TREE ? 3 * 256 : 1]; // + the lane's position: level << 28 | record
    uint32_t* const tlp = lds_tl_pending + (TREE ? threadIdx.x : 0u);
    auto pend_get = [&](uint32_t lvl) -> uint32_t { return (tlp[(lvl >> 2) * 256u] >> (8u * (lvl & 3u))) & 0xFFu; };
    auto pend_clear_bit = [&](uint32_t lvl, uint32_t k) { tlp[(lvl >> 2) * 256u] &= ~(1u << (8u * (lvl & 3u) + k)); };
    auto pend_set = [&](uint32_t lvl, uint32_t m) {
        uint32_t& w = tlp[(lvl >> 2) * 256u];
        w = (w & ~(0xFFu << (8u * (lvl & 3u)))) | (m << (8u * (lvl & 3u)));
    };
    // world-space origin and direction of each lane's ray, read again at every primitive it enters
    __shared__ float lds_ray[6 * 256];
    float* const ray = lds_ray + threadIdx.x;
    const uint32_t lane = lane_id();
    uint32_t* stack = lds_stack + (threadIdx.x >> 6) * (RT_LDS_LEVELS * RT_WAVE) + lane;
    // levels RT_LDS_LEVELS.. of a deep stack: one global dword column per thread of the grid
    // (addressed from the thread index at each use: a pointer per lane would cost two registers)
    const size_t spill_stride = (size_t)gridDim.x * blockDim.x;
    auto spill_at = [&](uint32_t level) -> uint32_t& {
        return a.spill[(size_t)(level - RT_LDS_LEVELS) * spill_stride + (blockIdx.x * blockDim.x + threadIdx.x)];
    };
    const uint32_t n = *a.count_in;
    const uint32_t depth = DEEP ? min(a.stack_depth, RT_MAX_STACK) : min(a.stack_depth, RT_LDS_LEVELS);
    const SceneDev& sc = a.sc;
    const void* const nodes = sc.nodes;
    const void* const tris = sc.tris;
    unsigned long long c_rays = 0, c_prims = 0, c_inner = 0, c_leaf = 0, c_tri = 0, c_alpha = 0, c_texel = 0, c_over = 0;
    unsigned long long r_prim = 0, r_inner = 0, r_leaf = 0, r_batch = 0; // wave-uniform

    // wave-uniform: the reserved run of queue slots not yet handed to a lane
    uint32_t run_next = 0, run_end = 0, cursor_seen = 0;
    bool drained = false;
#ifdef RT_PROBE
    const unsigned long long pr_start = wall_clock64();
    unsigned long long pr_drain = 0;
    uint32_t pr_steps = 0, pr_max = 0, pr_rounds_tail = 0, pr_live_at_drain = 0;
    uint32_t pr_n[3] = {0, 0, 0}, pr_t[3] = {0, 0, 0}; // after the drain: rounds and ticks per kind (inner, leaf, prim)
    unsigned long long pr_mark = 0;
#endif

    // per-lane
    // per-lane.  What a ray needs only between primitives is not held in registers, so that more
    // waves fit a SIMD: origin and direction in LDS, the closest hit so far and the seed in the hit
    // record (written at fetch, updated by every primitive that gives a closer hit, complete when
    // the ray retires).
    uint32_t st = ST_DONE;
    uint32_t slot = 0;
    uint32_t mask[4] = {0, 0, 0, 0};
    uint32_t cb = 0; // first primitive of the 128-primitive chunk the mask describes (MULTI); TREE: the next candidate primitive
    uint32_t& tl = tlp[TREE ? 512 : 0]; // TREE: level << 28 | record of the pyramid the lane stands at (in LDS, like the pending bytes)
    uint32_t li = 0;                  // LIST: next entry of the ray's candidate list, or RT_LIST_SCANNING once the scan has taken over
    uint32_t scan_from = 0;           // LIST: first primitive the scan still has to look at
    unsigned long long cmask = ~0ull; // bit c: the ray meets the union bound of chunk c (c < 64)
    uint32_t pm = 0; // RT_PM_PRIM: current primitive; RT_PM_OVERFLOW: a push of this ray was dropped; top bits: alpha mode
    uint32_t cur_first = 0, cur_count = 0, sp = 0;
    f3 lo = F3(0, 0, 0), ld = F3(0, 0, 1), lidir = ld;
    float scale = 1.0f, best_t = 0.0f;
    int32_t best_i = -1;

    // the ray is finished: its hit record is complete but for the overflow flag
    auto retire = [&]() {
        uint32_t* rec = reinterpret_cast<uint32_t*>(a.hits + slot);
        if (pm & RT_PM_OVERFLOW) rec[2] |= 0x80000000u;
        if (a.tap) {
            const uint32_t pix = __float_as_uint(a.in.s1[slot].w);
            if (pix < a.n_pixels) {
                const int32_t idx = (int32_t)rec[1];
                a.tap_tri[pix] = idx;
                a.tap_prim[pix] = idx >= 0 ? (int32_t)(rec[2] & 0x7FFFFFFFu) : -1;
                a.tap_t[pix] = __uint_as_float(rec[0]);
            }
        }
        st = ST_DONE;
    };
    // between primitives: next candidate of this chunk, else the next chunk, else the ray is done
    auto advance = [&]() {
        if (LIST && li != RT_LIST_SCANNING) {
            // cb holds the candidate itself while the list lasts
            const uint32_t e = li < a.list_len ? a.lists[(size_t)slot * a.list_len + li] : RT_LIST_END;
            if (e == RT_LIST_END) {
                retire();
            } else if (e == RT_LIST_MORE) {
                scan_from = li != 0u ? cb + 1u : 0u; // the last listed candidate was the last one entered
                li = RT_LIST_SCANNING;
                cmask = ~0ull;
                cb = scan_from & ~127u;
                if (cb < sc.n_prims) st = ST_SCAN;
                else retire();
            } else {
                cb = e;
                li++;
                st = ST_PRIM;
            }
        } else if (TREE) {
            // the next pending child, in ascending order: a primitive (level 0), a record to descend
            // into, or, when the top record is used up, the end of the ray
            const uint32_t top = lds_tlas_dir[0];
            for (;;) {
                const uint32_t lvl = tl >> 28, rec = tl & 0x0FFFFFFFu;
                const uint32_t pend = pend_get(lvl);
                if (pend == 0u) {
                    if (lvl >= top) { retire(); break; }
                    tl = ((lvl + 1u) << 28) | (rec >> 3);
                    continue;
                }
                const uint32_t k = (uint32_t)__builtin_ctz(pend);
                pend_clear_bit(lvl, k);
                const uint32_t ent = rec * 8u + k;
                if (lvl == 0u) { cb = ent; st = ST_PRIM; break; }
                tl = ((lvl - 1u) << 28) | ent;
                st = ST_TLAS;
                break;
            }
        } else if ((mask[0] | mask[1] | mask[2] | mask[3]) != 0u) {
            st = ST_PRIM;
        } else {
            // the next chunk: the one that follows inside the same super-chunk, else the first of
            // the next super-chunk whose union bound the ray meets (cmask)
            if (SCAN) {
                uint32_t next = (cb >> 7) + 1u;
                const uint32_t sh = sc.chunk_shift;
                if ((next & ((1u << sh) - 1u)) == 0u) {
                    uint32_t sup = next >> sh;
                    if (sup < 64u) {
                        const unsigned long long rest = cmask >> sup;
                        sup = rest ? sup + (uint32_t)__builtin_ctzll(rest) : 64u;
                    }
                    next = sup << sh;
                }
                cb = next * 128u;
                if (cb < sc.n_prims) st = ST_SCAN;
                else retire();
            } else {
                retire(); // at most 128 primitives: one chunk
            }
        }
    };
    // the traversal of primitive p ended: shader.wgsl:610-614
    auto finish = [&]() {
        if (!ANYHIT && best_i >= 0) { // (a shadow ray that finds a triangle retires on the spot, in the leaf step)
            float* rec = reinterpret_cast<float*>(a.hits + slot);
            rec[0] = best_t / scale;
            rec[1] = __int_as_float(best_i);
            rec[2] = __uint_as_float(pm & RT_PM_PRIM);
        }
        advance();
    };
    auto enter = [&](uint32_t first, uint32_t count) {
        cur_first = first;
        cur_count = count;
        st = count > 0u ? ST_LEAF : ST_INNER;
        if (COUNT && count > 0u) c_leaf++;
    };
    // enter the node a descriptor (stack entry format) names
    auto enter_desc = [&](uint32_t e) {
        uint32_t first = e & 0x07FFFFFFu, count = e >> 27;
        if (e & RT_STACK_RELOAD) {
            const float4 q = rec16(nodes, (e & 0x7FFFFFFFu) * 2u)[1];
            first = __float_as_uint(q.z);
            count = __float_as_uint(q.w);
        }
        enter(first, count);
    };
    auto pop_or_finish = [&]() {
        if (sp == 0u) {
            finish();
        } else {
            --sp;
            // two separate loads (a plain select of the two addresses compiles to one flat load, which
            // takes the vector-memory path for LDS addresses too); a spilled level is read once
            uint32_t e;
            if (DEEP && sp >= RT_LDS_LEVELS) e = *const_cast<volatile uint32_t*>(&spill_at(sp));
            else e = stack[sp * RT_WAVE];
            enter_desc(e);
        }
    };

    for (;;) {
        const uint32_t n_prim = (uint32_t)__popcll(__ballot(st == ST_PRIM));
        const uint32_t n_inner = (uint32_t)__popcll(__ballot(st == ST_INNER));
        const uint32_t n_leaf = (uint32_t)__popcll(__ballot(st == ST_LEAF));
        const unsigned long long scan_m = __ballot(st == ST_SCAN);
        const unsigned long long done_m = __ballot(st == ST_DONE);
        const uint32_t n_scan = (uint32_t)__popcll(scan_m), n_done = (uint32_t)__popcll(done_m);
        const uint32_t n_tlas = TREE ? (uint32_t)__popcll(__ballot(st == ST_TLAS)) : 0u;
        const uint32_t n_trav = n_prim + n_inner + n_leaf + n_tlas;

        // ---- refill: hand reserved queue slots to the lanes that wait for a ray
        if (n_done != 0u && !drained && (n_done >= RT_REFILL_LANES || n_trav + n_scan == 0u)) {
            if (run_next == run_end) {
                // Guided reservation: RT_RUN_SLOTS while the queue is long, then ever smaller runs
                // (what is left, by this wave's last sight of the cursor, over twice the waves of the
                // grid), so that the last rays of a launch are spread over all waves instead of
                // filling the lanes of a few: the tail of a launch is the time its last-fetched
                // batch needs, and a wave with a handful of rays is done long before one with 64.
                const uint32_t left = n > cursor_seen ? n - cursor_seen : 0u;
                const uint32_t want = min(RT_RUN_SLOTS, max(RT_RUN_MIN, left / (RT_RUN_DIV * 4u * gridDim.x)));
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.cursor, want);
                base = __builtin_amdgcn_readfirstlane(base);
                cursor_seen = base + want;
                if (base >= n) {
                    drained = true;
#ifdef RT_PROBE
                    pr_drain = wall_clock64();
#endif
                } else {
                    run_next = base;
                    run_end = min(base + want, n);
                    if (COUNT) r_batch += (run_end - run_next + 63u) >> 6;
                }
            }
            if (!drained) {
                const uint32_t avail = run_end - run_next;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(done_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)done_m, 0u));
                bool fresh = false;
                f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
                if (st == ST_DONE && rank < avail) {
                    slot = run_next + rank;
                    const float4 s0 = a.in.s0[slot], s1 = a.in.s1[slot];
                    if (__float_as_uint(s1.w) != RT_NO_PIXEL) { // slots outside the image carry no ray
                        o = F3(s0.x, s0.y, s0.z);
                        d = F3(s0.w, s1.x, s1.y);
                        if (!MULTI) { // the candidate mask the cull stage left in the ray's hit record
                            uint4 mk = make_uint4(sc.n_prims >= 32u ? ~0u : (1u << sc.n_prims) - 1u, 0u, 0u, 0u);
                            if (!a.no_cull) mk = *reinterpret_cast<const uint4*>(a.hits + slot);
                            mask[0] = mk.x; mask[1] = mk.y; mask[2] = mk.z; mask[3] = mk.w;
                        }
                        // no hit yet; `debug` (here: the overflow flag) is reset by every trace(), shader.wgsl:602
                        a.hits[slot] = make_float4(99999999.0f, __int_as_float(-1), 0.0f, s1.z);
                        ray[0] = o.x; ray[256] = o.y; ray[512] = o.z;
                        ray[768] = d.x; ray[1024] = d.y; ray[1280] = d.z;
                        pm = 0;
                        cb = 0;
                        if (SCAN) cmask = ~0ull;
                        st = ST_SCAN;
                        if (TREE) { tl = lds_tlas_dir[0] << 28; tlp[0] = 0u; tlp[256] = 0u; st = ST_TLAS; } // at the top record, untested
                        if (LIST) { li = 0u; scan_from = 0u; }
                        fresh = true;
#ifdef RT_PROBE
                        pr_max = max(pr_max, pr_steps);
                        pr_steps = 0;
#endif
                        if (COUNT) c_rays++;
                        // A direction of NaNs (normalize() of a zero vector in the shading code)
                        // fails no box test and passes no triangle test: the reference walks
                        // every node of every primitive and reports a miss.  One such ray would
                        // hold its wave for the whole tree walk, so its outcome (the miss, the
                        // dropped pushes, the step counts) is precomputed by the host.
                        if (d.x != d.x && d.y != d.y && d.z != d.z) {
                            if (a.nan_overflow != 0u) pm = RT_PM_OVERFLOW;
                            if (COUNT) {
                                c_prims += sc.n_prims;
                                c_inner += a.nan_inner;
                                c_leaf += a.nan_leaf;
                                c_tri += a.nan_tri;
                                c_over += a.nan_dropped;
                            }
                            retire();
                        } else if (!MULTI || LIST) {
                            advance(); // first candidate, or done if the ray meets no primitive's bound
                        }
                    }
                }
                run_next += min(avail, n_done);
                // scenes of more than one chunk: which (super-)chunks can the new rays touch at all
                // (wave-uniform over their union bounds, scalar loads)
                if (SCAN && !LIST) {
                    const f3 r = rcp3(d);
                    const f3 idir = F3(fminf(fmaxf(r.x, -1e30f), 1e30f), fminf(fmaxf(r.y, -1e30f), 1e30f), fminf(fmaxf(r.z, -1e30f), 1e30f));
                    const f3 oi = F3(o.x * idir.x, o.y * idir.y, o.z * idir.z);
                    unsigned long long cm = 0ull;
                    const uint32_t nc = min(sc.n_super, 64u);
                    for (uint32_t ch = 0; ch < nc; ch++) {
                        const rt_float8 u = *(rt_cfloat8_p)(sc.sbounds + (size_t)ch * 8u);
                        const float b[6] = {u[0], u[1], u[2], u[3], u[4], u[5]};
                        cm |= (overlaps_world_box(oi, idir, b) ? 1ull : 0ull) << ch;
                    }
                    if (nc < 64u) cm |= ~0ull << nc; // the chunks that do not exist end the walk
                    if (fresh && st == ST_SCAN) {
                        cmask = cm;
                        if (cm == 0ull) { // 64 super-chunks and the ray meets none of them
                            retire();
                        } else {
                            cb = ((uint32_t)__builtin_ctzll(cm) << sc.chunk_shift) * 128u;
                            if (cb >= sc.n_prims) retire();
                        }
                    }
                }
            }
            continue;
        }

        // ---- candidate scan for the lanes that wait for a mask (wave-uniform over primitives)
        if (SCAN && n_scan != 0u && (n_scan >= RT_SCAN_LANES || n_trav == 0u)) {
            // the chunk the first waiting lane wants; lanes that want another chunk wait for a later scan
            const uint32_t c0 = MULTI ? __builtin_amdgcn_readlane(cb, (int)__builtin_ctzll(scan_m)) : 0u;
            const bool mine = st == ST_SCAN && (!MULTI || cb == c0);
            f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
            if (mine) {
                o = F3(ray[0], ray[256], ray[512]);
                d = F3(ray[768], ray[1024], ray[1280]);
            }
            // A zero direction component makes 1/d infinite and b * inf - o * inf a NaN that would
            // silently drop that axis' bound on one side only; a huge finite reciprocal keeps the
            // sign logic of the slab test intact (the reference's own tests keep the true 1/d).
            const f3 r = rcp3(d);
            const f3 idir = F3(fminf(fmaxf(r.x, -1e30f), 1e30f), fminf(fmaxf(r.y, -1e30f), 1e30f), fminf(fmaxf(r.z, -1e30f), 1e30f));
            const f3 oi = F3(o.x * idir.x, o.y * idir.y, o.z * idir.z);
            // inside a super-chunk of several chunks: a chunk that none of the waiting rays meets
            // is passed over as a whole
            if (sc.chunk_shift != 0u) {
                const rt_float8 u = *(rt_cfloat8_p)(sc.cbounds + (size_t)(c0 >> 7) * 8u);
                const float ub[6] = {u[0], u[1], u[2], u[3], u[4], u[5]};
                if (__ballot(mine && overlaps_world_box(oi, idir, ub)) == 0ull) {
                    if (mine) {
                        mask[0] = 0u; mask[1] = 0u; mask[2] = 0u; mask[3] = 0u;
                        advance();
                    }
                    continue;
                }
            }
            // bounds come eight primitives at a time: six 32-byte scalar loads per group; in
            // scenes of more than one chunk a group whose union bound no waiting lane meets is
            // skipped as a whole
            uint32_t m[4];
            const uint32_t g0 = c0 >> 3;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t mw = 0;
#pragma unroll 1
                for (uint32_t gg = 0; gg < 4u; gg++) {
                    const uint32_t g = g0 + 4u * w + gg;
                    if (g >= sc.n_prim_groups) break;
                    if (MULTI) {
                        const rt_float8 u = *(rt_cfloat8_p)(sc.gbounds + (size_t)g * 8u);
                        const float ub[6] = {u[0], u[1], u[2], u[3], u[4], u[5]};
                        if (__ballot(mine && overlaps_world_box(oi, idir, ub)) == 0ull) continue;
                    }
                    const rt_cfloat8_p bp = (rt_cfloat8_p)(sc.wbounds + (size_t)g * 48u);
                    const rt_float8 lx = bp[0], ly = bp[1], lz = bp[2], hx = bp[3], hy = bp[4], hz = bp[5];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const float b[6] = {lx[k], ly[k], lz[k], hx[k], hy[k], hz[k]};
                        mw |= (overlaps_world_box(oi, idir, b) ? 1u : 0u) << (gg * 8u + k);
                    }
                }
                // the last group is padded to eight: drop the bits of primitives that do not exist
                const uint32_t first = c0 + 32u * w;
                const uint32_t have = sc.n_prims > first ? min(sc.n_prims - first, 32u) : 0u;
                m[w] = have >= 32u ? mw : (mw & ((1u << have) - 1u));
            }
            if (mine) {
                if (LIST) { // the scan took over behind the listed candidates: nothing below scan_from
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const uint32_t first = c0 + 32u * w;
                        if (scan_from > first) m[w] = scan_from - first >= 32u ? 0u : (m[w] & ~((1u << (scan_from - first)) - 1u));
                    }
                }
                mask[0] = m[0]; mask[1] = m[1]; mask[2] = m[2]; mask[3] = m[3];
                advance(); // ST_PRIM if a bit is set, else on to the next chunk or done
            }
            continue;
        }

        if (n_trav == 0u) break; // every lane is done or idle and the queue is drained
#ifdef RT_PROBE
        if (drained) { if (pr_rounds_tail == 0u) pr_live_at_drain = n_trav; pr_rounds_tail++; }
#endif

#ifdef RT_PROBE
        pr_mark = wall_clock64();
#endif
        if (TREE && n_tlas != 0u && n_tlas >= n_inner && n_tlas >= n_leaf && n_tlas >= n_prim) {
            if (st == ST_TLAS) {
                // which of the eight entities of the lane's record does the ray meet (padded world
                // bounds: conservative, like the candidate masks of the other instantiations)
                const f3 o = F3(ray[0], ray[256], ray[512]);
                const f3 r = rcp3(F3(ray[768], ray[1024], ray[1280])); // clamped like the scan's, see there
                const f3 idir = F3(fminf(fmaxf(r.x, -1e30f), 1e30f), fminf(fmaxf(r.y, -1e30f), 1e30f), fminf(fmaxf(r.z, -1e30f), 1e30f));
                const f3 oi = F3(o.x * idir.x, o.y * idir.y, o.z * idir.z);
                const uint32_t lvl = tl >> 28, rec = tl & 0x0FFFFFFFu;
                const float4* rp = rec16(a.tlas, (lds_tlas_dir[1u + lvl] + rec) * 12u);
                uint32_t m = 0;
#pragma unroll 1
                for (int half = 0; half < 2; half++) { // four entities at a time: six loads in flight, not twelve
                    const float4 lx = rp[0 + half], ly = rp[2 + half], lz = rp[4 + half];
                    const float4 hx = rp[6 + half], hy = rp[8 + half], hz = rp[10 + half];
                    const float b0[6] = {lx.x, ly.x, lz.x, hx.x, hy.x, hz.x}, b1[6] = {lx.y, ly.y, lz.y, hx.y, hy.y, hz.y};
                    const float b2[6] = {lx.z, ly.z, lz.z, hx.z, hy.z, hz.z}, b3[6] = {lx.w, ly.w, lz.w, hx.w, hy.w, hz.w};
                    m |= (overlaps_world_box(oi, idir, b0) ? 1u : 0u) << (4 * half);
                    m |= (overlaps_world_box(oi, idir, b1) ? 2u : 0u) << (4 * half);
                    m |= (overlaps_world_box(oi, idir, b2) ? 4u : 0u) << (4 * half);
                    m |= (overlaps_world_box(oi, idir, b3) ? 8u : 0u) << (4 * half);
                }
                // the last record of a level is padded to eight: drop what does not exist
                const uint32_t have = lds_tlas_dir[1u + RT_TLAS_MAX_LEVELS + lvl] - rec * 8u;
                if (have < 8u) m &= (1u << have) - 1u;
                pend_set(lvl, m);
                advance();
            }
        } else if (n_inner >= n_prim && n_inner >= n_leaf) {
#pragma unroll 1
            for (int burst = 0; burst < RT_INNER_BURST; burst++) {
                if (COUNT) r_inner++;
                if (st == ST_INNER) {
                    if (COUNT) c_inner++;
#ifdef RT_PROBE
                    pr_steps++;
#endif
                    // the sibling pair: 48 bytes of boxes and a descriptor dword per child
                    const float4* np = rec16(sc.pairs, cur_first * 2u);
#ifdef RT_DIAG_EXTRA_LOADS
                    // Diagnostic build: RT_DIAG_EXTRA_LOADS further dword loads per inner-node step, from the
                    // line the step reads anyway (or, with RT_DIAG_UNIFORM, from one address for the whole
                    // wave), issued ahead of the step's own loads: loads return in order, so the step's waits
                    // cover them.  What the stage's time does with them tells what a load costs it.
                    // RT_DIAG_WIDE makes them 16-byte loads.
#if defined(RT_DIAG_WIDE)
                    typedef float rt_diag4 __attribute__((ext_vector_type(4)));
                    rt_diag4 xv[RT_DIAG_EXTRA_LOADS];
#define RT_DIAG_LOAD(dst) asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(dst) : "v"(np) : "memory")
#elif defined(RT_DIAG_UNIFORM)
                    float xv[RT_DIAG_EXTRA_LOADS];
#define RT_DIAG_LOAD(dst) asm volatile("global_load_dword %0, %1, off offset:60" : "=v"(dst) : "v"(sc.pairs) : "memory")
#else
                    float xv[RT_DIAG_EXTRA_LOADS];
#define RT_DIAG_LOAD(dst) asm volatile("global_load_dword %0, %1, off offset:60" : "=v"(dst) : "v"(np) : "memory")
#endif
#pragma unroll
                    for (int x = 0; x < RT_DIAG_EXTRA_LOADS; x++) RT_DIAG_LOAD(xv[x]);
#endif
                    const float4 b0 = np[0], b1 = np[1], b2 = np[2];
                    const uint2 dsc = *reinterpret_cast<const uint2*>(np + 3);
                    const float left = hit_aabb6(lo, lidir, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y);
                    const float right = hit_aabb6(lo, lidir, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w);
#ifdef RT_DIAG_EXTRA_LOADS
#pragma unroll
                    for (int x = 0; x < RT_DIAG_EXTRA_LOADS; x++) asm volatile("" :: "v"(xv[x])); // keeps the registers until here
#endif
                    // a shadow ray has no distance bound: its best_t is +inf whatever the primitive (not kept in a register)
                    const float bound_t = ANYHIT ? __builtin_inff() : best_t;
                    const bool lmiss = (left < -0.5f || left > bound_t);
                    const bool rmiss = (right < -0.5f || right > bound_t);
                    if (lmiss && rmiss) {
                        pop_or_finish();
                    } else {
                        const bool go_left = rmiss ? true : (lmiss ? false : left < right);
                        if (!lmiss && !rmiss) {
                            if (sp < depth) {
                                const uint32_t e = go_left ? dsc.y : dsc.x; // the far child, postponed
                                if (DEEP && sp >= RT_LDS_LEVELS) spill_at(sp) = e;
                                else stack[sp * RT_WAVE] = e;
                                sp++;
                            }
                            else { pm |= RT_PM_OVERFLOW; if (COUNT) c_over++; } // shader.wgsl:306-313
                        }
                        enter_desc(go_left ? dsc.x : dsc.y);
                    }
                }
                if (__ballot(st == ST_INNER) == 0ull) break;
            }
#ifdef RT_PROBE
            if (drained) { pr_n[0]++; pr_t[0] += (uint32_t)(wall_clock64() - pr_mark); }
#endif
        } else if (n_leaf >= n_prim) {
            // up to two triangles of the lane's leaf per round (leaves hold 1.8 on average): both
            // fetches are in flight together and the lane is back among the inner-node lanes a
            // round earlier
            const bool two = st == ST_LEAF && cur_count > 1u;
            if (COUNT) r_leaf += __ballot(two) ? 2u : 1u;
            if (st == ST_LEAF) {
#ifdef RT_PROBE
                pr_steps++;
#endif
                if (ANYHIT) best_i = -1; // nothing found so far, or the ray would have retired: the register is free between rounds
                const float4* tp = rec16(tris, cur_first * 3u);
                // 12-byte loads: the fourth lane of a triangle's rows is padding, and the return path
                // charges a wave-level load by its width
                const rt_float3 a0 = ld12(tp), a1 = ld12(tp + 1), a2 = ld12(tp + 2);
                rt_float3 b0 = a0, b1 = a1, b2 = a2;
                if (two) { b0 = ld12(tp + 3); b1 = ld12(tp + 4); b2 = ld12(tp + 5); }
                auto test = [&](rt_float3 q0, rt_float3 q1, rt_float3 q2, uint32_t tri_i) {
                    if (COUNT) c_tri++;
                    float u, v;
                    const float t = hit_tri(lo, ld, F3(q0.x, q0.y, q0.z), F3(q1.x, q1.y, q1.z), F3(q2.x, q2.y, q2.z), u, v);
                    if (t >= 0.0f && (ANYHIT || t < best_t)) { // a finite t is below a shadow ray's +inf
                        bool accept = true;
                        if (!ANYHIT && pm >= (1u << RT_PM_MODE_SHIFT)) { // shader.wgsl:464-487
                            if (COUNT) c_alpha++;
                            const uint32_t mode = pm >> RT_PM_MODE_SHIFT;
                            const PrimX& px = sc.primx[pm & RT_PM_PRIM];
                            const float w0 = (1.0f - u) - v;
                            const uint32_t set = px.albedo_texcoord > 1u ? 1u : px.albedo_texcoord;
                            const float2* e = reinterpret_cast<const float2*>(sc.exts + 6 * (size_t)tri_i);
                            const float2 t0 = e[set], t1 = e[4 + set], t2 = e[8 + set];
                            float tcx = 0.0f, tcy = 0.0f;
                            tcx += w0 * t0.x; tcy += w0 * t0.y;
                            tcx += u * t1.x;  tcy += u * t1.y;
                            tcx += v * t2.x;  tcy += v * t2.y;
                            float alpha = 1.0f; // placeholder texture, shader.wgsl:215-226
                            const uint32_t asize = px.albedo_size;
                            if (asize != 0u) {
                                if (COUNT) c_texel++;
                                alpha = (float)(fetch_texel(sc, px.albedo_offset, asize, tcx, tcy) & 255u) / 255.0f;
                            }
                            if (mode == 1u) accept = !(alpha < (float)(px.alpha_settings >> 16) / 65536.0f);
                            else { // BLEND draws from the path's random stream: the seed lives in the hit record
                                uint32_t* sw = reinterpret_cast<uint32_t*>(a.hits + slot) + 3;
                                uint32_t seed = *sw;
                                accept = !(rnd(seed) > alpha);
                                *sw = seed;
                            }
                        }
                        if (accept) { best_i = (int32_t)tri_i; if (!ANYHIT) best_t = t; }
                    }
                };
                test(a0, a1, a2, cur_first);
                if (two) test(b0, b1, b2, cur_first + 1u);
                const uint32_t done = two ? 2u : 1u;
                cur_first += done;
                cur_count -= done;
                if (ANYHIT && best_i >= 0) { // something lies along the ray: no further node, no further primitive
                    reinterpret_cast<int32_t*>(a.hits + slot)[1] = best_i;
                    retire();
                } else if (cur_count == 0u) pop_or_finish();
            }
#ifdef RT_PROBE
            if (drained) { pr_n[1]++; pr_t[1] += (uint32_t)(wall_clock64() - pr_mark); }
#endif
        } else {
            if (COUNT) r_prim++;
            if (st == ST_PRIM) {
#ifdef RT_PROBE
                pr_steps++;
#endif
                // next candidate in index order
                uint32_t k = 0;
                if (!TREE && !(LIST && li != RT_LIST_SCANNING)) {
                    if (mask[0]) { k = __builtin_ctz(mask[0]); mask[0] &= mask[0] - 1u; }
                    else if (mask[1]) { k = 32u + __builtin_ctz(mask[1]); mask[1] &= mask[1] - 1u; }
                    else if (mask[2]) { k = 64u + __builtin_ctz(mask[2]); mask[2] &= mask[2] - 1u; }
                    else { k = 96u + __builtin_ctz(mask[3]); mask[3] &= mask[3] - 1u; }
                }
                const uint32_t p = (MULTI ? cb : 0u) + k;
                const f3 o = F3(ray[0], ray[256], ray[512]), d = F3(ray[768], ray[1024], ray[1280]);
                const float closest_t = ANYHIT ? __builtin_inff() : a.hits[slot].x; // a shadow ray has no distance bound
                const float4* pp = rec16(sc.primx, p * (uint32_t)(sizeof(PrimX) / 16));
                const float4 m0 = pp[0], m1 = pp[1], m2 = pp[2], r0 = pp[3];
                float4 r1 = pp[4]; // r0, r1: the root node, alpha mode and flags above its count
                const uint32_t settings = __float_as_uint(r1.w);
                r1.w = __uint_as_float(settings & RT_PRIMX_COUNT_MASK);
                const float inv[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
                // transform_ray + scale factor, shader.wgsl:604-607
                const f3 td = xform_dir(d, inv);
                scale = length(td);
                ld = td / scale;
                lo = xform_pos(o, inv);
                lidir = rcp3(ld);
                pm = (pm & RT_PM_OVERFLOW) | p | (((settings >> RT_PRIMX_MODE_SHIFT) & 3u) << RT_PM_MODE_SHIFT);
                const bool nested = (settings & RT_PRIMX_NESTED_ROOT) != 0u;
                if (COUNT) c_prims++;
                const float troot = hit_aabb_s(lo, lidir, r0, r1);
                const uint32_t root_first = __float_as_uint(r1.z), root_count = __float_as_uint(r1.w);
                if (!ANYHIT) { best_t = closest_t * scale; best_i = -1; }
                sp = 0;
                if (troot < -0.5f) { // shader.wgsl:448
                    advance();
                } else if (!ANYHIT && nested && troot > best_t) {
                    // The root (an inner node whose children lie inside its box) is entered beyond
                    // the best hit so far.  The reference would take one inner step here, find
                    // both children at least as far, hence beyond best_t, and leave with nothing:
                    // same outcome, one round saved.
                    if (COUNT) c_inner++;
                    advance();
                } else {
                    enter(root_first, root_count);
                }
            }
#ifdef RT_PROBE
            if (drained) { pr_n[2]++; pr_t[2] += (uint32_t)(wall_clock64() - pr_mark); }
#endif
        }
    }
#ifdef RT_PROBE
    if (a.probe) {
        pr_max = max(pr_max, pr_steps);
        for (int o = 32; o > 0; o >>= 1) pr_max = max(pr_max, (uint32_t)__shfl_xor((int)pr_max, o, 64));
        if (lane == 0) {
            unsigned long long* r = a.probe + 8ull * ((size_t)blockIdx.x * 4u + (threadIdx.x >> 6));
            r[4] = pr_n[0] | ((unsigned long long)pr_t[0] << 32); r[5] = pr_n[1] | ((unsigned long long)pr_t[1] << 32);
            r[6] = pr_n[2] | ((unsigned long long)pr_t[2] << 32); r[7] = 0;
            r[0] = pr_start; r[1] = pr_drain; r[2] = wall_clock64();
            r[3] = (unsigned long long)pr_max | ((unsigned long long)pr_rounds_tail << 20) | ((unsigned long long)pr_live_at_drain << 44);
        }
    }
#endif
    if (COUNT) {
        const unsigned long long v[8] = {c_rays, c_prims, c_inner, c_leaf, c_tri, c_alpha, c_texel, c_over};
        const int idx[8] = {C_RAYS, C_PRIMS, C_INNER, C_LEAF, C_TRI, C_ALPHA, C_TEXEL, C_OVERFLOW};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned long long sum = wave_sum(v[k]);
            if (lane == 0 && sum) atomicAdd(&a.counters[idx[k]], sum);
        }
        if (lane == 0) {
            atomicAdd(&a.counters[C_ROUNDS_PRIM], r_prim);
            atomicAdd(&a.counters[C_ROUNDS_INNER], r_inner);
            atomicAdd(&a.counters[C_ROUNDS_LEAF], r_leaf);
            atomicAdd(&a.counters[C_BATCHES], r_batch);
        }
    }
}

// ------------------------------------------------------------------ shade

struct ShadeArgs {
    SceneDev  sc;
    PathQueue in;
    PathQueue out;
    const float4* hits;
    const uint32_t* count_in;
    uint32_t* count_out;
    PathQueue shadow;       // SUN only: shadow rays s0=(o, d.x) s1=(d.y, d.z, seed, slot) s2=(sun term, overflow) s3=(lighting)
    uint32_t* shadow_count; // SUN only
    float4*   partial;      // one accumulation slot per (frame slot, pixel): written once, by the path's end
    uint32_t* tap_seed;     // null unless taps are on for this launch
    float     term_prob;
    uint32_t  last_bounce;  // 1: survivors of this bounce end here (loop bound, shader.wgsl:898)
    uint32_t  first_bounce; // 1: the queue holds camera paths (throughput 1, lighting 0: planes s2, s3 are not written)
    uint32_t  n_pixels;
    unsigned long long* counters;
};

// unpack_vec3_octrahedral, shader.wgsl:185-197
__device__ __forceinline__ f3 unpack_oct(float fx_in, float fy_in) {
    const float fx = fx_in * 2.0f - 1.0f, fy = fy_in * 2.0f - 1.0f;
    f3 n = F3(fx, fy, 1.0f - fabsf(fx) - fabsf(fy));
    const float t = fminf(fmaxf(-n.z, 0.0f), 1.0f);
    n.x -= t * sign11(n.x);
    n.y -= t * sign11(n.y);
    return normalize(n);
}

// rand_sphere, shader.wgsl:662-667
__device__ __forceinline__ f3 rand_sphere(uint32_t& seed) {
    const float theta = rnd(seed) * 2.0f * RT_PI;
    const float z = rnd(seed) * 2.0f - 1.0f;
    const float radius = sqrtf(1.0f - z * z);
    float sn, cs;
    sincosf(theta, &sn, &cs); // one argument reduction for both
    return F3(radius * cs, radius * sn, z);
}

__device__ __forceinline__ f3 unpack_rgb(uint32_t x) { // shader.wgsl:175-182
    return F3((float)((x >> 24) & 255u) / 255.0f, (float)((x >> 16) & 255u) / 255.0f, (float)((x >> 8) & 255u) / 255.0f);
}
// to_linear(unpack_rgba8(x).rgb), shader.wgsl:725-728 after 175-182: a channel is one of 256
// values, so pow(k / 255, 2.2) comes from a table that every block fills with that very
// expression at its start.
__device__ __forceinline__ f3 linear_rgb(const float* lut, uint32_t x) {
    return F3(lut[(x >> 24) & 255u], lut[(x >> 16) & 255u], lut[(x >> 8) & 255u]);
}

// What the bounce loop does with a traced ray (shader.wgsl:898-914), for one path: a miss gathers
// the environment and ends the path, the termination draw ends it (with the sun option: after
// sending a shadow ray), otherwise handle_surface_hit_brdf continues it.
#define RT_PATH_END    0   // lit is the path's result
#define RT_PATH_BOUNCE 1   // origin, dir, thr, lit, seed describe the next segment
#define RT_PATH_SHADOW 2   // SUN only: origin, dir are the shadow ray, sun_term what it carries, lit the light gathered so far
struct ShadeCounts { unsigned long long shade = 0, miss = 0, kill = 0, texel = 0; };

template <bool COUNT, bool SUN, bool TOGETHER = true>
__device__ __forceinline__ int shade_segment(const SceneDev& sc, const float* lin_lut, float term_prob, f3& origin, f3& dir, f3& thr, f3& lit,
                                             uint32_t& seed, int32_t tri, uint32_t prim, float hit_t, f3& sun_term, ShadeCounts& cnt) {
    if (tri < 0) { // handle_miss / sky / sample_env_map, shader.wgsl:883-885, 691-702
        if (COUNT) cnt.miss++;
        const float u = atan2f(dir.y, dir.x) / (RT_PI * 2.0f) + 0.5f;
        const float v = asinf(-dir.z) / RT_PI + 0.5f;
        uint32_t ex = f2u(u * (float)sc.env_w), ey = f2u(v * (float)sc.env_h);
        if (ex >= sc.env_w) ex = sc.env_w - 1u;
        if (ey >= sc.env_h) ey = sc.env_h - 1u;
        const float4 e = sc.env[(size_t)ey * sc.env_w + ex];
        lit = lit + thr * F3(e.x * 1.0f, e.y * 1.0f, e.z * 1.0f);
        return RT_PATH_END;
    }
    if (rnd(seed) < term_prob) { // shader.wgsl:907-909
        if (COUNT) cnt.kill++;
        thr = thr / term_prob;
        if (SUN) {
            // The sun block the reference keeps commented out (shader.wgsl:916-937), as an
            // option: a terminated path sends one shadow ray towards the sun.
            const PrimitiveD& pr = sc.prims[prim];
            const float* im = pr.inv_transform;
            const f3 td = F3((im[0] * dir.x + im[4] * dir.y) + im[8] * dir.z, (im[1] * dir.x + im[5] * dir.y) + im[9] * dir.z,
                             (im[2] * dir.x + im[6] * dir.y) + im[10] * dir.z);
            const f3 ld = normalize(td);
            const float4 a1 = sc.tris[3 * (size_t)tri + 1], a2 = sc.tris[3 * (size_t)tri + 2];
            f3 ng = normalize(cross(F3(a1.x, a1.y, a1.z), F3(a2.x, a2.y, a2.z)));
            ng = ng * (-sign11(dot(ng, ld)));
            const f3 n_geo = normalize(F3((im[0] * ng.x + im[1] * ng.y) + im[2] * ng.z, (im[4] * ng.x + im[5] * ng.y) + im[6] * ng.z,
                                          (im[8] * ng.x + im[9] * ng.y) + im[10] * ng.z));
            const float sl = sqrtf((0.5f * 0.5f + 0.4f * 0.4f) + 1.49f * 1.49f); // TO_SUN_DIR, shader.wgsl:20-24
            const f3 sun = F3(0.5f / sl, 0.4f / sl, 1.49f / sl);
            const f3 so = origin + (dir * hit_t + n_geo * 0.0001f);
            const f3 sd = normalize(sun + rand_sphere(seed) * 0.01f);
            const float lambert = fmaxf(dot(sun, n_geo), 0.0f) / RT_PI;
            if (lambert > 0.0f) {
                sun_term = ((thr * F3(1.0f * 5.0f, 0.5f * 5.0f, 0.3f * 5.0f)) * 12.0f) * lambert;
                origin = so;
                dir = sd;
                return RT_PATH_SHADOW;
            }
        }
        return RT_PATH_END;
    }
    thr = thr / (1.0f - term_prob);
    // ---- handle_surface_hit_brdf, shader.wgsl:803-881
    if (COUNT) cnt.shade++;
    const PrimitiveD& pr = sc.prims[prim];
    // Everything the hit needs from the primitive, the triangle and its vertex attributes is
    // read here, in one go, and (TOGETHER) the empty asm below is a fence for the scheduler: it
    // wants all thirteen values in registers at this point, so the reads are issued back to back
    // and waited for once.  Left alone, the compiler moves each read to its first use to save
    // registers, and the hit becomes seven memory round trips in a row.  Worth 5 % of the shade
    // stage; k_path, which has no register to spare (the fence costs it a spilled one) and gains
    // nothing measurable, calls without.
    const float4* icol = reinterpret_cast<const float4*>(pr.inv_transform); // column-major 4x4
    const float4* trow = sc.tris + 3 * (size_t)tri;
    const float4* ex = sc.exts + 6 * (size_t)tri;
    rt_float3 ic0 = ld12(icol), ic1 = ld12(icol + 1), ic2 = ld12(icol + 2), ic3 = ld12(icol + 3);
    rt_float3 tq0 = ld12(trow), tq1 = ld12(trow + 1), tq2 = ld12(trow + 2);
    rt_float4 v0a = ld16(ex), v1a = ld16(ex + 2), v2a = ld16(ex + 4);
    rt_float2 v0b = ld8(ex + 1), v1b = ld8(ex + 3), v2b = ld8(ex + 5);
    if (TOGETHER)
        asm volatile("" : "+v"(ic0), "+v"(ic1), "+v"(ic2), "+v"(ic3), "+v"(tq0), "+v"(tq1), "+v"(tq2),
                          "+v"(v0a), "+v"(v1a), "+v"(v2a), "+v"(v0b), "+v"(v1b), "+v"(v2b));
    const float im[15] = {ic0.x, ic0.y, ic0.z, 0.0f, ic1.x, ic1.y, ic1.z, 0.0f, ic2.x, ic2.y, ic2.z, 0.0f, ic3.x, ic3.y, ic3.z};
    // the primitive-space ray and the geometric normal, as trace() recomputes them
    // for the winning triangle (shader.wgsl:617-625, 376-412)
    const f3 td = F3((im[0] * dir.x + im[4] * dir.y) + im[8] * dir.z, (im[1] * dir.x + im[5] * dir.y) + im[9] * dir.z,
                     (im[2] * dir.x + im[6] * dir.y) + im[10] * dir.z);
    const f3 ld = normalize(td);
    const f3 lo = F3(((im[0] * origin.x + im[4] * origin.y) + im[8] * origin.z) + im[12],
                     ((im[1] * origin.x + im[5] * origin.y) + im[9] * origin.z) + im[13],
                     ((im[2] * origin.x + im[6] * origin.y) + im[10] * origin.z) + im[14]);
    const f3 p0 = F3(tq0.x, tq0.y, tq0.z), e1 = F3(tq1.x, tq1.y, tq1.z), e2 = F3(tq2.x, tq2.y, tq2.z);
    f3 ng = normalize(cross(e1, e2));
    ng = ng * (-sign11(dot(ng, ld)));
    float bu = 0.333f, bv = 0.333f, b0 = 0.333f;
    {
        float u, v;
        const float t = hit_tri(lo, ld, p0, e1, e2, u, v);
        if (t >= 0.0f) { b0 = (1.0f - u) - v; bu = u; bv = v; }
    }
    // transform_normal, shader.wgsl:585-587
    const f3 n_geo = normalize(F3((im[0] * ng.x + im[1] * ng.y) + im[2] * ng.z, (im[4] * ng.x + im[5] * ng.y) + im[6] * ng.z,
                                  (im[8] * ng.x + im[9] * ng.y) + im[10] * ng.z));
    (void)rnd(seed); // shader.wgsl:804
    origin = origin + dir * hit_t;
    origin = origin + n_geo * 0.001f;

    // tri_ext_interpolate, shader.wgsl:229-253
    float tc0x = 0.0f, tc0y = 0.0f, tc1x = 0.0f, tc1y = 0.0f;
    tc0x += b0 * v0a.x; tc0y += b0 * v0a.y; tc1x += b0 * v0a.z; tc1y += b0 * v0a.w;
    f3 sn = F3(0.0f, 0.0f, 0.0f) + unpack_oct(v0b.x, v0b.y) * b0;
    tc0x += bu * v1a.x; tc0y += bu * v1a.y; tc1x += bu * v1a.z; tc1y += bu * v1a.w;
    sn = sn + unpack_oct(v1b.x, v1b.y) * bu;
    tc0x += bv * v2a.x; tc0y += bv * v2a.y; tc1x += bv * v2a.z; tc1y += bv * v2a.w;
    sn = sn + unpack_oct(v2b.x, v2b.y) * bv;
    sn = normalize(sn);
    if (length(sn) > 0.001f) { // shader.wgsl:816-824
        const f3 nn = normalize(sn);
        sn = normalize(F3((im[0] * nn.x + im[1] * nn.y) + im[2] * nn.z, (im[4] * nn.x + im[5] * nn.y) + im[6] * nn.z,
                          (im[8] * nn.x + im[9] * nn.y) + im[10] * nn.z));
        if (dot(sn, n_geo) < 0.0f) sn = neg(sn);
    } else {
        sn = n_geo;
    }

    // The three texture reads of a hit go out together and without a branch: every index is
    // computed first (texel 0 for a texture that does not exist; its value is not used), then the
    // reads are issued back to back, then their values are used.  Read one after the other, each
    // behind its own branch, they were three memory round trips in a row, most of them misses of
    // every cache (and a wait in one branch cannot leave the reads of the other branches in flight).
    const MaterialD& m = pr.material;
    const uint32_t a_size = m.albedo.size, e_size = m.emissive.size, r_size = m.metallic_roughness.size;
    const bool a_s1 = m.albedo_texcoord >= 1u, e_s1 = m.emissive_texcoord >= 1u, r_s1 = m.metal_r_texcoord >= 1u;
    const uint64_t a_idx = texel_index(sc, m.albedo.offset, a_size, a_s1 ? tc1x : tc0x, a_s1 ? tc1y : tc0y);
    const uint64_t e_idx = texel_index(sc, m.emissive.offset, e_size, e_s1 ? tc1x : tc0x, e_s1 ? tc1y : tc0y);
    const uint64_t r_idx = texel_index(sc, m.metallic_roughness.offset, r_size, r_s1 ? tc1x : tc0x, r_s1 ? tc1y : tc0y);
    uint32_t a_tex = 0u, e_tex = 0u, r_tex = 0u;
    if (sc.n_texels) {
        a_tex = sc.texels[a_size != 0u ? a_idx : 0ull];
        e_tex = sc.texels[e_size != 0u ? e_idx : 0ull];
        r_tex = sc.texels[r_size != 0u ? r_idx : 0ull];
    }
    f3 albedo = F3(m.albedo_factor[0], m.albedo_factor[1], m.albedo_factor[2]);
    if (a_size != 0u) {
        if (COUNT) cnt.texel++;
        albedo = albedo * linear_rgb(lin_lut, a_tex);
    } else {
        albedo = albedo * F3(0.01f, 0.02f, 0.03f);
    }
    f3 emissive = F3(m.emissive_factor[0], m.emissive_factor[1], m.emissive_factor[2]);
    if (e_size != 0u) {
        if (COUNT) cnt.texel++;
        emissive = emissive * linear_rgb(lin_lut, e_tex);
    }
    float metallic_chance = m.metallic_factor;
    float roughness = m.roughness_factor;
    if (r_size != 0u) {
        if (COUNT) cnt.texel++;
        const f3 mr = unpack_rgb(r_tex);
        metallic_chance = mr.z;
        roughness *= mr.y;
    }

    const bool metal = rnd(seed) < metallic_chance;
    const float f0x = metal ? albedo.x : 0.04f;
    f3 hemi = rand_sphere(seed); // rand_hemisphere, shader.wgsl:675-681
    if (dot(hemi, sn) < 0.0f) hemi = hemi * -1.0f;
    const float mt = roughness * roughness * 0.8f;
    const f3 mn = normalize(F3(mixf(sn.x, hemi.x, mt), mixf(sn.y, hemi.y, mt), mixf(sn.z, hemi.z, mt)));
    const float pw = powf(1.0f - dot(mn, neg(dir)), 5.0f); // fresnel_schlick, shader.wgsl:791-793
    const float fresnel_x = f0x + (1.0f - f0x) * pw;
    const bool specular = rnd(seed) < fresnel_x;

    lit = lit + emissive * thr;
    if (!specular || metal) thr = thr * albedo;
    if (specular || metal) {
        f3 nd = dir - mn * (2.0f * dot(mn, dir)); // reflect
        if (dot(nd, sn) < 0.0f) nd = normalize(nd - sn * dot(nd, sn)); // project_to_hemisphere
        dir = nd;
    } else {
        dir = normalize(sn + rand_sphere(seed)); // sample_lambert, shader.wgsl:781-785
    }
    return RT_PATH_BOUNCE;
}

template <bool COUNT, bool SUN>
__global__ __launch_bounds__(256) void k_shade(ShadeArgs a) {
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t scratch2[8];
    __shared__ float lin_lut[256];
    lin_lut[threadIdx.x & 255u] = powf((float)(threadIdx.x & 255u) / 255.0f, 2.2f);
    __syncthreads();
    const uint32_t n = *a.count_in;
    const uint32_t n_pad = (n + 255u) & ~255u; // whole blocks stay in the loop together
    const uint32_t stride = gridDim.x * blockDim.x;
    ShadeCounts cnt;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_pad; slot += stride) {
        bool survive = false, to_shadow = false;
        float4 o0 = make_float4(0, 0, 0, 0), o1 = o0, o2 = o0, o3 = o0;
        if (slot < n && __float_as_uint(a.in.s1[slot].w) != RT_NO_PIXEL) {
            const float4 s0 = a.in.s0[slot], s1 = a.in.s1[slot];
            // camera paths all start with throughput 1 and no light gathered: not stored, not read
            float4 s2 = make_float4(1.0f, 1.0f, 1.0f, 0.0f), s3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (!a.first_bounce) { s2 = a.in.s2[slot]; s3 = a.in.s3[slot]; }
            const float4 h = a.hits[slot];
            f3 origin = F3(s0.x, s0.y, s0.z), dir = F3(s0.w, s1.x, s1.y);
            f3 thr = F3(s2.x, s2.y, s2.z), lit = F3(s3.x, s3.y, s3.z);
            const uint32_t pix = __float_as_uint(s1.w);
            uint32_t seed = __float_as_uint(h.w);
            const int32_t tri = __float_as_int(h.y);
            const uint32_t primw = __float_as_uint(h.z);
            const bool overflow = (primw >> 31) != 0u;
            f3 sun_term = F3(0.0f, 0.0f, 0.0f);
            const int what = shade_segment<COUNT, SUN>(a.sc, lin_lut, a.term_prob, origin, dir, thr, lit, seed, tri, primw & 0x7FFFFFFFu, h.x, sun_term, cnt);
            if (SUN && what == RT_PATH_SHADOW) {
                to_shadow = true;
                o0 = make_float4(origin.x, origin.y, origin.z, dir.x);
                o1 = make_float4(dir.y, dir.z, __uint_as_float(seed), __uint_as_float(pix));
                o2 = make_float4(sun_term.x, sun_term.y, sun_term.z, overflow ? 1.0f : 0.0f);
                o3 = make_float4(lit.x, lit.y, lit.z, 0.0f);
            } else if (what == RT_PATH_BOUNCE) {
                survive = !a.last_bounce;
                o0 = make_float4(origin.x, origin.y, origin.z, dir.x);
                o1 = make_float4(dir.y, dir.z, __uint_as_float(seed), __uint_as_float(pix));
                o2 = make_float4(thr.x, thr.y, thr.z, 0.0f);
                o3 = make_float4(lit.x, lit.y, lit.z, 0.0f);
            }

            if (!survive && !to_shadow) { // the path ends: shader.wgsl:941-946; k_accumulate applies 947-951
                if (overflow) lit = F3(1.0f, 1.0f, 0.0f);
                a.partial[pix] = make_float4(lit.x, lit.y, lit.z, 1.0f);
                if (a.tap_seed && pix < a.n_pixels) a.tap_seed[pix] = seed;
            }
        }
        const uint32_t dst = block_append(survive, a.count_out, scratch);
        if (survive) {
            a.out.s0[dst] = o0;
            a.out.s1[dst] = o1;
            a.out.s2[dst] = o2;
            a.out.s3[dst] = o3;
        }
        if (SUN) {
            const uint32_t sdst = block_append(to_shadow, a.shadow_count, scratch2);
            if (to_shadow) {
                a.shadow.s0[sdst] = o0;
                a.shadow.s1[sdst] = o1;
                a.shadow.s2[sdst] = o2;
                a.shadow.s3[sdst] = o3;
            }
        }
    }
    if (COUNT) {
        const unsigned long long s0 = wave_sum(cnt.shade), s1 = wave_sum(cnt.miss), s2 = wave_sum(cnt.kill), s3 = wave_sum(cnt.texel);
        if (lane_id() == 0) {
            if (s0) atomicAdd(&a.counters[C_SHADE], s0);
            if (s1) atomicAdd(&a.counters[C_MISS], s1);
            if (s2) atomicAdd(&a.counters[C_KILL], s2);
            if (s3) atomicAdd(&a.counters[C_TEXEL], s3);
        }
    }
}

// ------------------------------------------------------------------ shadow (sun option)

struct ShadowArgs {
    PathQueue in;           // shadow rays written by k_shade<., true>
    const float4* hits;     // their hit records from the ANYHIT instantiation of k_extend
    const uint32_t* count;
    float4*   partial;
    uint32_t* tap_seed;     // null unless taps are on for this launch
    uint32_t  n_pixels;
    unsigned long long* counters; // null unless counting
};

// The end of a path that sent a shadow ray towards the sun (shader.wgsl:916-937): lighting, plus
// the sun term when nothing lies along the ray, goes to the path's result slot.  The traversal
// itself (trace_shadow) is the ANYHIT instantiation of k_extend.
__global__ __launch_bounds__(256) void k_shadow_resolve(ShadowArgs a) {
    const uint32_t n = *a.count;
    unsigned long long c_shadow = 0;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        const float4 s1 = a.in.s1[slot], s2 = a.in.s2[slot], s3 = a.in.s3[slot];
        const uint32_t pix = __float_as_uint(s1.w);
        const float4 h = a.hits[slot];
        const bool occluded = __float_as_int(h.y) >= 0;
        c_shadow++;
        f3 lit = F3(s3.x, s3.y, s3.z);
        if (!occluded) lit = lit + F3(s2.x, s2.y, s2.z);
        // a dropped push in the path's last trace() or in trace_bvh_shadow itself (push() -> assert,
        // shader.wgsl:306-313): the pixel turns yellow (:941-943)
        if (s2.w != 0.0f || (__float_as_uint(h.z) >> 31) != 0u) lit = F3(1.0f, 1.0f, 0.0f);
        a.partial[pix] = make_float4(lit.x, lit.y, lit.z, 1.0f);
        if (a.tap_seed && pix < a.n_pixels) a.tap_seed[pix] = __float_as_uint(s1.z);
    }
    if (a.counters) {
        const unsigned long long sum = wave_sum(c_shadow);
        if (lane_id() == 0 && sum) atomicAdd(&a.counters[C_SHADOW], sum);
    }
}

// ------------------------------------------------------------------ accumulate

// shader.wgsl:947-951 for the frames of one launch group, in frame order:
// screen = reject_hist ? (L, 1) : screen + (L, 1).  Adding the frames one after the other keeps
// the float sum bit-identical to rendering them one launch at a time.
__global__ __launch_bounds__(256) void k_accumulate(float4* accum, const float4* partial, uint32_t n_pixels, uint32_t n_frames, uint32_t overwrite) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += stride) {
        float4 acc;
        uint32_t f = 0;
        if (overwrite) { acc = partial[i]; f = 1; }
        else acc = accum[i];
        for (; f < n_frames; f++) {
            const float4 p = partial[(size_t)f * n_pixels + i];
            acc = make_float4(acc.x + p.x, acc.y + p.y, acc.z + p.z, acc.w + p.w);
        }
        accum[i] = acc;
    }
}

// ------------------------------------------------------------------ resolve

// tonemap_pbr_neutral, shader.wgsl:969-992
__device__ __forceinline__ f3 tonemap_pbr_neutral(f3 c) {
    const float start_compression = 0.76f; // 0.8 - 0.04
    const float desaturation = 0.15f;
    const float x = fminf(c.x, fminf(c.y, c.z));
    float offset = 0.04f;
    if (x < 0.08f) offset = x - 6.25f * x * x;
    c = F3(c.x - offset, c.y - offset, c.z - offset);
    const float peak = fmaxf(c.x, fmaxf(c.y, c.z));
    if (peak < start_compression) return c;
    const float d = 1.0f - start_compression;
    const float new_peak = 1.0f - d * d / (peak + d - start_compression);
    c = c * (new_peak / peak);
    const float g = 1.0f - 1.0f / (desaturation * (peak - new_peak) + 1.0f);
    return F3(mixf(c.x, new_peak * 1.0f, g), mixf(c.y, new_peak * 1.0f, g), mixf(c.z, new_peak * 1.0f, g));
}

__device__ __forceinline__ uint32_t to_unorm8(float c, bool srgb) {
    if (!(c > 0.0f)) c = 0.0f;
    if (c > 1.0f) c = 1.0f;
    if (srgb) c = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    return (uint32_t)(c * 255.0f + 0.5f);
}

// fs_main, shader.wgsl:994-1012, followed by the sRGB render target's encode (main.rs:116).
// Output bytes r, g, b, a per pixel (little-endian packed word).
__global__ __launch_bounds__(256) void k_resolve(const float4* accum, uint32_t* rgba8, uint64_t n, int srgb) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 s = accum[i];
        const f3 col = tonemap_pbr_neutral(F3(s.x / s.w * 1.0f, s.y / s.w * 1.0f, s.z / s.w * 1.0f));
        rgba8[i] = to_unorm8(col.x, srgb != 0) | (to_unorm8(col.y, srgb != 0) << 8) | (to_unorm8(col.z, srgb != 0) << 16) | 0xFF000000u;
    }
}

// ------------------------------------------------------------------ ray queries on device buffers

// Caller rays (origin.xyz, direction.xyz per ray) and RNG states as the first two planes of a
// path queue, and the queue's control words: ctl[0] = number of rays, ctl[1] = the extend
// stage's cursor, which must start at zero.
__global__ __launch_bounds__(256) void k_pack_rays(const float* rays6, const uint32_t* seeds, uint32_t n, float4* s0, float4* s1, uint32_t* ctl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0u) { ctl[0] = n; ctl[1] = 0u; }
    if (i >= n) return;
    const float* r = rays6 + (size_t)i * 6u;
    s0[i] = make_float4(r[0], r[1], r[2], r[3]);
    s1[i] = make_float4(r[4], r[5], __uint_as_float(seeds ? seeds[i] : 0u), __uint_as_float(0u));
}

// Hit records (t, triangle, primitive | overflow << 31, seed) as the separate arrays of the C ABI.
__global__ __launch_bounds__(256) void k_unpack_hits(const float4* hits, uint32_t n, float* t, int32_t* tri_idx, int32_t* prim_idx, uint8_t* overflow, uint32_t* seeds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 h = hits[i];
    const int32_t tri = __float_as_int(h.y);
    const uint32_t primw = __float_as_uint(h.z);
    if (t) t[i] = h.x;
    if (tri_idx) tri_idx[i] = tri;
    if (prim_idx) prim_idx[i] = tri >= 0 ? (int32_t)(primw & 0x7FFFFFFFu) : -1;
    if (overflow) overflow[i] = (uint8_t)(primw >> 31);
    if (seeds) seeds[i] = __float_as_uint(h.w);
}

// ------------------------------------------------------------------ measurement aid: the gather ceiling

// What a CU's vector-memory pipeline sustains for the access shape of the extend stage when memory
// latency is out of the picture: every ACTIVE lane of a wave (lanes chosen by `mask`) reads the
// 16-byte pieces of RECS records of its own per iteration -- PIECES = 4 of a 64-byte record: the
// shape of an inner-node step; PIECES = 8 of a 128-byte record: what a step of a 4-wide tree with
// exact f32 boxes would read; RECS = 2: two records, i.e. twice the loads in flight per wave -- from a
// 16 KiB table that stays in the L1; the loads of an iteration do not depend on each other and eight
// waves share a SIMD, so the time is the throughput of the address / tag / return path, not a
// latency.  The records a lane reads change every iteration (a small LCG), as the nodes of a
// traversal do.
template <uint32_t PIECES, uint32_t RECS>
__global__ __launch_bounds__(256, 8) void k_gather_ceiling(const float4* table, unsigned long long mask, uint32_t iters, float* sink) {
    constexpr uint32_t RECORDS = 16384u / (16u * PIECES);
    const uint32_t lane = lane_id();
    float acc = 0.0f;
    if ((mask >> lane) & 1ull) {
        uint32_t idx = (threadIdx.x * 7u + blockIdx.x * 13u) & (RECORDS - 1u);
        for (uint32_t i = 0; i < iters; i++) {
            float4 v[RECS][PIECES];
#pragma unroll
            for (uint32_t r = 0; r < RECS; r++) {
                const float4* rec = table + ((idx + r * (RECORDS / 2u + 3u)) & (RECORDS - 1u)) * PIECES;
#pragma unroll
                for (uint32_t k = 0; k < PIECES; k++) v[r][k] = rec[k];
            }
#pragma unroll
            for (uint32_t r = 0; r < RECS; r++) {
#pragma unroll
                for (uint32_t k = 0; k < PIECES; k += 4u) acc += (v[r][k].x + v[r][k + 1u].y) + (v[r][k + 2u].z + v[r][k + 3u].w);
            }
            idx = (idx * 5u + 1u) & (RECORDS - 1u);
        }
    }
    if (acc == 12345.678f) sink[0] = acc; // never true: keeps the loads
}
