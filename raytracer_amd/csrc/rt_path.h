// The whole of cs_main (shader.wgsl:886-953) for small frame groups, in ONE persistent launch.
//
// A host that shows every sample (the reference's frame(), main.rs:356-407: one dispatch per
// redraw) hands over one frame per call, or a few.  The staged pipeline of rt_kernels.h then spends
// most of the call in the tails of its per-bounce launches: a persistent launch ends when its
// slowest wave does, and one frame has thirteen launches.  Here a lane owns a PATH instead of a
// ray: it makes the camera ray itself (camera_path, the generate stage's code), walks it through
// the primitives with the extend stage's lane state machine, and when the ray retires runs the
// shade stage's code (shade_segment) in place and either ends the path -- its result goes to the
// path's accumulation slot -- or carries on with the bounced ray.  No queues, no hit records, no
// launches between bounces; one tail per call.
//
// Everything a path computes is the code the staged pipeline runs for it (same functions, same
// operand order), so its result is bit-identical; paths only finish in another order, which no
// result depends on (one writer per accumulation slot, frames summed in frame order afterwards).
//
// Frame groups of at most RT_PATH_MAX_SLOTS camera paths (four 1080p frames; any group of a small
// image) and scenes of up to 128 primitives (one chunk of candidates); everything else keeps the
// staged pipeline, which is the faster one once its launches are long.
#pragma once

#include "rt_kernels.h"

#define ST_SHADE 7u   // its ray is traced: the bounce loop's miss / kill / BRDF is next

// four waves per SIMD (128 registers each): a CU's 160 KB of LDS gives each of its four blocks 40 KB
#ifndef RT_PATH_WAVES
#define RT_PATH_WAVES 4
#endif
#ifndef RT_PATH_MAX_SLOTS
// A frame group runs this way if it has at most this many camera paths (whatever its number of
// frames): beyond, the staged pipeline's tails weigh less than this kernel's lower rate.  Wall time
// of a call, this kernel / staged pipeline, 1080p headline scene on 256 CUs: 1080p x 1 / 2 / 3 / 4
// frames 2.73 / 4.66 / 6.65 / 8.62 ms against 3.88 / 5.55 / 7.22 / 8.63; 1440p x 1 / 2: 4.20 / 7.73
// against 5.25 / 7.93; 4K x 1: 8.58 against 8.69; 720p x 5 / 8: 5.14 / 7.82 against 6.10 / 8.13;
// 512 x 512 x 16 / 32: 4.69 / 8.71 against 5.82 / 8.94.
#define RT_PATH_MAX_SLOTS (8u << 20)
#endif
#ifndef RT_PATH_LDS_LEVELS
#define RT_PATH_LDS_LEVELS 23u   // the reference's stack depth (shader.wgsl:303)
#endif
// Lanes that must be waiting (with a traced ray, or without a path) before the wave runs a path
// round for them while others still have traversal work: this many, or one in RT_PATH_HEAVY_DIV of
// the lanes that are in play, whichever is fewer
#ifndef RT_PATH_HEAVY_LANES
#define RT_PATH_HEAVY_LANES 32
#endif
#ifndef RT_PATH_HEAVY_NUM
#define RT_PATH_HEAVY_NUM 1u
#endif
#ifndef RT_PATH_HEAVY_DIV
#define RT_PATH_HEAVY_DIV 2u
#endif
// camera paths left per wave of the grid below which a wave reserves only what its free lanes take at once
#ifndef RT_PATH_EXACT_LEFT
#define RT_PATH_EXACT_LEFT 0xFFFFFFFFu
#endif
// inner-node steps a lane takes in a round that the inner nodes have won (2 / 3 / 4 / 6: 2.66 / 2.62 /
// 2.59 / 2.65 ms of kernel per 1080p frame)
#ifndef RT_PATH_INNER_BURST
#define RT_PATH_INNER_BURST 4
#endif
// new camera rays in a path round from which on their candidate masks are made for the set first (0:
// never; 0 / 8 / 16: 2.613 / 2.597 / 2.579 ms of kernel per 1080p frame, 4.06 / 4.02 / 4.02 at 1440p)
#ifndef RT_PATH_CAMERA_BUNDLE
#define RT_PATH_CAMERA_BUNDLE 16u
#endif
#ifndef RT_PATH_RUN_SLOTS
#define RT_PATH_RUN_SLOTS 64u
#endif

// iterations after which a wave gives up and reports (a bound on the kernel's only unbounded loop: a
// wave does a few thousand per frame; the bound is reached after seconds, not minutes)
#define RT_PATH_WATCHDOG (1u << 23)

struct PathArgs {
    SceneDev  sc;
    GenArgs   gen;            // camera constants, frames of the call, slot -> pixel mapping (out / count_out unused)
    uint32_t* cursor;         // next camera path to hand out; must start at zero
    uint32_t  n_slots;        // tiles * 64 * frames
    uint32_t  max_bounces;
    uint32_t  stack_depth;
    float     term_prob;
    float4*   partial;        // one accumulation slot per (frame slot, pixel): written once, by the path's end
    float4*   accum;          // direct != 0 (a call of ONE frame): screen itself, shader.wgsl:947-951
    uint32_t  direct;
    uint32_t  overwrite;      // reject_hist
    uint32_t  no_cull;        // a handful of primitives: every primitive is a candidate
    uint32_t  tap;            // record primary hits and final seeds (frame slot 0)
    uint32_t  n_pixels;
    uint32_t  nan_overflow;
    int32_t*  tap_tri;
    int32_t*  tap_prim;
    float*    tap_t;
    uint32_t* tap_seed;
    uint32_t* spill;          // DEEP only: stack levels beyond the LDS column, one dword column per thread of the grid
    uint32_t* watchdog;       // set to 1 by a wave that ran into RT_PATH_WATCHDOG
    // The order in which the camera paths are handed out (see "hand-out order" below): position ->
    // queue slot, slots outside the image left out (n_slots then counts the slots that are in it);
    // null = position s renders slot s.  What a run's paths cost (wave rounds from their first to
    // their last, summed) is added to cost[run], from which the order of the next call on the same
    // view is made.
    const uint32_t* perm;
    uint32_t* cost;
    unsigned long long* counters;
    unsigned long long nan_inner, nan_leaf, nan_tri, nan_dropped; // see ExtArgs
#ifdef RT_PROBE
    unsigned long long* probe; // diagnostic build: eight words per wave, see the end of k_path
#endif
};

// SUN: the sun option (shader.wgsl:916-937).  A path that the termination draw ends sends one shadow
// ray towards the sun; here the lane walks it like any other ray of its path, in "shadow mode" (top
// bit of `bounce`): no distance bound, no alpha test, over at the first triangle it meets
// (trace_bvh_shadow, shader.wgsl:528-575), not counted by the traversal counters -- what the
// ANYHIT instantiation of k_extend does -- and the path round after it is the path's end.
#define RT_PATH_SHADOW_BIT 0x80000000u
// DEEP: stack_depth above the LDS column (trees of tens of millions of triangles): the levels beyond
// it live in a global-memory column per thread, as in k_extend.
template <bool COUNT, bool SUN, bool DEEP>
__global__ __launch_bounds__(256, RT_PATH_WAVES) void k_path(PathArgs a) {
    __shared__ uint32_t lds_stack[4 * RT_PATH_LDS_LEVELS * RT_WAVE];
    __shared__ float lds_ray[6 * 256];   // world-space origin and direction of the lane's current ray
    __shared__ float lds_path[6 * 256];  // throughput and lighting of the lane's path
    __shared__ float lin_lut[256];
    __shared__ uint32_t lds_meta[2 * 256]; // per lane: the run its path came from, and the wave's round count when it started
    uint32_t* const meta = lds_meta + threadIdx.x;
    lin_lut[threadIdx.x & 255u] = powf((float)(threadIdx.x & 255u) / 255.0f, 2.2f); // as k_shade
    __syncthreads();
    float* const ray = lds_ray + threadIdx.x;
    float* const pth = lds_path + threadIdx.x;
    const uint32_t lane = lane_id();
    uint32_t* const stack = lds_stack + (threadIdx.x >> 6) * (RT_PATH_LDS_LEVELS * RT_WAVE) + lane;
    const uint32_t n = a.n_slots;
    const uint32_t depth = DEEP ? min(a.stack_depth, RT_MAX_STACK) : min(a.stack_depth, RT_PATH_LDS_LEVELS);
    const size_t spill_stride = (size_t)gridDim.x * blockDim.x;
    auto spill_at = [&](uint32_t level) -> uint32_t& {
        return a.spill[(size_t)(level - RT_PATH_LDS_LEVELS) * spill_stride + (blockIdx.x * blockDim.x + threadIdx.x)];
    };
    const SceneDev& sc = a.sc;
    const void* const nodes = sc.nodes;
    const void* const tris = sc.tris;
    unsigned long long c_rays = 0, c_prims = 0, c_inner = 0, c_leaf = 0, c_tri = 0, c_alpha = 0, c_texel = 0, c_over = 0, c_paths = 0, c_shadow = 0;
    unsigned long long r_prim = 0, r_inner = 0, r_leaf = 0, r_batch = 0; // wave-uniform
    ShadeCounts cnt;

    uint32_t run_next = 0, run_end = 0, cursor_seen = 0, iter = 0;
#ifdef RT_PROBE2
    uint32_t pr_last_start = 0, pr_last_end = 0, pr_last_b = 0, pr_last_run = 0, pr2_start = 0, pr2_end = 0, pr2_b = 0, pr2_run = 0;
#endif
    bool drained = false;
#ifdef RT_PROBE
    const unsigned long long pr_start = wall_clock64();
    unsigned long long pr_drain = 0, pr_lanes = 0, pr_lanes_tail = 0;
    uint32_t pr_trav = 0, pr_shade = 0, pr_scan = 0, pr_refill = 0, pr_tail = 0, pr_shade_lanes = 0, pr_scan_lanes = 0;
    unsigned long long pr_heavy_ticks = 0, pr_ticks_shade = 0, pr_ticks_camera = 0;
#endif

    // per-lane: the path
    uint32_t st = ST_DONE;
    uint32_t pix = 0, bounce = 0, seed = 0;
    // per-lane: its current ray's closest hit so far (what the staged pipeline keeps in the hit record)
    float closest_t = 0.0f;
    int32_t hit_i = -1;
    uint32_t hit_prim = 0;
    // per-lane: the walk
    uint32_t mask[4] = {0, 0, 0, 0};
    uint32_t pm = 0;
    uint32_t cur_first = 0, cur_count = 0, sp = 0;
    f3 lo = F3(0, 0, 0), ld = F3(0, 0, 1), lidir = ld;
    float scale = 1.0f, best_t = 0.0f;
    int32_t best_i = -1;
    auto in_shadow = [&]() -> bool { return SUN && (bounce & RT_PATH_SHADOW_BIT) != 0u; };
    auto counted = [&]() -> bool { return COUNT && !in_shadow(); }; // the staged pipeline walks shadow rays without counting

    // the ray is traced
    auto retire = [&]() {
        if (pm & RT_PM_OVERFLOW) hit_prim |= 0x80000000u;
        if (a.tap && bounce == 0u && pix < a.n_pixels) {
            a.tap_tri[pix] = hit_i;
            a.tap_prim[pix] = hit_i >= 0 ? (int32_t)(hit_prim & 0x7FFFFFFFu) : -1;
            a.tap_t[pix] = closest_t;
        }
        st = ST_SHADE;
    };
    auto advance = [&]() {
        if ((mask[0] | mask[1] | mask[2] | mask[3]) != 0u) st = ST_PRIM;
        else retire();
    };
    auto finish = [&]() { // shader.wgsl:610-614
        if (best_i >= 0) {
            closest_t = best_t / scale;
            hit_i = best_i;
            hit_prim = pm & RT_PM_PRIM;
        }
        advance();
    };
    auto enter = [&](uint32_t first, uint32_t count) {
        cur_first = first;
        cur_count = count;
        st = count > 0u ? ST_LEAF : ST_INNER;
        if (counted() && count > 0u) c_leaf++;
    };
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
            uint32_t e; // two separate loads, see k_extend
            if (DEEP && sp >= RT_PATH_LDS_LEVELS) e = *const_cast<volatile uint32_t*>(&spill_at(sp));
            else e = stack[sp * RT_WAVE];
            enter_desc(e);
        }
    };
    // a new segment of the lane's path: what the extend stage does when it fetches a ray
    // (`carried`: a shadow ray starts with the overflow flag of the path's last trace(): either paints the pixel)
    auto begin_ray = [&](f3 o, f3 d, bool carried) {
        ray[0] = o.x; ray[256] = o.y; ray[512] = o.z;
        ray[768] = d.x; ray[1024] = d.y; ray[1280] = d.z;
        closest_t = in_shadow() ? __builtin_inff() : 99999999.0f; // shader.wgsl:600; a shadow ray has no distance bound
        hit_i = -1;
        hit_prim = 0u;
        pm = carried ? RT_PM_OVERFLOW : 0u;
        if (counted()) c_rays++;
        if (d.x != d.x && d.y != d.y && d.z != d.z) { // a direction of NaNs: see k_extend
            if (a.nan_overflow != 0u) pm = RT_PM_OVERFLOW;
            if (counted()) {
                c_prims += sc.n_prims;
                c_inner += a.nan_inner;
                c_leaf += a.nan_leaf;
                c_tri += a.nan_tri;
                c_over += a.nan_dropped;
            }
            retire();
        } else if (a.no_cull) {
            mask[0] = sc.n_prims >= 32u ? ~0u : (1u << sc.n_prims) - 1u;
            mask[1] = 0u; mask[2] = 0u; mask[3] = 0u;
            advance();
        } else {
            st = ST_SCAN;
        }
    };

    // ---- the three traversal steps (as in k_extend), on records the caller has fetched
    // an inner node: its two children (shader.wgsl:496-521)
    auto inner_step = [&](float4 b0, float4 b1, float4 b2, uint2 dsc) {
        if (counted()) c_inner++;
        const float left = hit_aabb6(lo, lidir, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y);
        const float right = hit_aabb6(lo, lidir, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w);
        const bool lmiss = (left < -0.5f || left > best_t);
        const bool rmiss = (right < -0.5f || right > best_t);
        if (lmiss && rmiss) {
            pop_or_finish();
        } else {
            const bool go_left = rmiss ? true : (lmiss ? false : left < right);
            if (!lmiss && !rmiss) {
                if (sp < depth) {
                    const uint32_t e = go_left ? dsc.y : dsc.x; // the far child, postponed
                    if (DEEP && sp >= RT_PATH_LDS_LEVELS) spill_at(sp) = e;
                    else stack[sp * RT_WAVE] = e;
                    sp++;
                }
                else { pm |= RT_PM_OVERFLOW; if (counted()) c_over++; } // shader.wgsl:306-313
            }
            enter_desc(go_left ? dsc.x : dsc.y);
        }
    };
    // up to two triangles of the lane's leaf (shader.wgsl:458-495)
    auto leaf_step = [&](rt_float3 a0, rt_float3 a1, rt_float3 a2, rt_float3 b0, rt_float3 b1, rt_float3 b2, bool two) {
        auto test = [&](rt_float3 q0, rt_float3 q1, rt_float3 q2, uint32_t tri_i) {
            if (counted()) c_tri++;
            float u, v;
            const float t = hit_tri(lo, ld, F3(q0.x, q0.y, q0.z), F3(q1.x, q1.y, q1.z), F3(q2.x, q2.y, q2.z), u, v);
            if (t >= 0.0f && t < best_t) {
                bool accept = true;
                if (pm >= (1u << RT_PM_MODE_SHIFT)) { // shader.wgsl:464-487
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
                    else accept = !(rnd(seed) > alpha); // BLEND draws from the path's random stream
                }
                if (accept) { best_i = (int32_t)tri_i; best_t = t; }
            }
        };
        test(a0, a1, a2, cur_first);
        if (two) test(b0, b1, b2, cur_first + 1u);
        const uint32_t done = two ? 2u : 1u;
        cur_first += done;
        cur_count -= done;
        if (in_shadow() && best_i >= 0) { // something lies along the shadow ray: no further node, no further primitive
            hit_i = best_i;
            retire();
        } else if (cur_count == 0u) pop_or_finish();
    };
    // the next candidate primitive, in index order
    auto next_candidate = [&]() -> uint32_t {
        uint32_t k;
        if (mask[0]) { k = __builtin_ctz(mask[0]); mask[0] &= mask[0] - 1u; }
        else if (mask[1]) { k = 32u + __builtin_ctz(mask[1]); mask[1] &= mask[1] - 1u; }
        else if (mask[2]) { k = 64u + __builtin_ctz(mask[2]); mask[2] &= mask[2] - 1u; }
        else { k = 96u + __builtin_ctz(mask[3]); mask[3] &= mask[3] - 1u; }
        return k;
    };
    // the ray enters primitive p: m0..m2 = rows of its inverse transform, r0, r1 = its root node
    auto prim_step = [&](uint32_t p, float4 m0, float4 m1, float4 m2, float4 r0, float4 r1) {
        const f3 o = F3(ray[0], ray[256], ray[512]), d = F3(ray[768], ray[1024], ray[1280]);
        const uint32_t settings = __float_as_uint(r1.w);
        r1.w = __uint_as_float(settings & RT_PRIMX_COUNT_MASK);
        const float inv[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
        // transform_ray + scale factor, shader.wgsl:604-607
        const f3 td = xform_dir(d, inv);
        scale = length(td);
        ld = td / scale;
        lo = xform_pos(o, inv);
        lidir = rcp3(ld);
        pm = (pm & RT_PM_OVERFLOW) | p | (in_shadow() ? 0u : ((settings >> RT_PRIMX_MODE_SHIFT) & 3u) << RT_PM_MODE_SHIFT); // no alpha test for shadow rays
        const bool nested = (settings & RT_PRIMX_NESTED_ROOT) != 0u;
        if (counted()) c_prims++;
        const float troot = hit_aabb_s(lo, lidir, r0, r1);
        const uint32_t root_first = __float_as_uint(r1.z), root_count = __float_as_uint(r1.w);
        best_t = closest_t * scale;
        best_i = -1;
        sp = 0;
        if (troot < -0.5f) { // shader.wgsl:448
            advance();
        } else if (nested && troot > best_t) { // see k_extend
            if (counted()) c_inner++;
            advance();
        } else {
            enter(root_first, root_count);
        }
    };

    for (;;) {
        if (++iter > RT_PATH_WATCHDOG) { // cannot happen; a bound on the loop all the same
            if (lane == 0) *a.watchdog = 1u;
            break;
        }
        const uint32_t n_prim = (uint32_t)__popcll(__ballot(st == ST_PRIM));
        const uint32_t n_inner = (uint32_t)__popcll(__ballot(st == ST_INNER));
        const uint32_t n_leaf = (uint32_t)__popcll(__ballot(st == ST_LEAF));
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(st == ST_SHADE));
        const uint32_t n_done = (uint32_t)__popcll(__ballot(st == ST_DONE));
        const uint32_t n_trav = n_prim + n_inner + n_leaf;

        // ---- path round: the lanes whose ray is traced run the bounce loop's body (shader.wgsl:900-913),
        // the lanes whose path has ended take a new camera path, and every lane that has a new ray
        // gets its candidate mask.  Run when enough lanes wait for it, measured against the lanes
        // that have traversal work: many while the wave is full, few once its paths run out.
        const uint32_t n_wait = n_shade + (drained ? 0u : n_done);
        if (n_wait != 0u && (n_trav == 0u || n_wait >= min((uint32_t)RT_PATH_HEAVY_LANES, max(1u, (n_trav + n_wait) * RT_PATH_HEAVY_NUM / RT_PATH_HEAVY_DIV)))) {
#ifdef RT_PROBE
            const unsigned long long pr_t0 = wall_clock64();
            pr_shade++; pr_shade_lanes += n_shade;
#endif
            bool fresh = false;
            f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
            bool carried = false;
            if (st == ST_SHADE) {
                f3 origin = F3(ray[0], ray[256], ray[512]), dir = F3(ray[768], ray[1024], ray[1280]);
                f3 thr = F3(pth[0], pth[256], pth[512]), lit = F3(pth[768], pth[1024], pth[1280]);
                const bool overflow = (hit_prim >> 31) != 0u;
                f3 sun_term = F3(0.0f, 0.0f, 0.0f);
                int what = RT_PATH_END;
                if (in_shadow()) {
                    // the shadow ray is traced: lighting, plus the sun term (kept where the throughput
                    // was) when nothing lies along the ray (shader.wgsl:930-936; k_shadow_resolve)
                    if (COUNT) c_shadow++;
                    if (hit_i < 0) lit = lit + thr;
                } else {
                    what = shade_segment<COUNT, SUN, false>(sc, lin_lut, a.term_prob, origin, dir, thr, lit, seed, hit_i, hit_prim & 0x7FFFFFFFu,
                                                            closest_t, sun_term, cnt);
                    if (what == RT_PATH_BOUNCE && bounce + 1u == a.max_bounces) what = RT_PATH_END; // loop bound, shader.wgsl:898
                }
                if (what == RT_PATH_BOUNCE) {
                    bounce++;
                    pth[0] = thr.x; pth[256] = thr.y; pth[512] = thr.z;
                    pth[768] = lit.x; pth[1024] = lit.y; pth[1280] = lit.z;
                    fresh = true;
                    o = origin;
                    d = dir;
                } else if (SUN && what == RT_PATH_SHADOW) {
                    bounce |= RT_PATH_SHADOW_BIT;
                    pth[0] = sun_term.x; pth[256] = sun_term.y; pth[512] = sun_term.z;
                    pth[768] = lit.x; pth[1024] = lit.y; pth[1280] = lit.z;
                    fresh = true;
                    carried = overflow;
                    o = origin;
                    d = dir;
                } else { // shader.wgsl:941-951
                    if (overflow) lit = F3(1.0f, 1.0f, 0.0f);
                    if (a.direct) {
                        float4 acc = make_float4(lit.x, lit.y, lit.z, 1.0f);
                        if (!a.overwrite) {
                            const float4 old = a.accum[pix];
                            acc = make_float4(old.x + lit.x, old.y + lit.y, old.z + lit.z, old.w + 1.0f);
                        }
                        a.accum[pix] = acc;
                    } else {
                        a.partial[pix] = make_float4(lit.x, lit.y, lit.z, 1.0f);
                    }
                    if (a.tap && pix < a.n_pixels) a.tap_seed[pix] = seed;
#ifdef RT_PROBE2
                    pr_last_start = meta[256]; pr_last_end = iter; pr_last_b = bounce & 0xFFu; pr_last_run = meta[0];
#endif
#ifdef RT_PATH_COST_MAX
                    if (a.cost) atomicMax(a.cost + meta[0], iter - meta[256]);
#else
                    if (a.cost) atomicAdd(a.cost + meta[0], iter - meta[256]);
#endif
                    st = ST_DONE;
                }
            }
#ifdef RT_PROBE
            const unsigned long long pr_t1 = wall_clock64();
            pr_ticks_shade += pr_t1 - pr_t0;
#endif
            // new camera paths for the lanes without a path (those that just ended theirs included)
            if (!drained) {
                const unsigned long long done_m = __ballot(st == ST_DONE);
                if (done_m != 0ull && run_next == run_end) {
                    // As many slots as lanes are free right now (one atomic per path round that hands
                    // out paths): a camera path that waits in a wave's private run for one of that
                    // wave's lanes, while lanes elsewhere idle, is what the end of the launch would
                    // wait for.  (Whole runs of 64 per atomic until RT_PATH_EXACT_LEFT paths per wave
                    // are left, then exact: 3.20 ms per frame with whole runs throughout, 3.16 / 3.13 /
                    // 3.12 with the switch at 64 / 256 / never whole runs.)
                    const uint32_t left = n > cursor_seen ? n - cursor_seen : 0u;
                    const uint32_t want = left > (unsigned long long)RT_PATH_EXACT_LEFT * 4u * gridDim.x ? RT_PATH_RUN_SLOTS : min(RT_PATH_RUN_SLOTS, (uint32_t)__popcll(done_m));
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
                if (done_m != 0ull && !drained) {
#ifdef RT_PROBE
                    pr_refill++;
#endif
                    const uint32_t avail = run_end - run_next;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(done_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)done_m, 0u));
                    if (st == ST_DONE && rank < avail) {
                        const uint32_t s_in = run_next + rank;
                        const uint32_t slot = a.perm ? a.perm[s_in] : s_in;
                        const uint32_t run = slot >> 6;
                        if (camera_path(a.gen, slot, o, d, seed, pix)) {
                            meta[0] = run;
                            meta[256] = iter;
                            if (COUNT) c_paths++;
                            bounce = 0u;
                            pth[0] = 1.0f; pth[256] = 1.0f; pth[512] = 1.0f;    // throughput, shader.wgsl:891
                            pth[768] = 0.0f; pth[1024] = 0.0f; pth[1280] = 0.0f; // lighting, :892
                            fresh = true;
                        }
                    }
                    run_next += min(avail, (uint32_t)__popcll(done_m));
                }
            }
            if (fresh) begin_ray(o, d, carried);
#ifdef RT_PROBE
            pr_ticks_camera += wall_clock64() - pr_t1;
#endif
            // candidate masks of the new rays (k_cull's loop: wave-uniform over the primitives)
            // Candidate masks of the new rays (k_cull's test on the same padded bounds), one ray at a
            // time against all primitives, one or two primitives per LANE: the ray's reciprocal
            // direction and scaled origin are broadcast, every lane tests its primitives, and a
            // ballot per 64 primitives is the mask.  2 tests per ray and lane instead of n_prims
            // tests per lane of a wave that is a third full; two rays per turn for two independent
            // dependency chains.
            const bool mine = st == ST_SCAN;
            unsigned long long todo = __ballot(mine);
            if (todo != 0ull) {
#ifdef RT_PROBE
                pr_scan++; pr_scan_lanes += (uint32_t)__popcll(todo);
#endif
                float b0[6] = {0, 0, 0, 0, 0, 0}, b1[6] = {0, 0, 0, 0, 0, 0};
                const bool have0 = lane < sc.n_prims, have1 = lane + 64u < sc.n_prims;
                if (have0) {
                    const float4 x = reinterpret_cast<const float4*>(sc.pbounds)[2u * lane], y = reinterpret_cast<const float4*>(sc.pbounds)[2u * lane + 1u];
                    b0[0] = x.x; b0[1] = x.y; b0[2] = x.z; b0[3] = x.w; b0[4] = y.x; b0[5] = y.y;
                }
                if (have1) {
                    const float4 x = reinterpret_cast<const float4*>(sc.pbounds)[2u * (lane + 64u)], y = reinterpret_cast<const float4*>(sc.pbounds)[2u * (lane + 64u) + 1u];
                    b1[0] = x.x; b1[1] = x.y; b1[2] = x.z; b1[3] = x.w; b1[4] = y.x; b1[5] = y.y;
                }
                f3 oi, idir;
                world_test_ray(o, d, oi, idir);
#if RT_PATH_CAMERA_BUNDLE
                // The new CAMERA rays of a path round come from one or two pixel tiles (neighbours in the
                // hand-out list): as k_cull_camera does for a tile of the staged pipeline, the wave bounds
                // them as a set (origin box, direction box), tests every primitive's bound against the set
                // in interval arithmetic -- one or two primitives per lane, conservative -- and runs the
                // exact test per ray only for the primitives that are left, the candidate's bound through
                // the scalar cache.  Same masks; taken when it is the cheaper way (few candidates).
                {
                    const bool cam = mine && bounce == 0u;
                    const unsigned long long cam_m = __ballot(cam);
                    const uint32_t n_cam = (uint32_t)__popcll(cam_m);
                    if (n_cam >= RT_PATH_CAMERA_BUNDLE) {
                        const float inf = __builtin_inff();
                        const float on[3] = {wave_min_f(cam ? o.x : inf), wave_min_f(cam ? o.y : inf), wave_min_f(cam ? o.z : inf)};
                        const float ox[3] = {wave_max_f(cam ? o.x : -inf), wave_max_f(cam ? o.y : -inf), wave_max_f(cam ? o.z : -inf)};
                        const float dn[3] = {wave_min_f(cam ? d.x : inf), wave_min_f(cam ? d.y : inf), wave_min_f(cam ? d.z : inf)};
                        const float dx[3] = {wave_max_f(cam ? d.x : -inf), wave_max_f(cam ? d.y : -inf), wave_max_f(cam ? d.z : -inf)};
                        unsigned long long cand[2];
#pragma unroll
                        for (int pass = 0; pass < 2; pass++) {
                            const float* b = pass == 0 ? b0 : b1;
                            float l_max = -inf, u_min = inf;
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                if (dn[k] > 1e-30f || dx[k] < -1e-30f) { // wave-uniform
                                    const float i0 = 1.0f / dx[k], i1 = 1.0f / dn[k];
                                    const float n0 = b[k] - ox[k], n1 = b[k] - on[k], n2 = b[3 + k] - ox[k], n3 = b[3 + k] - on[k];
                                    const float p0 = n0 * i0, p1 = n0 * i1, p2 = n1 * i0, p3 = n1 * i1;
                                    const float p4 = n2 * i0, p5 = n2 * i1, p6 = n3 * i0, p7 = n3 * i1;
                                    float lo_ = fminf(fminf(fminf(p0, p1), fminf(p2, p3)), fminf(fminf(p4, p5), fminf(p6, p7)));
                                    float hi_ = fmaxf(fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)), fmaxf(fmaxf(p4, p5), fmaxf(p6, p7)));
                                    lo_ -= fabsf(lo_) * 1e-5f + 1e-6f;
                                    hi_ += fabsf(hi_) * 1e-5f + 1e-6f;
                                    l_max = fmaxf(l_max, lo_);
                                    u_min = fminf(u_min, hi_);
                                }
                            }
                            const bool no_ray_meets = l_max > u_min || u_min < 0.0f;
                            cand[pass] = __ballot((pass == 0 ? have0 : have1) && !no_ray_meets);
                        }
                        const uint32_t n_cand = (uint32_t)__popcll(cand[0]) + (uint32_t)__popcll(cand[1]);
                        if (n_cand * 24u < n_cam * 28u) { // an exact test per candidate against two tests per ray and lane
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
                            if (cam) { mask[0] = m0; mask[1] = m1; mask[2] = m2; mask[3] = m3; }
                            todo &= ~cam_m;
                        }
                    }
                }
#endif
                auto bcast = [&](float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); };
                while (todo) {
                    const int s0 = (int)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const int s1 = todo ? (int)__builtin_ctzll(todo) : s0; // a second ray in the same turn (the first again when none is left)
                    todo &= todo - 1ull;
                    const f3 oi0 = F3(bcast(oi.x, s0), bcast(oi.y, s0), bcast(oi.z, s0)), id0 = F3(bcast(idir.x, s0), bcast(idir.y, s0), bcast(idir.z, s0));
                    const f3 oi1 = F3(bcast(oi.x, s1), bcast(oi.y, s1), bcast(oi.z, s1)), id1 = F3(bcast(idir.x, s1), bcast(idir.y, s1), bcast(idir.z, s1));
                    const unsigned long long lo0 = __ballot(have0 && overlaps_world_box(oi0, id0, b0));
                    const unsigned long long lo1 = __ballot(have0 && overlaps_world_box(oi1, id1, b0));
                    const unsigned long long hi0 = __ballot(have1 && overlaps_world_box(oi0, id0, b1));
                    const unsigned long long hi1 = __ballot(have1 && overlaps_world_box(oi1, id1, b1));
                    if ((int)lane == s0) {
                        mask[0] = (uint32_t)lo0; mask[1] = (uint32_t)(lo0 >> 32);
                        mask[2] = (uint32_t)hi0; mask[3] = (uint32_t)(hi0 >> 32);
                    }
                    if ((int)lane == s1) {
                        mask[0] = (uint32_t)lo1; mask[1] = (uint32_t)(lo1 >> 32);
                        mask[2] = (uint32_t)hi1; mask[3] = (uint32_t)(hi1 >> 32);
                    }
                }
                if (mine) advance();
            }
#ifdef RT_PROBE
            pr_heavy_ticks += wall_clock64() - pr_t0;
#endif
            continue;
        }

        if (n_trav == 0u) break; // every path has ended and no camera path is left
#ifdef RT_PROBE
        pr_trav++; pr_lanes += n_trav;
        if (drained) { pr_tail++; pr_lanes_tail += n_trav; }
#endif

        if (n_inner >= n_prim && n_inner >= n_leaf) {
#pragma unroll 1
            for (int burst = 0; burst < RT_PATH_INNER_BURST; burst++) {
                if (COUNT) r_inner++;
                if (st == ST_INNER) {
                    const float4* np = rec16(sc.pairs, cur_first * 2u);
                    const float4 b0 = np[0], b1 = np[1], b2 = np[2];
                    const uint2 dsc = *reinterpret_cast<const uint2*>(np + 3);
                    inner_step(b0, b1, b2, dsc);
                }
                if (__ballot(st == ST_INNER) == 0ull) break;
            }
        } else if (n_leaf >= n_prim) {
            const bool two = st == ST_LEAF && cur_count > 1u;
            if (COUNT) r_leaf += __ballot(two) ? 2u : 1u;
            if (st == ST_LEAF) {
                const float4* tp = rec16(tris, cur_first * 3u);
                const rt_float3 a0 = ld12(tp), a1 = ld12(tp + 1), a2 = ld12(tp + 2);
                rt_float3 b0 = a0, b1 = a1, b2 = a2;
                if (two) { b0 = ld12(tp + 3); b1 = ld12(tp + 4); b2 = ld12(tp + 5); }
                leaf_step(a0, a1, a2, b0, b1, b2, two);
            }
        } else {
            if (COUNT) r_prim++;
            if (st == ST_PRIM) {
                const uint32_t p = next_candidate();
                const float4* pp = rec16(sc.primx, p * (uint32_t)(sizeof(PrimX) / 16));
                const float4 m0 = pp[0], m1 = pp[1], m2 = pp[2], r0 = pp[3], r1 = pp[4];
                prim_step(p, m0, m1, m2, r0, r1);
            }
        }
    }
#ifdef RT_PROBE2
    { // the path of this wave that ended last: when it started, when it ended, its segments
        uint32_t best = pr_last_end, who = lane;
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t ob = (uint32_t)__shfl_xor((int)best, o, 64), ow = (uint32_t)__shfl_xor((int)who, o, 64);
            if (ob > best || (ob == best && ow < who)) { best = ob; who = ow; }
        }
        pr2_start = (uint32_t)__shfl((int)pr_last_start, (int)who, 64);
        pr2_end = best;
        pr2_b = (uint32_t)__shfl((int)pr_last_b, (int)who, 64);
        pr2_run = (uint32_t)__shfl((int)pr_last_run, (int)who, 64);
    }
#endif
#ifdef RT_PROBE
    if (a.probe && lane == 0) {
        unsigned long long* r = a.probe + 8ull * ((size_t)blockIdx.x * 4u + (threadIdx.x >> 6));
        r[0] = pr_start; r[1] = pr_drain; r[2] = wall_clock64();
        r[3] = pr_trav | ((unsigned long long)pr_tail << 32);                  // traversal rounds: all, after the drain
        r[4] = pr_lanes | ((pr_ticks_shade >> 4) << 40);                                                       // lanes with traversal work, summed over those rounds
        r[5] = pr_lanes_tail | (pr_heavy_ticks << 32);                         // ... after the drain; ticks spent in path rounds
        r[6] = pr_shade | ((unsigned long long)pr_shade_lanes << 32);
        r[7] = pr_scan | ((unsigned long long)(pr_ticks_camera >> 4) << 32);
#ifdef RT_PROBE2
        r[6] = pr2_start | ((unsigned long long)pr2_end << 32);
        r[7] = pr2_b | ((unsigned long long)pr2_run << 8) | ((unsigned long long)iter << 32);
#endif
    }
#endif
    if (COUNT) {
        const unsigned long long v[13] = {c_rays, c_prims, c_inner, c_leaf, c_tri, c_alpha, c_texel + cnt.texel, c_over, c_paths, cnt.shade, cnt.miss, cnt.kill, c_shadow};
        const int idx[13] = {C_RAYS, C_PRIMS, C_INNER, C_LEAF, C_TRI, C_ALPHA, C_TEXEL, C_OVERFLOW, C_PATHS, C_SHADE, C_MISS, C_KILL, C_SHADOW};
#pragma unroll
        for (int k = 0; k < 13; k++) {
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

// ------------------------------------------------------------------ hand-out order
//
// A persistent launch ends when its last paths do.  Two things decide how long a path keeps its lane:
// what a ray costs where the path runs (a tile full of foliage: several times the average), and how
// many segments the path has -- one of a single segment that starts with the last hand-out is over a
// few rounds later, one of four needs three more traces and three more path rounds (1080p headline
// scene, the median wave's end after the last hand-out: 0.03 / 0.23 / 0.46 ms with max_bounces 1 / 2 /
// 4).  The first is known from the call before: every path adds the wave rounds it was alive for to
// its run's word of `cost`, and k_order_items sorts the runs by it after a launch (counting sort over
// 256 levels, stable; whatever the costs are, every run gets exactly one place, because both passes
// compute the level the same way).  Of the second, one thing is known before the path is traced:
// whether it outlives its FIRST termination draw (shader.wgsl:907), which is the fifth draw of the
// pixel's random stream (after the four of camera_ray) unless BLEND alpha draws come before it.  So
// every run of 64 slots is two ITEMS -- its paths that go on after their first hit ("long") and those
// that end there or in the sky -- and in a call with many paths per lane the long items of all runs
// are handed out before the others (`split`), each kind in the order of the runs.  A call with few
// paths per lane keeps the two items of a run together: its waves start with a whole tile each, whose
// rays walk the same nodes, and that is worth more there than the order (512 x 512, one path per
// lane: 1.61 ms against 1.78 split; 1080p, eight per lane: 2.77 against 2.61).  Three small kernels
// before the launch turn places into the list of slots (`perm`): counts per item, a prefix over the
// places, the scatter.  All of this decides which lane traces which path when, nothing a path
// computes (one writer per accumulation slot); where BLEND draws shift the stream the split is merely
// a worse guess.  (Keys of cost x 2.75 for the long item and cost x 1 for the other, sorted together,
// were measured too: 2.75 ms at 1080p, i.e. the loss of coherence without the gain.)

// cost levels that count as one (a power of two: 1 = all 256 levels apart; within a level the runs keep
// their order in the image, and neighbours in the hand-out list are then neighbours in the image more
// often: 1 / 4 / 16 / 64: 2.571 / 2.581 / 2.557 / 2.579 ms of kernel at 1080p, 4.01 / 3.97 / 3.93 / 3.94 at 1440p)
#ifndef RT_ORDER_LEVEL_STEP
#define RT_ORDER_LEVEL_STEP 16u
#endif
// order[place] = run: the runs sorted by cost, expensive first.  One block; also clears the costs for
// the next call.
__global__ __launch_bounds__(1024) void k_order_runs(uint32_t* cost, uint32_t* order, uint32_t n_runs) {
    // a histogram per wave: most runs fall into a few dozen cost levels, and 1024 threads counting
    // into one set of 256 LDS words spend their time queueing for the same words
    __shared__ uint32_t bucket[16][256];
    __shared__ uint32_t red[1024];
    const uint32_t t = threadIdx.x, w = t >> 6;
    // (every pass reads eight costs per thread at a time: the reads of a plain loop wait for each other)
    uint32_t mx = 0;
    for (uint32_t base = t; base < n_runs; base += 8192u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) { const uint32_t i = base + k * 1024u; v[k] = i < n_runs ? cost[i] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) mx = max(mx, v[k]);
    }
    red[t] = mx;
    for (uint32_t k = t; k < 16u * 256u; k += 1024u) (&bucket[0][0])[k] = 0u;
    __syncthreads();
    for (uint32_t h = 512u; h > 0u; h >>= 1) {
        if (t < h) red[t] = max(red[t], red[t + h]);
        __syncthreads();
    }
    const uint32_t top = red[0];
    // level 0 = the most expensive
    const float to_level = top == 0u ? 0.0f : 255.0f / (float)top;
    auto level = [&](uint32_t c) -> uint32_t { return (255u - min(255u, (uint32_t)((float)c * to_level))) & ~(RT_ORDER_LEVEL_STEP - 1u); };
    for (uint32_t base = t; base < n_runs; base += 8192u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) { const uint32_t i = base + k * 1024u; v[k] = i < n_runs ? cost[i] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) if (base + k * 1024u < n_runs) atomicAdd(&bucket[w][level(v[k])], 1u);
    }
    __syncthreads();
    // exclusive prefix over (level, wave): level-major, so that the order is by level
    if (t < 256u) {
        uint32_t sum = 0u;
        for (uint32_t k = 0; k < 16u; k++) { const uint32_t c = bucket[k][t]; bucket[k][t] = sum; sum += c; }
        red[t] = sum; // runs at level t
    }
    __syncthreads();
    // exclusive prefix over the 256 levels: a scan inside each of the first four waves, then their totals
    uint32_t mine = 0u, incl = 0u;
    if (t < 256u) {
        mine = red[t];
        incl = mine;
        for (uint32_t o = 1u; o < 64u; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
            if ((t & 63u) >= o) incl += up;
        }
        if ((t & 63u) == 63u) red[256u + w] = incl;
    }
    __syncthreads();
    if (t < 256u) {
        uint32_t before = 0u;
        for (uint32_t k = 0; k < w; k++) before += red[256u + k];
        red[t] = before + incl - mine;
    }
    __syncthreads();
    for (uint32_t base = t; base < n_runs; base += 8192u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) { const uint32_t i = base + k * 1024u; v[k] = i < n_runs ? cost[i] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) {
            const uint32_t i = base + k * 1024u;
            if (i < n_runs) {
                const uint32_t l = level(v[k]);
                order[red[l] + atomicAdd(&bucket[w][l], 1u)] = i;
                cost[i] = 0u;
            }
        }
    }
}

// the order of a first call (no costs yet): the runs as they come
__global__ __launch_bounds__(256) void k_order_first(uint32_t* cost, uint32_t* order, uint32_t n_runs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_runs) { order[i] = i; cost[i] = 0u; }
}

struct PlanArgs {
    GenArgs gen;
    const uint32_t* order;   // place -> run
    uint32_t  n_runs;
    uint32_t  split;         // 1: the long items of all runs before the others; 0: the two items of a run next to each other
    float     term_prob;
    uint32_t* counts;        // per place: the slots of its long item, of its other item
    uint32_t* block_sums;    // per block of RT_PLAN_BLOCK_RUNS places: the same, summed
    uint32_t* perm;          // hand-out position -> queue slot
    uint32_t* cursor;        // k_path's hand-out cursor: cleared here
};
#define RT_PLAN_WAVE_RUNS  4u                            // places a wave takes, one after the other
#define RT_PLAN_BLOCK_RUNS (16u * RT_PLAN_WAVE_RUNS)   // a block of 1024 threads

// 0: the path of this slot outlives its first termination draw; 1: it does not; 2: the slot is outside the image
__device__ __forceinline__ uint32_t plan_class(const GenArgs& g, uint32_t gid, float term_prob) {
    uint32_t px, py, fs;
    if (!slot_pixel(g, gid, px, py, fs)) return 2u;
    uint32_t seed = hash21(hash21(px, py), g.first_frame + fs * g.frame_stride); // shader.wgsl:895
    (void)rnd(seed); (void)rnd(seed); (void)rnd(seed); (void)rnd(seed);        // camera_ray's four draws
    return rnd(seed) < term_prob ? 1u : 0u;                                     // shader.wgsl:907
}

// Two kernels make `perm`.  The first counts, per place, the slots of either kind, and per block of
// 64 places their sums; the second gives every block the sums of the blocks before it (each block
// adds them up for itself: a few hundred words out of the L2) and writes the slots out.
__global__ __launch_bounds__(1024) void k_plan_count(PlanArgs a) {
    __shared__ uint32_t cnt[2][RT_PLAN_BLOCK_RUNS];
    const uint32_t w = threadIdx.x >> 6;
#pragma unroll
    for (uint32_t i = 0; i < RT_PLAN_WAVE_RUNS; i++) {
        const uint32_t q = w * RT_PLAN_WAVE_RUNS + i, p = blockIdx.x * RT_PLAN_BLOCK_RUNS + q;
        uint32_t cls = 2u;
        if (p < a.n_runs) cls = plan_class(a.gen, (a.order[p] << 6) | lane_id(), a.term_prob);
        const uint32_t nl = (uint32_t)__popcll(__ballot(cls == 0u)), ns = (uint32_t)__popcll(__ballot(cls == 1u));
        if (lane_id() == 0) {
            cnt[0][q] = nl; cnt[1][q] = ns;
            if (p < a.n_runs) { a.counts[2u * p] = nl; a.counts[2u * p + 1u] = ns; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64u) { // RT_PLAN_BLOCK_RUNS = 64 places: one per lane of the first wave
        uint32_t sl = cnt[0][threadIdx.x], ss = cnt[1][threadIdx.x];
        for (int o = 32; o > 0; o >>= 1) { sl += (uint32_t)__shfl_xor((int)sl, o, 64); ss += (uint32_t)__shfl_xor((int)ss, o, 64); }
        if (threadIdx.x == 0) {
            a.block_sums[2u * blockIdx.x] = sl; a.block_sums[2u * blockIdx.x + 1u] = ss;
            if (blockIdx.x == 0) *a.cursor = 0u;
        }
    }
}

__global__ __launch_bounds__(1024) void k_plan_scatter(PlanArgs a) {
    __shared__ uint32_t red[3][16];
    __shared__ uint32_t cnt[2][RT_PLAN_BLOCK_RUNS];
    const uint32_t t = threadIdx.x, w = t >> 6, lane = lane_id(), b = blockIdx.x;
    // the long and the other slots of the blocks before this one, and the long slots of all blocks
    uint32_t sl = 0u, ss = 0u, tl = 0u;
    for (uint32_t j = t; j < gridDim.x; j += 1024u) {
        const uint2 v = reinterpret_cast<const uint2*>(a.block_sums)[j];
        tl += v.x;
        if (j < b) { sl += v.x; ss += v.y; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sl += (uint32_t)__shfl_xor((int)sl, o, 64); ss += (uint32_t)__shfl_xor((int)ss, o, 64); tl += (uint32_t)__shfl_xor((int)tl, o, 64);
    }
    if (lane == 0) { red[0][w] = sl; red[1][w] = ss; red[2][w] = tl; }
    if (t < RT_PLAN_BLOCK_RUNS) {
        const uint32_t p = b * RT_PLAN_BLOCK_RUNS + t;
        const uint2 v = p < a.n_runs ? reinterpret_cast<const uint2*>(a.counts)[p] : make_uint2(0u, 0u);
        cnt[0][t] = v.x; cnt[1][t] = v.y;
    }
    __syncthreads();
    sl = 0u; ss = 0u; tl = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 16u; k++) { sl += red[0][k]; ss += red[1][k]; tl += red[2][k]; }
    uint32_t bl = 0u, bs = 0u; // the places of this block before this wave's
    for (uint32_t k = 0; k < w * RT_PLAN_WAVE_RUNS; k++) { bl += cnt[0][k]; bs += cnt[1][k]; }
#pragma unroll
    for (uint32_t i = 0; i < RT_PLAN_WAVE_RUNS; i++) {
        const uint32_t q = w * RT_PLAN_WAVE_RUNS + i, p = b * RT_PLAN_BLOCK_RUNS + q;
        if (p >= a.n_runs) break;
        const uint32_t gid = (a.order[p] << 6) | lane;
        const uint32_t cls = plan_class(a.gen, gid, a.term_prob);
        const unsigned long long ml = __ballot(cls == 0u), ms = __ballot(cls == 1u);
        const unsigned long long mine = cls == 0u ? ml : ms;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
        uint32_t dst;
        if (a.split) dst = cls == 0u ? sl + bl + rank : tl + ss + bs + rank;
        else dst = sl + ss + bl + bs + (cls == 0u ? rank : cnt[0][q] + rank);
        if (cls < 2u) a.perm[dst] = gid;
        bl += cnt[0][q]; bs += cnt[1][q];
    }
}
