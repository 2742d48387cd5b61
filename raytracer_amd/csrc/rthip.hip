// C ABI of the MI355X path-tracing hot path (include/rthip.h): context, scene upload and
// repack, per-frame stage launches, read-back.  gfx950 only; compile with -ffp-contract=off.

#include "../../include/rthip.h"
#include "rt_kernels.h"
#include "rt_path.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

// Work memory of rthip_trace_rays_noalloc, the entry point that may not allocate: the queue
// planes, hit records and control words its kernels use, and the deep part of the traversal
// stacks.  One copy per GPU, held until the process ends; its device address is looked up once
// per context (hipGetSymbolAddress) -- the name below is only the host-side handle of the symbol.
#define RT_NOALLOC_MAX_RAYS   RTHIP_NOALLOC_MAX_RAYS
#define RT_NOALLOC_DEEP_GRID  128u   // blocks of 256 threads the extend stage gets when it needs deep stacks
#define RT_NOALLOC_CTL_BYTES  256u
#define RT_NOALLOC_PLANES_OFF ((size_t)RT_NOALLOC_CTL_BYTES)
#define RT_NOALLOC_HITS_OFF   (RT_NOALLOC_PLANES_OFF + (size_t)32 * RT_NOALLOC_MAX_RAYS)
#define RT_NOALLOC_SPILL_OFF  (RT_NOALLOC_HITS_OFF + (size_t)16 * RT_NOALLOC_MAX_RAYS)
#define RT_NOALLOC_SPILL_WORDS ((size_t)(RT_MAX_STACK - RT_LDS_LEVELS_SINGLE) * RT_NOALLOC_DEEP_GRID * 256u)
#define RT_NOALLOC_BYTES      (RT_NOALLOC_SPILL_OFF + 4 * RT_NOALLOC_SPILL_WORDS)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];
// Every call of rthip_trace_rays_noalloc on a GPU uses that GPU's one array, whatever context or
// stream it comes from.  One event per GPU, created with the first context on it and kept until the
// process ends, orders them: a call waits on the event before its first kernel and records it
// after its last, and a host mutex makes "wait, enqueue, record" one step.
#define RT_MAX_DEVICES 64
static hipEvent_t g_noalloc_done[RT_MAX_DEVICES];
static std::mutex g_noalloc_mutex[RT_MAX_DEVICES];

namespace {

thread_local std::string g_global_error;

struct EventPair {
    int        stage;
    hipEvent_t e0, e1;
};

} // namespace

struct HostNode { float aabb[6]; uint32_t first, count; };
struct HostTri { float v[3][4]; };

// Outcome of trace() (shader.wgsl:598-627) for a ray whose direction is NaN in every component:
// no box test fails (every comparison with NaN is false), so every node of every primitive is
// visited, far child first, and no triangle test passes.  Only the stack depth matters.
struct NanWalk { uint64_t inner = 0, leaf = 0, tri = 0, dropped = 0; };

struct rthip_ctx {
    int         device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int         n_cu = 0;
    std::string error;
    bool        tuning = false;  // RTHIP_TUNING was set when the context was made: the RTHIP_* experiment knobs are honoured

    // scene
    bool      have_scene = false;
    SceneDev  sc{};
    void*     d_nodes = nullptr;
    void*     d_tris = nullptr;
    void*     d_exts = nullptr;
    void*     d_primx = nullptr;
    void*     d_wbounds = nullptr;
    void*     d_gbounds = nullptr;
    void*     d_pairs = nullptr;
    void*     d_pbounds = nullptr;
    void*     d_cbounds = nullptr;
    void*     d_sbounds = nullptr;
    void*     d_tlas = nullptr;      // the bounds pyramid of a scene of many primitives, and its directory
    void*     d_tlas_dir = nullptr;
    bool      tree = false;          // the extend stage walks the pyramid instead of scanning chunks
    void*     d_ptree = nullptr;     // spatial tree over the primitives' bounds, for the candidate lists
    bool      lists = false;         // candidates come from per-ray lists (k_cull_list)
    uint32_t* cand_lists = nullptr;  // [queue slot][list_len]
    uint64_t  cand_cap = 0;          // in dwords
    uint32_t  list_len = 0;          // entries per list of the uploaded scene
    void*     d_prims = nullptr;
    void*     d_texels = nullptr;
    void*     d_env = nullptr;
    uint64_t  n_tris = 0, n_nodes = 0, n_dev_nodes = 0;
    std::vector<uint32_t> node_index_map;

    // frame state
    bool      have_uniforms = false;
    uint8_t   uniforms[608];
    uint32_t  res_x = 0, res_y = 0;
    uint32_t  reject_hist = 1;
    uint64_t  n_pixels = 0;

    // per-resolution buffers
    float4*   accum_own = nullptr;
    float4*   accum = nullptr;       // own or bound
    uint64_t  accum_bound_pixels = 0;
    bool      accum_is_bound = false;
    float4*   queue_mem = nullptr;   // 2 queues x 4 streams x n_pixels
    float4*   hits = nullptr;
    // host copy of the device tree, for the precomputed walk of a NaN-direction ray
    std::vector<HostNode> host_nodes;
    std::vector<uint32_t> prim_roots;
    NanWalk   nan_walks[RT_MAX_STACK + 1]; // by stack depth; filled on first use (no heap: the no-alloc entry point uses it too)
    bool      nan_walk_valid[RT_MAX_STACK + 1] = {};
    float4*   partial = nullptr;
    float4*   shadow_mem = nullptr; // sun option: shadow-ray queue (4 planes of queue_cap), allocated on first use
    uint64_t  shadow_cap = 0;
    uint32_t* spill = nullptr;      // deep-stack columns (levels beyond the LDS part), allocated on first use
    uint64_t  spill_words = 0;
    uint32_t  frames_cap = 0;       // frames one launch group can hold
    uint32_t* ctrl = nullptr;        // [0..63] queue counts per bounce, [64..127] cursors, [128..191] shadow-ray counts, [192..193] ray queries, [256..319] shadow-ray cursors
    int32_t*  tap_tri = nullptr;
    int32_t*  tap_prim = nullptr;
    float*    tap_t = nullptr;
    uint32_t* tap_seed = nullptr;
    bool      taps_valid = false;
    uint32_t  buf_res_x = 0, buf_res_y = 0; // resolution the work buffers are sized for
    uint64_t  accum_own_pixels = 0;
    uint64_t  queue_cap = 0;

    unsigned long long* d_counters = nullptr;

    // k_path: what each run of camera paths cost in a call of the single launch, and the runs in the
    // order (expensive first) made from that.  Two sets, used in turn: a call hands out by the order
    // that is ready -- the one the call before left if that has been made by now, else the one of the
    // call before that -- and never waits for k_order_runs of the call before while it has an older one.
    struct OrderSet {
        uint32_t* cost = nullptr;    // zero between calls (k_order_runs clears what it has read)
        uint32_t* order = nullptr;
        uint32_t  runs = 0, res_x = 0, res_y = 0, frames = 0, spread = 0; // what the order was made for (0 runs: none)
        hipEvent_t done = nullptr;   // the k_order_runs that writes this set has finished
        bool      pending = false;   // ... and the stream of the calls has not been told to wait for it yet
    };
    OrderSet  oset[2];
    uint32_t  oset_turn = 0;         // the set the next call of the single launch writes
    uint32_t  tile_order_set = 0;    // the set whose k_order_runs the copy in tile_order follows
    uint64_t  path_runs_cap = 0;
    uint32_t* path_plan = nullptr;   // k_plan_*: per place and per block of places, the slots of either kind
    uint32_t* path_perm = nullptr;   // hand-out position -> queue slot (64 words per run)
    uint32_t* tile_order = nullptr;  // the staged pipeline's copy of the order a one-frame single launch left behind (tiles, expensive first)
    uint32_t  tile_order_cap = 0, tile_order_tiles = 0, tile_order_res_x = 0, tile_order_res_y = 0; // 0 tiles: none
    // k_order_runs prepares the NEXT call of the single launch, so it runs on a stream of its own
    // behind the launch: the caller's synchronisation does not wait for it, the next call's kernels do
    hipStream_t side_stream = nullptr;
    hipEvent_t  ev_path_done = nullptr;
    bool      side_failed = false;
    uint32_t* watchdog_host = nullptr; // pinned host word that a kernel sets when one of its bounded loops ran out
    uint32_t* watchdog_dev = nullptr;  // the same word as the device sees it
    unsigned char* noalloc_scratch = nullptr; // this device's copy of `scratch` (rthip_trace_rays_noalloc)
    // grow-only staging of rthip_trace_rays and rthip_resolve
    float4*   tr_q = nullptr;
    float4*   tr_hits = nullptr;
    uint64_t  tr_cap = 0;
    uint32_t* tr_lists = nullptr;    // candidate lists of the rays of rthip_trace_rays (scenes that use them)
    uint64_t  tr_lists_cap = 0;      // in dwords
    uint32_t* resolve_out = nullptr;
    uint64_t  resolve_cap = 0;

    std::vector<EventPair>  pending_events;
    std::vector<hipEvent_t> free_events;
    rthip_timing            timing{};
};

namespace {

int fail(rthip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    else g_global_error = buf;
    return code;
}

// The RTHIP_* environment knobs (candidate walk, frames in flight, grid sizes ...) exist for
// experiments and tests.  They are looked at only by contexts created while RTHIP_TUNING was set to
// a non-zero value: a stray variable in a production environment changes nothing, and a render
// reads no environment at all.
const char* knob(const rthip_ctx* c, const char* name) { return c->tuning ? std::getenv(name) : nullptr; }

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, RTHIP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void free_dev(void*& p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}
template <typename T> void free_dev_t(T*& p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

void free_scene(rthip_ctx* c) {
    free_dev(c->d_nodes); free_dev(c->d_tris); free_dev(c->d_exts); free_dev(c->d_primx); free_dev(c->d_wbounds); free_dev(c->d_gbounds); free_dev(c->d_pairs); free_dev(c->d_pbounds); free_dev(c->d_cbounds); free_dev(c->d_sbounds); free_dev(c->d_tlas); free_dev(c->d_tlas_dir); c->tree = false; free_dev(c->d_ptree); c->lists = false;
    free_dev(c->d_prims); free_dev(c->d_texels); free_dev(c->d_env);
    c->have_scene = false;
}

// the per-frame work buffers: everything sized by the 8x8-tile slot count or the frame group
void free_work_buffers(rthip_ctx* c) {
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream); // k_order_runs may still be reading the buffers freed below
    for (auto& s : c->oset) { s.pending = false; s.runs = 0; free_dev_t(s.cost); free_dev_t(s.order); }
    free_dev_t(c->queue_mem); free_dev_t(c->hits); free_dev_t(c->partial); free_dev_t(c->spill); free_dev_t(c->shadow_mem);
    free_dev_t(c->cand_lists);
    free_dev_t(c->path_plan);
    free_dev_t(c->path_perm);
    free_dev_t(c->tile_order);
    c->tile_order_cap = 0;
    c->tile_order_tiles = 0;
    c->path_runs_cap = 0;
    c->cand_cap = 0;
    c->shadow_cap = 0;
    c->spill_words = 0;
    c->frames_cap = 0;
    c->queue_cap = 0;
    free_dev_t(c->tap_tri); free_dev_t(c->tap_prim); free_dev_t(c->tap_t); free_dev_t(c->tap_seed);
    c->taps_valid = false;
    c->buf_res_x = c->buf_res_y = 0;
}

void free_frame_buffers(rthip_ctx* c) {
    free_work_buffers(c);
    free_dev_t(c->accum_own);
    c->accum_own_pixels = 0;
    if (!c->accum_is_bound) c->accum = nullptr;
}

// double-precision 4x4 inverse (column-major); false when singular or non-finite
bool invert4(const double m[16], double out[16]) {
    double inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    for (int i = 0; i < 16; i++) {
        out[i] = inv[i] / det;
        if (!std::isfinite(out[i])) return false;
    }
    return true;
}

// Frames (samples per pixel) that travel through the stages together.  More frames in flight keep
// the thin late bounces wide enough to fill the chip; the cost is queue memory (144 bytes per
// path).  RTHIP_FRAMES_IN_FLIGHT overrides the default.
// queues and per-frame results of a frame group: at most this much of the GPU's 288 GB (and at most half
// of what is free); 96 GiB hold a 64-frame group at 3840 x 2160 (2946 Mrays/s there against 2907 with the
// 38-frame groups that 48 GiB allow)
#ifndef RT_QUEUE_BUDGET_GIB
#define RT_QUEUE_BUDGET_GIB 96
#endif
uint32_t frames_in_flight_limit(const rthip_ctx* c) {
    uint32_t want = 128; // 64 / 96 / 128 frames per group at 1080p: 2793 / 2827 / 2840 Mrays/s (the launches' tails weigh half as much)
    if (const char* e = knob(c, "RTHIP_FRAMES_IN_FLIGHT")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 128) want = (uint32_t)v;
    }
    const uint64_t slots = (uint64_t)((c->res_x + 7) / 8) * ((c->res_y + 7) / 8) * 64;
    // 160 bytes of queues and results per path slot: at most RT_QUEUE_BUDGET_GIB, and at most half of
    // what the device has free beyond what this context already holds for the purpose
    uint64_t budget = (uint64_t)RT_QUEUE_BUDGET_GIB << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const uint64_t held = (c->queue_mem ? c->queue_cap * 160ull : 0ull) + c->cand_cap * 4ull;
        budget = std::min<uint64_t>(budget, (free_b + held) / 2);
    }
    const uint64_t per_slot = 160ull + (c->lists ? 4ull * c->list_len : 0ull); // + a candidate list where the scene uses them
    while (want > 1 && (slots * want >= 0x7FFFFFFFull || c->n_pixels * want >= 0x7FFFFFFFull || slots * want * per_slot > budget)) want--;
    return want;
}

// `screen` for the current resolution: the caller's bound buffer, or a library-owned one that a
// change of resolution replaces by a zeroed one of the new size (update_resolution, main.rs:79-94).
int ensure_accum(rthip_ctx* c) {
    const uint64_t n = c->n_pixels;
    if (c->accum_is_bound) {
        if (c->accum_own) { // not in use while a caller's buffer is bound
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            free_dev_t(c->accum_own);
            c->accum_own_pixels = 0;
        }
        return RTHIP_OK;
    }
    if (c->accum_own && c->accum_own_pixels != n) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        free_dev_t(c->accum_own);
        c->accum_own_pixels = 0;
        c->accum = nullptr;
    }
    if (!c->accum_own) {
        HIP_TRY(c, hipMalloc((void**)&c->accum_own, sizeof(float4) * n));
        c->accum_own_pixels = n;
        HIP_TRY(c, hipMemsetAsync(c->accum_own, 0, sizeof(float4) * n, c->stream));
    }
    c->accum = c->accum_own;
    return RTHIP_OK;
}

// Queue slots are handed out per 8x8 pixel tile, so the work buffers are keyed on the resolution
// itself: two resolutions of equal pixel count can differ in tiles (1000x500 has 7875, 2000x250 has
// 8000).  The library-owned accumulator is keyed on the pixel count and survives a re-allocation
// of the work buffers (a larger frame group must not lose the image).
int ensure_frame_buffers(rthip_ctx* c, uint32_t frames) {
    const uint64_t n = c->n_pixels;
    const bool same_res = c->buf_res_x == c->res_x && c->buf_res_y == c->res_y;
    if (!(same_res && c->queue_mem && c->frames_cap >= frames)) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        free_work_buffers(c);
        const uint64_t per_frame = (uint64_t)((c->res_x + 7) / 8) * ((c->res_y + 7) / 8) * 64;
        const uint64_t cap = per_frame * frames;
        HIP_TRY(c, hipMalloc((void**)&c->queue_mem, sizeof(float4) * 8 * cap));
        HIP_TRY(c, hipMalloc((void**)&c->hits, sizeof(float4) * cap));
        HIP_TRY(c, hipMalloc((void**)&c->partial, sizeof(float4) * n * frames));
        HIP_TRY(c, hipMalloc((void**)&c->tap_tri, sizeof(int32_t) * n));
        HIP_TRY(c, hipMalloc((void**)&c->tap_prim, sizeof(int32_t) * n));
        HIP_TRY(c, hipMalloc((void**)&c->tap_t, sizeof(float) * n));
        HIP_TRY(c, hipMalloc((void**)&c->tap_seed, sizeof(uint32_t) * n));
        // only now do the sizes count: a failure above leaves frames_cap = 0 and the next call retries
        c->queue_cap = cap;
        c->frames_cap = frames;
        c->buf_res_x = c->res_x;
        c->buf_res_y = c->res_y;
    }
    return ensure_accum(c);
}

hipEvent_t get_event(rthip_ctx* c) {
    if (!c->free_events.empty()) {
        hipEvent_t e = c->free_events.back();
        c->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct StageTimer {
    rthip_ctx* c;
    bool       on;
    EventPair  p{};
    StageTimer(rthip_ctx* ctx, bool enabled, int stage) : c(ctx), on(enabled) {
        if (!on) return;
        p.stage = stage;
        p.e0 = get_event(c);
        p.e1 = get_event(c);
        (void)hipEventRecord(p.e0, c->stream);
    }
    ~StageTimer() {
        if (!on) return;
        (void)hipEventRecord(p.e1, c->stream);
        c->pending_events.push_back(p);
    }
};

int drain_events(rthip_ctx* c) {
    if (c->pending_events.empty()) return RTHIP_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (const EventPair& p : c->pending_events) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            c->timing.ms[p.stage] += (double)ms;
            c->timing.launches[p.stage] += 1;
        }
        c->free_events.push_back(p.e0);
        c->free_events.push_back(p.e1);
    }
    c->pending_events.clear();
    return RTHIP_OK;
}

template <typename K> int occupancy_blocks(K kernel, int block) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) nb = 2;
    return nb;
}

// k_extend has three compile-time switches; these pick the instantiation for run-time flags.
// which instantiation walks a scene's primitives: one chunk with the masks of the cull stage, the
// chunk scan, or the bounds pyramid
enum ExtMode : int { EXT_SINGLE = 0, EXT_SCAN = 1, EXT_TREE = 2, EXT_LIST = 3 };
template <bool COUNT, bool DEEP> int extend_occupancy_t(ExtMode m) {
    if (m == EXT_LIST) return occupancy_blocks(k_extend<COUNT, DEEP, true, false, false, true>, 256);
    if (m == EXT_TREE) return occupancy_blocks(k_extend<COUNT, DEEP, true, false, true>, 256);
    return m == EXT_SCAN ? occupancy_blocks(k_extend<COUNT, DEEP, true>, 256) : occupancy_blocks(k_extend<COUNT, DEEP, false>, 256);
}
int extend_occupancy(bool count, bool deep, ExtMode m) {
    if (count) return deep ? extend_occupancy_t<true, true>(m) : extend_occupancy_t<true, false>(m);
    return deep ? extend_occupancy_t<false, true>(m) : extend_occupancy_t<false, false>(m);
}
template <bool COUNT, bool DEEP> void extend_launch_t(ExtMode m, uint32_t grid, hipStream_t st, const ExtArgs& ea) {
    if (m == EXT_LIST) hipLaunchKernelGGL((k_extend<COUNT, DEEP, true, false, false, true>), dim3(grid), dim3(256), 0, st, ea);
    else if (m == EXT_TREE) hipLaunchKernelGGL((k_extend<COUNT, DEEP, true, false, true>), dim3(grid), dim3(256), 0, st, ea);
    else if (m == EXT_SCAN) hipLaunchKernelGGL((k_extend<COUNT, DEEP, true>), dim3(grid), dim3(256), 0, st, ea);
    else hipLaunchKernelGGL((k_extend<COUNT, DEEP, false>), dim3(grid), dim3(256), 0, st, ea);
}
template <bool DEEP> void shadow_extend_launch_t(ExtMode m, uint32_t grid, hipStream_t st, const ExtArgs& ea) {
    if (m == EXT_LIST) hipLaunchKernelGGL((k_extend<false, DEEP, true, true, false, true>), dim3(grid), dim3(256), 0, st, ea);
    else if (m == EXT_TREE) hipLaunchKernelGGL((k_extend<false, DEEP, true, true, true>), dim3(grid), dim3(256), 0, st, ea);
    else if (m == EXT_SCAN) hipLaunchKernelGGL((k_extend<false, DEEP, true, true>), dim3(grid), dim3(256), 0, st, ea);
    else hipLaunchKernelGGL((k_extend<false, DEEP, false, true>), dim3(grid), dim3(256), 0, st, ea);
}
template <bool COUNT, bool SUN> void path_launch_t(bool deep, uint32_t grid, hipStream_t st, const PathArgs& pa) {
    if (deep) hipLaunchKernelGGL((k_path<COUNT, SUN, true>), dim3(grid), dim3(256), 0, st, pa);
    else hipLaunchKernelGGL((k_path<COUNT, SUN, false>), dim3(grid), dim3(256), 0, st, pa);
}
void path_launch(bool count, bool sun, bool deep, uint32_t grid, hipStream_t st, const PathArgs& pa) {
    if (count) { if (sun) path_launch_t<true, true>(deep, grid, st, pa); else path_launch_t<true, false>(deep, grid, st, pa); }
    else { if (sun) path_launch_t<false, true>(deep, grid, st, pa); else path_launch_t<false, false>(deep, grid, st, pa); }
}
template <bool COUNT, bool SUN> int path_occupancy_t(bool deep) {
    return deep ? occupancy_blocks(k_path<COUNT, SUN, true>, 256) : occupancy_blocks(k_path<COUNT, SUN, false>, 256);
}
int path_occupancy(bool count, bool sun, bool deep) {
    if (count) return sun ? path_occupancy_t<true, true>(deep) : path_occupancy_t<true, false>(deep);
    return sun ? path_occupancy_t<false, true>(deep) : path_occupancy_t<false, false>(deep);
}
void cull_list_launch(uint32_t list_len, uint32_t grid, hipStream_t st, const CullListArgs& la) {
    if (list_len == RT_LIST_LEN_LONG) hipLaunchKernelGGL((k_cull_list<RT_LIST_LEN_LONG>), dim3(grid), dim3(256), 0, st, la);
    else hipLaunchKernelGGL((k_cull_list<RT_LIST_LEN>), dim3(grid), dim3(256), 0, st, la);
}
void shadow_extend_launch(bool deep, ExtMode m, uint32_t grid, hipStream_t st, const ExtArgs& ea) {
    if (deep) shadow_extend_launch_t<true>(m, grid, st, ea);
    else shadow_extend_launch_t<false>(m, grid, st, ea);
}
void extend_launch(bool count, bool deep, ExtMode m, uint32_t grid, hipStream_t st, const ExtArgs& ea) {
    if (count) { if (deep) extend_launch_t<true, true>(m, grid, st, ea); else extend_launch_t<true, false>(m, grid, st, ea); }
    else { if (deep) extend_launch_t<false, true>(m, grid, st, ea); else extend_launch_t<false, false>(m, grid, st, ea); }
}


// After the stream is drained: did a kernel run into the bound on its loops?  (It cannot by
// construction; the bound is there so that a defect ends as an error code and not as a hung GPU.)
int check_watchdog(rthip_ctx* c) {
    if (c->watchdog_host && *const_cast<volatile uint32_t*>(c->watchdog_host) != 0u) {
        *c->watchdog_host = 0u;
        return fail(c, RTHIP_ERR_HIP, "a wave of the path kernel gave up at its iteration bound: the image of the last render is incomplete");
    }
    return RTHIP_OK;
}

} // namespace

extern "C" {

int rthip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* rthip_last_error(const rthip_ctx* ctx) { return ctx ? ctx->error.c_str() : g_global_error.c_str(); }

int rthip_create(int device_ordinal, rthip_ctx** out) {
    if (!out) return fail(nullptr, RTHIP_ERR_INVALID, "rthip_create: out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(nullptr, RTHIP_ERR_NO_DEVICE, "no HIP device is visible");
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, RTHIP_ERR_NO_DEVICE, "device ordinal %d out of range (0..%d)", device_ordinal, n - 1);
    DeviceGuard g(device_ordinal);
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RTHIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device_ordinal, prop.gcnArchName);
    rthip_ctx* c = new rthip_ctx();
    c->device = device_ordinal;
    c->n_cu = prop.multiProcessorCount;
    if (const char* e = std::getenv("RTHIP_TUNING")) c->tuning = std::atoi(e) != 0;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(nullptr, RTHIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->stream = c->own_stream;
    e = hipMalloc((void**)&c->ctrl, sizeof(uint32_t) * 512);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_counters, sizeof(Counters));
    if (e == hipSuccess) e = hipMemset(c->d_counters, 0, sizeof(Counters));
    if (e != hipSuccess) { rthip_destroy(c); return fail(nullptr, RTHIP_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    // a pinned host word the device can write: a kernel whose bounded loop ran out says so here, and
    // the calls that synchronise report it (without it short calls keep the staged pipeline)
    if (hipHostMalloc((void**)&c->watchdog_host, sizeof(uint32_t), hipHostMallocMapped) == hipSuccess) {
        *c->watchdog_host = 0u;
        if (hipHostGetDevicePointer((void**)&c->watchdog_dev, c->watchdog_host, 0) != hipSuccess) c->watchdog_dev = nullptr;
    } else {
        c->watchdog_host = nullptr;
        (void)hipGetLastError();
    }
    // every GPU holds its own copy of the static work array at its own address
    e = hipGetSymbolAddress((void**)&c->noalloc_scratch, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !c->noalloc_scratch) { rthip_destroy(c); return fail(nullptr, RTHIP_ERR_HIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e)); }
    if (device_ordinal < RT_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_noalloc_mutex[device_ordinal]);
        if (!g_noalloc_done[device_ordinal] && hipEventCreateWithFlags(&g_noalloc_done[device_ordinal], hipEventDisableTiming) != hipSuccess) {
            g_noalloc_done[device_ordinal] = nullptr; // without it overlapping calls race, as documented
            (void)hipGetLastError();
        }
    }
    *out = c;
    return RTHIP_OK;
}

void rthip_destroy(rthip_ctx* c) {
    if (!c) return;
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending_events) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (auto& e : c->free_events) (void)hipEventDestroy(e);
    free_scene(c);
    free_frame_buffers(c);
    free_dev_t(c->ctrl);
    if (c->ev_path_done) (void)hipEventDestroy(c->ev_path_done);
    for (auto& s : c->oset) if (s.done) (void)hipEventDestroy(s.done);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    free_dev_t(c->d_counters);
    if (c->watchdog_host) (void)hipHostFree(c->watchdog_host);
    free_dev_t(c->tr_q); free_dev_t(c->tr_hits); free_dev_t(c->tr_lists); free_dev_t(c->resolve_out);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int rthip_set_stream(rthip_ctx* c, void* hip_stream) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->side_stream) { // what still runs behind the last call of the single launch ends before a new stream takes over
        HIP_TRY(c, hipStreamSynchronize(c->side_stream));
        for (auto& s : c->oset) s.pending = false;
    }
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RTHIP_OK;
}

// Walks the device tree the way a NaN-direction ray does, with a stack of `depth` entries whose
// overflowing pushes are dropped (shader.wgsl:306-313), and counts the steps.
static const NanWalk& nan_walk(rthip_ctx* c, uint32_t depth) {
    depth = std::min<uint32_t>(depth, RT_MAX_STACK);
    if (c->nan_walk_valid[depth]) return c->nan_walks[depth];
    NanWalk w;
    uint32_t stack[RT_MAX_STACK + 1];
    for (uint32_t root : c->prim_roots) {
        uint32_t sp = 0;
        uint32_t first = c->host_nodes[root].first, count = c->host_nodes[root].count;
        for (;;) {
            if (count > 0) { // leaf: every triangle is tested, none is hit
                w.leaf++;
                w.tri += count;
                if (sp == 0) break;
                const HostNode& n = c->host_nodes[stack[--sp]];
                first = n.first;
                count = n.count;
            } else {         // inner: both children "hit"; left < right is false, so right first
                w.inner++;
                if (sp < depth) stack[sp++] = first;
                else w.dropped++;
                const HostNode& n = c->host_nodes[first + 1];
                first = n.first;
                count = n.count;
            }
        }
    }
    c->nan_walks[depth] = w;
    c->nan_walk_valid[depth] = true;
    return c->nan_walks[depth];
}

int rthip_upload_scene(rthip_ctx* c, const rthip_scene_desc* s) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!s) return fail(c, RTHIP_ERR_INVALID, "upload_scene: null scene");
    if (!s->prims || s->n_prims == 0) return fail(c, RTHIP_ERR_INVALID, "upload_scene: no primitives");
    if (!s->nodes || s->n_nodes == 0) return fail(c, RTHIP_ERR_INVALID, "upload_scene: no BVH nodes");
    if (s->n_tris && (!s->tris || !s->tri_exts)) return fail(c, RTHIP_ERR_INVALID, "upload_scene: null triangle arrays");
    if (!s->env || s->env_w == 0 || s->env_h == 0) return fail(c, RTHIP_ERR_INVALID, "upload_scene: no environment map");
    if (s->n_texels && !s->texels) return fail(c, RTHIP_ERR_INVALID, "upload_scene: null texel array");
    if (s->n_nodes >= 0x7FFFFFFFull || s->n_tris >= 0x7FFFFFFFull || s->n_prims >= 0x7FFFFFFFu)
        return fail(c, RTHIP_ERR_INVALID, "upload_scene: more than 2^31-1 nodes, triangles or primitives");
    // the kernels address records by a 32-bit count of 16-byte units: 64 GiB per array
    if (s->n_tris * 3ull >= 0xFFFFFFFFull || s->n_prims * 7ull >= 0xFFFFFFFFull)
        return fail(c, RTHIP_ERR_UNSUPPORTED, "upload_scene: triangle or primitive array of 64 GiB or more (%llu triangles, %u primitives)",
                    (unsigned long long)s->n_tris, s->n_prims);
    // the extend stage keeps the primitive index in 29 bits of a packed word
    if (s->n_prims > RT_PM_PRIM)
        return fail(c, RTHIP_ERR_UNSUPPORTED, "upload_scene: more than 2^29-1 primitives (%u)", s->n_prims);

    const HostNode*   nodes = static_cast<const HostNode*>(s->nodes);
    const HostTri*    tris = static_cast<const HostTri*>(s->tris);
    const PrimitiveD* prims = static_cast<const PrimitiveD*>(s->prims);

    // The kernels walk the tree without bounds checks, so the tree is validated here: every
    // leaf range inside the triangle array, every child pair inside the node array and behind
    // its parent (which also rules out cycles; the reference builder appends children after
    // their parent, scene.rs:1041-1046).
    for (uint64_t i = 0; i < s->n_nodes; i++) {
        const HostNode& n = nodes[i];
        if (n.count > 0) {
            if ((uint64_t)n.first + n.count > s->n_tris)
                return fail(c, RTHIP_ERR_INVALID, "upload_scene: leaf node %llu covers triangles [%u, %u) of %llu", (unsigned long long)i, n.first, n.first + n.count, (unsigned long long)s->n_tris);
        } else {
            if ((uint64_t)n.first + 1 >= s->n_nodes || (uint64_t)n.first <= i)
                return fail(c, RTHIP_ERR_INVALID, "upload_scene: inner node %llu has children at %u (must lie after it, inside %llu nodes)", (unsigned long long)i, n.first, (unsigned long long)s->n_nodes);
        }
    }
    std::vector<uint32_t> roots;
    for (uint32_t p = 0; p < s->n_prims; p++) {
        if (prims[p].bvh_idx >= s->n_nodes) return fail(c, RTHIP_ERR_INVALID, "upload_scene: primitive %u has bvh_idx %u of %llu nodes", p, prims[p].bvh_idx, (unsigned long long)s->n_nodes);
        roots.push_back(prims[p].bvh_idx);
    }
    roots.push_back(0);
    std::sort(roots.begin(), roots.end());
    roots.erase(std::unique(roots.begin(), roots.end()), roots.end());

    // Device node layout: same records, same tree.  Each BVH (a contiguous run starting at a
    // primitive's root) is shifted so that its root sits at an odd index; the builder appends
    // children in pairs after the root, so every sibling pair then starts on a 64-byte line.
    std::vector<uint32_t> map(s->n_nodes);
    uint64_t dcur = 0;
    for (size_t k = 0; k < roots.size(); k++) {
        const uint64_t seg_begin = roots[k], seg_end = (k + 1 < roots.size()) ? roots[k + 1] : s->n_nodes;
        if ((dcur & 1ull) == 0) dcur++;
        for (uint64_t i = seg_begin; i < seg_end; i++) map[i] = (uint32_t)(dcur + (i - seg_begin));
        dcur += seg_end - seg_begin;
    }
    const uint64_t n_dev_nodes = dcur;
    if (n_dev_nodes >= 0x7FFFFFFFull) return fail(c, RTHIP_ERR_INVALID, "upload_scene: device node array exceeds 2^31-1 records");
    std::vector<HostNode> dev_nodes(n_dev_nodes);
    std::memset(dev_nodes.data(), 0, sizeof(HostNode) * n_dev_nodes);
    for (uint64_t i = 0; i < s->n_nodes; i++) {
        HostNode n = nodes[i];
        if (n.count == 0) n.first = map[n.first];
        dev_nodes[map[i]] = n;
    }
    // padding records stay zero; nothing references them

    // Sibling-pair records for the inner-node step: boxes of the two children and a descriptor
    // each, at the position of the first child.  A node with two parents would make two records
    // overlap; the reference's builder cannot produce one.
    std::vector<uint32_t> pairs((size_t)n_dev_nodes * 8, 0u);
    {
        std::vector<uint8_t> claimed(n_dev_nodes, 0);
        for (uint64_t i = 0; i < n_dev_nodes; i++) {
            const HostNode& n = dev_nodes[i];
            if (n.count != 0 || n.first == 0) continue; // leaves and padding
            for (uint32_t ch = 0; ch < 2; ch++) {
                const uint64_t ci = (uint64_t)n.first + ch;
                if (claimed[ci]) return fail(c, RTHIP_ERR_INVALID, "upload_scene: node %llu (device numbering) is the child of two nodes: not a tree", (unsigned long long)ci);
                claimed[ci] = 1;
                const HostNode& cn = dev_nodes[ci];
                std::memcpy(&pairs[(size_t)n.first * 8 + 6 * ch], cn.aabb, 24);
                pairs[(size_t)n.first * 8 + 12 + ch] = (cn.count < 16u && cn.first < 0x08000000u) ? (cn.first | (cn.count << 27)) : (RT_STACK_RELOAD | (uint32_t)ci);
            }
        }
    }

    // Triangles: (p0, e1, e2); the edges are the same f32 subtractions the shader performs per
    // test (shader.wgsl:340-341), done once.  Order = uploaded order.
    std::vector<float> dev_tris((size_t)s->n_tris * 12);
    for (uint64_t i = 0; i < s->n_tris; i++) {
        const HostTri& t = tris[i];
        float* o = &dev_tris[(size_t)i * 12];
        for (int k = 0; k < 3; k++) {
            o[k] = t.v[0][k];
            o[4 + k] = t.v[1][k] - t.v[0][k];
            o[8 + k] = t.v[2][k] - t.v[0][k];
        }
        o[3] = o[7] = o[11] = 0.0f;
    }

    std::vector<PrimX> primx(s->n_prims);
    const uint32_t n_prim_groups = (s->n_prims + 7u) / 8u;
    std::vector<float> wbounds((size_t)n_prim_groups * 48, 0.0f); // padding slots stay zero; the scan masks them off
    std::vector<float> pbounds((size_t)s->n_prims * 8, 0.0f);
    for (uint32_t p = 0; p < s->n_prims; p++) {
        const PrimitiveD& pr = prims[p];
        PrimX& x = primx[p];
        std::memset(&x, 0, sizeof x);
        for (int col = 0; col < 4; col++)
            for (int row = 0; row < 3; row++) x.inv[col * 3 + row] = pr.inv_transform[col * 4 + row];
        x.root = map[pr.bvh_idx];
        {
            const HostNode& rn = dev_nodes[x.root];
            std::memcpy(x.root_box, rn.aabb, sizeof x.root_box);
            x.root_first = rn.first;
            if (rn.count > RT_PRIMX_COUNT_MASK)
                return fail(c, RTHIP_ERR_UNSUPPORTED, "upload_scene: primitive %u has a root leaf of %u triangles", p, rn.count);
            x.root_word = rn.count | ((pr.material.alpha_settings & 3u) << RT_PRIMX_MODE_SHIFT);
        }
        x.alpha_settings = pr.material.alpha_settings;
        {
            const HostNode& rn = dev_nodes[map[pr.bvh_idx]];
            if (rn.count == 0) {
                bool inside = true;
                for (int ch = 0; ch < 2; ch++) {
                    const HostNode& cn = dev_nodes[rn.first + ch];
                    for (int k = 0; k < 3; k++) inside = inside && cn.aabb[k] >= rn.aabb[k] && cn.aabb[3 + k] <= rn.aabb[3 + k];
                }
                if (inside) x.root_word |= RT_PRIMX_NESTED_ROOT;
            }
        }
        x.albedo_offset = pr.material.albedo.offset;
        x.albedo_size = pr.material.albedo.size;
        x.albedo_texcoord = pr.material.albedo_texcoord;
        // world bound of the root box through the inverse of inv_transform (in double), padded
        const float inf = std::numeric_limits<float>::infinity();
        float wb[6] = {-inf, -inf, -inf, inf, inf, inf};
        double im[16], fw[16];
        for (int k = 0; k < 16; k++) im[k] = (double)pr.inv_transform[k];
        const HostNode& root = nodes[pr.bvh_idx];
        bool finite_box = true;
        for (int k = 0; k < 6; k++) finite_box = finite_box && std::isfinite(root.aabb[k]);
        if (finite_box && invert4(im, fw)) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            double mag = 0.0;
            bool ok = true;
            for (int corner = 0; corner < 8; corner++) {
                const double x0 = root.aabb[(corner & 1) ? 3 : 0], y0 = root.aabb[(corner & 2) ? 4 : 1], z0 = root.aabb[(corner & 4) ? 5 : 2];
                const double w = fw[3] * x0 + fw[7] * y0 + fw[11] * z0 + fw[15];
                for (int r = 0; r < 3; r++) {
                    double v = fw[r] * x0 + fw[4 + r] * y0 + fw[8 + r] * z0 + fw[12 + r];
                    if (!(std::fabs(w - 1.0) < 1e-9)) ok = false; // projective: give up on culling
                    if (!std::isfinite(v)) ok = false;
                    lo[r] = std::min(lo[r], v);
                    hi[r] = std::max(hi[r], v);
                    mag = std::max(mag, std::fabs(v));
                }
            }
            if (ok) {
                const double ext = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]});
                const double pad = 1e-4 * (ext + mag) + 1e-5;
                for (int r = 0; r < 3; r++) {
                    wb[r] = (float)(lo[r] - pad);
                    wb[3 + r] = (float)(hi[r] + pad);
                    wb[r] = std::nextafter(wb[r], -inf);
                    wb[3 + r] = std::nextafter(wb[3 + r], inf);
                }
            }
        }
        float* grp = &wbounds[(size_t)(p >> 3) * 48];
        for (int k = 0; k < 6; k++) grp[k * 8 + (p & 7u)] = wb[k];
        for (int k = 0; k < 6; k++) pbounds[(size_t)p * 8 + k] = wb[k];
    }

    // union bounds per group of 8 and per chunk of 128 primitives
    const uint32_t n_chunks = (s->n_prims + 127u) / 128u;
    std::vector<float> gbounds((size_t)n_prim_groups * 8, 0.0f), cbounds((size_t)n_chunks * 8, 0.0f);
    {
        const float big = std::numeric_limits<float>::infinity();
        for (uint32_t g = 0; g < n_prim_groups; g++) for (int k = 0; k < 3; k++) { gbounds[(size_t)g * 8 + k] = big; gbounds[(size_t)g * 8 + 3 + k] = -big; }
        for (uint32_t ch = 0; ch < n_chunks; ch++) for (int k = 0; k < 3; k++) { cbounds[(size_t)ch * 8 + k] = big; cbounds[(size_t)ch * 8 + 3 + k] = -big; }
        for (uint32_t p = 0; p < s->n_prims; p++) {
            const float* grp = &wbounds[(size_t)(p >> 3) * 48];
            float* gb = &gbounds[(size_t)(p >> 3) * 8];
            float* cb = &cbounds[(size_t)(p >> 7) * 8];
            for (int k = 0; k < 3; k++) {
                const float lo = grp[k * 8 + (p & 7u)], hi = grp[(3 + k) * 8 + (p & 7u)];
                gb[k] = std::min(gb[k], lo); gb[3 + k] = std::max(gb[3 + k], hi);
                cb[k] = std::min(cb[k], lo); cb[3 + k] = std::max(cb[3 + k], hi);
            }
        }
    }

    // at most 64 super-chunks of 2^chunk_shift chunks each
    uint32_t chunk_shift = 0;
    while (((n_chunks + (1u << chunk_shift) - 1u) >> chunk_shift) > 64u) chunk_shift++;
    const uint32_t n_super = (n_chunks + (1u << chunk_shift) - 1u) >> chunk_shift;
    std::vector<float> sbounds((size_t)n_super * 8, 0.0f);
    {
        const float big = std::numeric_limits<float>::infinity();
        for (uint32_t su = 0; su < n_super; su++) for (int k = 0; k < 3; k++) { sbounds[(size_t)su * 8 + k] = big; sbounds[(size_t)su * 8 + 3 + k] = -big; }
        for (uint32_t ch = 0; ch < n_chunks; ch++) {
            float* sb = &sbounds[(size_t)(ch >> chunk_shift) * 8];
            const float* cb = &cbounds[(size_t)ch * 8];
            for (int k = 0; k < 3; k++) { sb[k] = std::min(sb[k], cb[k]); sb[3 + k] = std::max(sb[3 + k], cb[3 + k]); }
        }
    }

    // Scenes of very many primitives: an 8-ary pyramid of union bounds over the primitives in index
    // order, for the TREE instantiation of the extend kernel.  Level 0 = the records of wbounds; an
    // entity of level L + 1 is the union of the existing slots of record e of level L.
    // It pays from about 500 primitives on (measured on instanced terrain, scan vs pyramid: 4.30 vs
    // 4.17 Grays/s at 392 primitives, 4.23 vs 4.40 at 512, 3.47 vs 3.99 at 1568, 1.1 vs 2.5 at
    // 64800) -- if consecutive primitives are neighbours in space.  For
    // a list in incoherent order every union bound spans the scene, the walk tests every record
    // lane by lane, and the wave-uniform chunk scan is the cheaper way to test everything (by a
    // third at 8192 shuffled instances).  Estimate of the box tests per ray: a record is tested when
    // the ray meets its entity, with probability area(entity) / area(scene) for a random ray.  The
    // estimate is made for every scene of more than 128 primitives: it also decides between the scan
    // and the per-ray candidate lists below.
    uint32_t tree_min = 511u;
    const char* tree_min_env = knob(c, "RTHIP_TREE_MIN_PRIMS"); // tuning / test knob: pyramid above, scan up to it
    const bool tree_min_forced = tree_min_env != nullptr;
    if (tree_min_forced) tree_min = (uint32_t)std::strtoul(tree_min_env, nullptr, 10);
    const bool pyramid_possible = s->n_prims > 8u && (uint64_t)s->n_prims <= (1ull << (3u * RT_TLAS_MAX_LEVELS));
    bool tree = s->n_prims > tree_min && pyramid_possible;
    bool coherent = true;
    std::vector<float> tlas;
    std::vector<uint32_t> tlas_dir(1u + 2u * RT_TLAS_MAX_LEVELS, 0u);
    if (pyramid_possible && (tree || s->n_prims > 128u)) {
        tlas = wbounds;
        std::vector<double> areas; // of the entities above level 0, all levels
        auto half_area = [](const float* b) {
            const double dx = (double)b[3] - b[0], dy = (double)b[4] - b[1], dz = (double)b[5] - b[2];
            return dx * dy + dy * dz + dz * dx;
        };
        const float inf = std::numeric_limits<float>::infinity();
        float scene_box[6] = {inf, inf, inf, -inf, -inf, -inf};
        uint32_t level = 0, first = 0, cnt = s->n_prims;
        for (;;) {
            const uint32_t n_rec = (cnt + 7u) / 8u;
            tlas_dir[1u + level] = first;
            tlas_dir[1u + RT_TLAS_MAX_LEVELS + level] = cnt;
            const bool top = cnt <= 8u;
            const uint32_t up_first = first + n_rec, up_rec = (n_rec + 7u) / 8u;
            if (!top) tlas.resize((size_t)(up_first + up_rec) * 48, 0.0f);
            for (uint32_t e = 0; e < n_rec; e++) {
                const float* rec = &tlas[(size_t)(first + e) * 48];
                float b[6] = {inf, inf, inf, -inf, -inf, -inf};
                const uint32_t have = std::min(8u, cnt - e * 8u);
                for (uint32_t k = 0; k < have; k++)
                    for (int ax = 0; ax < 3; ax++) { b[ax] = std::min(b[ax], rec[ax * 8 + k]); b[3 + ax] = std::max(b[3 + ax], rec[(3 + ax) * 8 + k]); }
                if (top) { std::memcpy(scene_box, b, sizeof b); break; }
                areas.push_back(half_area(b));
                float* up = &tlas[(size_t)(up_first + (e >> 3)) * 48];
                for (int k = 0; k < 6; k++) up[k * 8 + (e & 7u)] = b[k];
            }
            if (top) break;
            first = up_first;
            cnt = n_rec;
            level++;
        }
        tlas_dir[0] = level;
        const double scene_area = half_area(scene_box);
        double tests = 8.0;
        for (double a : areas) {
            const double pr = a / scene_area;
            tests += 8.0 * (pr < 1.0 ? pr : 1.0); // a NaN (unbounded primitives) counts as 1
        }
        coherent = tests < 0.5 * (double)s->n_prims;
        if (!tree_min_forced && !coherent) tree = false;
    }
    // A long list that is not coherent gets per-ray candidate lists from a spatial tree over the
    // primitives' bounds instead (k_cull_list): a 4-ary tree from median splits of the centres along
    // the longest axis (two per node), one primitive per leaf, stored level by level (node j of level
    // d: children 4j .. 4j + 3 of level d + 1).  Primitives without a finite bound cannot be placed in
    // it; such a scene keeps the scan.
    // (shuffled instances, scan against lists: 288 primitives 2.6 against 3.3 Grays/s, 512: 1.5 against 3.2.)
    bool lists = !tree && !coherent && s->n_prims > 128u && !tree_min_forced;
    if (const char* e = knob(c, "RTHIP_CANDIDATE_WALK")) { // test / tuning knob: scan, pyramid or list for any scene of more than 128 primitives
        const std::string want(e);
        if (s->n_prims > 128u && (want == "scan" || want == "pyramid" || want == "list")) {
            lists = want == "list";
            // no pyramid was built (more than 8^RT_TLAS_MAX_LEVELS primitives, or a bound that is not finite): such a scene keeps the scan
            tree = want == "pyramid" && !tlas.empty();
        }
    }
    uint32_t ptree_depth = 0; // inner levels: 4^depth >= primitives
    while ((1ull << (2u * ptree_depth)) < s->n_prims) ptree_depth++;
    if (ptree_depth > RT_LIST_MAX_DEPTH) lists = false;
    for (uint32_t p = 0; p < s->n_prims && lists; p++)
        for (int k = 0; k < 6; k++) lists = lists && std::isfinite(pbounds[(size_t)p * 8 + k]);
    std::vector<uint32_t> ptree;
    if (lists) {
        const uint32_t np = s->n_prims;
        std::vector<uint32_t> order(np);
        for (uint32_t p = 0; p < np; p++) order[p] = p;
        auto centre = [&](uint32_t p, int ax) { return 0.5f * pbounds[(size_t)p * 8 + ax] + 0.5f * pbounds[(size_t)p * 8 + 3 + ax]; };
        // median split of order[b, e) along the longest axis of its centres; returns the middle
        auto halve = [&](uint32_t b, uint32_t e) -> uint32_t {
            if (e - b < 2u) return e;
            float lo[3], hi[3];
            for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; }
            for (uint32_t i = b; i < e; i++)
                for (int k = 0; k < 3; k++) { const float v = centre(order[i], k); lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v); }
            int ax = 0;
            if (hi[1] - lo[1] > hi[ax] - lo[ax]) ax = 1;
            if (hi[2] - lo[2] > hi[ax] - lo[ax]) ax = 2;
            const uint32_t mid = b + (e - b + 1u) / 2;
            std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e,
                             [&](uint32_t x, uint32_t y) { const float cx = centre(x, ax), cy = centre(y, ax); return cx < cy || (cx == cy && x < y); });
            return mid;
        };
        struct Range { uint32_t begin, end, level, node; };
        ptree.assign((size_t)32 * rt_list_level_base(ptree_depth), 0u); // records of 128 bytes, levels 0 .. depth - 1
        std::vector<Range> todo{{0u, np, 0u, 0u}};
        while (!todo.empty()) {
            const Range r = todo.back();
            todo.pop_back();
            // two median splits: four children of at most ceil(count / 4) primitives
            const uint32_t m1 = halve(r.begin, r.end);
            const uint32_t m0 = halve(r.begin, m1), m2 = halve(m1, r.end);
            const uint32_t cut[5] = {r.begin, m0, m1, m2, r.end};
            uint32_t* rec = &ptree[(size_t)(rt_list_level_base(r.level) + r.node) * 32];
            float* box = reinterpret_cast<float*>(rec);
            for (uint32_t ch = 0; ch < 4; ch++) {
                const uint32_t b = cut[ch], e = cut[ch + 1];
                if (b == e) { rec[24 + ch] = RT_LIST_NONE; continue; }
                const float inf = std::numeric_limits<float>::infinity();
                float bb[6] = {inf, inf, inf, -inf, -inf, -inf};
                for (uint32_t i = b; i < e; i++)
                    for (int k = 0; k < 3; k++) {
                        bb[k] = std::min(bb[k], pbounds[(size_t)order[i] * 8 + k]);
                        bb[3 + k] = std::max(bb[3 + k], pbounds[(size_t)order[i] * 8 + 3 + k]);
                    }
                for (int k = 0; k < 6; k++) box[4 * k + ch] = bb[k];
                if (e - b == 1u) {
                    rec[24 + ch] = RT_LIST_LEAF | order[b];
                } else {
                    rec[24 + ch] = RT_LIST_INNER;
                    todo.push_back({b, e, r.level + 1u, 4u * r.node + ch});
                }
            }
        }
    }

    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    auto upload = [&](void*& dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(&dst, bytes ? bytes : 16);
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    HIP_TRY(c, upload(c->d_nodes, dev_nodes.data(), sizeof(HostNode) * n_dev_nodes));
    HIP_TRY(c, upload(c->d_pairs, pairs.data(), sizeof(uint32_t) * pairs.size()));
    HIP_TRY(c, upload(c->d_tris, dev_tris.data(), sizeof(float) * dev_tris.size()));
    HIP_TRY(c, upload(c->d_primx, primx.data(), sizeof(PrimX) * primx.size()));
    HIP_TRY(c, upload(c->d_wbounds, wbounds.data(), sizeof(float) * wbounds.size()));
    HIP_TRY(c, upload(c->d_gbounds, gbounds.data(), sizeof(float) * gbounds.size()));
    HIP_TRY(c, upload(c->d_pbounds, pbounds.data(), sizeof(float) * pbounds.size()));
    HIP_TRY(c, upload(c->d_cbounds, cbounds.data(), sizeof(float) * cbounds.size()));
    HIP_TRY(c, upload(c->d_sbounds, sbounds.data(), sizeof(float) * sbounds.size()));
    if (lists) {
        HIP_TRY(c, upload(c->d_ptree, ptree.data(), sizeof(uint32_t) * ptree.size()));
        c->lists = true;
        // longer lists pay where rays cross many bounds (8192 shuffled instances: 1476 -> 1777 Mrays/s)
        // and cost where they do not (2048: 2669 -> 2623): the cull kernel's LDS and the bytes double
        c->list_len = s->n_prims >= 4096u ? RT_LIST_LEN_LONG : RT_LIST_LEN;
        if (const char* e = knob(c, "RTHIP_LIST_LEN")) { // tuning / test knob
            const int v = std::atoi(e);
            if (v == (int)RT_LIST_LEN || v == (int)RT_LIST_LEN_LONG) c->list_len = (uint32_t)v;
        }
    }
    if (tree) {
        HIP_TRY(c, upload(c->d_tlas, tlas.data(), sizeof(float) * tlas.size()));
        HIP_TRY(c, upload(c->d_tlas_dir, tlas_dir.data(), sizeof(uint32_t) * tlas_dir.size()));
        c->tree = true;
    }
    HIP_TRY(c, upload(c->d_exts, s->tri_exts, (size_t)96 * s->n_tris));
    HIP_TRY(c, upload(c->d_prims, s->prims, (size_t)240 * s->n_prims));
    HIP_TRY(c, upload(c->d_texels, s->texels, sizeof(uint32_t) * s->n_texels));
    HIP_TRY(c, upload(c->d_env, s->env, sizeof(float) * 4 * (size_t)s->env_w * s->env_h));

    c->sc.nodes = static_cast<const float4*>(c->d_nodes);
    c->sc.tris = static_cast<const float4*>(c->d_tris);
    c->sc.exts = static_cast<const float4*>(c->d_exts);
    c->sc.primx = static_cast<const PrimX*>(c->d_primx);
    c->sc.wbounds = static_cast<const float*>(c->d_wbounds);
    c->sc.n_prim_groups = n_prim_groups;
    c->sc.pairs = static_cast<const float4*>(c->d_pairs);
    c->sc.pbounds = static_cast<const float*>(c->d_pbounds);
    c->sc.gbounds = static_cast<const float*>(c->d_gbounds);
    c->sc.cbounds = static_cast<const float*>(c->d_cbounds);
    c->sc.n_chunks = n_chunks;
    c->sc.chunk_shift = chunk_shift;
    c->sc.n_super = n_super;
    c->sc.sbounds = static_cast<const float*>(c->d_sbounds);
    c->sc.prims = static_cast<const PrimitiveD*>(c->d_prims);
    c->sc.texels = static_cast<const uint32_t*>(c->d_texels);
    c->sc.env = static_cast<const float4*>(c->d_env);
    c->sc.n_texels = s->n_texels;
    c->sc.env_w = s->env_w;
    c->sc.env_h = s->env_h;
    c->sc.n_prims = s->n_prims;
    c->n_tris = s->n_tris;
    c->n_nodes = s->n_nodes;
    c->n_dev_nodes = n_dev_nodes;
    c->prim_roots.resize(s->n_prims);
    for (uint32_t p = 0; p < s->n_prims; p++) c->prim_roots[p] = primx[p].root;
    c->host_nodes.swap(dev_nodes);
    for (bool& v : c->nan_walk_valid) v = false;
    (void)nan_walk(c, 23); // the default stack depth; other depths are walked at their first render
    c->node_index_map.swap(map);
    c->have_scene = true;
    c->reject_hist = 1;
    for (auto& s : c->oset) s.runs = 0; // what the runs of camera paths cost in the old scene says nothing about this one
    c->tile_order_tiles = 0;
    return RTHIP_OK;
}

int rthip_set_frame_uniforms(rthip_ctx* c, const void* blob608) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!blob608) return fail(c, RTHIP_ERR_INVALID, "set_frame_uniforms: null blob");
    uint32_t res[2], reject;
    std::memcpy(res, static_cast<const uint8_t*>(blob608) + 576, 8);
    std::memcpy(&reject, static_cast<const uint8_t*>(blob608) + 592, 4);
    if (res[0] == 0 || res[1] == 0 || (uint64_t)res[0] * res[1] >= 0x7FFFFFFFull)
        return fail(c, RTHIP_ERR_INVALID, "set_frame_uniforms: resolution %ux%u is empty or above 2^31-1 pixels", res[0], res[1]);
    const uint64_t np = (uint64_t)res[0] * res[1];
    if (c->accum_is_bound && np != c->accum_bound_pixels)
        return fail(c, RTHIP_ERR_INVALID, "set_frame_uniforms: %ux%u does not match the bound accumulator (%llu pixels)", res[0], res[1], (unsigned long long)c->accum_bound_pixels);
    std::memcpy(c->uniforms, blob608, 608);
    const bool resized = (res[0] != c->res_x || res[1] != c->res_y);
    c->res_x = res[0];
    c->res_y = res[1];
    c->n_pixels = np;
    if (reject == 1 || resized) c->reject_hist = 1;
    c->have_uniforms = true;
    return RTHIP_OK;
}

void rthip_default_params(rthip_render_params* p) {
    if (!p) return;
    p->max_bounces = 8;
    p->term_prob = 0.5f;
    p->aperture_radius = 0.025f;
    p->stack_depth = 23;
    p->enable_sun = 0;
    p->flags = 0;
}

int rthip_render(rthip_ctx* c, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride, const rthip_render_params* params) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->have_scene) return fail(c, RTHIP_ERR_STATE, "render: no scene uploaded");
    if (!c->have_uniforms) return fail(c, RTHIP_ERR_STATE, "render: no frame uniforms set");
    rthip_render_params P;
    if (params) P = *params;
    else rthip_default_params(&P);
    if (P.max_bounces == 0 || P.max_bounces > 63) return fail(c, RTHIP_ERR_INVALID, "render: max_bounces %u outside 1..63", P.max_bounces);
    if (!(P.term_prob >= 0.0f && P.term_prob < 1.0f)) return fail(c, RTHIP_ERR_INVALID, "render: term_prob %g outside [0,1)", (double)P.term_prob);
    if (P.stack_depth == 0 || P.stack_depth > RTHIP_MAX_STACK_DEPTH) return fail(c, RTHIP_ERR_INVALID, "render: stack_depth %u outside 1..%u", P.stack_depth, RTHIP_MAX_STACK_DEPTH);
    if (P.enable_sun > 1) return fail(c, RTHIP_ERR_INVALID, "render: enable_sun %u is neither 0 nor 1", P.enable_sun);
    uint32_t prim_count;
    std::memcpy(&prim_count, c->uniforms + 600, 4);
    if (prim_count != c->sc.n_prims) return fail(c, RTHIP_ERR_INVALID, "render: uniforms say %u primitives, scene has %u", prim_count, c->sc.n_prims);
    (void)hipGetLastError(); // an error left behind by other code of the process is not this call's

    DeviceGuard g(c->device);
    if (n_frames == 0) {
        // No sample, but the reject_hist protocol still holds: an accumulator that was to be
        // overwritten by the first frame now holds zero samples (a GPU of a group that gets no
        // frame of a short render must not discard, later, what a reduce left in its buffer).
        if (c->reject_hist == 1 && c->accum && (c->accum_is_bound || c->accum_own_pixels == c->n_pixels)) {
            HIP_TRY(c, hipMemsetAsync(c->accum, 0, sizeof(float4) * c->n_pixels, c->stream));
            c->reject_hist = 0;
        }
        return RTHIP_OK;
    }
    const uint32_t group = std::min<uint32_t>(n_frames, frames_in_flight_limit(c));
    int rc = ensure_frame_buffers(c, group);
    if (rc != RTHIP_OK) return rc;
    if (c->accum_is_bound && c->accum_bound_pixels != c->n_pixels) return fail(c, RTHIP_ERR_STATE, "render: bound accumulator size mismatch");

    const bool count = (P.flags & RTHIP_FLAG_COUNTERS) != 0;
    const bool taps = (P.flags & RTHIP_FLAG_TAPS) != 0;
    const bool timing = (P.flags & RTHIP_FLAG_TIMING) != 0;

    // camera constants of camera_ray (shader.wgsl:744-755), identical for every pixel
    float cam[12];
    std::memcpy(cam, c->uniforms + 512, 48);
    GenArgs ga{};
    const float fwd[3] = {cam[0], cam[1], cam[2]};
    const float fovy = cam[3];
    ga.fov_factor = (std::sin(fovy / 2.0f) / std::cos(fovy / 2.0f)) * 2.0f;
    {
        // right = normalize(cross(forward, (0,0,1))), up = normalize(cross(right, forward))
        float r[3] = {fwd[1] * 1.0f - fwd[2] * 0.0f, fwd[2] * 0.0f - fwd[0] * 1.0f, fwd[0] * 0.0f - fwd[1] * 0.0f};
        float rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        for (int k = 0; k < 3; k++) r[k] = r[k] / rl;
        float u[3] = {r[1] * fwd[2] - r[2] * fwd[1], r[2] * fwd[0] - r[0] * fwd[2], r[0] * fwd[1] - r[1] * fwd[0]};
        float ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        for (int k = 0; k < 3; k++) u[k] = u[k] / ul;
        std::memcpy(ga.right, r, sizeof r);
        std::memcpy(ga.up, u, sizeof u);
    }
    std::memcpy(ga.forward, fwd, sizeof fwd);
    ga.origin[0] = cam[4]; ga.origin[1] = cam[5]; ga.origin[2] = cam[6];
    ga.focus = cam[7];
    ga.aperture_radius = P.aperture_radius;
    ga.aspect = (float)c->res_x / (float)c->res_y;
    ga.res_x = c->res_x;
    ga.res_y = c->res_y;
    ga.tiles_x = (c->res_x + 7) / 8;
    ga.n_tiles = ga.tiles_x * ((c->res_y + 7) / 8);
    ga.n_pixels = (uint32_t)c->n_pixels;
    ga.frame_stride = frame_stride;
    ga.counters = count ? c->d_counters : nullptr;
    // Calls of one or two frames (an interactive host: one sample per displayed frame) are too
    // short to average out the cost of expensive image regions over a wave's many reservations:
    // their runs are built from 4x4 blocks of four far-apart tiles (with the guided reservations of
    // the extend stage: 4.8 -> 4.2 ms for one 1080p frame of the atrium).  Longer calls keep whole
    // tiles, which is worth 14 % of their throughput.  RTHIP_SPREAD (0, 2, 4) overrides for experiments.
    int spread_env = -1;
    if (const char* e = knob(c, "RTHIP_SPREAD")) spread_env = std::atoi(e);
    int path_blocks_env = 0;
    if (const char* e = knob(c, "RTHIP_PATH_BLOCKS_PER_CU")) path_blocks_env = std::atoi(e);
    bool staged_only = (P.flags & RTHIP_FLAG_STAGED) != 0;
    if (const char* e = knob(c, "RTHIP_FORCE_STAGED")) staged_only = staged_only || std::atoi(e) != 0; // tests: a whole module through the staged pipeline
    int tile_lpt_env = 1;
    if (const char* e = knob(c, "RTHIP_STAGED_TILE_ORDER")) tile_lpt_env = std::atoi(e); // experiment knob: 0 = the staged pipeline queues its tiles as they lie in the image
    int keep_env = 1;
    if (const char* e = knob(c, "RTHIP_PATH_KEEP_ORDER")) keep_env = std::atoi(e); // experiment knob: 0 = a call that discards the image starts from the plain order
    int split_env = -1;
    if (const char* e = knob(c, "RTHIP_PATH_SPLIT")) split_env = std::atoi(e); // experiment knob: 1 / 0 = long paths of all runs first / every run's paths together
    int lpt_env = 1;
    if (const char* e = knob(c, "RTHIP_PATH_LONGEST_FIRST")) lpt_env = std::atoi(e); // experiment knob: 0 = runs as they come

    PathQueue q[2];
    for (int k = 0; k < 2; k++) {
        const uint64_t cap = c->queue_cap;
        float4* base = c->queue_mem + (size_t)k * 4 * cap;
        q[k].s0 = base; q[k].s1 = base + cap; q[k].s2 = base + 2 * cap; q[k].s3 = base + 3 * cap;
    }

    const NanWalk& nanw = nan_walk(c, std::min<uint32_t>(P.stack_depth, RT_MAX_STACK));
    const bool multi = c->sc.n_chunks > 1u;
    const ExtMode ext_mode = !multi ? EXT_SINGLE : (c->tree ? EXT_TREE : (c->lists ? EXT_LIST : EXT_SCAN));
    const bool deep = P.stack_depth > rt_lds_levels(multi, c->tree);
    // with a handful of primitives the candidate test saves less than it costs
    const bool cull = !multi && c->sc.n_prims > RT_CULL_MIN_PRIMS;
    bool cull_camera = true;
    if (const char* e = knob(c, "RTHIP_CULL_CAMERA")) cull_camera = std::atoi(e) != 0; // diagnostic: 0 = the general cull kernel for camera rays too
    bool fuse_generate = true;
    if (const char* e = knob(c, "RTHIP_GENERATE_IN_CULL")) fuse_generate = std::atoi(e) != 0; // experiment knob: 0 = k_generate, then the camera cull
    int ext_blocks = extend_occupancy(count, deep, ext_mode);
    bool ext_blocks_fixed = false;
    if (const char* e = knob(c, "RTHIP_EXTEND_BLOCKS_PER_CU")) { // tuning knob; the default comes from the occupancy query
        const int v = std::atoi(e);
        if (v > 0 && v <= 16) { ext_blocks = v; ext_blocks_fixed = true; }
    }

    // The staged pipeline queues its tiles in the order of what they cost when the single launch last
    // rendered one frame of this shape (expensive first: the last rays of every bounce's queue are then
    // cheap ones, and the extend launches end sooner after their queues do: 20 frames 28.5 -> 27.9 ms,
    // 64 frames 84.8 -> 84.3).  A call that would go through the staged pipeline and finds no such
    // order renders its first frame as a single launch, which leaves one behind.
    auto have_tile_order = [&]() {
        return c->tile_order != nullptr && c->tile_order_tiles == ga.n_tiles && c->tile_order_res_x == c->res_x && c->tile_order_res_y == c->res_y;
    };
    const uint64_t path_max_slots = (uint64_t)RT_PATH_MAX_SLOTS * (uint64_t)c->n_cu / 256u;
    const bool tile_order_wanted = tile_lpt_env != 0 && lpt_env != 0 && spread_env <= 0 && !staged_only && !multi && c->watchdog_dev != nullptr;
    uint32_t nf = 0;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += nf) {
        nf = std::min(group, n_frames - f0);
        if (tile_order_wanted && nf > 1u && (uint64_t)ga.n_tiles * 64 * nf > path_max_slots && (uint64_t)ga.n_tiles * 64 <= path_max_slots && !have_tile_order()) {
            nf = 1u;
            // (what is left of the call then runs as it would have: staged, unless it has become small enough for the single launch)
        }
        const uint64_t slots = (uint64_t)ga.n_tiles * 64 * nf;
        // A call of one to three frames does not fill the resident waves of the chip with rays (one
        // 1080p frame: four camera rays per lane, fewer in every bounce after): fewer, fuller waves
        // finish it sooner.  One 1080p frame per call, blocks per CU 8 / 6 / 5 / 4 / 3: 4.09 / 3.80 /
        // 3.74 / 3.78 / 3.95 ms; two frames 8 / 6: 5.57 / 5.39; from four frames on all 8 are best.
        int blocks = ext_blocks;
        if (!ext_blocks_fixed) blocks = std::min<int>(ext_blocks, 4 + (int)std::min<uint64_t>((slots + (1u << 21) - 1) >> 21, 64));
        const uint32_t ext_grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * blocks, (slots + 255) / 256);
        if (P.enable_sun && c->shadow_cap < c->queue_cap) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            free_dev_t(c->shadow_mem);
            c->shadow_cap = 0;
            HIP_TRY(c, hipMalloc((void**)&c->shadow_mem, sizeof(float4) * 4 * c->queue_cap));
            c->shadow_cap = c->queue_cap;
        }
        if (ext_mode == EXT_LIST && c->cand_cap < c->queue_cap * c->list_len) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            free_dev_t(c->cand_lists);
            c->cand_cap = 0;
            HIP_TRY(c, hipMalloc((void**)&c->cand_lists, sizeof(uint32_t) * c->list_len * c->queue_cap));
            c->cand_cap = c->queue_cap * c->list_len;
        }
        if (deep) { // stack levels beyond the LDS part: one dword column per thread of the extend grid
            const uint64_t words = (uint64_t)(RT_MAX_STACK - rt_lds_levels(multi, c->tree)) * ext_grid * 256;
            if (c->spill_words < words) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                free_dev_t(c->spill);
                c->spill_words = 0;
                HIP_TRY(c, hipMalloc((void**)&c->spill, sizeof(uint32_t) * words));
                c->spill_words = words;
            }
        }
        const uint32_t shade_grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * 8, (slots + 255) / 256);
        const uint32_t gen_grid = (uint32_t)((slots + 255) / 256);
        const bool overwrite = (c->reject_hist == 1);
        const bool tap_group = taps && f0 == 0;
        ga.out = q[0];
        ga.count_out = c->ctrl;
        ga.first_frame = first_frame + f0 * frame_stride;
        ga.n_frames = nf;
        const bool short_call = nf <= 2u && ga.n_tiles >= 64u;
        ga.spread_shift = short_call ? 2u : 0u;
        if (spread_env == 0 || ((spread_env == 2 || spread_env == 4) && ga.n_tiles >= 64u)) ga.spread_shift = (uint32_t)spread_env;
        // A small frame group runs as ONE persistent launch in which a lane owns a whole path
        // (rt_path.h): the staged pipeline below would spend most of such a call in the tails of its
        // per-bounce launches.  Same code per path, same results; scenes of one candidate chunk only.
        // (the bound was measured on 256 CUs; a partition of the chip gets its share of it)
        const bool fused = !staged_only && slots <= path_max_slots && !multi && c->watchdog_dev != nullptr;
        // queue counts and cursors start at zero (the single launch has one cursor, which the kernels that make its hand-out order clear)
        if (!(fused && lpt_env != 0)) HIP_TRY(c, hipMemsetAsync(c->ctrl, 0, sizeof(uint32_t) * 512, c->stream));
        if (fused) {
            PathArgs pa{};
            pa.sc = c->sc;
            pa.gen = ga;
            pa.gen.tile_order = nullptr; // the single launch has its own list (perm)
            pa.gen.spread_shift = (ga.n_tiles >= 64u && spread_env > 0) ? (spread_env == 4 ? 4u : 2u) : 0u;
            pa.gen.counters = nullptr;
            pa.cursor = c->ctrl + 64;
            pa.n_slots = (uint32_t)slots;
            pa.max_bounces = P.max_bounces;
            pa.stack_depth = P.stack_depth;
            pa.term_prob = P.term_prob;
            pa.partial = c->partial;
            pa.accum = c->accum;
            pa.direct = nf == 1u ? 1u : 0u;
            pa.overwrite = overwrite ? 1u : 0u;
            pa.no_cull = cull ? 0u : 1u;
            pa.tap = tap_group ? 1u : 0u;
            pa.n_pixels = (uint32_t)c->n_pixels;
            pa.tap_tri = c->tap_tri;
            pa.tap_prim = c->tap_prim;
            pa.tap_t = c->tap_t;
            pa.tap_seed = c->tap_seed;
            pa.counters = c->d_counters;
            pa.nan_inner = nanw.inner;
            pa.nan_leaf = nanw.leaf;
            pa.nan_tri = nanw.tri;
            pa.nan_dropped = nanw.dropped;
            pa.nan_overflow = nanw.dropped ? 1u : 0u;
            const bool sun = P.enable_sun != 0, pdeep = P.stack_depth > RT_PATH_LDS_LEVELS;
            int pblocks = path_occupancy(count, sun, pdeep);
            if (path_blocks_env > 0) pblocks = std::min(pblocks, path_blocks_env);
            const uint32_t pgrid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * pblocks, (slots + 255) / 256);
            pa.watchdog = c->watchdog_dev;
            pa.spill = nullptr;
            if (pdeep) { // stack levels beyond the LDS column: one dword column per thread of this grid
                const uint64_t words = (uint64_t)(RT_MAX_STACK - RT_PATH_LDS_LEVELS) * pgrid * 256;
                if (c->spill_words < words) {
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    free_dev_t(c->spill);
                    c->spill_words = 0;
                    HIP_TRY(c, hipMalloc((void**)&c->spill, sizeof(uint32_t) * words));
                    c->spill_words = words;
                }
                pa.spill = c->spill;
            }
            // Hand-out order (rt_path.h): the camera paths of expensive tiles first, and of every tile
            // first those that outlive their first termination draw, by what the runs cost in the
            // last call of this shape (resolution, frames per call).  A new shape or a new scene starts
            // without costs.
            const uint32_t n_runs = (uint32_t)(slots >> 6);
            // long paths of all runs first where a lane gets six camera paths or more (measured: 1080p and up on 256 CUs)
            uint32_t split = slots >= 6ull * 256ull * pgrid ? 1u : 0u;
            if (split_env >= 0) split = split_env ? 1u : 0u;
            if (lpt_env != 0) {
                // (the second stream is made before anything of this call is launched: a queue that
                // appears while a persistent kernel runs had that kernel's first launch take 10 ms instead of 3)
                if (!c->side_stream && !c->side_failed) {
                    if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
                        hipEventCreateWithFlags(&c->ev_path_done, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&c->oset[0].done, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&c->oset[1].done, hipEventDisableTiming) != hipSuccess) {
                        (void)hipGetLastError();
                        if (c->side_stream) { (void)hipStreamDestroy(c->side_stream); c->side_stream = nullptr; }
                        c->side_failed = true;
                    }
                }
                if (nf == 1u && pa.gen.spread_shift == 0u && c->tile_order_cap < n_runs) {
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    if (c->side_stream) HIP_TRY(c, hipStreamSynchronize(c->side_stream));
                    free_dev_t(c->tile_order);
                    c->tile_order_cap = 0; c->tile_order_tiles = 0;
                    HIP_TRY(c, hipMalloc((void**)&c->tile_order, sizeof(uint32_t) * n_runs));
                    c->tile_order_cap = n_runs;
                }
                if (c->path_runs_cap < n_runs) {
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    if (c->side_stream) HIP_TRY(c, hipStreamSynchronize(c->side_stream));
                    for (auto& s : c->oset) { s.pending = false; s.runs = 0; free_dev_t(s.cost); free_dev_t(s.order); }
                    free_dev_t(c->path_plan);
                    free_dev_t(c->path_perm);
                    c->path_runs_cap = 0;
                    for (auto& s : c->oset) {
                        HIP_TRY(c, hipMalloc((void**)&s.cost, sizeof(uint32_t) * n_runs));
                        HIP_TRY(c, hipMalloc((void**)&s.order, sizeof(uint32_t) * (size_t)n_runs));
                    }
                    HIP_TRY(c, hipMalloc((void**)&c->path_plan, sizeof(uint32_t) * (2 * (size_t)n_runs + 2 * ((size_t)n_runs / RT_PLAN_BLOCK_RUNS + 1))));
                    HIP_TRY(c, hipMalloc((void**)&c->path_perm, sizeof(uint32_t) * 64 * (size_t)n_runs));
                    c->path_runs_cap = n_runs;
                }
                // This call's costs, and the order made from them behind it, go to set `mine`; `last` is
                // what the call before wrote.  (A call that discards the image keeps the order: while
                // the camera moves every call discards, and what the tiles cost a frame or two ago
                // still says more than nothing.)
                rthip_ctx::OrderSet& mine = c->oset[c->oset_turn];
                rthip_ctx::OrderSet& last = c->oset[c->oset_turn ^ 1u];
                auto fits = [&](const rthip_ctx::OrderSet& s) {
                    return s.runs == n_runs && s.res_x == c->res_x && s.res_y == c->res_y && s.frames == nf && s.spread == pa.gen.spread_shift &&
                           (keep_env != 0 || !overwrite);
                };
                auto wait_for = [&](rthip_ctx::OrderSet& s) -> hipError_t {
                    if (!s.pending) return hipSuccess;
                    s.pending = false;
                    return hipStreamWaitEvent(c->stream, s.done, 0);
                };
                // (an event that has been reached needs no wait in the stream)
                for (auto& s : c->oset)
                    if (s.pending && hipEventQuery(s.done) == hipSuccess) s.pending = false;
                (void)hipGetLastError(); // hipErrorNotReady is not an error here
                // the kernels of this call read and write `mine`: whatever the side stream still does with it comes first
                HIP_TRY(c, wait_for(mine));
                const uint32_t* order = nullptr;
                if (fits(last) && (!last.pending || !fits(mine))) {
                    HIP_TRY(c, wait_for(last));      // made by now, or the only one there is
                    order = last.order;
                    if (!fits(mine)) hipLaunchKernelGGL(k_order_first, dim3((n_runs + 255u) / 256u), dim3(256), 0, c->stream, mine.cost, mine.order, n_runs); // clears the costs
                } else if (fits(mine)) {
                    order = mine.order;              // the call before is still being ordered: the order of the one before it
                } else {
                    hipLaunchKernelGGL(k_order_first, dim3((n_runs + 255u) / 256u), dim3(256), 0, c->stream, mine.cost, mine.order, n_runs);
                    order = mine.order;              // a new shape or a new scene starts without costs
                }
                mine.runs = n_runs; mine.res_x = c->res_x; mine.res_y = c->res_y; mine.frames = nf; mine.spread = pa.gen.spread_shift;
                PlanArgs pl{};
                pl.gen = pa.gen;
                pl.order = order;
                pl.n_runs = n_runs;
                pl.split = split;
                pl.term_prob = P.term_prob;
                pl.counts = c->path_plan;
                pl.block_sums = c->path_plan + 2 * (size_t)n_runs;
                pl.perm = c->path_perm;
                pl.cursor = pa.cursor;
                const uint32_t plan_grid = (n_runs + RT_PLAN_BLOCK_RUNS - 1u) / RT_PLAN_BLOCK_RUNS;
                hipLaunchKernelGGL(k_plan_count, dim3(plan_grid), dim3(1024), 0, c->stream, pl);
                hipLaunchKernelGGL(k_plan_scatter, dim3(plan_grid), dim3(1024), 0, c->stream, pl);
                pa.perm = c->path_perm;
                pa.cost = mine.cost;
                pa.n_slots = (uint32_t)(c->n_pixels * nf); // the slots that are inside the image
            }
#ifdef RT_PROBE
            // diagnostic build only (tools/path_probe.py): per-wave records of this launch appended to $RTHIP_PROBE_FILE
            static unsigned long long* path_probe_buf = nullptr;
            if (!path_probe_buf) (void)hipMalloc((void**)&path_probe_buf, sizeof(unsigned long long) * 8 * 4 * 65536);
            pa.probe = std::getenv("RTHIP_PROBE_FILE") ? path_probe_buf : nullptr;
#endif
            {
                StageTimer t(c, timing, RTHIP_STAGE_PATH);
                path_launch(count, sun, pdeep, pgrid, c->stream, pa);
            }
#ifdef RT_PROBE
            if (pa.probe) {
                std::vector<unsigned long long> h((size_t)pgrid * 32);
                (void)hipStreamSynchronize(c->stream);
                (void)hipMemcpy(h.data(), path_probe_buf, h.size() * 8, hipMemcpyDeviceToHost);
                if (FILE* f = std::fopen(std::getenv("RTHIP_PROBE_FILE"), "ab")) {
                    const unsigned long long hdr[2] = {1000, (unsigned long long)pgrid * 4};
                    std::fwrite(hdr, 8, 2, f);
                    std::fwrite(h.data(), 8, h.size(), f);
                    std::fclose(f);
                }
            }
#endif
            if (!pa.direct) {
                StageTimer t(c, timing, RTHIP_STAGE_ACCUMULATE);
                const uint32_t acc_grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * 8, (c->n_pixels + 255) / 256);
                hipLaunchKernelGGL(k_accumulate, dim3(acc_grid), dim3(256), 0, c->stream, c->accum, (const float4*)c->partial,
                                   (uint32_t)c->n_pixels, nf, overwrite ? 1u : 0u);
            }
            if (pa.cost) {
                // the order for the calls to come: behind this launch, beside whatever the caller does next
                rthip_ctx::OrderSet& mine = c->oset[c->oset_turn];
                hipStream_t os = c->stream;
                if (c->side_stream && c->ev_path_done && mine.done) {
                    HIP_TRY(c, hipEventRecord(c->ev_path_done, c->stream));
                    HIP_TRY(c, hipStreamWaitEvent(c->side_stream, c->ev_path_done, 0));
                    os = c->side_stream;
                }
                hipLaunchKernelGGL(k_order_runs, dim3(1), dim3(1024), 0, os, mine.cost, mine.order, n_runs);
                if (nf == 1u && pa.gen.spread_shift == 0u && c->tile_order != nullptr && c->tile_order_cap >= n_runs) {
                    // one frame, whole tiles: the order of the runs is an order of the tiles; the staged pipeline keeps a copy
                    HIP_TRY(c, hipMemcpyAsync(c->tile_order, mine.order, sizeof(uint32_t) * n_runs, hipMemcpyDeviceToDevice, os));
                    c->tile_order_tiles = n_runs; c->tile_order_res_x = c->res_x; c->tile_order_res_y = c->res_y;
                    c->tile_order_set = c->oset_turn;
                }
                if (os != c->stream) {
                    HIP_TRY(c, hipEventRecord(mine.done, os));
                    mine.pending = true;
                }
                c->oset_turn ^= 1u;
            }
            c->reject_hist = 0; // main.rs:403
            if (tap_group) c->taps_valid = true;
            continue;
        }
        ga.tile_order = nullptr;
        if (tile_order_wanted && ga.spread_shift == 0u && have_tile_order()) {
            rthip_ctx::OrderSet& ts = c->oset[c->tile_order_set]; // (a later k_order_runs of the set only makes the wait longer)
            if (ts.pending) {
                HIP_TRY(c, hipStreamWaitEvent(c->stream, ts.done, 0));
                ts.pending = false;
            }
            ga.tile_order = c->tile_order;
        }
        // camera rays and their candidate masks in one launch where the camera cull runs at all
        const bool generate_in_cull = cull && cull_camera && fuse_generate && P.max_bounces > 0u;
        if (!generate_in_cull) {
            StageTimer t(c, timing, RTHIP_STAGE_GENERATE);
            hipLaunchKernelGGL(k_generate, dim3(gen_grid), dim3(256), 0, c->stream, ga);
        }
        for (uint32_t b = 0; b < P.max_bounces; b++) {
            ExtArgs ea{};
            ea.sc = c->sc;
            ea.tlas = static_cast<const float4*>(c->d_tlas);
            ea.tlas_dir = static_cast<const uint32_t*>(c->d_tlas_dir);
            ea.lists = c->cand_lists;
            ea.list_len = c->list_len;
            ea.in = q[b & 1];
            ea.hits = c->hits;
            ea.spill = c->spill;
            ea.count_in = c->ctrl + b;
            ea.cursor = c->ctrl + 64 + b;
            ea.stack_depth = P.stack_depth;
            ea.tap = (tap_group && b == 0) ? 1u : 0u;
            ea.n_pixels = (uint32_t)c->n_pixels;
            ea.tap_tri = c->tap_tri;
            ea.tap_prim = c->tap_prim;
            ea.tap_t = c->tap_t;
            ea.counters = c->d_counters;
            ea.nan_inner = nanw.inner;
            ea.nan_leaf = nanw.leaf;
            ea.nan_tri = nanw.tri;
            ea.nan_dropped = nanw.dropped;
            ea.nan_overflow = nanw.dropped ? 1u : 0u;
            ea.no_cull = cull ? 0u : 1u;
            if (cull) {
                CullArgs ca{};
                ca.sc = c->sc;
                ca.in = q[b & 1];
                ca.masks = reinterpret_cast<uint4*>(c->hits);
                ca.count_in = c->ctrl + b;
                StageTimer t(c, timing, RTHIP_STAGE_CULL);
                // camera rays: a wave's slots are one pixel tile, whose rays meet few primitives
                if (b == 0 && generate_in_cull) hipLaunchKernelGGL(k_cull_camera<true>, dim3(shade_grid), dim3(256), 0, c->stream, ca, ga);
                else if (b == 0 && cull_camera) hipLaunchKernelGGL(k_cull_camera<false>, dim3(shade_grid), dim3(256), 0, c->stream, ca, ga);
                else hipLaunchKernelGGL(k_cull, dim3(shade_grid), dim3(256), 0, c->stream, ca);
            }
            if (ext_mode == EXT_LIST) {
                CullListArgs la{static_cast<const float4*>(c->d_ptree), q[b & 1], c->cand_lists, c->ctrl + b};
                StageTimer t(c, timing, RTHIP_STAGE_CULL);
                cull_list_launch(c->list_len, shade_grid, c->stream, la);
            }
#ifdef RT_PROBE
            // diagnostic build only (tools/tail_probe.py): per-wave timestamps of this launch appended to $RTHIP_PROBE_FILE
            static unsigned long long* probe_buf = nullptr;
            if (!probe_buf) (void)hipMalloc((void**)&probe_buf, sizeof(unsigned long long) * 8 * 4 * 65536);
            ea.probe = std::getenv("RTHIP_PROBE_FILE") ? probe_buf : nullptr;
#endif
            {
                StageTimer t(c, timing, RTHIP_STAGE_EXTEND);
                extend_launch(count, deep, ext_mode, ext_grid, c->stream, ea);
            }
#ifdef RT_PROBE
            if (ea.probe) {
                std::vector<unsigned long long> h((size_t)ext_grid * 32);
                (void)hipStreamSynchronize(c->stream);
                (void)hipMemcpy(h.data(), probe_buf, h.size() * 8, hipMemcpyDeviceToHost);
                if (FILE* f = std::fopen(std::getenv("RTHIP_PROBE_FILE"), "ab")) {
                    const unsigned long long hdr[2] = {b, (unsigned long long)ext_grid * 4};
                    std::fwrite(hdr, 8, 2, f);
                    std::fwrite(h.data(), 8, h.size(), f);
                    std::fclose(f);
                }
            }
#endif
            ShadeArgs sa{};
            sa.sc = c->sc;
            sa.in = q[b & 1];
            sa.out = q[(b + 1) & 1];
            sa.hits = c->hits;
            sa.count_in = c->ctrl + b;
            sa.count_out = c->ctrl + b + 1;
            sa.partial = c->partial;
            sa.tap_seed = tap_group ? c->tap_seed : nullptr;
            sa.term_prob = P.term_prob;
            sa.last_bounce = (b + 1 == P.max_bounces) ? 1u : 0u;
            sa.first_bounce = (b == 0) ? 1u : 0u;
            sa.n_pixels = (uint32_t)c->n_pixels;
            sa.counters = c->d_counters;
            {
                StageTimer t(c, timing, RTHIP_STAGE_SHADE);
                if (P.enable_sun) {
                    const uint64_t cap = c->shadow_cap;
                    sa.shadow.s0 = c->shadow_mem; sa.shadow.s1 = c->shadow_mem + cap;
                    sa.shadow.s2 = c->shadow_mem + 2 * cap; sa.shadow.s3 = c->shadow_mem + 3 * cap;
                    sa.shadow_count = c->ctrl + 128 + b;
                    if (count) hipLaunchKernelGGL((k_shade<true, true>), dim3(shade_grid), dim3(256), 0, c->stream, sa);
                    else hipLaunchKernelGGL((k_shade<false, true>), dim3(shade_grid), dim3(256), 0, c->stream, sa);
                } else {
                    if (count) hipLaunchKernelGGL((k_shade<true, false>), dim3(shade_grid), dim3(256), 0, c->stream, sa);
                    else hipLaunchKernelGGL((k_shade<false, false>), dim3(shade_grid), dim3(256), 0, c->stream, sa);
                }
            }
            if (P.enable_sun) {
                // trace_shadow for the shadow rays of this bounce: candidate masks, the any-hit walk
                // (the hit records of this bounce have been consumed by the shade stage), the paths' ends
                StageTimer t(c, timing, RTHIP_STAGE_SHADOW);
                ExtArgs xa = ea;
                xa.in = sa.shadow;
                xa.count_in = c->ctrl + 128 + b;
                xa.cursor = c->ctrl + 256 + b;
                xa.tap = 0u;
                if (cull) {
                    CullArgs ca{};
                    ca.sc = c->sc;
                    ca.in = sa.shadow;
                    ca.masks = reinterpret_cast<uint4*>(c->hits);
                    ca.count_in = xa.count_in;
                    hipLaunchKernelGGL(k_cull, dim3(shade_grid), dim3(256), 0, c->stream, ca);
                }
                if (ext_mode == EXT_LIST) {
                    CullListArgs la{static_cast<const float4*>(c->d_ptree), sa.shadow, c->cand_lists, xa.count_in};
                    cull_list_launch(c->list_len, shade_grid, c->stream, la);
                }
                shadow_extend_launch(deep, ext_mode, ext_grid, c->stream, xa);
                ShadowArgs wa{};
                wa.in = sa.shadow;
                wa.hits = c->hits;
                wa.count = c->ctrl + 128 + b;
                wa.partial = c->partial;
                wa.tap_seed = tap_group ? c->tap_seed : nullptr;
                wa.n_pixels = (uint32_t)c->n_pixels;
                wa.counters = count ? c->d_counters : nullptr;
                hipLaunchKernelGGL(k_shadow_resolve, dim3(shade_grid), dim3(256), 0, c->stream, wa);
            }
        }
        {
            StageTimer t(c, timing, RTHIP_STAGE_ACCUMULATE);
            const uint32_t acc_grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * 8, (c->n_pixels + 255) / 256);
            hipLaunchKernelGGL(k_accumulate, dim3(acc_grid), dim3(256), 0, c->stream, c->accum, (const float4*)c->partial,
                               (uint32_t)c->n_pixels, nf, overwrite ? 1u : 0u);
        }
        c->reject_hist = 0; // main.rs:403
        if (tap_group) c->taps_valid = true;
    }
    HIP_TRY(c, hipGetLastError());
    return RTHIP_OK;
}

int rthip_frames_in_flight(rthip_ctx* c, uint32_t* n_frames) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!n_frames) return fail(c, RTHIP_ERR_INVALID, "frames_in_flight: null output");
    if (!c->have_uniforms) return fail(c, RTHIP_ERR_STATE, "frames_in_flight: set_frame_uniforms first (the resolution sizes the buffers)");
    DeviceGuard g(c->device);
    *n_frames = frames_in_flight_limit(c);
    return RTHIP_OK;
}

int rthip_candidate_walk(rthip_ctx* c, uint32_t* kind) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!kind) return fail(c, RTHIP_ERR_INVALID, "candidate_walk: null output");
    if (!c->have_scene) return fail(c, RTHIP_ERR_STATE, "candidate_walk: no scene uploaded");
    *kind = c->sc.n_chunks <= 1u ? RTHIP_WALK_MASK : (c->tree ? RTHIP_WALK_PYRAMID : (c->lists ? RTHIP_WALK_LIST : RTHIP_WALK_SCAN));
    return RTHIP_OK;
}

int rthip_reserve(rthip_ctx* c, uint32_t n_frames) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->have_uniforms) return fail(c, RTHIP_ERR_STATE, "reserve: set_frame_uniforms first (the resolution sizes the buffers)");
    if (n_frames == 0) return fail(c, RTHIP_ERR_INVALID, "reserve: n_frames = 0");
    DeviceGuard g(c->device);
    return ensure_frame_buffers(c, std::min<uint32_t>(n_frames, frames_in_flight_limit(c)));
}

int rthip_trace_rays(rthip_ctx* c, const float* rays6, uint32_t* seeds, uint64_t n, const rthip_render_params* params,
                     float* t, int32_t* tri_idx, int32_t* prim_idx, uint8_t* overflow) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->have_scene) return fail(c, RTHIP_ERR_STATE, "trace_rays: no scene uploaded");
    if (n == 0) return RTHIP_OK;
    if (!rays6) return fail(c, RTHIP_ERR_INVALID, "trace_rays: null rays");
    if (n >= 0x7FFFFFFFull) return fail(c, RTHIP_ERR_INVALID, "trace_rays: more than 2^31-1 rays");
    rthip_render_params P;
    if (params) P = *params; else rthip_default_params(&P);
    if (P.stack_depth == 0 || P.stack_depth > RTHIP_MAX_STACK_DEPTH) return fail(c, RTHIP_ERR_INVALID, "trace_rays: stack_depth %u outside 1..%u", P.stack_depth, RTHIP_MAX_STACK_DEPTH);
    DeviceGuard g(c->device);

    // the rays as the first two planes of a path queue
    std::vector<float> planes((size_t)n * 8);
    for (uint64_t i = 0; i < n; i++) {
        float* s0 = &planes[(size_t)i * 4];
        float* s1 = &planes[(size_t)(n + i) * 4];
        const float* r = rays6 + i * 6;
        s0[0] = r[0]; s0[1] = r[1]; s0[2] = r[2]; s0[3] = r[3];
        s1[0] = r[4]; s1[1] = r[5];
        const uint32_t seed = seeds ? seeds[i] : 0u, slot = 0u;
        std::memcpy(&s1[2], &seed, 4);
        std::memcpy(&s1[3], &slot, 4);
    }
    // staging grows to the largest query seen and is kept (no allocation per call)
    hipError_t e = hipSuccess;
    if (c->tr_cap < n) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        free_dev_t(c->tr_q); free_dev_t(c->tr_hits);
        c->tr_cap = 0;
        e = hipMalloc((void**)&c->tr_q, sizeof(float4) * 2 * n);
        if (e == hipSuccess) e = hipMalloc((void**)&c->tr_hits, sizeof(float4) * n);
        if (e != hipSuccess) { free_dev_t(c->tr_q); free_dev_t(c->tr_hits); return fail(c, RTHIP_ERR_HIP, "trace_rays: %s", hipGetErrorString(e)); }
        c->tr_cap = n;
    }
    float4* const d_q = c->tr_q;
    float4* const d_hits = c->tr_hits;
    uint32_t* const d_ctl = c->ctrl + 192; // [0] = ray count, [1] = cursor (words no render uses)
    const uint32_t ctl[2] = {(uint32_t)n, 0u};
    if (e == hipSuccess) e = hipMemcpyAsync(d_q, planes.data(), sizeof(float4) * 2 * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ctl, ctl, sizeof ctl, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return fail(c, RTHIP_ERR_HIP, "trace_rays: %s", hipGetErrorString(e));

    const bool count = (P.flags & RTHIP_FLAG_COUNTERS) != 0;
    const bool timing = (P.flags & RTHIP_FLAG_TIMING) != 0;
    const bool multi = c->sc.n_chunks > 1u;
    const ExtMode ext_mode = !multi ? EXT_SINGLE : (c->tree ? EXT_TREE : (c->lists ? EXT_LIST : EXT_SCAN));
    const bool deep = P.stack_depth > rt_lds_levels(multi, c->tree);
    const bool cull = !multi && c->sc.n_prims > RT_CULL_MIN_PRIMS;
    const NanWalk& nanw = nan_walk(c, std::min<uint32_t>(P.stack_depth, RT_MAX_STACK));
    const int blocks = extend_occupancy(count, deep, ext_mode);
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * blocks, (n + 255) / 256);
    if (deep) {
        const uint64_t words = (uint64_t)(RT_MAX_STACK - rt_lds_levels(multi, c->tree)) * grid * 256;
        if (c->spill_words < words) {
            (void)hipStreamSynchronize(c->stream);
            free_dev_t(c->spill);
            c->spill_words = 0;
            e = hipMalloc((void**)&c->spill, sizeof(uint32_t) * words);
            if (e != hipSuccess) return fail(c, RTHIP_ERR_HIP, "trace_rays: %s", hipGetErrorString(e));
            c->spill_words = words;
        }
    }
    if (ext_mode == EXT_LIST && c->tr_lists_cap < n * c->list_len) {
        (void)hipStreamSynchronize(c->stream);
        free_dev_t(c->tr_lists);
        c->tr_lists_cap = 0;
        e = hipMalloc((void**)&c->tr_lists, sizeof(uint32_t) * n * c->list_len);
        if (e != hipSuccess) return fail(c, RTHIP_ERR_HIP, "trace_rays: %s", hipGetErrorString(e));
        c->tr_lists_cap = n * c->list_len;
    }
    ExtArgs ea{};
    ea.sc = c->sc;
    ea.tlas = static_cast<const float4*>(c->d_tlas);
    ea.tlas_dir = static_cast<const uint32_t*>(c->d_tlas_dir);
    ea.lists = c->tr_lists;
    ea.list_len = c->list_len;
    ea.in.s0 = d_q; ea.in.s1 = d_q + n; ea.in.s2 = d_q; ea.in.s3 = d_q;
    ea.hits = d_hits;
    ea.spill = c->spill;
    ea.count_in = d_ctl;
    ea.cursor = d_ctl + 1;
    ea.stack_depth = P.stack_depth;
    ea.counters = c->d_counters;
    ea.nan_inner = nanw.inner; ea.nan_leaf = nanw.leaf; ea.nan_tri = nanw.tri; ea.nan_dropped = nanw.dropped;
    ea.nan_overflow = nanw.dropped ? 1u : 0u;
    (void)hipGetLastError(); // not this call's
    ea.no_cull = cull ? 0u : 1u;
    if (cull) {
        CullArgs ca{};
        ca.sc = c->sc;
        ca.in = ea.in;
        ca.masks = reinterpret_cast<uint4*>(d_hits);
        ca.count_in = d_ctl;
        StageTimer tm(c, timing, RTHIP_STAGE_CULL);
        hipLaunchKernelGGL(k_cull, dim3((uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * 8, (n + 255) / 256)), dim3(256), 0, c->stream, ca);
    }
    if (ext_mode == EXT_LIST) {
        CullListArgs la{static_cast<const float4*>(c->d_ptree), ea.in, c->tr_lists, d_ctl};
        StageTimer tm(c, timing, RTHIP_STAGE_CULL);
        cull_list_launch(c->list_len, (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * 8, (n + 255) / 256), c->stream, la);
    }
    {
        StageTimer tm(c, timing, RTHIP_STAGE_EXTEND);
        extend_launch(count, deep, ext_mode, grid, c->stream, ea);
    }
    std::vector<float> hits((size_t)n * 4);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(hits.data(), d_hits, sizeof(float4) * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, RTHIP_ERR_HIP, "trace_rays: %s", hipGetErrorString(e));
    for (uint64_t i = 0; i < n; i++) {
        const float* h = &hits[(size_t)i * 4];
        int32_t tri;
        uint32_t primw, seed;
        std::memcpy(&tri, &h[1], 4);
        std::memcpy(&primw, &h[2], 4);
        std::memcpy(&seed, &h[3], 4);
        if (t) t[i] = h[0];
        if (tri_idx) tri_idx[i] = tri;
        if (prim_idx) prim_idx[i] = tri >= 0 ? (int32_t)(primw & 0x7FFFFFFFu) : -1;
        if (overflow) overflow[i] = (uint8_t)(primw >> 31);
        if (seeds) seeds[i] = seed;
    }
    return RTHIP_OK;
}

int rthip_trace_rays_noalloc(rthip_ctx* c, const float* d_rays6, uint32_t* d_seeds, uint64_t n, const rthip_render_params* params,
                             float* d_t, int32_t* d_tri_idx, int32_t* d_prim_idx, uint8_t* d_overflow) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->have_scene) return fail(c, RTHIP_ERR_STATE, "trace_rays_noalloc: no scene uploaded");
    if (n == 0) return RTHIP_OK;
    if (!d_rays6) return fail(c, RTHIP_ERR_INVALID, "trace_rays_noalloc: null rays");
    if (n > RT_NOALLOC_MAX_RAYS)
        return fail(c, RTHIP_ERR_INVALID, "trace_rays_noalloc: %llu rays exceed the fixed work memory of %u rays per call (split the query, or use rthip_trace_rays)",
                    (unsigned long long)n, RT_NOALLOC_MAX_RAYS);
    rthip_render_params P;
    if (params) P = *params; else rthip_default_params(&P);
    if (P.stack_depth == 0 || P.stack_depth > RTHIP_MAX_STACK_DEPTH) return fail(c, RTHIP_ERR_INVALID, "trace_rays_noalloc: stack_depth %u outside 1..%u", P.stack_depth, RTHIP_MAX_STACK_DEPTH);
    DeviceGuard g(c->device);

    // the pieces of this device's static work array (addresses from hipGetSymbolAddress)
    unsigned char* const w = c->noalloc_scratch;
    uint32_t* const ctl = reinterpret_cast<uint32_t*>(w);
    float4* const planes = reinterpret_cast<float4*>(w + RT_NOALLOC_PLANES_OFF);
    float4* const hits = reinterpret_cast<float4*>(w + RT_NOALLOC_HITS_OFF);
    uint32_t* const spill = reinterpret_cast<uint32_t*>(w + RT_NOALLOC_SPILL_OFF);

    const uint32_t nn = (uint32_t)n;
    const bool count = (P.flags & RTHIP_FLAG_COUNTERS) != 0;
    const bool multi = c->sc.n_chunks > 1u;
    const ExtMode ext_mode = !multi ? EXT_SINGLE : (c->tree ? EXT_TREE : EXT_SCAN);
    const bool deep = P.stack_depth > rt_lds_levels(multi, c->tree);
    const bool cull = !multi && c->sc.n_prims > RT_CULL_MIN_PRIMS;
    const NanWalk& nanw = nan_walk(c, std::min<uint32_t>(P.stack_depth, RT_MAX_STACK));
    const int blocks = extend_occupancy(count, deep, ext_mode);
    uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * blocks, (n + 255) / 256);
    if (deep) grid = std::min<uint32_t>(grid, RT_NOALLOC_DEEP_GRID); // the deep stack columns are a fixed part of the work array
    (void)hipGetLastError(); // not this call's

    // Order this call behind the last one that used the GPU's work array (another context, another
    // stream).  Not while the stream is being captured into a graph: an event recorded outside the
    // capture cannot be waited on inside it, and the order of a graph's nodes is the caller's.
    hipEvent_t order = c->device < RT_MAX_DEVICES ? g_noalloc_done[c->device] : nullptr;
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &capture) != hipSuccess) { capture = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
    if (capture != hipStreamCaptureStatusNone) order = nullptr;
    std::unique_lock<std::mutex> lock;
    if (order) {
        lock = std::unique_lock<std::mutex>(g_noalloc_mutex[c->device]);
        HIP_TRY(c, hipStreamWaitEvent(c->stream, order, 0));
    }
    const uint32_t flat_grid = (nn + 255u) / 256u;
    hipLaunchKernelGGL(k_pack_rays, dim3(flat_grid), dim3(256), 0, c->stream, d_rays6, (const uint32_t*)d_seeds, nn, planes, planes + RT_NOALLOC_MAX_RAYS, ctl);
    ExtArgs ea{};
    ea.sc = c->sc;
    ea.tlas = static_cast<const float4*>(c->d_tlas);
    ea.tlas_dir = static_cast<const uint32_t*>(c->d_tlas_dir);
    ea.in.s0 = planes; ea.in.s1 = planes + RT_NOALLOC_MAX_RAYS; ea.in.s2 = planes; ea.in.s3 = planes;
    ea.hits = hits;
    ea.spill = spill;
    ea.count_in = ctl;
    ea.cursor = ctl + 1;
    ea.stack_depth = P.stack_depth;
    ea.counters = c->d_counters;
    ea.nan_inner = nanw.inner; ea.nan_leaf = nanw.leaf; ea.nan_tri = nanw.tri; ea.nan_dropped = nanw.dropped;
    ea.nan_overflow = nanw.dropped ? 1u : 0u;
    ea.no_cull = cull ? 0u : 1u;
    if (cull) {
        CullArgs ca{};
        ca.sc = c->sc;
        ca.in = ea.in;
        ca.masks = reinterpret_cast<uint4*>(hits);
        ca.count_in = ctl;
        hipLaunchKernelGGL(k_cull, dim3(std::min<uint32_t>((uint32_t)c->n_cu * 8u, flat_grid)), dim3(256), 0, c->stream, ca);
    }
    extend_launch(count, deep, ext_mode, grid, c->stream, ea);
    hipLaunchKernelGGL(k_unpack_hits, dim3(flat_grid), dim3(256), 0, c->stream, (const float4*)hits, nn, d_t, d_tri_idx, d_prim_idx, d_overflow, d_seeds);
    if (order) HIP_TRY(c, hipEventRecord(order, c->stream));
    HIP_TRY(c, hipGetLastError());
    return RTHIP_OK;
}

int rthip_bind_accum(rthip_ctx* c, void* device_ptr, uint64_t n_pixels) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!device_ptr) {
        c->accum_is_bound = false;
        c->accum_bound_pixels = 0;
        c->accum = c->accum_own;
        c->reject_hist = 1;
        return RTHIP_OK;
    }
    if ((reinterpret_cast<uintptr_t>(device_ptr) & 15u) != 0) return fail(c, RTHIP_ERR_INVALID, "bind_accum: pointer is not 16-byte aligned");
    if (c->have_uniforms && n_pixels != c->n_pixels) return fail(c, RTHIP_ERR_INVALID, "bind_accum: %llu pixels, current resolution has %llu", (unsigned long long)n_pixels, (unsigned long long)c->n_pixels);
    c->accum_is_bound = true;
    c->accum_bound_pixels = n_pixels;
    c->accum = static_cast<float4*>(device_ptr);
    return RTHIP_OK;
}

int rthip_accum_device_ptr(rthip_ctx* c, void** device_ptr, uint64_t* n_pixels) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->have_uniforms) return fail(c, RTHIP_ERR_STATE, "accum_device_ptr: no frame uniforms set");
    DeviceGuard g(c->device);
    int rc = ensure_frame_buffers(c, std::max(c->frames_cap, 1u));
    if (rc != RTHIP_OK) return rc;
    if (device_ptr) *device_ptr = c->accum;
    if (n_pixels) *n_pixels = c->n_pixels;
    return RTHIP_OK;
}

int rthip_read_accum(rthip_ctx* c, float* rgba, uint64_t n_pixels) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!rgba) return fail(c, RTHIP_ERR_INVALID, "read_accum: null output");
    if (!c->have_uniforms || n_pixels != c->n_pixels) return fail(c, RTHIP_ERR_INVALID, "read_accum: %llu pixels requested, accumulator has %llu", (unsigned long long)n_pixels, (unsigned long long)(c->have_uniforms ? c->n_pixels : 0));
    DeviceGuard g(c->device);
    // a change of resolution that no render has followed yet: the accumulator of the new size (zeroed)
    int rc = ensure_accum(c);
    if (rc != RTHIP_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(rgba, c->accum, sizeof(float4) * n_pixels, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return check_watchdog(c);
}

int rthip_write_accum(rthip_ctx* c, const float* rgba, uint64_t n_pixels) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!rgba) return fail(c, RTHIP_ERR_INVALID, "write_accum: null input");
    if (!c->have_uniforms || n_pixels != c->n_pixels) return fail(c, RTHIP_ERR_INVALID, "write_accum: size mismatch");
    DeviceGuard g(c->device);
    int rc = ensure_frame_buffers(c, std::max(c->frames_cap, 1u));
    if (rc != RTHIP_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->accum, rgba, sizeof(float4) * n_pixels, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->reject_hist = 0;
    return RTHIP_OK;
}

int rthip_read_taps(rthip_ctx* c, int32_t* tri_idx, int32_t* prim_idx, float* t, uint32_t* final_seed, uint64_t n_pixels) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->taps_valid) return fail(c, RTHIP_ERR_STATE, "read_taps: no render with RTHIP_FLAG_TAPS has been issued");
    if (n_pixels != c->n_pixels) return fail(c, RTHIP_ERR_INVALID, "read_taps: size mismatch");
    DeviceGuard g(c->device);
    if (tri_idx) HIP_TRY(c, hipMemcpyAsync(tri_idx, c->tap_tri, 4 * n_pixels, hipMemcpyDeviceToHost, c->stream));
    if (prim_idx) HIP_TRY(c, hipMemcpyAsync(prim_idx, c->tap_prim, 4 * n_pixels, hipMemcpyDeviceToHost, c->stream));
    if (t) HIP_TRY(c, hipMemcpyAsync(t, c->tap_t, 4 * n_pixels, hipMemcpyDeviceToHost, c->stream));
    if (final_seed) HIP_TRY(c, hipMemcpyAsync(final_seed, c->tap_seed, 4 * n_pixels, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTHIP_OK;
}

int rthip_read_nodes_device_layout(rthip_ctx* c, void* nodes_out, uint64_t capacity, uint64_t* n_records) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->have_scene) return fail(c, RTHIP_ERR_STATE, "no scene uploaded");
    if (n_records) *n_records = c->n_dev_nodes;
    if (!nodes_out) return RTHIP_OK;
    if (capacity < c->n_dev_nodes) return fail(c, RTHIP_ERR_INVALID, "read_nodes_device_layout: capacity %llu < %llu records", (unsigned long long)capacity, (unsigned long long)c->n_dev_nodes);
    DeviceGuard g(c->device);
    HIP_TRY(c, hipMemcpy(nodes_out, c->d_nodes, 32 * c->n_dev_nodes, hipMemcpyDeviceToHost));
    return RTHIP_OK;
}

int rthip_read_node_index_map(rthip_ctx* c, uint32_t* out, uint64_t n_nodes) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->have_scene) return fail(c, RTHIP_ERR_STATE, "no scene uploaded");
    if (!out || n_nodes != c->n_nodes) return fail(c, RTHIP_ERR_INVALID, "read_node_index_map: size mismatch");
    std::memcpy(out, c->node_index_map.data(), sizeof(uint32_t) * n_nodes);
    return RTHIP_OK;
}

int rthip_read_tri_order(rthip_ctx* c, uint32_t* out, uint64_t n_tris) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->have_scene) return fail(c, RTHIP_ERR_STATE, "no scene uploaded");
    if (!out || n_tris != c->n_tris) return fail(c, RTHIP_ERR_INVALID, "read_tri_order: size mismatch");
    for (uint64_t i = 0; i < n_tris; i++) out[i] = (uint32_t)i;
    return RTHIP_OK;
}

int rthip_read_tris_device_layout(rthip_ctx* c, void* out, uint64_t n_tris) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!c->have_scene) return fail(c, RTHIP_ERR_STATE, "no scene uploaded");
    if (!out || n_tris != c->n_tris) return fail(c, RTHIP_ERR_INVALID, "read_tris_device_layout: size mismatch");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipMemcpy(out, c->d_tris, 48 * n_tris, hipMemcpyDeviceToHost));
    return RTHIP_OK;
}

int rthip_resolve(rthip_ctx* c, uint8_t* rgba8, uint64_t n_pixels, int apply_srgb_oetf) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!rgba8) return fail(c, RTHIP_ERR_INVALID, "resolve: null output");
    if (!c->have_uniforms || n_pixels != c->n_pixels) return fail(c, RTHIP_ERR_INVALID, "resolve: size mismatch");
    DeviceGuard g(c->device);
    {
        int rc = ensure_accum(c);
        if (rc != RTHIP_OK) return rc;
    }
    // The output may be memory of this GPU (a presentation buffer: what the reference's render pass
    // writes, main.rs:395-401): the image is then written there by the kernel, nothing is copied
    // and nothing allocated.  Anything else (pageable or pinned host memory) gets a copy.
    bool on_device = false;
    {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, rgba8) == hipSuccess) on_device = at.type == hipMemoryTypeDevice && at.device == c->device && (reinterpret_cast<uintptr_t>(rgba8) & 3u) == 0;
        else (void)hipGetLastError(); // an ordinary host pointer the runtime has never seen
    }
    if (!on_device && c->resolve_cap < n_pixels) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        free_dev_t(c->resolve_out);
        c->resolve_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->resolve_out, 4 * n_pixels));
        c->resolve_cap = n_pixels;
    }
    uint32_t* const d_out = on_device ? reinterpret_cast<uint32_t*>(rgba8) : c->resolve_out;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * 8, (n_pixels + 255) / 256);
    hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(256), 0, c->stream, (const float4*)c->accum, d_out, n_pixels, apply_srgb_oetf);
    hipError_t e = hipSuccess;
    if (!on_device) e = hipMemcpyAsync(rgba8, d_out, 4 * n_pixels, hipMemcpyDefault, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, RTHIP_ERR_HIP, "resolve: %s", hipGetErrorString(e));
    return check_watchdog(c);
}

int rthip_get_counters(rthip_ctx* c, rthip_counters* out) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!out) return fail(c, RTHIP_ERR_INVALID, "get_counters: null output");
    DeviceGuard g(c->device);
    Counters h;
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    out->rays = h.v[C_RAYS]; out->prims_visited = h.v[C_PRIMS]; out->inner_steps = h.v[C_INNER];
    out->leaf_steps = h.v[C_LEAF]; out->tri_tests = h.v[C_TRI]; out->alpha_tests = h.v[C_ALPHA];
    out->texel_fetches = h.v[C_TEXEL]; out->shade_calls = h.v[C_SHADE]; out->misses = h.v[C_MISS];
    out->kills = h.v[C_KILL]; out->stack_overflows = h.v[C_OVERFLOW]; out->paths = h.v[C_PATHS];
    out->rounds_prim = h.v[C_ROUNDS_PRIM]; out->rounds_inner = h.v[C_ROUNDS_INNER]; out->rounds_leaf = h.v[C_ROUNDS_LEAF];
    out->batches = h.v[C_BATCHES];
    out->shadow_rays = h.v[C_SHADOW];
    return RTHIP_OK;
}

int rthip_reset_counters(rthip_ctx* c) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, sizeof(Counters), c->stream));
    return RTHIP_OK;
}

int rthip_get_timing(rthip_ctx* c, rthip_timing* out) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!out) return fail(c, RTHIP_ERR_INVALID, "get_timing: null output");
    DeviceGuard g(c->device);
    int rc = drain_events(c);
    if (rc != RTHIP_OK) return rc;
    *out = c->timing;
    return RTHIP_OK;
}

int rthip_reset_timing(rthip_ctx* c) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    DeviceGuard g(c->device);
    int rc = drain_events(c);
    if (rc != RTHIP_OK) return rc;
    c->timing = rthip_timing{};
    return RTHIP_OK;
}

int rthip_copy_bandwidth(rthip_ctx* c, uint64_t bytes, uint32_t repeats, double* gb_per_s) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!gb_per_s || bytes == 0 || repeats == 0) return fail(c, RTHIP_ERR_INVALID, "copy_bandwidth: null result, zero bytes or zero repeats");
    DeviceGuard g(c->device);
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&a, bytes);
    if (e == hipSuccess) e = hipMalloc(&b, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(a, 1, bytes, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream); // untimed first touch
    if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
    for (uint32_t r = 0; r < repeats && e == hipSuccess; r++) e = hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (e != hipSuccess) return fail(c, RTHIP_ERR_HIP, "copy_bandwidth: %s", hipGetErrorString(e));
    *gb_per_s = ms > 0.0f ? 2.0 * (double)bytes * repeats / ((double)ms * 1e-3) / 1e9 : 0.0;
    return RTHIP_OK;
}

int rthip_gather_ceiling(rthip_ctx* c, uint32_t active_lanes, uint32_t record_bytes, uint32_t records_in_flight, double* lane_loads_per_s) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    if (!lane_loads_per_s || active_lanes == 0 || active_lanes > 64) return fail(c, RTHIP_ERR_INVALID, "gather_ceiling: null result, or active_lanes outside 1..64");
    if (record_bytes != 64 && record_bytes != 128) return fail(c, RTHIP_ERR_INVALID, "gather_ceiling: record_bytes %u is neither 64 nor 128", record_bytes);
    if (records_in_flight != 1 && records_in_flight != 2) return fail(c, RTHIP_ERR_INVALID, "gather_ceiling: records_in_flight %u is neither 1 nor 2", records_in_flight);
    const bool wide = record_bytes == 128, two = records_in_flight == 2;
    DeviceGuard g(c->device);
    // `active_lanes` lanes spread over the wave by a fixed permutation (scattered, as the lanes that
    // share a step kind are)
    unsigned long long mask = 0ull;
    for (uint32_t k = 0, lane = 5u; k < active_lanes; k++, lane = (lane + 37u) & 63u) mask |= 1ull << lane; // 37 is odd: a permutation of 0..63
    const uint32_t iters = 2000u, blocks = (uint32_t)c->n_cu * 8u;
    void* table = nullptr;
    float* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&table, 256 * 64);
    if (e == hipSuccess) e = hipMalloc((void**)&sink, sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(table, 0, 256 * 64, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    auto launch = [&](uint32_t n) {
        const float4* t = (const float4*)table;
        if (wide && two) hipLaunchKernelGGL((k_gather_ceiling<8u, 2u>), dim3(blocks), dim3(256), 0, c->stream, t, mask, n, sink);
        else if (wide) hipLaunchKernelGGL((k_gather_ceiling<8u, 1u>), dim3(blocks), dim3(256), 0, c->stream, t, mask, n, sink);
        else if (two) hipLaunchKernelGGL((k_gather_ceiling<4u, 2u>), dim3(blocks), dim3(256), 0, c->stream, t, mask, n, sink);
        else hipLaunchKernelGGL((k_gather_ceiling<4u, 1u>), dim3(blocks), dim3(256), 0, c->stream, t, mask, n, sink);
    };
    if (e == hipSuccess) {
        launch(50u); // untimed first: code and table in the caches
        e = hipEventRecord(e0, c->stream);
    }
    if (e == hipSuccess) {
        launch(iters);
        e = hipEventRecord(e1, c->stream);
    }
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (table) (void)hipFree(table);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return fail(c, RTHIP_ERR_HIP, "gather_ceiling: %s", hipGetErrorString(e));
    // blocks x 4 waves x iters iterations x (4 or 8 loads) x (1 or 2 records), each by active_lanes lanes
    *lane_loads_per_s = ms > 0.0f ? (double)blocks * 4.0 * iters * (wide ? 8.0 : 4.0) * (two ? 2.0 : 1.0) * active_lanes / ((double)ms * 1e-3) : 0.0;
    return RTHIP_OK;
}

int rthip_synchronize(rthip_ctx* c) {
    if (!c) return fail(nullptr, RTHIP_ERR_INVALID, "null context");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return check_watchdog(c);
}

} // extern "C"
