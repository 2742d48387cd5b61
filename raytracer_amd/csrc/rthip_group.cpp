// Several GPUs of one node from one host process: one rthip context, HIP stream and host
// thread per GPU, frames striped over the GPUs, one RCCL reduce of the accumulators.
// See include/rthip_group.h for the contract and the reference lines it splits.
#include "../../include/rthip_group.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct rthip_group {
    std::vector<int>          devices;
    std::vector<rthip_ctx*>   ctx;
    std::vector<hipStream_t>  stream;
    std::vector<float*>       accum;     // group-owned accumulators, bound to the contexts
    std::vector<ncclComm_t>   comm;      // empty until the first reduce
    uint64_t                  n_pixels = 0;
    uint32_t                  res_x = 0, res_y = 0;
    std::string               error;
};

namespace {

thread_local std::string g_create_error;

int fail(rthip_group* g, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->error = buf;
    else g_create_error = buf;
    return code;
}

// Runs fn(rank) for every rank on its own host thread (the per-GPU calls copy and repack on
// the host, or enqueue many launches); returns the first failing rank's status and message.
template <class F>
int for_each_rank(rthip_group* g, const char* what, F fn) {
    const int n = (int)g->ctx.size();
    std::vector<int> rc(n, RTHIP_OK);
    if (n == 1) {
        rc[0] = fn(0);
    } else {
        std::vector<std::thread> th;
        th.reserve(n);
        for (int r = 0; r < n; r++) th.emplace_back([&, r] { rc[r] = fn(r); });
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < n; r++)
        if (rc[r] != RTHIP_OK)
            return fail(g, rc[r], "%s: GPU %d (rank %d): %s", what, g->devices[r], r, rthip_last_error(g->ctx[r]));
    return RTHIP_OK;
}

void release(rthip_group* g) {
    for (size_t r = 0; r < g->comm.size(); r++) ncclCommDestroy(g->comm[r]);
    g->comm.clear();
    for (size_t r = 0; r < g->ctx.size(); r++) {
        if (g->ctx[r]) rthip_destroy(g->ctx[r]);
        const bool holds = (r < g->accum.size() && g->accum[r]) || (r < g->stream.size() && g->stream[r]);
        if (holds && hipSetDevice(g->devices[r]) == hipSuccess) {
            if (r < g->accum.size() && g->accum[r]) (void)hipFree(g->accum[r]);
            if (r < g->stream.size() && g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
        }
    }
    g->ctx.clear();
}

} // namespace

extern "C" {

int rthip_group_create(const int* device_ordinals, int n_devices, rthip_group** out) {
    if (!out) return fail(nullptr, RTHIP_ERR_INVALID, "group_create: null output");
    *out = nullptr;
    if (!device_ordinals || n_devices <= 0) return fail(nullptr, RTHIP_ERR_INVALID, "group_create: no devices");
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < i; j++)
            if (device_ordinals[i] == device_ordinals[j])
                return fail(nullptr, RTHIP_ERR_INVALID, "group_create: device %d listed twice", device_ordinals[i]);
    rthip_group* g = new rthip_group;
    g->devices.assign(device_ordinals, device_ordinals + n_devices);
    g->ctx.assign(n_devices, nullptr);
    g->stream.assign(n_devices, nullptr);
    g->accum.assign(n_devices, nullptr);
    for (int r = 0; r < n_devices; r++) {
        int rc = rthip_create(g->devices[r], &g->ctx[r]);
        if (rc != RTHIP_OK) {
            fail(nullptr, rc, "group_create: device %d: %s", g->devices[r], rthip_last_error(nullptr));
            release(g);
            delete g;
            return rc;
        }
        hipError_t e = hipSetDevice(g->devices[r]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream[r], hipStreamNonBlocking);
        if (e != hipSuccess) {
            fail(nullptr, RTHIP_ERR_HIP, "group_create: device %d: %s", g->devices[r], hipGetErrorString(e));
            release(g);
            delete g;
            return RTHIP_ERR_HIP;
        }
        rc = rthip_set_stream(g->ctx[r], g->stream[r]);
        if (rc != RTHIP_OK) {
            fail(nullptr, rc, "group_create: device %d: %s", g->devices[r], rthip_last_error(g->ctx[r]));
            release(g);
            delete g;
            return rc;
        }
    }
    *out = g;
    return RTHIP_OK;
}

void rthip_group_destroy(rthip_group* g) {
    if (!g) return;
    release(g);
    delete g;
}

const char* rthip_group_last_error(const rthip_group* g) { return g ? g->error.c_str() : g_create_error.c_str(); }

int rthip_group_size(const rthip_group* g) { return g ? (int)g->ctx.size() : 0; }

rthip_ctx* rthip_group_ctx(rthip_group* g, int rank) {
    if (!g || rank < 0 || rank >= (int)g->ctx.size()) return nullptr;
    return g->ctx[rank];
}

int rthip_group_upload_scene(rthip_group* g, const rthip_scene_desc* scene) {
    if (!g) return fail(nullptr, RTHIP_ERR_INVALID, "null group");
    return for_each_rank(g, "upload_scene", [&](int r) { return rthip_upload_scene(g->ctx[r], scene); });
}

int rthip_group_set_frame_uniforms(rthip_group* g, const void* blob608) {
    if (!g) return fail(nullptr, RTHIP_ERR_INVALID, "null group");
    if (!blob608) return fail(g, RTHIP_ERR_INVALID, "set_frame_uniforms: null blob");
    uint32_t res[2], reject;
    std::memcpy(res, static_cast<const uint8_t*>(blob608) + 576, 8);   // FrameUniforms.res, main.rs:31-43
    std::memcpy(&reject, static_cast<const uint8_t*>(blob608) + 592, 4); // FrameUniforms.reject_hist
    const uint64_t np = (uint64_t)res[0] * res[1];
    if (np == 0 || np >= 0x7FFFFFFFull) return fail(g, RTHIP_ERR_INVALID, "set_frame_uniforms: resolution %ux%u", res[0], res[1]);
    // a change of shape at equal pixel count (WxH -> HxW) is a new image too
    const bool resized = res[0] != g->res_x || res[1] != g->res_y;
    const bool realloc = np != g->n_pixels;
    const int n = (int)g->ctx.size();
    if (realloc) {
        // update_resolution (main.rs:79-94): a new accumulator, owned here so that it exists
        // (zeroed) on a GPU that gets no frame before a reduce.  A context accepts a new
        // resolution only while no accumulator of another size is bound to it.
        g->n_pixels = 0;
        for (int r = 0; r < n; r++) {
            const int rc = rthip_bind_accum(g->ctx[r], nullptr, 0); // synchronises the stream
            if (rc != RTHIP_OK) return fail(g, rc, "set_frame_uniforms: rank %d: %s", r, rthip_last_error(g->ctx[r]));
            hipError_t e = hipSetDevice(g->devices[r]);
            if (e == hipSuccess && g->accum[r]) e = hipFree(g->accum[r]);
            g->accum[r] = nullptr;
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->accum[r]), np * 16);
            if (e != hipSuccess) return fail(g, RTHIP_ERR_HIP, "set_frame_uniforms: rank %d: %s", r, hipGetErrorString(e));
        }
    }
    int rc = for_each_rank(g, "set_frame_uniforms", [&](int r) { return rthip_set_frame_uniforms(g->ctx[r], blob608); });
    if (rc != RTHIP_OK) return rc;
    for (int r = 0; r < n; r++) {
        if (realloc) {
            rc = rthip_bind_accum(g->ctx[r], g->accum[r], np);
            if (rc != RTHIP_OK) return fail(g, rc, "set_frame_uniforms: rank %d: %s", r, rthip_last_error(g->ctx[r]));
        }
        if (resized || reject == 1) {
            hipError_t e = hipSetDevice(g->devices[r]);
            if (e == hipSuccess) e = hipMemsetAsync(g->accum[r], 0, np * 16, g->stream[r]);
            if (e != hipSuccess) return fail(g, RTHIP_ERR_HIP, "set_frame_uniforms: rank %d: %s", r, hipGetErrorString(e));
        }
    }
    g->n_pixels = np;
    g->res_x = res[0];
    g->res_y = res[1];
    return RTHIP_OK;
}

void rthip_group_plan(uint32_t first_frame, uint32_t n_frames, int n_ranks, int rank, uint32_t* first, uint32_t* count) {
    uint32_t f = first_frame, c = 0;
    if (n_ranks > 0 && rank >= 0 && rank < n_ranks) {
        f = first_frame + (uint32_t)rank;
        c = n_frames > (uint32_t)rank ? (n_frames - (uint32_t)rank + (uint32_t)n_ranks - 1u) / (uint32_t)n_ranks : 0u;
    }
    if (first) *first = f;
    if (count) *count = c;
}

int rthip_group_render(rthip_group* g, uint32_t first_frame, uint32_t n_frames, const rthip_render_params* params) {
    if (!g) return fail(nullptr, RTHIP_ERR_INVALID, "null group");
    if (g->n_pixels == 0) return fail(g, RTHIP_ERR_STATE, "render: no frame uniforms set");
    const int n = (int)g->ctx.size();
    return for_each_rank(g, "render", [&](int r) {
        uint32_t first, count;
        rthip_group_plan(first_frame, n_frames, n, r, &first, &count);
        // a GPU without a frame still takes part: a render of zero frames settles its reject_hist
        // state, so that it does not overwrite, later, a sum that a reduce left in its accumulator
        return rthip_render(g->ctx[r], first, count, (uint32_t)n, params);
    });
}

int rthip_group_synchronize(rthip_group* g) {
    if (!g) return fail(nullptr, RTHIP_ERR_INVALID, "null group");
    return for_each_rank(g, "synchronize", [&](int r) { return rthip_synchronize(g->ctx[r]); });
}

int rthip_group_reduce(rthip_group* g, int root) {
    if (!g) return fail(nullptr, RTHIP_ERR_INVALID, "null group");
    const int n = (int)g->ctx.size();
    if (root < 0 || root >= n) return fail(g, RTHIP_ERR_INVALID, "reduce: root %d of %d", root, n);
    if (g->n_pixels == 0) return fail(g, RTHIP_ERR_STATE, "reduce: no frame uniforms set");
    if (g->comm.empty()) {
        g->comm.assign(n, nullptr);
        const ncclResult_t e = ncclCommInitAll(g->comm.data(), n, g->devices.data());
        if (e != ncclSuccess) {
            g->comm.clear();
            return fail(g, RTHIP_ERR_HIP, "reduce: ncclCommInitAll: %s", ncclGetErrorString(e));
        }
    }
    // one collective, enqueued behind each GPU's rendering on that GPU's stream
    ncclResult_t e = ncclGroupStart();
    for (int r = 0; r < n && e == ncclSuccess; r++)
        e = ncclReduce(g->accum[r], g->accum[r], (size_t)g->n_pixels * 4, ncclFloat32, ncclSum, root, g->comm[r], g->stream[r]);
    const ncclResult_t e2 = ncclGroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) return fail(g, RTHIP_ERR_HIP, "reduce: ncclReduce: %s", ncclGetErrorString(e));
    for (int r = 0; r < n; r++) {
        hipError_t h = hipSetDevice(g->devices[r]);
        // the samples of the other GPUs now live on the root: they start again from zero
        if (h == hipSuccess && r != root) h = hipMemsetAsync(g->accum[r], 0, g->n_pixels * 16, g->stream[r]);
        if (h == hipSuccess) h = hipStreamSynchronize(g->stream[r]);
        if (h != hipSuccess) return fail(g, RTHIP_ERR_HIP, "reduce: rank %d: %s", r, hipGetErrorString(h));
    }
    return RTHIP_OK;
}

} // extern "C"
