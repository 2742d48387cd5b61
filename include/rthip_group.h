/*
 * rthip_group.h -- several GPUs of one node driven from ONE host process (one rthip context,
 * one HIP stream and one host thread per GPU), for hosts that have no process launcher or
 * collective library of their own.  The reference renders on a single wgpu device
 * (gpu.rs:508-590); what is split here is the sample loop of its frame driver: frame() is
 * called once per sample with `frame` incremented (main.rs:356-407), the samples are
 * independent given (pixel, frame) (shader.wgsl:895), so GPU r of n renders frames
 * first + r, first + r + n, ... into its own `screen` accumulator (shader.wgsl:6, 947-951) and
 * one RCCL reduce over xGMI sums the accumulators on a root GPU.  The union of samples equals
 * the single-GPU run; only the order of the float additions differs.
 *
 * Lives in its own library (librthip_group.so: links librthip.so and librccl.so) so that the
 * single-GPU library has no collective dependency.  Process-per-GPU hosts (torch.distributed,
 * MPI) use rthip_render's frame_stride and rthip_bind_accum directly instead.
 *
 * Status codes are rthip_status; rthip_group_last_error gives the text.  A group is not
 * thread-safe: drive it from one thread.
 */
#ifndef RTHIP_GROUP_H
#define RTHIP_GROUP_H

#include "rthip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rthip_group rthip_group;

/* One context per listed device ordinal (no duplicates).  The collective communicator is
 * created by the first rthip_group_reduce. */
int  rthip_group_create(const int* device_ordinals, int n_devices, rthip_group** out);
void rthip_group_destroy(rthip_group* g);
const char* rthip_group_last_error(const rthip_group* g);
int  rthip_group_size(const rthip_group* g);
/* The context of GPU `rank` (borrowed; for read-backs, counters, resolve). */
rthip_ctx* rthip_group_ctx(rthip_group* g, int rank);

/* Context::upload_scene (main.rs:328-353) on every GPU: the scene is replicated. */
int rthip_group_upload_scene(rthip_group* g, const rthip_scene_desc* scene);
/* queue.write_buffer(frame_uniforms_buffer) (main.rs:380) on every GPU. */
int rthip_group_set_frame_uniforms(rthip_group* g, const void* blob608);

/* Which of the n_frames frames first_frame, first_frame + 1, ... GPU `rank` of `n_ranks`
 * renders: frames *first, *first + n_ranks, ... (*count of them).  Pure arithmetic. */
void rthip_group_plan(uint32_t first_frame, uint32_t n_frames, int n_ranks, int rank,
                      uint32_t* first, uint32_t* count);

/* n_frames calls of frame() (main.rs:356-407), striped over the GPUs as rthip_group_plan says.
 * Asynchronous like rthip_render. */
int rthip_group_render(rthip_group* g, uint32_t first_frame, uint32_t n_frames,
                       const rthip_render_params* params);

/* Sum of all accumulators into GPU `root`'s accumulator (ncclReduce, f32 sum, in place, on the
 * contexts' streams).  The other GPUs' accumulators are cleared afterwards, so that rendering
 * may continue and a later reduce adds only the new samples.  Synchronous. */
int rthip_group_reduce(rthip_group* g, int root);

int rthip_group_synchronize(rthip_group* g);

#ifdef __cplusplus
}
#endif
#endif
