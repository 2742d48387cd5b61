//! Bindings to `include/rthip.h`.  NOT BUILT in the authoring environment (no cargo/rustc there);
//! kept thin on purpose: `extern "C"` declarations that mirror the header one to one, and a safe
//! `Context` whose methods take the slices the reference's `Scene` already owns
//! (`Vec<Tri>`, `Vec<TriExt>`, `Vec<BvhNode>`, `Vec<Primitive>`, `Vec<u32>`, `Vec<[f32; 4]>`).
#![allow(non_camel_case_types)]

use std::ffi::{c_char, c_int, c_void, CStr};

#[repr(C)]
pub struct rthip_ctx {
    _private: [u8; 0],
}

#[repr(C)]
pub struct rthip_scene_desc {
    pub tris: *const c_void,
    pub n_tris: u64,
    pub tri_exts: *const c_void,
    pub nodes: *const c_void,
    pub n_nodes: u64,
    pub prims: *const c_void,
    pub n_prims: u32,
    pub texels: *const u32,
    pub n_texels: u64,
    pub env: *const f32,
    pub env_w: u32,
    pub env_h: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rthip_render_params {
    pub max_bounces: u32,
    pub term_prob: f32,
    pub aperture_radius: f32,
    pub stack_depth: u32,
    pub enable_sun: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rthip_counters {
    pub rays: u64,
    pub prims_visited: u64,
    pub inner_steps: u64,
    pub leaf_steps: u64,
    pub tri_tests: u64,
    pub alpha_tests: u64,
    pub texel_fetches: u64,
    pub shade_calls: u64,
    pub misses: u64,
    pub kills: u64,
    pub stack_overflows: u64,
    pub paths: u64,
    pub rounds_prim: u64,
    pub rounds_inner: u64,
    pub rounds_leaf: u64,
    pub batches: u64,
    pub shadow_rays: u64,
}

/// Mirrors `rthip_timing`: summed kernel time and launch count per stage
/// (generate, extend, shade, resolve, accumulate, shadow, cull, path).
pub const RTHIP_NUM_STAGES: usize = 8;
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rthip_timing {
    pub ms: [f64; RTHIP_NUM_STAGES],
    pub launches: [u64; RTHIP_NUM_STAGES],
}

pub const RTHIP_FLAG_COUNTERS: u32 = 1;
pub const RTHIP_FLAG_TAPS: u32 = 2;
pub const RTHIP_FLAG_TIMING: u32 = 4;
pub const RTHIP_FLAG_STAGED: u32 = 8;

extern "C" {
    pub fn rthip_device_count() -> c_int;
    pub fn rthip_create(device_ordinal: c_int, out: *mut *mut rthip_ctx) -> c_int;
    pub fn rthip_destroy(ctx: *mut rthip_ctx);
    pub fn rthip_last_error(ctx: *const rthip_ctx) -> *const c_char;
    pub fn rthip_set_stream(ctx: *mut rthip_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn rthip_upload_scene(ctx: *mut rthip_ctx, scene: *const rthip_scene_desc) -> c_int;
    pub fn rthip_set_frame_uniforms(ctx: *mut rthip_ctx, blob608: *const c_void) -> c_int;
    pub fn rthip_default_params(p: *mut rthip_render_params);
    pub fn rthip_render(ctx: *mut rthip_ctx, first_frame: u32, n_frames: u32, frame_stride: u32,
                        params: *const rthip_render_params) -> c_int;
    pub fn rthip_reserve(ctx: *mut rthip_ctx, n_frames: u32) -> c_int;
    pub fn rthip_trace_rays(ctx: *mut rthip_ctx, rays6: *const f32, seeds: *mut u32, n: u64, params: *const rthip_render_params,
                            t: *mut f32, tri_idx: *mut i32, prim_idx: *mut i32, overflow: *mut u8) -> c_int;
    /// All pointers are device memory; allocates nothing; at most 1 << 18 rays per call.
    pub fn rthip_trace_rays_noalloc(ctx: *mut rthip_ctx, d_rays6: *const f32, d_seeds: *mut u32, n: u64, params: *const rthip_render_params,
                                    d_t: *mut f32, d_tri_idx: *mut i32, d_prim_idx: *mut i32, d_overflow: *mut u8) -> c_int;
    pub fn rthip_frames_in_flight(ctx: *mut rthip_ctx, n_frames: *mut u32) -> c_int;
    /// 0 = one mask per ray, 1 = chunk scan, 2 = bounds pyramid, 3 = per-ray candidate lists.
    pub fn rthip_candidate_walk(ctx: *mut rthip_ctx, kind: *mut u32) -> c_int;
    pub fn rthip_copy_bandwidth(ctx: *mut rthip_ctx, bytes: u64, repeats: u32, gb_per_s: *mut f64) -> c_int;
    /// Lane-loads per second of the traversal's access shape out of the L1 (a measured ceiling).
    pub fn rthip_gather_ceiling(ctx: *mut rthip_ctx, active_lanes: u32, record_bytes: u32, records_in_flight: u32, lane_loads_per_s: *mut f64) -> c_int;
    pub fn rthip_bind_accum(ctx: *mut rthip_ctx, device_ptr: *mut c_void, n_pixels: u64) -> c_int;
    pub fn rthip_accum_device_ptr(ctx: *mut rthip_ctx, device_ptr: *mut *mut c_void, n_pixels: *mut u64) -> c_int;
    pub fn rthip_read_accum(ctx: *mut rthip_ctx, rgba: *mut f32, n_pixels: u64) -> c_int;
    pub fn rthip_write_accum(ctx: *mut rthip_ctx, rgba: *const f32, n_pixels: u64) -> c_int;
    pub fn rthip_read_taps(ctx: *mut rthip_ctx, tri_idx: *mut i32, prim_idx: *mut i32, t: *mut f32,
                           final_seed: *mut u32, n_pixels: u64) -> c_int;
    pub fn rthip_read_nodes_device_layout(ctx: *mut rthip_ctx, nodes_out: *mut c_void, capacity: u64,
                                          n_records: *mut u64) -> c_int;
    pub fn rthip_read_node_index_map(ctx: *mut rthip_ctx, out: *mut u32, n_nodes: u64) -> c_int;
    pub fn rthip_read_tri_order(ctx: *mut rthip_ctx, out: *mut u32, n_tris: u64) -> c_int;
    pub fn rthip_read_tris_device_layout(ctx: *mut rthip_ctx, out: *mut c_void, n_tris: u64) -> c_int;
    pub fn rthip_resolve(ctx: *mut rthip_ctx, rgba8: *mut u8, n_pixels: u64, apply_srgb_oetf: c_int) -> c_int;
    pub fn rthip_get_counters(ctx: *mut rthip_ctx, out: *mut rthip_counters) -> c_int;
    pub fn rthip_reset_counters(ctx: *mut rthip_ctx) -> c_int;
    pub fn rthip_get_timing(ctx: *mut rthip_ctx, out: *mut rthip_timing) -> c_int;
    pub fn rthip_reset_timing(ctx: *mut rthip_ctx) -> c_int;
    pub fn rthip_synchronize(ctx: *mut rthip_ctx) -> c_int;
}

/// `include/rthip_group.h` (librthip_group.so): all GPUs of a node from this one process.
#[repr(C)]
pub struct rthip_group {
    _private: [u8; 0],
}

#[link(name = "rthip_group")]
extern "C" {
    pub fn rthip_group_create(device_ordinals: *const c_int, n_devices: c_int, out: *mut *mut rthip_group) -> c_int;
    pub fn rthip_group_destroy(g: *mut rthip_group);
    pub fn rthip_group_last_error(g: *const rthip_group) -> *const c_char;
    pub fn rthip_group_size(g: *const rthip_group) -> c_int;
    pub fn rthip_group_ctx(g: *mut rthip_group, rank: c_int) -> *mut rthip_ctx;
    pub fn rthip_group_upload_scene(g: *mut rthip_group, scene: *const rthip_scene_desc) -> c_int;
    pub fn rthip_group_set_frame_uniforms(g: *mut rthip_group, blob608: *const c_void) -> c_int;
    pub fn rthip_group_plan(first_frame: u32, n_frames: u32, n_ranks: c_int, rank: c_int, first: *mut u32, count: *mut u32);
    pub fn rthip_group_render(g: *mut rthip_group, first_frame: u32, n_frames: u32, params: *const rthip_render_params) -> c_int;
    pub fn rthip_group_reduce(g: *mut rthip_group, root: c_int) -> c_int;
    pub fn rthip_group_synchronize(g: *mut rthip_group) -> c_int;
}


#[derive(Debug)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

/// One GPU.  Not `Sync`: drive a context from one thread.
pub struct Context {
    raw: *mut rthip_ctx,
}

impl Context {
    pub fn new(device_ordinal: i32) -> Result<Self, Error> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { rthip_create(device_ordinal, &mut raw) };
        if rc != 0 {
            let message = unsafe { CStr::from_ptr(rthip_last_error(std::ptr::null())) }.to_string_lossy().into_owned();
            return Err(Error { code: rc, message });
        }
        Ok(Self { raw })
    }

    fn check(&self, rc: c_int) -> Result<(), Error> {
        if rc == 0 {
            return Ok(());
        }
        let message = unsafe { CStr::from_ptr(rthip_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(Error { code: rc, message })
    }

    /// Replaces `Context::upload_scene` (main.rs:328-353).  The slices are the reference's own
    /// pod vectors viewed as bytes (`bytemuck::cast_slice`).
    pub fn upload_scene(&mut self, tris: &[u8], tri_exts: &[u8], nodes: &[u8], prims: &[u8], texels: &[u32],
                        env: &[[f32; 4]], env_w: u32, env_h: u32) -> Result<(), Error> {
        assert!(tris.len() % 48 == 0 && tri_exts.len() == tris.len() * 2 && nodes.len() % 32 == 0 && prims.len() % 240 == 0);
        assert!(env.len() == env_w as usize * env_h as usize);
        let d = rthip_scene_desc {
            tris: tris.as_ptr().cast(), n_tris: (tris.len() / 48) as u64, tri_exts: tri_exts.as_ptr().cast(),
            nodes: nodes.as_ptr().cast(), n_nodes: (nodes.len() / 32) as u64,
            prims: prims.as_ptr().cast(), n_prims: (prims.len() / 240) as u32,
            texels: texels.as_ptr(), n_texels: texels.len() as u64,
            env: env.as_ptr().cast(), env_w, env_h,
        };
        self.check(unsafe { rthip_upload_scene(self.raw, &d) })
    }

    /// Replaces `queue.write_buffer(&frame_uniforms_buffer, ..)` (main.rs:380).
    pub fn set_frame_uniforms(&mut self, frame_uniforms: &[u8; 608]) -> Result<(), Error> {
        self.check(unsafe { rthip_set_frame_uniforms(self.raw, frame_uniforms.as_ptr().cast()) })
    }

    /// Replaces the compute pass of `frame()` (main.rs:383-393) for frames
    /// `first, first + stride, ...`.
    pub fn render(&mut self, first_frame: u32, n_frames: u32, frame_stride: u32,
                  params: Option<&rthip_render_params>) -> Result<(), Error> {
        let p = params.map_or(std::ptr::null(), |p| p as *const _);
        self.check(unsafe { rthip_render(self.raw, first_frame, n_frames, frame_stride, p) })
    }

    /// Replaces the render pass of `frame()` (main.rs:395-401): tonemapped sRGB bytes.
    pub fn resolve(&mut self, rgba8: &mut [u8]) -> Result<(), Error> {
        self.check(unsafe { rthip_resolve(self.raw, rgba8.as_mut_ptr(), (rgba8.len() / 4) as u64, 1) })
    }

    pub fn read_accum(&mut self, rgba: &mut [f32]) -> Result<(), Error> {
        self.check(unsafe { rthip_read_accum(self.raw, rgba.as_mut_ptr(), (rgba.len() / 4) as u64) })
    }

    pub fn counters(&mut self) -> Result<rthip_counters, Error> {
        let mut c = rthip_counters::default();
        self.check(unsafe { rthip_get_counters(self.raw, &mut c) })?;
        Ok(c)
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { rthip_destroy(self.raw) }
    }
}
