//! Headless run of the reference's compute pass on a software adapter.
//!
//! What the reference does per window frame (main.rs:356-407) without the window: the same
//! bind-group interface (shader.wgsl:1-9; main.rs:196-221), the same dispatch
//! (ceil(w/8) x ceil(h/8) workgroups of 8x8, main.rs:382-393), `frame` incremented before every
//! dispatch and `reject_hist` cleared after the first (main.rs:372, 403).  Only `cs_main` is built
//! into a pipeline; `vs_main`/`fs_main` need a surface and are not part of what is compared.
use std::borrow::Cow;
use std::fs;
use std::io::Write;
use std::time::Instant;

struct RtScene {
    tris: Vec<u8>,
    tri_exts: Vec<u8>,
    nodes: Vec<u8>,
    prims: Vec<u8>,
    texels: Vec<u8>,
    env: Vec<u8>,
    env_w: u32,
    env_h: u32,
}

fn pad16(n: usize) -> usize {
    (n + 15) & !15
}

/// "RTSCENE1", u64 n_tris n_nodes n_texels, u32 n_prims env_w env_h n_cameras, 16 zero bytes,
/// then tris (48 B), tri_exts (96 B), nodes (32 B), prims (240 B), texels (u32), env (rgba32f),
/// each padded to 16 bytes.
fn read_rtscene(path: &str) -> RtScene {
    let b = fs::read(path).unwrap_or_else(|e| panic!("cannot read {path}: {e}"));
    assert!(b.len() >= 64 && &b[0..8] == b"RTSCENE1", "{path} is not an RTSCENE1 file");
    let u64_at = |o: usize| u64::from_le_bytes(b[o..o + 8].try_into().unwrap()) as usize;
    let u32_at = |o: usize| u32::from_le_bytes(b[o..o + 4].try_into().unwrap());
    let (n_tris, n_nodes, n_texels) = (u64_at(8), u64_at(16), u64_at(24));
    let (n_prims, env_w, env_h) = (u32_at(32) as usize, u32_at(36), u32_at(40));
    let mut pos = 64usize;
    let mut take = |bytes: usize| {
        let v = b[pos..pos + bytes].to_vec();
        pos += pad16(bytes);
        v
    };
    RtScene {
        tris: take(48 * n_tris),
        tri_exts: take(96 * n_tris),
        nodes: take(32 * n_nodes),
        prims: take(240 * n_prims),
        texels: take(4 * n_texels),
        env: take(16 * env_w as usize * env_h as usize),
        env_w,
        env_h,
    }
}

fn arg(args: &[String], name: &str) -> Option<String> {
    args.iter().position(|a| a == name).and_then(|i| args.get(i + 1).cloned())
}

fn storage_buffer(device: &wgpu::Device, queue: &wgpu::Queue, bytes: &[u8]) -> wgpu::Buffer {
    // an empty array cannot be bound: keep at least one element's worth of zeroes
    let size = bytes.len().max(256) as u64;
    let buf = device.create_buffer(&wgpu::BufferDescriptor {
        label: None,
        size,
        usage: wgpu::BufferUsages::STORAGE | wgpu::BufferUsages::COPY_DST | wgpu::BufferUsages::COPY_SRC,
        mapped_at_creation: false,
    });
    if !bytes.is_empty() {
        queue.write_buffer(&buf, 0, bytes); // what upload_scene does, main.rs:330-334
    }
    buf
}

fn layout_entry(binding: u32, ty: wgpu::BindingType) -> wgpu::BindGroupLayoutEntry {
    wgpu::BindGroupLayoutEntry { binding, visibility: wgpu::ShaderStages::COMPUTE, ty, count: None }
}

async fn run() {
    let args: Vec<String> = std::env::args().collect();
    let reference_dir = arg(&args, "--reference-dir").expect("--reference-dir <checkout of hartleyblakey/raytracer>");
    let scene = read_rtscene(&arg(&args, "--rtscene").expect("--rtscene <file>"));
    let mut uniforms = fs::read(arg(&args, "--uniforms").expect("--uniforms <608-byte FrameUniforms>")).unwrap();
    assert_eq!(uniforms.len(), 608, "FrameUniforms is 608 bytes (main.rs:31-43)");
    let frames: u32 = arg(&args, "--frames").map(|s| s.parse().unwrap()).unwrap_or(1);
    let repeats: usize = arg(&args, "--repeats").map(|s| s.parse().unwrap()).unwrap_or(5);
    let out = arg(&args, "--out").expect("--out <dump file>");
    let shader_src = fs::read_to_string(format!("{reference_dir}/src/shader.wgsl")).expect("src/shader.wgsl of the reference");

    let u32_at = |b: &[u8], o: usize| u32::from_le_bytes(b[o..o + 4].try_into().unwrap());
    let (w, h) = (u32_at(&uniforms, 576), u32_at(&uniforms, 580));
    let first_frame = u32_at(&uniforms, 584); // the index the FIRST dispatch sees (the reference starts at 1, main.rs:372)

    // software adapter, no surface (gpu.rs:508-545 asks for a surface-compatible hardware one)
    let instance = wgpu::Instance::default();
    let adapter = instance
        .request_adapter(&wgpu::RequestAdapterOptions {
            power_preference: wgpu::PowerPreference::default(),
            force_fallback_adapter: true,
            compatible_surface: None,
        })
        .await
        .expect("no software (fallback) adapter: lavapipe baseline unavailable");
    let info = adapter.get_info();
    eprintln!("adapter: {} ({:?}, {:?}), driver {} {}", info.name, info.device_type, info.backend, info.driver, info.driver_info);
    let (device, queue) = adapter
        .request_device(
            &wgpu::DeviceDescriptor {
                label: None,
                required_features: wgpu::Features::empty(),
                required_limits: adapter.limits(),
                memory_hints: wgpu::MemoryHints::MemoryUsage,
            },
            None,
        )
        .await
        .expect("device");

    let module = device.create_shader_module(wgpu::ShaderModuleDescriptor {
        label: Some("reference shader.wgsl, unmodified"),
        source: wgpu::ShaderSource::Wgsl(Cow::Owned(shader_src)),
    });

    // group 0: FrameUniforms; group 1: six read_write storage buffers + the environment texture
    let bgl0 = device.create_bind_group_layout(&wgpu::BindGroupLayoutDescriptor {
        label: None,
        entries: &[layout_entry(0, wgpu::BindingType::Buffer { ty: wgpu::BufferBindingType::Uniform, has_dynamic_offset: false, min_binding_size: None })],
    });
    let storage = wgpu::BindingType::Buffer { ty: wgpu::BufferBindingType::Storage { read_only: false }, has_dynamic_offset: false, min_binding_size: None };
    let bgl1 = device.create_bind_group_layout(&wgpu::BindGroupLayoutDescriptor {
        label: None,
        entries: &[
            layout_entry(0, storage),
            layout_entry(1, storage),
            layout_entry(2, storage),
            layout_entry(3, storage),
            layout_entry(4, storage),
            layout_entry(5, storage),
            layout_entry(6, wgpu::BindingType::Texture {
                sample_type: wgpu::TextureSampleType::Float { filterable: false }, // Rgba32Float
                view_dimension: wgpu::TextureViewDimension::D2,
                multisampled: false,
            }),
        ],
    });
    let pipeline_layout = device.create_pipeline_layout(&wgpu::PipelineLayoutDescriptor { label: None, bind_group_layouts: &[&bgl0, &bgl1], push_constant_ranges: &[] });
    let pipeline = device.create_compute_pipeline(&wgpu::ComputePipelineDescriptor {
        label: None,
        layout: Some(&pipeline_layout),
        module: &module,
        entry_point: Some("cs_main"),
        compilation_options: Default::default(),
        cache: None,
    });

    let uniform_buf = device.create_buffer(&wgpu::BufferDescriptor {
        label: None,
        size: 608,
        usage: wgpu::BufferUsages::UNIFORM | wgpu::BufferUsages::COPY_DST,
        mapped_at_creation: false,
    });
    let tris = storage_buffer(&device, &queue, &scene.tris);
    let exts = storage_buffer(&device, &queue, &scene.tri_exts);
    let nodes = storage_buffer(&device, &queue, &scene.nodes);
    let texels = storage_buffer(&device, &queue, &scene.texels);
    let prims = storage_buffer(&device, &queue, &scene.prims);
    let screen_bytes = w as u64 * h as u64 * 16;
    let screen = device.create_buffer(&wgpu::BufferDescriptor {
        label: None,
        size: screen_bytes,
        usage: wgpu::BufferUsages::STORAGE | wgpu::BufferUsages::COPY_SRC,
        mapped_at_creation: false,
    });
    let readback = device.create_buffer(&wgpu::BufferDescriptor {
        label: None,
        size: screen_bytes,
        usage: wgpu::BufferUsages::MAP_READ | wgpu::BufferUsages::COPY_DST,
        mapped_at_creation: false,
    });
    let env_size = wgpu::Extent3d { width: scene.env_w, height: scene.env_h, depth_or_array_layers: 1 };
    let env = device.create_texture(&wgpu::TextureDescriptor {
        label: None,
        size: env_size,
        mip_level_count: 1,
        sample_count: 1,
        dimension: wgpu::TextureDimension::D2,
        format: wgpu::TextureFormat::Rgba32Float,
        usage: wgpu::TextureUsages::TEXTURE_BINDING | wgpu::TextureUsages::COPY_DST,
        view_formats: &[],
    });
    queue.write_texture(
        wgpu::ImageCopyTexture { texture: &env, mip_level: 0, origin: wgpu::Origin3d::ZERO, aspect: wgpu::TextureAspect::All },
        &scene.env,
        wgpu::ImageDataLayout { offset: 0, bytes_per_row: Some(16 * scene.env_w), rows_per_image: Some(scene.env_h) },
        env_size,
    ); // main.rs:336-349
    let env_view = env.create_view(&wgpu::TextureViewDescriptor::default());

    let bg0 = device.create_bind_group(&wgpu::BindGroupDescriptor {
        label: None,
        layout: &bgl0,
        entries: &[wgpu::BindGroupEntry { binding: 0, resource: uniform_buf.as_entire_binding() }],
    });
    let bg1 = device.create_bind_group(&wgpu::BindGroupDescriptor {
        label: None,
        layout: &bgl1,
        entries: &[
            wgpu::BindGroupEntry { binding: 0, resource: tris.as_entire_binding() },
            wgpu::BindGroupEntry { binding: 1, resource: exts.as_entire_binding() },
            wgpu::BindGroupEntry { binding: 2, resource: nodes.as_entire_binding() },
            wgpu::BindGroupEntry { binding: 3, resource: screen.as_entire_binding() },
            wgpu::BindGroupEntry { binding: 4, resource: texels.as_entire_binding() },
            wgpu::BindGroupEntry { binding: 5, resource: prims.as_entire_binding() },
            wgpu::BindGroupEntry { binding: 6, resource: wgpu::BindingResource::TextureView(&env_view) },
        ],
    });

    // one pass over `frames` samples; repeated for the timing (the accumulator restarts each time)
    let mut per_frame_ms: Vec<f64> = Vec::new();
    for _ in 0..repeats.max(1) {
        let t0 = Instant::now();
        for f in 0..frames {
            uniforms[584..588].copy_from_slice(&(first_frame + f).to_le_bytes());
            uniforms[592..596].copy_from_slice(&(if f == 0 { 1u32 } else { 0u32 }).to_le_bytes());
            queue.write_buffer(&uniform_buf, 0, &uniforms); // main.rs:380
            let mut encoder = device.create_command_encoder(&wgpu::CommandEncoderDescriptor { label: None });
            {
                let mut cpass = encoder.begin_compute_pass(&wgpu::ComputePassDescriptor::default());
                cpass.set_pipeline(&pipeline);
                cpass.set_bind_group(0, &bg0, &[]);
                cpass.set_bind_group(1, &bg1, &[]);
                cpass.dispatch_workgroups((w + 7) / 8, (h + 7) / 8, 1); // main.rs:382-393
            }
            queue.submit(Some(encoder.finish()));
            device.poll(wgpu::Maintain::Wait);
        }
        per_frame_ms.push(t0.elapsed().as_secs_f64() * 1e3 / frames as f64);
    }
    per_frame_ms.sort_by(|a, b| a.partial_cmp(b).unwrap());
    let median = per_frame_ms[per_frame_ms.len() / 2];

    let mut encoder = device.create_command_encoder(&wgpu::CommandEncoderDescriptor { label: None });
    encoder.copy_buffer_to_buffer(&screen, 0, &readback, 0, screen_bytes);
    queue.submit(Some(encoder.finish()));
    let slice = readback.slice(..);
    slice.map_async(wgpu::MapMode::Read, |r| r.expect("map"));
    device.poll(wgpu::Maintain::Wait);
    let data = slice.get_mapped_range();

    let threads = std::env::var("LP_NUM_THREADS").ok().and_then(|s| s.parse::<u32>().ok())
        .unwrap_or_else(|| std::thread::available_parallelism().map(|n| n.get() as u32).unwrap_or(0));
    let mut f = fs::File::create(&out).unwrap();
    f.write_all(b"RTDUMP01").unwrap();
    for v in [w, h, frames, first_frame] {
        f.write_all(&v.to_le_bytes()).unwrap();
    }
    f.write_all(&median.to_le_bytes()).unwrap();
    f.write_all(&threads.to_le_bytes()).unwrap();
    f.write_all(&0u32.to_le_bytes()).unwrap();
    f.write_all(&data).unwrap();
    println!(
        "{{\"adapter\": \"{}\", \"resolution\": \"{}x{}\", \"frames\": {}, \"median_ms_per_frame\": {:.3}, \"threads\": {}, \"dump\": \"{}\"}}",
        info.name, w, h, frames, median, threads, out
    );
}

fn main() {
    pollster::block_on(run());
}
