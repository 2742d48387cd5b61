"""Sample-parallel rendering across the GPUs of one node: one process per GPU.

The scene is replicated in every GPU's HBM; the samples of a pixel are independent given
(pixel, frame) (shader.wgsl:895), so rank r of W renders frames r+1, r+1+W, ... into a private
float4 accumulator and one sum-reduce to rank 0 (RCCL over xGMI: torch's "nccl" backend on ROCm)
merges them.  The union of samples equals the single-GPU run; only the order of the float
additions differs.  The reference itself is single-device (main.rs:356-407).
"""


def frame_stripe(total_frames, rank, world, first_frame=1):
    """Frames of `rank`: (first, count, stride).  Frame indices start at 1 (main.rs:372)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    count = (total_frames - rank + world - 1) // world if total_frames > rank else 0
    return first_frame + rank, count, world


def render_striped(render_fn, accum, total_frames, rank, world, first_frame=1, reduce_fn=None):
    """Renders this rank's stripe through `render_fn(first, count, stride)` (which accumulates
    into `accum`, overwriting on its first frame) and merges with `reduce_fn(accum)` (e.g.
    torch.distributed.reduce to rank 0).  A rank with no frames contributes zeros."""
    first, count, stride = frame_stripe(total_frames, rank, world, first_frame)
    if count > 0:
        render_fn(first, count, stride)
    else:
        accum.zero_()
    if reduce_fn is not None and world > 1:
        reduce_fn(accum)
    return accum


class DeviceRenderer:
    """Binds a raytracer_amd.device.Context to a torch CUDA tensor so that torch.distributed can
    reduce the accumulator in place."""

    def __init__(self, ctx, torch, res, device):
        self.ctx = ctx
        self.torch = torch
        self.res = res
        self.accum = torch.zeros(res[0] * res[1] * 4, dtype=torch.float32, device=device)
        ctx.bind_accum(self.accum.data_ptr(), res[0] * res[1])

    def render(self, scene, total_frames, rank, world, params=None, first_frame=1, dist=None, root=0, reduce_fn=None):
        """Renders this rank's stripe and sums the accumulators into rank `root`.  `dist` =
        torch.distributed with an initialised group whose backend reduces device tensors in place
        (RCCL: "nccl"); or pass reduce_fn(tensor) for another way of summing (e.g. an all-reduce
        through a backend that has no device-side reduce)."""
        self.ctx.set_frame_uniforms(scene.frame_uniforms(self.res, frame=first_frame, reject_hist=1))

        def render_fn(first, count, stride):
            self.ctx.render(first, count, stride, params)
            self.ctx.synchronize()

        if reduce_fn is None and dist is not None:
            def reduce_fn(t):
                dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)

        return render_striped(render_fn, self.accum, total_frames, rank, world, first_frame, reduce_fn)
