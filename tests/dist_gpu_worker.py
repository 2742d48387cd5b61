"""One rank of the two-process test of the sample-split path (tests/test_gpu_dist_two_ranks.py): a
fresh process that renders its stripe of frames through raytracer_amd.dist.DeviceRenderer on the GPU
it is given and takes part in the sum of the accumulators.  Not a test module.
    python tests/dist_gpu_worker.py <rank> <world> <port> <total_frames> <root> <device> <out.npy>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    rank, world, port, total_frames, root, dev = (int(v) for v in sys.argv[1:7])
    out_path = sys.argv[7]
    import torch
    import torch.distributed as dist
    from raytracer_amd import device, dist as rdist, procedural
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # gloo: both ranks may share one GPU (RCCL refuses two ranks on one device); its all-reduce takes
    # device tensors, which is all the path needs from a backend
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(dev)
    s = procedural.material_test_scene(tex=32, env_size=(64, 32))
    res = (160, 96)
    ctx = device.Context(dev)
    ctx.upload_scene(s.arrays())
    r = rdist.DeviceRenderer(ctx, torch, res, f"cuda:{dev}")

    def all_reduce_sum(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    acc = r.render(s, total_frames, rank, world, params=device.default_params(max_bounces=4), root=root, reduce_fn=all_reduce_sum)
    torch.cuda.synchronize()
    assert ctx.accum_device_ptr()[0] == r.accum.data_ptr()      # the context rendered into the tensor itself
    if rank == root:
        np.save(out_path, acc.cpu().numpy().reshape(res[1], res[0], 4))
        # the root resolves the summed image through the bound tensor (rthip_resolve on rank root only)
        img = ctx.resolve()
        np.save(out_path + ".rgba8.npy", img)
    dist.barrier()
    ctx.bind_accum(0, 0)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
