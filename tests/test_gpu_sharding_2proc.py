"""The HIP path under two ranks: two PROCESSES, each with its own context on the card, render their shares of the
global sample range through the C ABI, project to XYZ on the device, and the XYZ sums meet in an all-reduce (gloo,
staged through the host: the box has one GPU, so the ranks share it - on a node each would own one and the reduction
would run over RCCL).  The image must be the one a single context renders, the ray counts must add up, for the weak
split (step-major ranges, bench.py's default) and the strong one (contiguous parts of a fixed job, configs[4]); the 3d
image likewise (its merge is the plain sum of the ranks' f32[W*H][3] buffers)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd.api import params2d, params3d
    from raytracer_amd.sharding import sample_offset, strong_range, total_spp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    torch.cuda.set_device(0)
    ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    W, H, spp, steps, ls = 96, 72, 3, 2, 2
    scene = rt.Scene2D(ctx, rt.World2D("complex_world", seed=1))
    out = {}

    def merge(buf, n):
        xyz = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
        scene.project_xyz(buf.data_ptr(), W, H, xyz.data_ptr())
        host = xyz.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        xyz.copy_(host)
        return scene.resolve_xyz(xyz.data_ptr(), W, H, n)[0]

    # weak: step-major, rank-minor sample ranges
    buf = torch.zeros(W * H * rt.NBINS, dtype=torch.float32, device="cuda")
    for step in range(steps):
        scene.accumulate(params2d(W, H, spp, ls, 10, seed=5, sample_offset=sample_offset(step, rank, world_size, spp)), buf.data_ptr())
    n = total_spp(steps, world_size, spp)
    out["weak_rgb"] = merge(buf, n)
    # strong: contiguous parts of a fixed job of 7 samples per pixel
    lo, hi = strong_range(rank, world_size, 7)
    buf.zero_()
    scene.accumulate(params2d(W, H, hi - lo, ls, 10, seed=5, sample_offset=lo), buf.data_ptr())
    out["strong_rgb"] = merge(buf, 7)
    rays = torch.tensor([ctx.stats()["rays"]], dtype=torch.int64)
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    # 3d: the image is the sum
    s3 = rt.Scene3D(ctx, rt.Standalone3D(64, 48, use_bvh=True))
    img = torch.zeros(64 * 48 * 3, dtype=torch.float32, device="cuda")
    s3.accumulate(params3d(hi - lo, 10, seed=2, sample_offset=lo), img.data_ptr())
    ctx.synchronize()
    host = img.cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.savez(out_path, weak_rgb=out["weak_rgb"], strong_rgb=out["strong_rgb"], rays=rays.numpy(), img3d=host.numpy(), n_weak=n)
    s3.close()
    scene.close()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_on_the_card_render_one_image(gpu_ctx, tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    import raytracer_amd as rt
    from raytracer_amd.api import params2d, params3d
    out = str(tmp_path / "two_ranks.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    W, H = 96, 72
    scene = rt.Scene2D(gpu_ctx, rt.World2D("complex_world", seed=1))
    gpu_ctx.reset_stats()
    weak = scene.global_render(params2d(W, H, int(r["n_weak"]), 2, 10, seed=5))[0]
    strong = scene.global_render(params2d(W, H, 7, 2, 10, seed=5))[0]
    rays = gpu_ctx.stats()["rays"]
    scene.close()
    assert weak.any() and np.allclose(r["weak_rgb"], weak, rtol=1e-4, atol=1e-5)
    assert np.allclose(r["strong_rgb"], strong, rtol=1e-4, atol=1e-5)
    assert int(r["rays"][0]) == rays                      # the same paths, whichever process traced them
    s3 = rt.Scene3D(gpu_ctx, rt.Standalone3D(64, 48, use_bvh=True))
    img = s3.render(params3d(7, 10, seed=2))
    s3.close()
    assert np.allclose(r["img3d"].reshape(48, 64, 3), img, rtol=1e-5, atol=1e-7)
