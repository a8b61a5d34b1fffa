"""The multi-GPU decomposition on CPU: two processes over gloo.  Each rank renders its own
global sample ranges (raytracer_amd.sharding, as bench.py does), projects its spectral sums on
the colour matching functions, the per-rank XYZ sums are added with all_reduce and encoded, and
the result must equal one process rendering the whole range (summing the spectra themselves is
checked too).  The per-rank renderer here is the CPU checker standing in for the GPU
library; what is under test is the sharding arithmetic and the reduction."""
import os
import socket
import sys

import numpy as np
import pytest

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
    from raytracer_amd.api import params2d
    from raytracer_amd.sharding import sample_offset, total_spp
    from tests.oracle_ffi import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    oracle = Oracle(os.path.join(ROOT, "oracle", "build", "librt_oracle.so"))
    world = rt.World2D("complex_world", seed=1)
    W, H, spp, steps = 24, 16, 2, 2
    acc = np.zeros((H, W, 95), np.float32)
    for step in range(steps):
        off = sample_offset(step, rank, world_size, spp)
        p = params2d(W, H, spp, 2, 10, seed=5, sample_offset=off)
        spec = oracle.render2d(world, p, threads=1)[2]
        acc += spec * np.float32(spp)          # compute_pixel returns the mean; the buffer holds the sum
    xyz = torch.from_numpy(oracle.project_xyz(acc))    # what bench.py exchanges: 12 bytes per pixel
    dist.all_reduce(xyz, op=dist.ReduceOp.SUM)
    t = torch.from_numpy(acc)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        n = total_spp(steps, world_size, spp)
        whole_rgb, _, whole, _ = oracle.render2d(world, params2d(W, H, n, 2, 10, seed=5), threads=1)
        np.savez(out_path, sharded=t.numpy() / np.float32(n), whole=whole,
                 sharded_rgb=oracle.encode_xyz(xyz.numpy(), n).reshape(H, W, 3), whole_rgb=whole_rgb)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    assert r["whole"].any()
    assert np.allclose(r["sharded"], r["whole"], rtol=1e-5, atol=1e-7)
    assert r["whole_rgb"].any()
    assert np.allclose(r["sharded_rgb"], r["whole_rgb"], rtol=1e-5, atol=1e-6)


def test_shard_offsets_partition_the_sample_range():
    from raytracer_amd.sharding import sample_offset, total_spp
    for world_size in (1, 2, 4, 8):
        spp, steps = 64, 3
        covered = []
        for step in range(steps):
            for rank in range(world_size):
                off = sample_offset(step, rank, world_size, spp)
                covered.extend(range(off, off + spp))
        assert sorted(covered) == list(range(total_spp(steps, world_size, spp)))
    with pytest.raises(ValueError):
        sample_offset(0, 2, 2, 64)


def test_strong_ranges_partition_the_fixed_job():
    from raytracer_amd.sharding import strong_range
    for total in (4096, 4097, 7, 0):
        for world_size in (1, 2, 3, 4, 8):
            parts = [strong_range(r, world_size, total) for r in range(world_size)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        strong_range(2, 2, 64)


def _worker3d(rank, world_size, port, out_path):
    """configs[4]'s 3d leg on CPU: every rank sums its contiguous sample range of the 3d image, one all-reduce adds the
    f32[W*H][3] images (World::compute_pixel is a SUM over spp, 3d/src/rt.rs:239-246, so the image needs no division)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd.api import params3d
    from raytracer_amd.sharding import strong_range
    from tests.oracle_ffi import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    oracle = Oracle(os.path.join(ROOT, "oracle", "build", "librt_oracle.so"))
    scene = rt.Standalone3D(32, 24, use_bvh=False)
    total = 5                                   # odd on purpose: the parts differ in length
    lo, hi = strong_range(rank, world_size, total)
    img = oracle.render3d(scene, params3d(hi - lo, 10, seed=9, sample_offset=lo), threads=1)[0]
    t = torch.from_numpy(np.ascontiguousarray(img, np.float32))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        whole = oracle.render3d(scene, params3d(total, 10, seed=9), threads=1)[0]
        np.savez(out_path, sharded=t.numpy(), whole=whole)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sum_the_3d_image(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    out = str(tmp_path / "result3d.npz")
    mp.spawn(_worker3d, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    assert r["whole"].any()
    assert np.allclose(r["sharded"], r["whole"], rtol=1e-5, atol=1e-6)
