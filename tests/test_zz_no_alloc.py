"""rt2d_forward_accumulate_static - the entry point that allocates nothing (its scratch is a fixed array inside the
library) - against rt2d_forward_accumulate on the same inputs, through the C ABI: the number of lines drawn and the
weight plane (how many line items touched each pixel) are integers and must be equal; coverage sums are f32 atomics
added in a different order and must agree to 1e-5 relative.  Two calls in a row on the same GPU, the second on a
smaller image with fewer rays: the scratch keeps the first call's contents and the second must not see them."""
import numpy as np
import pytest
import torch

import raytracer_amd as rt
from raytracer_amd import _ffi

pytestmark = pytest.mark.gpu


def forward_pair(ctx, world_name, W, H, rays, depth, seed, offset):
    scene = rt.Scene2D(ctx, rt.World2D(world_name, seed=1))
    p = scene.forward_params(W, H, rays, depth, seed=seed, sample_offset=offset)
    start = torch.rand(W * H * 4, device="cuda")          # planes that already hold something: the entry points ADD
    a, b = start.clone(), start.clone()
    d_lines = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    want_lines = scene.forward_accumulate(p, a.data_ptr(), a.data_ptr() + 4 * 3 * W * H, count_lines=True)
    scene.forward_accumulate_static(p, b.data_ptr(), b.data_ptr() + 4 * 3 * W * H, d_lines.data_ptr())
    ctx.synchronize()
    torch.cuda.synchronize()
    scene.close()
    a, b, start = a.cpu().numpy(), b.cpu().numpy(), start.cpu().numpy()
    n = W * H
    return (a[:3 * n], a[3 * n:], want_lines), (b[:3 * n], b[3 * n:], int(d_lines.item())), (start[:3 * n], start[3 * n:])


def check(want, got, start):
    assert got[2] == want[2] and want[2] > 0, "lines drawn"
    # weight += number of items: integers far below 2^24 on top of a start value in [0, 1): compare the increments
    assert np.array_equal(np.rint(got[1] - start[1]), np.rint(want[1] - start[1]))
    assert np.allclose(got[1], want[1], rtol=1e-6, atol=1e-6)
    assert (np.rint(want[1] - start[1]) > 0).sum() > 100
    assert np.allclose(got[0], want[0], rtol=1e-5, atol=1e-5)


def test_static_entry_point_matches_on_two_calls_in_a_row(gpu_ctx):
    gpu_ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        # 49 emitters x 2000 rays = 98 000 rays: two chunks of the static staging buffer
        check(*forward_pair(gpu_ctx, "complex_world", 800, 600, 2000, 5, seed=7, offset=0))
        # smaller image (pixel == world unit: the image is the world's corner), fewer rays, another seed: stale
        # coverage or a stale counter from the call above would show
        check(*forward_pair(gpu_ctx, "complex_world", 320, 240, 300, 5, seed=8, offset=100))
    finally:
        gpu_ctx.set_stream(None)


def test_static_entry_point_rejects_what_its_scratch_cannot_hold(gpu_ctx):
    scene = rt.Scene2D(gpu_ctx, rt.World2D("cornell_box", seed=1))
    p = scene.forward_params(1025, 1024, 10, 5)
    buf = torch.zeros(1025 * 1024 * 4, device="cuda")
    with pytest.raises(_ffi.RtError) as e:
        scene.forward_accumulate_static(p, buf.data_ptr(), buf.data_ptr() + 4 * 3 * 1025 * 1024)
    assert e.value.code == -5 and "static scratch" in str(e.value)   # RT_E_LIMIT
    scene.close()


def test_static_entry_point_allocates_nothing_and_can_be_captured(gpu_ctx):
    """The contract beyond the result: no device allocation (free memory is the same before and after a call), and no
    event, allocation or synchronisation on the way - so the call can be recorded into a HIP graph, where any of those
    fails the capture, and the replayed graph adds what the eager call adds."""
    W, H = 400, 300
    scene = rt.Scene2D(gpu_ctx, rt.World2D("complex_world", seed=1))
    p = scene.forward_params(W, H, 500, 5, seed=3)
    side = torch.cuda.Stream()
    gpu_ctx.set_stream(side.cuda_stream)
    try:
        eager = torch.zeros(W * H * 4, device="cuda")
        replayed = torch.zeros(W * H * 4, device="cuda")
        d_lines = torch.zeros(1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        scene.forward_accumulate_static(p, eager.data_ptr(), eager.data_ptr() + 4 * 3 * W * H, d_lines.data_ptr())   # warm-up: module load
        gpu_ctx.synchronize()
        eager.zero_()
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        scene.forward_accumulate_static(p, eager.data_ptr(), eager.data_ptr() + 4 * 3 * W * H, d_lines.data_ptr())
        gpu_ctx.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
        lines_eager = int(d_lines.item())
        assert lines_eager > 0
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            scene.forward_accumulate_static(p, replayed.data_ptr(), replayed.data_ptr() + 4 * 3 * W * H, d_lines.data_ptr())
        torch.cuda.synchronize()
        assert not replayed.any()                       # recording runs nothing
        d_lines.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert int(d_lines.item()) == lines_eager
        a, b = eager.cpu().numpy(), replayed.cpu().numpy()
        assert np.array_equal(np.rint(a[3 * W * H:]), np.rint(b[3 * W * H:]))          # items per pixel
        assert np.allclose(a, b, rtol=1e-5, atol=1e-5)
        graph.replay()                                  # a second replay adds the same again
        torch.cuda.synchronize()
        assert np.allclose(replayed.cpu().numpy(), 2.0 * a, rtol=1e-5, atol=1e-5)
    finally:
        gpu_ctx.set_stream(None)
        scene.close()
