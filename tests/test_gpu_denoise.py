"""rt2d_denoise (one pass of ReverseRenderer::denoise, 2d/src/librt2d_reverse.rs:337-422) against the CPU checker:
the same pixels are selected (weights exact, recomputed-set size and ray count exact), the added colours agree to
the f32 summation tolerance of a pixel's samples."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def two_passes(scene, w, h, spp, ls, seed):
    """merged and loop images as endless_render has them when it calls denoise (:463-467): first pass + second pass."""
    first = scene.global_render(params2d(w, h, spp, ls, 10, seed=seed, sample_offset=0))
    loop = scene.global_render(params2d(w, h, spp, ls, 10, seed=seed, sample_offset=spp))
    merged_rgb = first[0] + loop[0]                      # RawImage Add sums values and drops weights (img.rs:257-274)
    merged_w = np.zeros((h, w), np.float32)
    return merged_rgb.astype(np.float32), merged_w, loop[0].astype(np.float32)


@pytest.mark.parametrize("name,w,h,top,mask,over", [
    ("sample_world", 96, 72, 0.02, 2, 1.0),
    ("complex_world", 80, 60, 0.05, 1, 2.5),
    ("cornell_box", 64, 40, 0.1, 0, 1.0),
    ("simple_world", 33, 21, 0.3, 3, 1.0),       # a mask that reaches outside the image on every side
])
def test_denoise_pass_matches_checker(name, w, h, top, mask, over, gpu_ctx, oracle):
    world = rt.World2D(name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    merged_rgb, merged_w, loop_rgb = two_passes(scene, w, h, 4, 2, seed=9)
    p = params2d(w, h, 3, 2, 10, seed=9, sample_offset=8)
    want_rgb, want_w, key, want_n, want_rays = oracle.denoise2d(world, p, top, mask, over, merged_rgb, merged_w, loop_rgb)
    gpu_ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        d_rgb = torch.from_numpy(merged_rgb).cuda().contiguous()
        d_w = torch.from_numpy(merged_w).cuda().contiguous()
        d_loop = torch.from_numpy(loop_rgb).cuda().contiguous()
        gpu_ctx.reset_stats()
        n = scene.denoise_device(p, top, mask, over, d_rgb.data_ptr(), d_w.data_ptr(), d_loop.data_ptr())
        torch.cuda.synchronize()
        rays = gpu_ctx.stats()["rays"]
        got_rgb, got_w = d_rgb.cpu().numpy(), d_w.cpu().numpy()
    finally:
        gpu_ctx.set_stream(None)
    assert n == want_n and n > 0
    assert rays == want_rays
    assert np.array_equal(got_w, want_w)                  # exactly the same pixels were re-sampled
    assert (got_w > 0).sum() >= int(np.float32(w * h) * np.float32(top))
    assert np.allclose(got_rgb, want_rgb, rtol=2e-4, atol=2e-5)
    # the host-plane entry does the same
    h_rgb, h_w, h_n = scene.denoise(p, top, mask, over, merged_rgb, merged_w, loop_rgb)
    assert h_n == n and np.array_equal(h_w, want_w) and np.allclose(h_rgb, want_rgb, rtol=2e-4, atol=2e-5)
    scene.close()


def test_denoise_arguments(gpu_ctx):
    world = rt.World2D("simple_world")
    scene = rt.Scene2D(gpu_ctx, world)
    p = params2d(16, 16, 1, 1, 3, seed=1)
    buf = torch.zeros(16 * 16 * 3, device="cuda")
    wt = torch.zeros(16 * 16, device="cuda")
    gpu_ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        assert scene.denoise_device(p, 0.0, 2, 1.0, buf.data_ptr(), wt.data_ptr(), buf.data_ptr()) == 0   # top = 0: nothing to do
        with pytest.raises(rt.RtError):
            scene.denoise_device(p, -0.1, 2, 1.0, buf.data_ptr(), wt.data_ptr(), buf.data_ptr())
        with pytest.raises(rt.RtError):
            scene.denoise_device(p, 0.1, -1, 1.0, buf.data_ptr(), wt.data_ptr(), buf.data_ptr())
        with pytest.raises(rt.RtError):
            scene.denoise_device(p, 0.1, 1, 1.0, 0, wt.data_ptr(), buf.data_ptr())
        # top >= 1 selects every pixel
        n = scene.denoise_device(p, 2.0, 0, 1.0, buf.data_ptr(), wt.data_ptr(), buf.data_ptr())
        torch.cuda.synchronize()
        assert n == 256 and bool((wt == 1.0).all())
    finally:
        gpu_ctx.set_stream(None)
