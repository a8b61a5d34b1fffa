"""rt2d_forward_render / rt2d_forward_accumulate (ForwardRenderer::global_render, 2d/src/librt2d_forward.rs:198-249)
against the CPU checker: the same rays escape (line and ray counts exact, hit-count plane exact), coverage sums agree
to the f32 summation tolerance of a pixel's items (lines of different rays meet in atomics)."""
import numpy as np
import pytest

import raytracer_amd as rt
from tests.test_forward_oracle import segment_emitter_world

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("name,w,h,rays,depth", [
    ("simple_world", 200, 150, 400, 5),
    ("sample_world", 800, 600, 300, 5),
    ("complex_world", 800, 600, 100, 5),          # 200 circles, many emitters, mirrors and glass
    ("cornell_box", 800, 600, 500, 9),            # Cauchy glass at the tracer's lambda = 1.
    ("spectrum_world", 320, 240, 250, 1),         # depth 1: only rays that leave at once are drawn
])
def test_forward_render_matches_checker(name, w, h, rays, depth, gpu_ctx, oracle):
    world = rt.World2D(name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    p = scene.forward_params(w, h, rays, depth, seed=11, sample_offset=7)
    rc, want_rgb, want_w, _, want_n, want_rays = oracle.forward_render(world, w, h, rays, depth, seed=11, sample_offset=7)
    assert rc == 0
    gpu_ctx.reset_stats()
    rgb, weight, n = scene.forward_render(p)
    assert n == want_n and n > 0
    assert gpu_ctx.stats()["rays"] == want_rays
    assert np.array_equal(weight, want_w)                       # every line crossed exactly the same pixels
    assert np.allclose(rgb, want_rgb, rtol=1e-5, atol=1e-5)
    assert np.allclose(rgb[..., 0], rgb[..., 1], rtol=1e-5, atol=1e-5)   # three atomics per item, each in its own order
    scene.close()


def test_forward_accumulate_adds_passes(gpu_ctx, oracle):
    world = rt.World2D("sample_world", seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    w, h = 400, 300
    d_rgb = torch.zeros(h * w * 3, device="cuda")
    d_w = torch.ones(h * w, device="cuda")
    gpu_ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        total = 0
        want_rgb = np.zeros((h, w, 3), np.float32)
        want_w = np.ones((h, w), np.float32)
        for k in range(3):                                       # merged + loop + loop (librt2d_forward.rs:263-296)
            total += scene.forward_accumulate(scene.forward_params(w, h, 200, 5, seed=5, sample_offset=200 * k),
                                              d_rgb.data_ptr(), d_w.data_ptr(), count_lines=True)
            rc, r, ww, _, _, _ = oracle.forward_render(world, w, h, 200, 5, seed=5, sample_offset=200 * k)
            want_rgb += r
            want_w += ww - 1.0
        torch.cuda.synchronize()
        assert total > 0
        assert np.array_equal(d_w.cpu().numpy().reshape(h, w), want_w)
        assert np.allclose(d_rgb.cpu().numpy().reshape(h, w, 3), want_rgb, rtol=1e-5, atol=1e-5)
        # nothing to do: no rays, or no depth
        assert scene.forward_accumulate(scene.forward_params(w, h, 0, 5), d_rgb.data_ptr(), d_w.data_ptr(), count_lines=True) == 0
        assert scene.forward_accumulate(scene.forward_params(w, h, 10, 0), d_rgb.data_ptr(), d_w.data_ptr(), count_lines=True) == 0
    finally:
        gpu_ctx.set_stream(None)
    scene.close()


def test_forward_arguments(gpu_ctx):
    scene = rt.Scene2D(gpu_ctx, rt.World2D("simple_world"))
    for bad in (dict(width=0), dict(rays_per_emitter=-1), dict(depth=-1), dict(depth=5000), dict(sample_offset=-1)):
        kw = dict(width=32, height=32, rays_per_emitter=4, depth=5)
        kw.update(bad)
        with pytest.raises(rt.RtError):
            scene.forward_render(scene.forward_params(**kw))
    scene.close()
    # an emissive segment: the reference panics in sample_out_ray, the library refuses
    lit = rt.Scene2D(gpu_ctx, segment_emitter_world())
    with pytest.raises(rt.RtError, match="not a circle"):
        lit.forward_render(lit.forward_params(64, 48, 4, 5))
    # the gather tracer still renders that scene
    from raytracer_amd.api import params2d
    rgb, _ = lit.global_render(params2d(64, 48, 2, 1, 5, seed=1, px_to_world=1.0))
    assert np.isfinite(rgb).all()
    lit.close()
