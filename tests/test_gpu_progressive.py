"""rt2d_progressive_* (the loop of ReverseRenderer::endless_render, 2d/src/librt2d_reverse.rs:424-510, with the merge
done on spectral sums and per-pixel sample counts) against the CPU checker's statement of the same loop.  Sample
counts, re-sampled pixel sets and ray counts are exact; sums and colours agree to the f32 summation tolerance of a
pixel's samples (the GPU adds them with atomics, in no fixed order).  The selection of a re-sampling pass is checked
on the planes the GPU itself produced, so that both sides select from identical inputs."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi
from raytracer_amd.api import params2d

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-4, 2e-5


def close_enough(a, b):
    return np.allclose(a, b, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("name,w,h,spp,ls,top,mask,over", [
    ("sample_world", 64, 48, 3, 2, 0.03, 2, 2.0),
    ("complex_world", 48, 36, 2, 3, 0.1, 1, 1.5),
    ("cornell_box", 40, 24, 2, 16, 0.2, 3, 0.4),     # floor(2 * 0.4) = 0 -> at least one sample
])
def test_passes_and_resampling_match_checker(name, w, h, spp, ls, top, mask, over, gpu_ctx, oracle):
    world = rt.World2D(name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    first = 5
    p = params2d(w, h, spp, ls, 10, seed=4, sample_offset=first)
    prog = scene.progressive(p, denoiser=(top, mask, over, 1))
    want_sum = np.zeros((h, w, rt.NBINS), np.float32)
    want_count = np.zeros((h, w), np.float32)
    for k in range(2):
        gpu_ctx.reset_stats()
        prog.render_pass()
        got_sum, got_count, got_merged, got_loop = prog.read()
        q = params2d(w, h, spp, ls, 10, seed=4, sample_offset=first + k * spp)
        want_merged, want_loop, want_rays = oracle.progressive_pass(world, q, want_sum, want_count)
        assert gpu_ctx.stats()["rays"] == want_rays
        assert np.array_equal(got_count, want_count) and float(got_count[0, 0]) == (k + 1) * spp
        assert close_enough(got_sum, want_sum)
        assert close_enough(got_loop, want_loop) and close_enough(got_merged, want_merged)
    # one adaptive pass, selected from the GPU's own planes
    spp_r = max(1, int(np.float32(spp) * np.float32(over)))
    q = params2d(w, h, spp_r, ls, 10, seed=4, sample_offset=first + 2 * spp)
    o_sum, o_count, o_merged = got_sum.copy(), got_count.copy(), got_merged.copy()
    want_n, want_rays = oracle.progressive_resample(world, q, top, mask, o_sum, o_count, o_merged, got_loop)
    gpu_ctx.reset_stats()
    n = prog.resample()
    assert n == want_n and n > 0
    assert gpu_ctx.stats()["rays"] == want_rays
    r_sum, r_count, r_merged, r_loop = prog.read()
    assert np.array_equal(r_count, o_count)                       # the same pixels, the same number of new samples
    assert (r_count > 2 * spp).sum() >= int(np.float32(w * h) * np.float32(top))
    assert set(np.unique(r_count)) <= {2.0 * spp, 2.0 * spp + spp_r}
    assert close_enough(r_sum, o_sum) and close_enough(r_merged, o_merged)
    assert np.array_equal(r_loop, got_loop)                       # the loop image belongs to the last whole pass
    untouched = r_count == 2 * spp
    assert np.array_equal(r_sum[untouched], got_sum[untouched]) and np.array_equal(r_merged[untouched], got_merged[untouched])
    # RenderProgress hand-off: Reinhard RGBA8 of the merged plane
    want_img = oracle.reinhard(r_merged.reshape(-1, 3), 1.5, 255)
    assert np.array_equal(prog.image(1.5).reshape(-1, 4), want_img.reshape(-1, 4))
    prog.close()
    scene.close()


def test_step_follows_the_reference_loop(gpu_ctx):
    world = rt.World2D("sample_world", seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    spp, over, passes = 4, 2.5, 2
    prog = scene.progressive(params2d(48, 36, spp, 2, 10, seed=2), denoiser=(0.05, 1, over, passes))
    s1 = prog.step()                     # the initial global_render: no denoiser passes (librt2d_reverse.rs:436)
    assert s1["loops"] == 1 and s1["recomputed"] == 0 and s1["next_sample"] == spp and s1["rays"] > 0
    s2 = prog.step()                     # loop 1: pass + `passes` adaptive passes (:463-488)
    assert s2["loops"] == 2 and s2["recomputed"] > 0
    assert s2["next_sample"] == 2 * spp + passes * int(spp * over)
    _, count, merged, _ = prog.read(sum=False, loop=False)
    assert count.min() == 2 * spp and count.max() <= 2 * spp + passes * int(spp * over) and count.max() > 2 * spp
    assert np.isfinite(merged).all()
    # without a denoiser a step is a pass
    plain = scene.progressive(params2d(48, 36, spp, 2, 10, seed=2))
    for k in range(3):
        s = plain.step()
        assert s["loops"] == k + 1 and s["recomputed"] == 0 and s["next_sample"] == (k + 1) * spp
    # and equals accumulate + resolve over the same samples
    sum3, count3, merged3, _ = plain.read(loop=False)
    rgb, _ = scene.global_render(params2d(48, 36, 3 * spp, 2, 10, seed=2))
    assert close_enough(merged3, rgb) and float(count3.max()) == 3 * spp == float(count3.min())
    with pytest.raises(rt.RtError):
        plain.resample()                 # no denoiser
    fresh = scene.progressive(params2d(48, 36, spp, 2, 10, seed=2), denoiser=(0.05, 1, 1.0, 1))
    with pytest.raises(rt.RtError):
        fresh.resample()                 # nothing rendered yet
    # a scene stays alive while a progressive renderer is bound to it
    assert _ffi.lib().rt2d_scene_destroy(scene._sc) != 0
    for g in (prog, plain, fresh):
        g.close()
    scene.close()


def test_progressive_arguments(gpu_ctx):
    world = rt.World2D("simple_world")
    scene = rt.Scene2D(gpu_ctx, world)
    p = params2d(16, 16, 2, 1, 3, seed=1)
    for bad in ((-0.1, 1, 1.0, 1), (0.1, -1, 1.0, 1), (0.1, 1, 0.0, 1), (0.1, 1, 1.0, -1), (0.1, 1, float("inf"), 1)):
        with pytest.raises(rt.RtError):
            scene.progressive(p, denoiser=bad)
    g = scene.progressive(params2d(16, 16, 2, 1, 3, seed=1, sample_offset=2**31 - 4))
    g.render_pass()
    with pytest.raises(rt.RtError):
        g.render_pass()                  # the global sample index would pass 2^31
    g.close()
    scene.close()
