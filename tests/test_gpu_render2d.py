"""GPU renders vs the CPU checker through the C ABI: whole images
(rt2d_render = global_render), pixel lists (rt2d_compute_pixels = compute_pixel /
render_column) and the accumulate/resolve pair used for sample-batch sharding.

Paths are bit-identical on both sides (same Philox streams, same f64
arithmetic); what differs is the f32 summation order of a pixel's samples (f32
atomics on the GPU) and the forward vs recursive order of the f32 factor
product, so spectra are compared with rtol 1e-4 + atol 1e-6 and ray counts
exactly."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name,w,h,spp,ls", [
    ("sample_world", 80, 60, 4, 2),
    ("complex_world", 48, 36, 6, 2),
    ("cornell_box", 40, 30, 3, 5),          # lambda_samples > 2: second Philox call for wavelength draws
    ("cornell_box_absorption", 37, 29, 2, 1),  # ragged tiles (not multiples of 8)
    ("spectrum_world", 64, 8, 3, 2),
    ("cornell_box", 24, 18, 2, 21),         # more lambda-samples than the ray pool's 16-entry wavelength table
    ("simple_world", 9, 5, 70, 1),          # a single partial tile, many samples
])
def test_global_render_matches_checker(name, w, h, spp, ls, gpu_ctx, oracle):
    world = rt.World2D(name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    p = params2d(w, h, spp, ls, 10, seed=21)
    gpu_ctx.reset_stats()
    rgb, weight, spec = scene.global_render(p, want_spectrum=True)
    st = gpu_ctx.stats()
    orgb, oweight, ospec, orays = oracle.render2d(world, p)
    assert st["rays"] == orays
    assert st["samples"] == w * h * spp * ls
    assert np.allclose(spec, ospec, **TOL)
    assert np.array_equal(weight, oweight) and weight[0, 0] == spp
    assert np.allclose(rgb, orgb, rtol=2e-4, atol=2e-5)
    scene.close()


def test_reference_resolution_identity_scale(gpu_ctx, oracle):
    """At px_to_world = 1 a pixel is a world unit, as in the reference (800x600 world)."""
    world = rt.World2D("sample_world")
    scene = rt.Scene2D(gpu_ctx, world)
    p = params2d(800, 600, 1, 1, 10, seed=1, px_to_world=1.0)
    px = np.arange(0, 800, 37, dtype=np.int32)
    py = (np.arange(len(px)) * 29 % 600).astype(np.int32)
    got = scene.compute_pixels(p, px, py)
    for k in range(len(px)):
        want, _ = oracle.compute_pixel2d(world, p, px[k], py[k])
        assert np.allclose(got[k], want, **TOL)


def test_compute_pixels_and_render_column(gpu_ctx, oracle):
    world = rt.World2D("cornell_box")
    scene = rt.Scene2D(gpu_ctx, world)
    p = params2d(100, 75, 5, 3, 10, seed=8)
    col = scene.render_column(p, 40)
    assert col.shape == (75, 95)
    for j in (0, 1, 37, 74):
        want, _ = oracle.compute_pixel2d(world, p, 40, j)
        assert np.allclose(col[j], want, **TOL)
    # arbitrary list incl. pixels outside the image (the denoiser's mask may step outside) and an empty list
    px = np.array([0, 99, -3, 120, 50], np.int32)
    py = np.array([0, 74, 10, -7, 200], np.int32)
    got = scene.compute_pixels(p, px, py)
    for k in range(len(px)):
        want, _ = oracle.compute_pixel2d(world, p, px[k], py[k])
        assert np.allclose(got[k], want, **TOL)
    assert scene.compute_pixels(p, [], []).shape == (0, 95)
    one = scene.compute_pixel(p, 50, 40)
    assert np.allclose(one, oracle.compute_pixel2d(world, p, 50, 40)[0], **TOL)


def test_sample_ranges_add_up(gpu_ctx):
    """Sharding by sample batch: accumulating [0,a) and [a,n) equals one pass over [0,n) (linearity)."""
    torch = pytest.importorskip("torch")
    world = rt.World2D("complex_world", seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    w, h, n = 64, 40, 8
    gpu_ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        whole = torch.zeros(h * w * 95, device="cuda")
        parts = torch.zeros_like(whole)
        scene.accumulate(params2d(w, h, n, 2, 10, seed=5), whole.data_ptr())
        scene.accumulate(params2d(w, h, 3, 2, 10, seed=5, sample_offset=0), parts.data_ptr())
        scene.accumulate(params2d(w, h, 5, 2, 10, seed=5, sample_offset=3), parts.data_ptr())
        torch.cuda.synchronize()
        assert torch.allclose(whole, parts, rtol=1e-4, atol=1e-6)
        assert float(whole.sum()) > 0
        rgb_a, wt = scene.resolve(whole.data_ptr(), w, h, n)
        rgb_b, _ = scene.resolve(parts.data_ptr(), w, h, n)
        assert np.allclose(rgb_a, rgb_b, rtol=2e-4, atol=2e-5) and wt[0, 0] == n
    finally:
        gpu_ctx.set_stream(None)


def test_render_is_reproducible(gpu_ctx):
    world = rt.World2D("sample_world")
    scene = rt.Scene2D(gpu_ctx, world)
    p = params2d(96, 72, 4, 2, 10, seed=77)
    a = scene.global_render(p, want_spectrum=True)[2]
    b = scene.global_render(p, want_spectrum=True)[2]
    assert np.allclose(a, b, rtol=1e-5, atol=1e-7)  # same paths; only the f32 atomic order may move the last bit
    c = scene.global_render(params2d(96, 72, 4, 2, 10, seed=78), want_spectrum=True)[2]
    assert not np.allclose(a, c, rtol=1e-3, atol=1e-5)


def test_resolve_and_tonemap_match_checker(gpu_ctx, oracle):
    rng = np.random.default_rng(2)
    spec = (rng.random((50, 95)) * rng.random((50, 1)) * 2).astype(np.float32)
    spec[0] = 0
    world = rt.World2D("simple_world")
    scene = rt.Scene2D(gpu_ctx, world)
    torch = pytest.importorskip("torch")
    d = torch.from_numpy(spec * 4).cuda()  # accumulated over 4 samples
    rgb, _ = scene.resolve(d.data_ptr(), 50, 1, 4)
    want = np.stack([oracle.resolve_pixel(s) for s in spec])
    assert np.allclose(rgb.reshape(50, 3), want, rtol=1e-5, atol=1e-6)
    vals = np.concatenate([want, [[-1.0, 0.0, 1e9], [np.nan, np.inf, 0.5], [1.0, 254.9 / 0.1, 3.0]]]).astype(np.float32)
    for param, alpha in ((1.0, 255), (0.5, 254)):
        assert np.array_equal(gpu_ctx.tonemap_reinhard(vals, param, alpha), oracle.reinhard(vals, param, alpha))


def test_two_step_resolve_for_a_sum_over_gpus(gpu_ctx, oracle):
    """rt2d_project_xyz + (sum over ranks) + rt2d_resolve_xyz: two partial sums of one image give the image."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    W, H = 37, 5
    parts = [(rng.random((H * W, 95)) * rng.random((H * W, 1)) * 3).astype(np.float32) for _ in range(2)]
    world = rt.World2D("simple_world")
    scene = rt.Scene2D(gpu_ctx, world)
    total = torch.zeros(H * W * 3, device="cuda")
    for part in parts:                                     # what each rank does before the all-reduce
        d = torch.from_numpy(part).cuda()
        xyz = torch.empty(H * W * 3, device="cuda")
        scene.project_xyz(d.data_ptr(), W, H, xyz.data_ptr())
        gpu_ctx.synchronize()
        assert np.array_equal(xyz.cpu().numpy().reshape(-1, 3), oracle.project_xyz(part))   # same order of operations
        total += xyz
    rgb, weight = scene.resolve_xyz(total.data_ptr(), W, H, 12)
    # powf of the sRGB transfer function is the platform's on both sides: last-bit differences
    assert np.allclose(rgb.reshape(-1, 3), oracle.encode_xyz(total.cpu().numpy(), 12), rtol=1e-6, atol=1e-7)
    assert np.all(weight == 12.0)
    # against the one-step resolve of the summed spectra only the rounding order differs
    both = torch.from_numpy(parts[0] + parts[1]).cuda()
    one, _ = scene.resolve(both.data_ptr(), W, H, 12)
    assert np.allclose(rgb, one, rtol=1e-5, atol=1e-6)
    with pytest.raises(rt.RtError):
        scene.resolve_xyz(total.data_ptr(), W, H, 0)
    with pytest.raises(rt.RtError):
        scene.project_xyz(0, W, H, total.data_ptr())


def test_zero_depth_and_bad_params(gpu_ctx):
    world = rt.World2D("simple_world")
    scene = rt.Scene2D(gpu_ctx, world)
    rgb, _, spec = scene.global_render(params2d(16, 16, 2, 1, 0, seed=1), want_spectrum=True)
    assert not spec.any()
    for bad in (dict(width=0), dict(spp=0), dict(lambda_samples=0), dict(recursion_limit=-1)):
        kw = dict(width=16, height=16, spp=1, lambda_samples=1, recursion_limit=3)
        kw.update(bad)
        with pytest.raises(rt.RtError):
            scene.global_render(params2d(**kw))


def test_lambda_samples_share_their_primary_search_and_nothing_else(gpu_ctx, oracle):
    """The lambda-samples of a (pixel, sample) pair trace ONE primary ray (librt2d_reverse.rs:90-101); the kernel
    searches its nearest hit once.  rays (trace_ray invocations) must still equal the checker's count, which searches
    every invocation; searches = rays - (lambda_samples - 1) x pairs whose ray was traced; spectra as the checker's."""
    world = rt.World2D("complex_world", seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    for ls in (1, 2, 5, 19):
        p = params2d(48, 36, 3, ls, 10, seed=8)
        gpu_ctx.reset_stats()
        _, _, spec = scene.global_render(p, want_spectrum=True)
        st = gpu_ctx.stats()
        _, _, ospec, orays = oracle.render2d(world, p, threads=4)
        assert st["rays"] == orays
        assert st["searches"] == orays - (ls - 1) * 48 * 36 * 3
        assert np.allclose(spec, ospec, rtol=1e-4, atol=1e-6)
    scene.close()
