"""3d sphere path tracer on the GPU vs the CPU checker through the C ABI.  The
reference has no BVH: the BVH path must return exactly the linear scan's hits."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params3d

pytestmark = pytest.mark.gpu


def camera_like_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = np.tile([0.0, 2.0, 0.0], (n, 1)) + rng.normal(0, 0.5, (n, 3))
    d = rng.normal(0, 1, (n, 3)) * [1.0, 0.5, 1.0] + [0.0, -0.1, -1.5]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def compare(got, want):
    rgb, fp, ft, nb = got
    orgb, ofp, oft, onb = want
    assert np.array_equal(fp, ofp)
    assert np.array_equal(nb, onb)
    assert np.array_equal(ft, oft)
    assert np.allclose(rgb, orgb, rtol=1e-5, atol=1e-30)


@pytest.mark.parametrize("use_bvh", [False, True])
def test_fed_batch_matches_checker(use_bvh, gpu_ctx, oracle):
    s = rt.Standalone3D(64, 48, use_bvh=use_bvh)
    scene = rt.Scene3D(gpu_ctx, s)
    o, d = camera_like_rays(100_000, seed=3)
    compare(scene.trace_batch(o, d, 10, seed=9), oracle.trace_batch3d(s, o, d, 10, seed=9))
    # rays from inside the grid, all directions
    rng = np.random.default_rng(5)
    o2 = rng.uniform(-10, 8, (50_000, 3)) + [0, 2, -20]
    d2 = rng.normal(0, 1, (50_000, 3))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    compare(scene.trace_batch(o2, d2, 10, seed=1), oracle.trace_batch3d(s, o2, d2, 10, seed=1))
    # axis-parallel rays through sphere centres and tangent to the ground
    o3 = np.array([[0.0, 0.0, 0.0], [3.0, 3.0, 10.0], [-9.0, -9.0, -40.0], [0.0, 0.0, -20.0], [0.0, 1e-9, 0.0]])
    d3 = np.array([[0.0, 0.0, -1.0], [0.0, 0.0, -1.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])
    compare(scene.trace_batch(o3, d3, 10, seed=2), oracle.trace_batch3d(s, o3, d3, 10, seed=2))
    for bl in (0, 1, 3):
        compare(scene.trace_batch(o[:5000], d[:5000], bl, seed=4), oracle.trace_batch3d(s, o[:5000], d[:5000], bl, seed=4))


def test_bvh_equals_linear_scan(gpu_ctx):
    lin = rt.Scene3D(gpu_ctx, rt.Standalone3D(64, 48, use_bvh=False))
    bvh = rt.Scene3D(gpu_ctx, rt.Standalone3D(64, 48, use_bvh=True))
    o, d = camera_like_rays(300_000, seed=11)
    a, b = lin.trace_batch(o, d, 10, seed=3), bvh.trace_batch(o, d, 10, seed=3)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("use_bvh,w,h,spp", [(False, 40, 30, 4), (True, 40, 30, 4), (True, 37, 21, 3)])
def test_render_matches_checker(use_bvh, w, h, spp, gpu_ctx, oracle):
    s = rt.Standalone3D(w, h, use_bvh=use_bvh)
    scene = rt.Scene3D(gpu_ctx, s)
    p = params3d(spp, 10, seed=6)
    gpu_ctx.reset_stats()
    img = scene.render(p)
    st = gpu_ctx.stats()
    oimg, orays = oracle.render3d(s, p)
    assert st["rays"] == orays
    assert np.allclose(img, oimg, rtol=1e-5, atol=1e-7)
    # the reference leaves the 10x10 corner (i<10, j<10) black; j is flipped on output
    assert not img[h - 10:, :10].any() and img[: h - 10].any()


def test_camera_must_be_normalised(gpu_ctx):
    s = rt.Standalone3D(32, 24)
    s.desc.cam_dir[0] = 0.5
    with pytest.raises(rt.RtError):
        rt.Scene3D(gpu_ctx, s)


@pytest.mark.parametrize("use_bvh", [False, True])
def test_two_contexts_render_sample_ranges_that_sum_to_the_single_render(use_bvh, gpu_ctx, oracle):
    """The sample-batch sharding of the 3d image on one device: two contexts (as two ranks would own) accumulate the
    global sample ranges [0, n/2) and [n/2, n) into their own f32[W*H][3] buffers; their sum - what the all-reduce
    produces - is the render of [0, n) by one context (World::compute_pixel is a sum over spp, 3d/src/rt.rs:239-246),
    with the same number of rays, and both match the CPU checker."""
    torch = pytest.importorskip("torch")
    from raytracer_amd.sharding import strong_range
    w, h, n = 48, 36, 7                                    # odd: the two parts differ in length
    s = rt.Standalone3D(w, h, use_bvh=use_bvh)
    parts, rays = [], 0
    for rank in range(2):
        ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        scene = rt.Scene3D(ctx, s)
        lo, hi = strong_range(rank, 2, n)
        buf = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
        scene.accumulate(params3d(hi - lo, 10, seed=6, sample_offset=lo), buf.data_ptr())
        ctx.synchronize()
        rays += ctx.stats()["rays"]
        parts.append(buf)
        scene.close()
        ctx.close()
    total = (parts[0] + parts[1]).cpu().numpy().reshape(h, w, 3)
    scene = rt.Scene3D(gpu_ctx, s)
    gpu_ctx.reset_stats()
    whole = scene.render(params3d(n, 10, seed=6))
    assert gpu_ctx.stats()["rays"] == rays
    scene.close()
    assert whole.any() and np.allclose(total, whole, rtol=1e-5, atol=1e-7)
    oimg, orays = oracle.render3d(s, params3d(n, 10, seed=6))
    assert orays == rays and np.allclose(total, oimg, rtol=1e-5, atol=1e-7)


def test_walk_counters_describe_the_tree_walk(gpu_ctx):
    """rt_stats.bvh_node_visits / sphere_tests: zero node visits for the linear scan, a few per ray for the BVH, and
    every ray tests at least the sphere kept out of the tree (the ground)."""
    for use_bvh in (False, True):
        scene = rt.Scene3D(gpu_ctx, rt.Standalone3D(64, 48, use_bvh=use_bvh))
        gpu_ctx.reset_stats()
        scene.render(params3d(4, 10, seed=1))
        st = gpu_ctx.stats()
        scene.close()
        assert st["rays"] > 0 and st["searches"] == st["rays"]
        if use_bvh:
            assert st["rays"] < st["bvh_node_visits"] < 20 * st["rays"] and st["rays"] <= st["sphere_tests"] < 10 * st["rays"]
        else:
            assert st["bvh_node_visits"] == 0 and st["sphere_tests"] > 0


@pytest.mark.parametrize("use_bvh", [False, True])
def test_directions_with_zero_components(use_bvh, gpu_ctx, oracle):
    """Rays parallel to one or two axes (direction components exactly +-0), from inside and from outside the slabs of
    the tree's boxes along the axes they do not move on: the box test must neither lose a hit nor depend on 0 x inf."""
    s = rt.Standalone3D(64, 48, use_bvh=use_bvh)
    scene = rt.Scene3D(gpu_ctx, s)
    rng = np.random.default_rng(21)
    m = 60_000
    # origins in and around the grid of spheres (x, y in -7.5 .. 7.5 about (0, 2), z in -27.5 .. -12.5) and far from it
    o = rng.uniform(-14.0, 14.0, (m, 3)) * rng.choice([1.0, 1.0, 3.0], (m, 1)) + [0.0, 2.0, -20.0]
    # some origins exactly on the lattice of sphere centres' coordinates: the rays then run through centres
    snap = rng.random(m) < 0.3
    o[snap] = np.round((o[snap] - [0.0, 2.0, -20.0]) / 3.0) * 3.0 + [-1.5, 0.5, -21.5] + rng.choice([0.0, 0.8, -0.8], (int(snap.sum()), 3))
    d = np.zeros((m, 3))
    axis = rng.integers(0, 3, m)
    d[np.arange(m), axis] = rng.choice([-1.0, 1.0], m)
    two = rng.random(m) < 0.4                       # a second non-zero component: the ray lies in an axis plane
    a2 = (axis + rng.integers(1, 3, m)) % 3
    d[np.arange(m)[two], a2[two]] = rng.normal(0, 1, int(two.sum()))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[rng.random(m) < 0.5] *= 1.0                    # keep +0
    neg = rng.random((m, 3)) < 0.5
    d = np.where((d == 0.0) & neg, -0.0, d)          # and -0
    got, want = scene.trace_batch(o, d, 6, seed=12), oracle.trace_batch3d(s, o, d, 6, seed=12)
    compare(got, want)
    assert 0.1 < (want[1] >= 0).mean() < 0.999
    scene.close()
