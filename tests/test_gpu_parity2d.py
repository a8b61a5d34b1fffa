"""GPU vs CPU checker on identical fed ray / wavelength batches, through the C ABI
(rt2d_trace_batch).  Bar: first-hit primitive ids and bounce counts bit-exact,
hit distance and radiance within 1e-5 relative (the kernels are designed to
reproduce the distance bit for bit, which is asserted too)."""
import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu

SCENES = list(rt.World2D.NAMES)


def random_rays(n, seed, span=(800.0, 600.0)):
    rng = np.random.default_rng(seed)
    ox, oy = rng.random(n) * span[0], rng.random(n) * span[1]
    ang = rng.random(n) * 2 * np.pi
    return ox, oy, np.cos(ang), np.sin(ang), rng.integers(0, 95, n).astype(np.int32)


def compare(got, want, exact_t=True):
    rad, fp, ft, nb = got
    orad, ofp, oft, onb = want
    assert np.array_equal(fp, ofp)
    assert np.array_equal(nb, onb)
    if exact_t:
        assert np.array_equal(ft, oft)
    assert np.allclose(ft, oft, rtol=1e-5, atol=0.0)
    assert np.allclose(rad, orad, rtol=1e-5, atol=1e-30)


@pytest.mark.parametrize("name", SCENES)
def test_fed_batch_matches_checker(name, gpu_ctx, oracle):
    world = rt.World2D(name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    rays = random_rays(200_000, seed=SCENES.index(name))
    got = scene.trace_batch(*rays, depth=10, seed=11)
    want = oracle.trace_batch2d(world, *rays, depth=10, seed=11)
    compare(got, want)
    if name != "simple_world":
        assert want[3].max() == 10  # the depth limit is exercised
    scene.close()


@pytest.mark.parametrize("depth", [0, 1, 2, 5])
def test_depth_limits(depth, gpu_ctx, oracle):
    world = rt.World2D("sample_world")
    scene = rt.Scene2D(gpu_ctx, world)
    rays = random_rays(20_000, seed=depth)
    compare(scene.trace_batch(*rays, depth=depth, seed=1), oracle.trace_batch2d(world, *rays, depth=depth, seed=1))


def test_ragged_and_tiny_batches(gpu_ctx, oracle):
    world = rt.World2D("cornell_box")
    scene = rt.Scene2D(gpu_ctx, world)
    for n in (1, 63, 64, 65, 255, 257, 1000):
        rays = random_rays(n, seed=n)
        compare(scene.trace_batch(*rays, depth=10, seed=2), oracle.trace_batch2d(world, *rays, depth=10, seed=2))
    # empty batch is a no-op
    e = np.zeros(0)
    out = scene.trace_batch(e, e, e, e, np.zeros(0, np.int32), depth=10)
    assert all(len(a) == 0 for a in out)


def test_rays_starting_on_surfaces_and_vertices(gpu_ctx, oracle):
    """Origins exactly on primitives, at shared strip vertices and circle centres; axis-parallel directions."""
    world = rt.World2D("sample_world")
    a = world.arrays()
    scene = rt.Scene2D(gpu_ctx, world)
    ox = np.concatenate([a["seg_ax"], a["seg_bx"], a["circle_cx"], a["circle_cx"] + a["circle_r"]])
    oy = np.concatenate([a["seg_ay"], a["seg_by"], a["circle_cy"], a["circle_cy"]])
    dirs = [(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0), (np.sqrt(0.5), np.sqrt(0.5)), (-0.6, 0.8)]
    for dx, dy in dirs:
        n = len(ox)
        li = (np.arange(n) % 95).astype(np.int32)
        args = (ox, oy, np.full(n, dx), np.full(n, dy), li)
        compare(scene.trace_batch(*args, depth=10, seed=4), oracle.trace_batch2d(world, *args, depth=10, seed=4))


def test_unnormalised_directions(gpu_ctx, oracle):
    world = rt.World2D("complex_world", seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    ox, oy, dx, dy, li = random_rays(50_000, seed=9)
    s = np.random.default_rng(1).uniform(0.1, 30.0, len(ox))
    # depth 1: the reference's refract/reflect assume unit directions only after the first hit
    compare(scene.trace_batch(ox, oy, dx * s, dy * s, li, depth=1, seed=0),
            oracle.trace_batch2d(world, ox, oy, dx * s, dy * s, li, depth=1, seed=0))


def test_all_wavelengths_dispersion(gpu_ctx, oracle):
    """Same geometric rays at every wavelength bin: Cauchy dispersion must split the paths identically."""
    world = rt.World2D("cornell_box")
    scene = rt.Scene2D(gpu_ctx, world)
    ox, oy, dx, dy, _ = random_rays(2_000, seed=3)
    for li in (0, 1, 47, 93, 94):
        lam = np.full(len(ox), li, np.int32)
        compare(scene.trace_batch(ox, oy, dx, dy, lam, depth=10, seed=6),
                oracle.trace_batch2d(world, ox, oy, dx, dy, lam, depth=10, seed=6))


def test_invalid_arguments_are_rejected(gpu_ctx):
    world = rt.World2D("simple_world")
    scene = rt.Scene2D(gpu_ctx, world)
    one = np.ones(4)
    with pytest.raises(rt.RtError):
        scene.trace_batch(one * np.nan, one, one, one, np.zeros(4, np.int32), depth=3)
    with pytest.raises(rt.RtError):
        scene.trace_batch(one, one, one, one, np.full(4, 95, np.int32), depth=3)
    with pytest.raises(rt.RtError):
        scene.trace_batch(one, one, one, one, np.zeros(4, np.int32), depth=-1)


def _grazing_rays(world, n, seed, span=(800.0, 600.0), spread=1.0):
    """Rays aimed at the rim of circles (discriminant ~ 0) and rays leaving from circle rims (d.(o-c) ~ 0 or
    origin on the boundary): the cases the conservative first-phase filter must never drop."""
    a = world.arrays()
    cx, cy, r = a["circle_cx"], a["circle_cy"], a["circle_r"]
    rng = np.random.default_rng(seed)
    k = rng.integers(0, len(cx), n)
    eps = np.concatenate([[0.0], 10.0 ** np.arange(-16, -2.0)])
    d = rng.choice(np.concatenate([eps, -eps]), n)
    half = n // 2
    # (1) tangent lines from random origins
    ox = (rng.random(n) - 0.5) * spread * span[0] + span[0] / 2
    oy = (rng.random(n) - 0.5) * spread * span[1] + span[1] / 2
    dist = np.hypot(cx[k] - ox, cy[k] - oy)
    phi = np.arctan2(cy[k] - oy, cx[k] - ox)
    alpha = np.arcsin(np.clip(r[k] / np.maximum(dist, r[k] * 1.0001), -1, 1))
    ang = phi + rng.choice([-1.0, 1.0], n) * alpha * (1.0 + d)
    # (2) origins on (or a hair off) a rim, leaving radially, tangentially or a hair inward
    th = rng.random(n) * 2 * np.pi
    rim_x = cx[k] + r[k] * (1.0 + d) * np.cos(th)
    rim_y = cy[k] + r[k] * (1.0 + d) * np.sin(th)
    rim_ang = th + rng.choice([0.0, np.pi / 2, -np.pi / 2, np.pi], n) + rng.choice(np.concatenate([eps, -eps]), n)
    sel = np.arange(n) >= half
    ox, oy, ang = np.where(sel, rim_x, ox), np.where(sel, rim_y, oy), np.where(sel, rim_ang, ang)
    return ox, oy, np.cos(ang), np.sin(ang), rng.integers(0, 95, n).astype(np.int32)


@pytest.mark.parametrize("name", ["complex_world", "sample_world", "cornell_box"])
def test_grazing_and_rim_rays(name, gpu_ctx, oracle):
    world = rt.World2D(name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    rays = _grazing_rays(world, 100_000, seed=5)
    compare(scene.trace_batch(*rays, depth=10, seed=3), oracle.trace_batch2d(world, *rays, depth=10, seed=3))
    scene.close()


@pytest.mark.parametrize("spread", [3.0, 7.9, 40.0, 1e6])
def test_origins_far_outside_the_scene(spread, gpu_ctx, oracle):
    """First origins up to and beyond the range the first-phase filter is built for (those take the unfiltered path)."""
    world = rt.World2D("complex_world", seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    n = 20_000
    rays = _grazing_rays(world, n, seed=8, spread=spread)
    sel = np.arange(n) < n // 2   # only the far-origin half; the rim half is covered above
    rays = tuple(x[sel] for x in rays)
    compare(scene.trace_batch(*rays, depth=10, seed=3), oracle.trace_batch2d(world, *rays, depth=10, seed=3))
    scene.close()


def test_scene_far_from_the_coordinate_origin(gpu_ctx, oracle):
    """The same circles moved to (1e7, -3e6): hit tests lose absolute precision there, and must lose it identically."""
    base = rt.World2D("complex_world", seed=1).arrays()
    a = dict(base)
    a["circle_cx"] = base["circle_cx"] + 1.0e7
    a["circle_cy"] = base["circle_cy"] - 3.0e6
    world = rt.ArrayWorld2D(a)
    scene = rt.Scene2D(gpu_ctx, world)
    ox, oy, dx, dy, li = _grazing_rays(rt.World2D("complex_world", seed=1), 50_000, seed=2)
    args = (ox + 1.0e7, oy - 3.0e6, dx, dy, li)
    compare(scene.trace_batch(*args, depth=10, seed=3), oracle.trace_batch2d(world, *args, depth=10, seed=3))
    scene.close()


@pytest.mark.timeout(60)
def test_a_hit_point_on_a_circles_centre_ends_the_path(gpu_ctx, oracle):
    """Diffuse circles far smaller than the spacing of f64 at their coordinates (radius 1e-9 at 1e9): a ray through
    the centre hits at the centre itself, the normal is NaN, and no direction of the reference's rejection loop
    (rt_common.rs:376-391) would ever be accepted.  The kernels end such a path (as the checker does) instead of
    looping; the rays that miss, and the bounces off an ordinary circle beside it, are unaffected."""
    spectra = np.stack([np.full(95, 0.5, np.float32), np.full(95, 2.0, np.float32)])
    mats = np.array([[2, 0, 1.5, 0.0, 0.0, 1.0, 0], [2, 0, 1.5, 0.0, 0.0, 1.0, 0], [0, 0, 1.0, 0.0, 0.0, 1.0, 1]], float)
    z, zi = np.zeros(0), np.zeros(0, np.int32)
    a = dict(materials=mats, spectra=spectra,
             circle_cx=np.array([1e9, 1e9 + 40.0, 1e9 - 200.0]), circle_cy=np.array([1e9, 1e9 + 3.0, 1e9]),
             circle_r=np.array([1e-9, 5.0, 100.0]), circle_obj=np.array([0, 1, 2], np.int32), circle_pid=np.array([0, 1, 2], np.int32),
             seg_ax=z, seg_ay=z, seg_bx=z, seg_by=z, seg_nx=z, seg_ny=z, seg_obj=zi, seg_pid=zi)
    world = rt.ArrayWorld2D(a)
    scene = rt.Scene2D(gpu_ctx, world)
    n = 4096
    rng = np.random.default_rng(5)
    # from points on a circle of radius 10 around the tiny circle, straight at its centre (half of them), or anywhere
    phi = rng.random(n) * 2 * np.pi
    ox, oy = 1e9 + 10.0 * np.cos(phi), 1e9 + 10.0 * np.sin(phi)
    aim = rng.random(n) < 0.5
    ang = np.where(aim, np.arctan2(1e9 - oy, 1e9 - ox), rng.random(n) * 2 * np.pi)
    rays = (ox, oy, np.cos(ang), np.sin(ang), rng.integers(0, 95, n).astype(np.int32))
    got, want = scene.trace_batch(*rays, depth=10, seed=2), oracle.trace_batch2d(world, *rays, depth=10, seed=2)
    compare(got, want)
    dead = (got[1] == 0) & (got[3] == 1)
    assert dead.sum() > 10 and (got[0][dead] == 0.0).all()      # paths that ended on the tiny circle
    assert (got[0] > 0).sum() > 100                              # and others that found the light
    scene.close()


def test_context_outlives_its_scenes():
    from raytracer_amd import _ffi
    ctx = rt.Context(0)
    scene = rt.Scene2D(ctx, rt.World2D("simple_world"))
    assert _ffi.lib().rt_ctx_destroy(ctx._ctx) == -1  # RT_E_INVALID: a scene is still alive
    assert b"scene" in _ffi.lib().rt_last_error()
    ctx.close()            # closes the scene first, then the context
    assert scene._sc is None and ctx._ctx is None
