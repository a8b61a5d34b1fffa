"""Randomly generated scenes through the C ABI against the CPU checker: every material kind on circles and
segments, objects made of several primitives, primitive counts around the kernels' 32-slot words and
256-slot tiles, odd circle counts, degenerate primitives; 3d sphere sets with and without an oversized sphere,
linear scan and BVH.  Same bar as the fixed scenes: primitive ids, bounce counts and hit distances bit-exact,
radiance to 1e-5 relative."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d, params3d

pytestmark = pytest.mark.gpu

EMISSIVE, DIRECTIONAL, DIFFUSE, REFLECTIVE, DIELECTRIC = range(5)


def random_world2d(rng, n_circ, n_seg, span=(800.0, 600.0), degenerate=False):
    """Objects as the reference has them (rt_common.rs:20-29): a circle, or a strip = a run of 1..4 segments;
    scan order = object order, then position in the strip, and a primitive's pid is its place in that order."""
    runs = []
    left = n_seg
    while left > 0:
        k = min(left, int(rng.integers(1, 5)))
        runs.append(k)
        left -= k
    kinds = ["c"] * n_circ + ["s"] * len(runs)
    kinds = [kinds[i] for i in rng.permutation(len(kinds))]
    n_obj = max(1, len(kinds))
    n_sp = 6
    spectra = rng.random((n_sp, 95)).astype(np.float32)
    spectra[0] = 0.0                       # a black absorber: Diffuse short-circuits on it
    spectra[1] *= 0.02                     # a weak Beer-Lambert absorption
    mats = np.zeros((n_obj, 7))
    for i in range(n_obj):
        mats[i] = (rng.integers(0, 5), rng.integers(0, 2), rng.uniform(1.1, 1.8), rng.uniform(0.0, 0.05),
                   rng.uniform(-1.0, 1.0), rng.uniform(0.1, 10.0), rng.integers(0, n_sp))
    cobj, cpid, sobj, spid = [], [], [], []
    pid, run = 0, 0
    for o, kind in enumerate(kinds):
        if kind == "c":
            cobj.append(o); cpid.append(pid); pid += 1
        else:
            for _ in range(runs[run]):
                sobj.append(o); spid.append(pid); pid += 1
            run += 1
    a = {"materials": mats, "spectra": spectra}
    a["circle_cx"] = rng.random(n_circ) * span[0]
    a["circle_cy"] = rng.random(n_circ) * span[1]
    a["circle_r"] = np.exp(rng.uniform(np.log(0.5), np.log(40.0), n_circ))
    a["circle_obj"] = np.array(cobj, np.int32)
    a["circle_pid"] = np.array(cpid, np.int32)
    ax, ay = rng.random(n_seg) * span[0], rng.random(n_seg) * span[1]
    ang, ln = rng.random(n_seg) * 2 * np.pi, rng.uniform(1.0, 150.0, n_seg)
    bx, by = ax + ln * np.cos(ang), ay + ln * np.sin(ang)
    # chain the segments of a strip end to start, as a polyline (shared vertices are hit by two segments at once)
    for k in range(1, n_seg):
        if sobj[k] == sobj[k - 1]:
            dxk, dyk = bx[k] - ax[k], by[k] - ay[k]
            ax[k], ay[k] = bx[k - 1], by[k - 1]
            bx[k], by[k] = ax[k] + dxk, ay[k] + dyk
    if degenerate and n_seg >= 2:
        bx[0], by[0] = ax[0], ay[0]        # zero-length segment: every ray is "parallel" to it
        by[1] = ay[1]                      # axis-parallel
    if degenerate and n_circ >= 2:
        a["circle_r"][0] = 1e-9            # next to a point (the ABI rejects radii below 2^-400)
        a["circle_cx"][1], a["circle_cy"][1] = a["circle_cx"][0], a["circle_cy"][0]  # concentric pair
    # Segment::new: n = (b - a).perp().normalize() = (-sy, sx) * (1 / len)   (rt_common.rs:233-240)
    sx, sy = bx - ax, by - ay
    with np.errstate(divide="ignore", invalid="ignore"):
        rl = 1.0 / np.sqrt((-sy) * (-sy) + sx * sx)
        nx, ny = (-sy) * rl, sx * rl
    bad = ~np.isfinite(nx) | ~np.isfinite(ny)
    nx[bad], ny[bad] = 1.0, 0.0            # the ABI wants finite values; a zero-length segment is never hit
    a.update(seg_ax=ax, seg_ay=ay, seg_bx=bx, seg_by=by, seg_nx=nx, seg_ny=ny)
    a["seg_obj"] = np.array(sobj, np.int32)
    a["seg_pid"] = np.array(spid, np.int32)
    return rt.ArrayWorld2D(a)


def rays2d(world, n, rng, span=(800.0, 600.0)):
    ox, oy = rng.random(n) * span[0], rng.random(n) * span[1]
    ang = rng.random(n) * 2 * np.pi
    a = world._a
    if len(a["circle_cx"]):
        # a third of the rays graze a circle's rim
        k = rng.integers(0, len(a["circle_cx"]), n)
        dist = np.hypot(a["circle_cx"][k] - ox, a["circle_cy"][k] - oy)
        r = a["circle_r"][k]
        phi = np.arctan2(a["circle_cy"][k] - oy, a["circle_cx"][k] - ox)
        alpha = np.arcsin(np.clip(r / np.maximum(dist, r * 1.0001 + 1e-300), -1, 1))
        d = rng.choice([0.0, 1e-15, -1e-15, 1e-12, -1e-12, 1e-8, -1e-8, 1e-4, -1e-4], n)
        graze = phi + rng.choice([-1.0, 1.0], n) * alpha * (1.0 + d)
        ang = np.where(rng.random(n) < 1 / 3, graze, ang)
    if len(a["seg_ax"]):
        # and some start exactly on a segment end point or aim at one
        k = rng.integers(0, len(a["seg_ax"]), n)
        on = rng.random(n) < 0.1
        ox, oy = np.where(on, a["seg_bx"][k], ox), np.where(on, a["seg_by"][k], oy)
        aim = (rng.random(n) < 0.1) & ~on
        ang = np.where(aim, np.arctan2(a["seg_ay"][k] - oy, a["seg_ax"][k] - ox), ang)
    return ox, oy, np.cos(ang), np.sin(ang), rng.integers(0, 95, n).astype(np.int32)


def compare2d(got, want):
    rad, fp, ft, nb = got
    orad, ofp, oft, onb = want
    assert np.array_equal(fp, ofp)
    assert np.array_equal(nb, onb)
    assert np.array_equal(ft, oft)
    assert np.allclose(rad, orad, rtol=1e-5, atol=1e-30)


@pytest.mark.parametrize("n_circ,n_seg", [(1, 0), (0, 1), (31, 1), (33, 0), (200, 57), (255, 2), (256, 0), (1, 256),
                                          (300, 300), (701, 5),
                                          # word boundaries of the filters: all-circle and all-segment words of 32 slots take
                                          # straight-line copies; 63 circles put the slot that pads an odd count inside one
                                          (32, 0), (0, 32), (32, 32), (0, 64), (63, 32), (64, 33), (96, 31), (31, 97)])
def test_random_world_fed_batch(n_circ, n_seg, gpu_ctx, oracle):
    rng = np.random.default_rng(1000 * n_circ + n_seg)
    world = random_world2d(rng, n_circ, n_seg, degenerate=(n_circ + n_seg) > 60)
    scene = rt.Scene2D(gpu_ctx, world)
    rays = rays2d(world, 30_000, rng)
    compare2d(scene.trace_batch(*rays, depth=10, seed=5), oracle.trace_batch2d(world, *rays, depth=10, seed=5))
    scene.close()


@pytest.mark.parametrize("scale,offset", [(1e-3, (0.0, 0.0)), (1.0, (1e7, -3e6)), (1e4, (0.0, 0.0)), (1e-2, (-5e4, 5e4)),
                                          (1.0, (3e11, 3e11))])
def test_world_at_other_scales_and_far_from_the_origin(scale, offset, gpu_ctx, oracle):
    """The f32 filters work in a frame centred on the scene, whatever its place and size in world coordinates; at
    3e11 from the origin an f64 coordinate resolves 6e-5, so circles and segments are visibly quantised - on both sides alike."""
    rng = np.random.default_rng(4242)
    base = random_world2d(rng, 70, 45, degenerate=True)
    ox, oy, dx, dy, li = rays2d(base, 30_000, rng)
    a = {k: np.array(v, copy=True) for k, v in base._a.items()}
    for k in ("circle_cx", "seg_ax", "seg_bx"):
        a[k] = a[k] * scale + offset[0]
    for k in ("circle_cy", "seg_ay", "seg_by"):
        a[k] = a[k] * scale + offset[1]
    a["circle_r"] = a["circle_r"] * scale
    # Segment::new on the moved end points (rt_common.rs:233-240)
    sx, sy = a["seg_bx"] - a["seg_ax"], a["seg_by"] - a["seg_ay"]
    with np.errstate(divide="ignore", invalid="ignore"):
        rl = 1.0 / np.sqrt((-sy) * (-sy) + sx * sx)
        nx, ny = (-sy) * rl, sx * rl
    bad = ~np.isfinite(nx) | ~np.isfinite(ny)
    nx[bad], ny[bad] = 1.0, 0.0
    a["seg_nx"], a["seg_ny"] = nx, ny
    world = rt.ArrayWorld2D(a)
    scene = rt.Scene2D(gpu_ctx, world)
    rays = (ox * scale + offset[0], oy * scale + offset[1], dx, dy, li)
    got, want = scene.trace_batch(*rays, depth=10, seed=9), oracle.trace_batch2d(world, *rays, depth=10, seed=9)
    compare2d(got, want)
    assert (got[1] >= 0).mean() > 0.5     # most rays do hit something: the moved rays meet the moved scene
    scene.close()


# one case per compiled shape of the render kernel: circles and segments in mixed words / words of segments only /
# one word with segments / one word of circles only / circles only
@pytest.mark.parametrize("n_circ,n_seg,w,h", [(40, 25, 24, 16), (7, 300, 19, 9), (9, 14, 21, 13), (12, 0, 23, 11), (70, 0, 17, 12)])
def test_random_world_render(n_circ, n_seg, w, h, gpu_ctx, oracle):
    rng = np.random.default_rng(77 + n_circ)
    world = random_world2d(rng, n_circ, n_seg)
    scene = rt.Scene2D(gpu_ctx, world)
    p = params2d(w, h, 3, 2, 10, seed=21)
    gpu_ctx.reset_stats()
    rgb, weight, spec = scene.global_render(p, want_spectrum=True)
    st = gpu_ctx.stats()
    orgb, ow, ospec, orays = oracle.render2d(world, p, threads=4, want_spectrum=True)
    assert st["rays"] == orays
    assert np.allclose(spec, ospec, rtol=1e-5, atol=1e-7)
    assert np.allclose(rgb, orgb, rtol=2e-4, atol=2e-5)   # a channel near zero is a small difference of large XYZ terms
    scene.close()


def random_scene3d(rng, n, giant, use_bvh, w=32, h=24):
    c = rng.uniform(-20.0, 20.0, (n, 3))
    r = np.exp(rng.uniform(np.log(0.1), np.log(3.0), n))
    if giant and n >= 2:
        c[n - 1], r[n - 1] = (0.0, -5020.0, 0.0), 5000.0
    if n >= 4:
        c[1] = c[0]                        # concentric pair, then an exact duplicate: ties go to the lower index
        c[3], r[3] = c[2], r[2]
    d = np.array([0.2, -0.1, -1.0])
    d /= np.linalg.norm(d)
    up = np.cross(np.cross(d, [0.0, 1.0, 0.0]), d)
    up /= np.linalg.norm(up)
    return rt.ArrayScene3D(c[:, 0], c[:, 1], c[:, 2], r, (1.0, 3.0, 30.0), d, up, 1.5, w, h, use_bvh=use_bvh)


def compare3d(got, want):
    got, want = list(got), list(want)
    for x, y in zip(got[1:], want[1:]):
        assert np.array_equal(x, y)
    assert np.allclose(got[0], want[0], rtol=1e-5, atol=1e-30)


@pytest.mark.parametrize("use_bvh", [False, True])
@pytest.mark.parametrize("n,giant", [(1, False), (2, True), (3, False), (4, False), (5, False), (6, True), (9, False), (64, True), (257, False), (1000, True)])
def test_random_spheres_fed_batch(n, giant, use_bvh, gpu_ctx, oracle):
    rng = np.random.default_rng(10 * n + giant)
    s = random_scene3d(rng, n, giant, use_bvh)
    scene = rt.Scene3D(gpu_ctx, s)
    m = 30_000
    o = rng.uniform(-25.0, 25.0, (m, 3))
    d = rng.normal(0, 1, (m, 3))
    # a third of the rays aim at a sphere's rim
    k = rng.integers(0, n, m)
    c = np.stack([s._a[0][k], s._a[1][k], s._a[2][k]], 1)
    t = rng.normal(0, 1, (m, 3))
    t -= (t * (c - o)).sum(1, keepdims=True) * (c - o) / ((c - o) ** 2).sum(1, keepdims=True)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    rim = c + t * s._a[3][k][:, None] * (1.0 + rng.choice([0.0, 1e-15, -1e-15, 1e-9, -1e-9, 1e-4], m))[:, None]
    d = np.where((rng.random(m) < 1 / 3)[:, None], rim - o, d)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    compare3d(scene.trace_batch(o, d, 10, seed=8), oracle.trace_batch3d(s, o, d, 10, seed=8))
    scene.close()


def timed_grays(ctx, scene, spp=16):
    """Kernel rate of one small render (HIP events around the launch), Grays/s."""
    scene.render(params3d(1, 10, seed=1))
    ctx.reset_stats()
    scene.render(params3d(spp, 10, seed=1))
    st = ctx.stats()
    return st["rays"] / st["kernel_ms"] / 1e6


@pytest.mark.parametrize("n", [5_000, 20_000])
def test_large_scenes_split_by_binned_areas(n, gpu_ctx, oracle, capsys):
    """Scenes above the 4096 spheres up to which a node is split by the full surface-area sweep: the top of the tree is
    split over binned areas.  The BVH must return the linear scan's hits bit for bit (both on the GPU, 40 k rays), and both
    the checker's on a part of the rays; the two kernels' rates at this size are printed for the record."""
    rng = np.random.default_rng(n)
    c = rng.uniform(-20.0, 20.0, (n, 3))
    r = np.exp(rng.uniform(np.log(0.05), np.log(0.5), n))
    c[n - 1], r[n - 1] = (0.0, -5020.0, 0.0), 5000.0
    c[1] = c[0]
    c[3], r[3] = c[2], r[2]
    d0 = np.array([0.2, -0.1, -1.0]) / np.linalg.norm([0.2, -0.1, -1.0])
    up = np.cross(np.cross(d0, [0.0, 1.0, 0.0]), d0)
    up /= np.linalg.norm(up)
    scenes = {b: rt.ArrayScene3D(c[:, 0], c[:, 1], c[:, 2], r, (1.0, 3.0, 30.0), d0, up, 1.5, 640, 480, use_bvh=b) for b in (False, True)}
    lin, bvh = rt.Scene3D(gpu_ctx, scenes[False]), rt.Scene3D(gpu_ctx, scenes[True])
    m = 40_000
    o = rng.uniform(-25.0, 25.0, (m, 3))
    d = rng.normal(0, 1, (m, 3))
    k = rng.integers(0, n, m)
    t = rng.normal(0, 1, (m, 3))
    t -= (t * (c[k] - o)).sum(1, keepdims=True) * (c[k] - o) / ((c[k] - o) ** 2).sum(1, keepdims=True)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    rim = c[k] + t * r[k][:, None] * (1.0 + rng.choice([0.0, 1e-15, -1e-15, 1e-9, -1e-9, 1e-4], m))[:, None]
    d = np.where((rng.random(m) < 1 / 3)[:, None], rim - o, d)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = lin.trace_batch(o, d, 10, seed=8), bvh.trace_batch(o, d, 10, seed=8)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    compare3d((x[:4000] for x in b), oracle.trace_batch3d(scenes[True], o[:4000], d[:4000], 10, seed=8))
    assert (b[1] >= 0).mean() > 0.5
    rates = {name: timed_grays(gpu_ctx, sc) for name, sc in (("linear scan", lin), ("BVH", bvh))}
    with capsys.disabled():
        print(f"\n[3d, {n} spheres, 640x480, 16 spp] linear scan {rates['linear scan']:.2f} Grays/s, BVH {rates['BVH']:.2f} Grays/s")
    lin.close()
    bvh.close()


@pytest.mark.parametrize("kind", ["coincident", "collinear", "onion"])
def test_degenerate_scenes_build_and_match(kind, gpu_ctx, oracle):
    """Inputs that drive a surface-area tree deep: thousands of spheres with ONE centre (every split position costs
    the same), centres on a line, and spheres nested like onion skins (each split peels one off).  The builder must
    stay within the traversal stack - balanced ties, median splits as the last resort - and the walk must return the
    linear scan's hits, lowest index first among equals."""
    rng = np.random.default_rng(5)
    n = 6000
    if kind == "coincident":
        c = np.tile([1.0, 2.0, -3.0], (n, 1))
        r = np.full(n, 1.5)
        r[n // 2:] = 0.75                                   # two radii: exact ties within each half
    elif kind == "collinear":
        c = np.stack([np.linspace(-30.0, 30.0, n), np.full(n, 2.0), np.full(n, -3.0)], 1)
        r = np.full(n, 0.2)
    else:
        c = np.tile([0.0, 0.0, 0.0], (n, 1)) + rng.normal(0, 1e-3, (n, 3))
        r = np.exp(np.linspace(np.log(0.01), np.log(40.0), n))
    d0 = np.array([0.0, 0.0, -1.0])
    scenes = {b: rt.ArrayScene3D(c[:, 0], c[:, 1], c[:, 2], r, (0.0, 0.0, 60.0), d0, (0.0, 1.0, 0.0), 1.5, 32, 24, use_bvh=b) for b in (False, True)}
    lin, bvh = rt.Scene3D(gpu_ctx, scenes[False]), rt.Scene3D(gpu_ctx, scenes[True])
    m = 20_000
    o = rng.uniform(-45.0, 45.0, (m, 3))
    aim = c[rng.integers(0, n, m)] + rng.normal(0, 1, (m, 3)) * r[rng.integers(0, n, m)][:, None]
    d = np.where((rng.random(m) < 0.7)[:, None], aim - o, rng.normal(0, 1, (m, 3)))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = lin.trace_batch(o, d, 4, seed=2), bvh.trace_batch(o, d, 4, seed=2)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    compare3d((x[:3000] for x in b), oracle.trace_batch3d(scenes[True], o[:3000], d[:3000], 4, seed=2))
    assert (b[1] >= 0).mean() > 0.3
    lin.close()
    bvh.close()


@pytest.mark.parametrize("use_bvh", [False, True])
@pytest.mark.parametrize("reach", [3e2, 1e4, 1e6])
def test_rays_from_far_away_aimed_at_rims(reach, use_bvh, gpu_ctx, oracle):
    """Origins far outside the scene: the linear scan's f32 filter widens its pad with the origin's distance, the BVH's
    box test with the origin's magnitude; neither may lose a grazing hit or invent one."""
    rng = np.random.default_rng(int(reach))
    n = 150
    s = random_scene3d(rng, n, True, use_bvh)
    scene = rt.Scene3D(gpu_ctx, s)
    m = 20_000
    o = rng.normal(0, 1, (m, 3))
    o *= (reach * rng.uniform(0.3, 1.0, m) / np.linalg.norm(o, axis=1))[:, None]
    k = rng.integers(0, n - 1, m)                      # not the giant sphere
    c = np.stack([s._a[0][k], s._a[1][k], s._a[2][k]], 1)
    t = rng.normal(0, 1, (m, 3))
    t -= (t * (c - o)).sum(1, keepdims=True) * (c - o) / ((c - o) ** 2).sum(1, keepdims=True)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    rim = c + t * s._a[3][k][:, None] * (1.0 + rng.choice([0.0, 1e-13, -1e-13, 1e-9, -1e-9, 1e-5, -1e-5, -0.5], m))[:, None]
    d = rim - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    got, want = scene.trace_batch(o, d, 6, seed=3), oracle.trace_batch3d(s, o, d, 6, seed=3)
    compare3d(got, want)
    assert (want[1] >= 0).mean() > 0.3                 # a good share of the rays does hit something
    scene.close()


@pytest.mark.parametrize("use_bvh", [False, True])
def test_random_spheres_render(use_bvh, gpu_ctx, oracle):
    rng = np.random.default_rng(4)
    s = random_scene3d(rng, 300, True, use_bvh, w=36, h=20)
    scene = rt.Scene3D(gpu_ctx, s)
    p = params3d(3, 10, seed=2)
    gpu_ctx.reset_stats()
    img = scene.render(p)
    oimg, orays = oracle.render3d(s, p)
    assert gpu_ctx.stats()["rays"] == orays
    assert np.allclose(img, oimg, rtol=1e-5, atol=1e-7)
    scene.close()


def test_empty_worlds(gpu_ctx, oracle):
    """No primitives at all: every ray misses (2d: radiance 0; 3d: the background), on both sides."""
    z = np.zeros(0)
    world = rt.ArrayWorld2D(dict(circle_cx=z, circle_cy=z, circle_r=z, circle_obj=z, circle_pid=z, seg_ax=z, seg_ay=z, seg_bx=z,
                                 seg_by=z, seg_nx=z, seg_ny=z, seg_obj=z, seg_pid=z, materials=np.zeros((0, 7)),
                                 spectra=np.zeros((0, 95), np.float32)))
    scene = rt.Scene2D(gpu_ctx, world)
    p = params2d(20, 12, 3, 2, 10, seed=1)
    gpu_ctx.reset_stats()
    rgb, weight, spec = scene.global_render(p, want_spectrum=True)
    st = gpu_ctx.stats()
    assert not spec.any() and not rgb.any() and np.all(weight == 3)
    assert st["rays"] == 20 * 12 * 3 * 2 == st["samples"]          # one hit search per sample, all misses
    orgb, _, ospec, orays = oracle.render2d(world, p)
    assert not ospec.any() and orays == st["rays"]
    rays = (np.array([1.0, 2.0]), np.array([3.0, 4.0]), np.array([1.0, 0.0]), np.array([0.0, 1.0]), np.zeros(2, np.int32))
    rad, fp, ft, nb = scene.trace_batch(*rays, depth=4)
    assert not rad.any() and np.all(fp == -1) and np.all(nb == 1)
    rgbf, wf, n = scene.forward_render(scene.forward_params(16, 16, 10, 5))   # no emitters: nothing to draw
    assert n == 0 and not rgbf.any() and np.all(wf == 1.0)
    scene.close()
    for use_bvh in (False, True):
        s3 = rt.ArrayScene3D(z, z, z, z, (0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 1.0, 24, 16, use_bvh=use_bvh)
        sc3 = rt.Scene3D(gpu_ctx, s3)
        img = sc3.render(params3d(2, 5, seed=1))
        oimg, _ = oracle.render3d(s3, params3d(2, 5, seed=1))
        assert np.allclose(img, oimg, rtol=1e-6, atol=1e-7) and img.any()
        sc3.close()
