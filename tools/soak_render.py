#!/usr/bin/env python3
"""Many small renders of random scenes through the C ABI against the CPU checker (test infrastructure), beyond the
cases the test suite runs: python tools/soak_render.py [renders]
2d: random worlds (every compiled shape of the render kernel), image sizes that do and do not fill whole 8x6 tiles,
1..6 samples, 1..17 lambda-samples, recursion limits 0..10, sample offsets, pixel-to-world scales; the trace_ray count
must be equal, spectra agree to 1e-5 (f32 sums in another order), colours to 2e-3 (see below).
3d: random sphere sets, both kernels, 1..9 samples, bounce limits 0..10; ray counts equal, colours to 1e-5."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from raytracer_amd.api import params2d, params3d
from tests import oracle_ffi
from tests.test_gpu_random_scenes import random_scene3d, random_world2d

n_renders = int(sys.argv[1]) if len(sys.argv) > 1 else 200
oracle = oracle_ffi.Oracle(os.path.join(ROOT, "oracle", "build", "librt_oracle.so"))
ctx = rt.Context(0)
t0 = time.time()
bad = 0
for w in range(n_renders):
    rng = np.random.default_rng(50_000 + w)
    if w % 4 != 3:
        n_circ, n_seg = [(40, 25), (7, 300), (9, 14), (12, 0), (70, 0), (0, 5), (200, 0), (33, 64), (1, 1), (0, 31)][int(rng.integers(0, 10))]
        world = random_world2d(rng, n_circ, n_seg)
        W, H = int(rng.integers(1, 70)), int(rng.integers(1, 50))
        spp, ls = int(rng.integers(1, 7)), int(rng.choice([1, 2, 3, 4, 7, 16, 17]))
        depth = int(rng.choice([0, 1, 2, 3, 10]))
        p = params2d(W, H, spp, ls, depth, seed=int(rng.integers(0, 1 << 30)), sample_offset=int(rng.choice([0, 1, 1000, 123456])),
                     px_to_world=float(rng.choice([1.0, 800.0 / W, 0.37])))
        scene = rt.Scene2D(ctx, world)
        ctx.reset_stats()
        rgb, weight, spec = scene.global_render(p, want_spectrum=True)
        rays = ctx.stats()["rays"]
        scene.close()
        orgb, ow, ospec, orays = oracle.render2d(world, p, threads=8, want_spectrum=True)
        # the spectra are what is compared closely; a colour channel near zero is a difference of XYZ terms a thousand times
        # its size (the sRGB matrix), encoded with a slope of 12.92: one unit in the last place of a spectral sum moves it by
        # 1e-3 of its value (seen on 3 of 40 000 renders at 1e-4)
        same = (rays == orays and np.allclose(spec, ospec, rtol=1e-5, atol=1e-7) and np.allclose(rgb, orgb, rtol=2e-3, atol=1e-5)
                and np.array_equal(weight, ow))
        what = f"2d {n_circ} circles {n_seg} segments {W}x{H} spp {spp} lambda {ls} depth {depth}"
    else:
        n = int(rng.choice([1, 4, 5, 9, 64, 218, 700]))
        W, H = int(rng.integers(1, 70)), int(rng.integers(1, 50))
        use_bvh = bool(rng.integers(0, 2))
        s = random_scene3d(rng, n, bool(rng.integers(0, 2)), use_bvh, w=W, h=H)
        p = params3d(int(rng.integers(1, 10)), int(rng.choice([0, 1, 2, 10])), seed=int(rng.integers(0, 1 << 30)))
        scene = rt.Scene3D(ctx, s)
        ctx.reset_stats()
        rgb = scene.render(p)
        rays = ctx.stats()["rays"]
        scene.close()
        orgb, orays = oracle.render3d(s, p, threads=8)
        same = rays == orays and np.allclose(rgb, orgb, rtol=1e-5, atol=1e-7)
        what = f"3d {n} spheres {'BVH' if use_bvh else 'linear scan'} {W}x{H}"
    if not same:
        bad += 1
        print(f"render {w} ({what}): MISMATCH rays {rays} vs {orays}", flush=True)
    if w % 100 == 99:
        print(f"{w + 1} renders, {bad} with a mismatch, {time.time() - t0:.0f} s", flush=True)
ctx.close()
print(f"done: {n_renders} renders, {bad} with a mismatch")
sys.exit(1 if bad else 0)
