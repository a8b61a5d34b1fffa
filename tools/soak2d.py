#!/usr/bin/env python3
"""Many random 2d worlds through the C ABI against the CPU checker (test infrastructure), beyond the handful of seeds
the test suite runs: python tools/soak2d.py [worlds] [rays per world]
Same bar as tests/test_gpu_random_scenes.py - first-hit ids, bounce counts and hit distances bit for bit, radiance to
1e-5 - with extra rays that pass a segment's end points at relative distances of 1e-15 .. 1e-3 on either side (what
the segment filter's margin is for), at several scales and places of the world."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from tests import oracle_ffi
from tests.test_gpu_random_scenes import random_world2d, rays2d

n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000
oracle = oracle_ffi.Oracle(os.path.join(ROOT, "oracle", "build", "librt_oracle.so"))
ctx = rt.Context(0)
t0 = time.time()
bad = 0
hits = 0
for w in range(n_worlds):
    rng = np.random.default_rng(90_000 + w)
    n_circ = int(rng.choice([0, 1, 5, 31, 40, 64, 120, 200]))
    n_seg = int(rng.choice([1, 2, 7, 32, 33, 64, 115, 300]))
    base = random_world2d(rng, n_circ, n_seg, degenerate=bool(w % 3 == 0))
    ox, oy, dx, dy, li = rays2d(base, n_rays, rng)
    a = base._a
    # a quarter of the rays pass an end point of a segment at a tiny signed distance
    k = rng.integers(0, n_seg, n_rays)
    end = rng.random(n_rays) < 0.5
    ex, ey = np.where(end, a["seg_ax"][k], a["seg_bx"][k]), np.where(end, a["seg_ay"][k], a["seg_by"][k])
    sx, sy = a["seg_bx"][k] - a["seg_ax"][k], a["seg_by"][k] - a["seg_ay"][k]
    rel = rng.choice([0.0, 1e-15, -1e-15, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6, 1e-3, -1e-3], n_rays)
    tx, ty = ex + sx * rel, ey + sy * rel
    near = rng.random(n_rays) < 0.25
    ang = np.arctan2(ty - oy, tx - ox)
    dx, dy = np.where(near, np.cos(ang), dx), np.where(near, np.sin(ang), dy)
    scale, off = [(1.0, (0.0, 0.0)), (1e-3, (0.0, 0.0)), (1e3, (2e6, -1e6)), (1.0, (1e9, 1e9))][w % 4]
    b = {key: np.array(v, copy=True) for key, v in a.items()}
    for key in ("circle_cx", "seg_ax", "seg_bx"):
        b[key] = b[key] * scale + off[0]
    for key in ("circle_cy", "seg_ay", "seg_by"):
        b[key] = b[key] * scale + off[1]
    b["circle_r"] = b["circle_r"] * scale
    ssx, ssy = b["seg_bx"] - b["seg_ax"], b["seg_by"] - b["seg_ay"]
    with np.errstate(divide="ignore", invalid="ignore"):
        rl = 1.0 / np.sqrt((-ssy) * (-ssy) + ssx * ssx)
        nx, ny = (-ssy) * rl, ssx * rl
    deg = ~np.isfinite(nx) | ~np.isfinite(ny)
    nx[deg], ny[deg] = 1.0, 0.0
    b["seg_nx"], b["seg_ny"] = nx, ny
    world = rt.ArrayWorld2D(b)
    scene = rt.Scene2D(ctx, world)
    rays = (ox * scale + off[0], oy * scale + off[1], dx, dy, li)
    got = scene.trace_batch(*rays, depth=10, seed=w)
    want = oracle.trace_batch2d(world, *rays, depth=10, seed=w)
    scene.close()
    same = (np.array_equal(got[1], want[1]) and np.array_equal(got[3], want[3]) and np.array_equal(got[2], want[2])
            and np.allclose(got[0], want[0], rtol=1e-5, atol=1e-30))
    hits += int((want[1] >= 0).sum())
    if not same:
        bad += 1
        print(f"world {w} ({n_circ} circles, {n_seg} segments, scale {scale}, offset {off}): MISMATCH "
              f"ids {int((got[1] != want[1]).sum())} bounces {int((got[3] != want[3]).sum())} t {int((got[2] != want[2]).sum())}", flush=True)
    if w % 100 == 99:
        print(f"{w + 1} worlds, {bad} with a mismatch, {time.time() - t0:.0f} s", flush=True)
ctx.close()
print(f"done: {n_worlds} worlds x {n_rays} rays ({hits / (n_worlds * n_rays):.2f} of them hit something first), {bad} with a mismatch")
sys.exit(1 if bad else 0)
