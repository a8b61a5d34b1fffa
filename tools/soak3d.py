#!/usr/bin/env python3
"""Many random 3d sphere sets through the C ABI - the BVH kernel against the linear scan (both on the GPU, every ray)
and both against the CPU checker (test infrastructure; a part of the rays) - beyond the seeds the test suite runs:
python tools/soak3d.py [scenes] [rays per scene]
Same bar as tests/test_gpu_random_scenes.py: first-hit sphere, bounce count and hit distance bit for bit, colour to
1e-5.  Sphere counts around the leaf size and in the thousands, an oversized sphere or none, concentric and duplicate
spheres, tiny spheres, scenes at several scales and places; a third of the rays aim at a sphere's rim."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from tests import oracle_ffi

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000
oracle = oracle_ffi.Oracle(os.path.join(ROOT, "oracle", "build", "librt_oracle.so"))
ctx = rt.Context(0)
t0 = time.time()
bad = 0
hits = 0
nans = 0
for w in range(n_scenes):
    rng = np.random.default_rng(70_000 + w)
    n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 33, 64, 100, 218, 500, 2000]))
    scale, off = [(1.0, (0.0, 0.0, 0.0)), (1e-3, (0.0, 0.0, 0.0)), (1e3, (2e6, -1e6, 5e5)), (1.0, (1e8, 1e8, -1e8))][w % 4]
    c = rng.uniform(-20.0, 20.0, (n, 3))
    r = np.exp(rng.uniform(np.log(0.05), np.log(3.0), n))
    if w % 2 == 0 and n >= 2:
        c[n - 1], r[n - 1] = (0.0, -5020.0, 0.0), 5000.0
    if n >= 4:
        c[1] = c[0]
        c[3], r[3] = c[2], r[2]
    if n >= 6 and w % 3 == 0:
        r[5] = 1e-9                                        # next to a point
    if n >= 40 and w % 5 == 0:
        c[8:40, 1] = c[8, 1]                               # a coplanar run
        c[8:40, 2] = c[8, 2]                               # ... on one line
    c, r = c * scale + np.array(off), r * scale
    d0 = np.array([0.2, -0.1, -1.0]) / np.linalg.norm([0.2, -0.1, -1.0])
    up = np.cross(np.cross(d0, [0.0, 1.0, 0.0]), d0)
    up /= np.linalg.norm(up)
    pos = tuple(np.array([1.0, 3.0, 30.0]) * scale + np.array(off))
    scenes = {b: rt.ArrayScene3D(c[:, 0], c[:, 1], c[:, 2], r, pos, d0, up, 1.5, 32, 24, use_bvh=b) for b in (False, True)}
    lin, bvh = rt.Scene3D(ctx, scenes[False]), rt.Scene3D(ctx, scenes[True])
    m = n_rays
    o = rng.uniform(-25.0, 25.0, (m, 3)) * scale + np.array(off)
    d = rng.normal(0, 1, (m, 3))
    k = rng.integers(0, n, m)
    t = rng.normal(0, 1, (m, 3))
    co = c[k] - o
    t -= (t * co).sum(1, keepdims=True) * co / np.maximum((co ** 2).sum(1, keepdims=True), 1e-300)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    rim = c[k] + t * r[k][:, None] * (1.0 + rng.choice([0.0, 1e-15, -1e-15, 1e-12, -1e-12, 1e-9, -1e-9, 1e-4, -1e-4], m))[:, None]
    d = np.where((rng.random(m) < 1 / 3)[:, None], rim - o, d)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = lin.trace_batch(o, d, 10, seed=w), bvh.trace_batch(o, d, 10, seed=w)
    part = min(m, 4000)
    want = oracle.trace_batch3d(scenes[True], o[:part], d[:part], 10, seed=w)
    lin.close()
    bvh.close()
    # colours may be NaN on every side alike: a hit point that rounds onto a sphere's centre gives `normalize` a zero
    # vector (3d/src/rt.rs:90), and the reference carries the NaN through to the pixel
    same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))
    got = [x[:part] for x in b]
    same_o = (all(np.array_equal(x, y) for x, y in zip(got[1:], list(want)[1:]))
              and np.allclose(got[0], want[0], rtol=1e-5, atol=1e-30, equal_nan=True))
    nans += int(np.isnan(b[0]).any(1).sum())
    hits += int((b[1] >= 0).sum())
    if not (same and same_o):
        bad += 1
        print(f"scene {w} ({n} spheres, scale {scale}, offset {off}): MISMATCH bvh-vs-linear {not same}, gpu-vs-checker {not same_o}", flush=True)
    if w % 100 == 99:
        print(f"{w + 1} scenes, {bad} with a mismatch, {time.time() - t0:.0f} s", flush=True)
ctx.close()
print(f"done: {n_scenes} scenes x {n_rays} rays ({hits / (n_scenes * n_rays):.2f} of them hit something first, {nans} NaN colours on both sides), {bad} with a mismatch")
sys.exit(1 if bad else 0)
