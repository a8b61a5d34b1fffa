#!/usr/bin/env python3
"""Many forward-tracer passes over random worlds through the C ABI against the CPU checker (test infrastructure):
python tools/soak_forward.py [passes]
Same bar as tests/test_gpu_forward.py: the number of lines drawn, the trace_ray count and the plane of items per pixel
are equal, coverage sums agree to 1e-5.  Emissive materials are kept on circles (the tracer emits from circles only)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from tests import oracle_ffi
from tests.test_gpu_random_scenes import random_world2d

n_passes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
oracle = oracle_ffi.Oracle(os.path.join(ROOT, "oracle", "build", "librt_oracle.so"))
ctx = rt.Context(0)
t0 = time.time()
bad = lines = 0
for w in range(n_passes):
    rng = np.random.default_rng(30_000 + w)
    n_circ, n_seg = [(40, 25), (7, 300), (9, 14), (12, 0), (70, 0), (3, 5), (200, 0), (33, 64), (1, 1)][int(rng.integers(0, 9))]
    W, H = int(rng.integers(8, 400)), int(rng.integers(8, 300))
    base = random_world2d(rng, n_circ, n_seg, span=(float(W), float(H)))     # a pixel is a world unit in this tracer
    a = {k: np.array(v, copy=True) for k, v in base._a.items()}
    strips = np.unique(a["seg_obj"])
    mats = a["materials"]
    for o in strips:
        if mats[o, 0] < 2:
            mats[o, 0] = 2 + (o % 3)                   # no emissive strips
    emitters = [o for o in a["circle_obj"] if mats[o, 0] < 2]
    if len(emitters) > 48:                             # the checker's line buffer is sized for 64 emitters
        for o in emitters[48:]:
            mats[o, 0] = 3
    world = rt.ArrayWorld2D(a)
    rays, depth = int(rng.integers(1, 400)), int(rng.choice([1, 2, 5, 9]))
    seed, off = int(rng.integers(0, 1 << 30)), int(rng.choice([0, 7, 100000]))
    scene = rt.Scene2D(ctx, world)
    p = scene.forward_params(W, H, rays, depth, seed=seed, sample_offset=off)
    rc, want_rgb, want_w, _, want_n, want_rays = oracle.forward_render(world, W, H, rays, depth, seed=seed, sample_offset=off)
    ctx.reset_stats()
    rgb, weight, n = scene.forward_render(p)
    got_rays = ctx.stats()["rays"]
    scene.close()
    same = rc == 0 and n == want_n and got_rays == want_rays and np.array_equal(weight, want_w) and np.allclose(rgb, want_rgb, rtol=1e-5, atol=1e-5)
    lines += n
    if not same:
        bad += 1
        print(f"pass {w} ({n_circ} circles {n_seg} segments {W}x{H} rays {rays} depth {depth}): MISMATCH lines {n} vs {want_n}, rays {got_rays} vs {want_rays}, "
              f"items {int((weight != want_w).sum())}", flush=True)
    if w % 100 == 99:
        print(f"{w + 1} passes, {bad} with a mismatch, {lines} lines drawn, {time.time() - t0:.0f} s", flush=True)
ctx.close()
print(f"done: {n_passes} passes, {lines} lines drawn, {bad} with a mismatch")
sys.exit(1 if bad else 0)
