#!/usr/bin/env python3
"""Many adaptive re-sampling passes (rt2d_denoise, host planes) on random images against the CPU checker (test
infrastructure): python tools/soak_denoise.py [passes]
The merged and loop images are random planes - half of them quantised, so that many noise keys tie at the selection
threshold - of sizes from 3x3 up; top fractions from "one pixel" to "more than all", masks that reach outside the
image.  Same bar as tests/test_gpu_denoise.py: the same pixels are re-sampled (weights and counts equal, trace_ray
count equal), colours agree to 2e-4."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from raytracer_amd.api import params2d
from tests import oracle_ffi

n_passes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
oracle = oracle_ffi.Oracle(os.path.join(ROOT, "oracle", "build", "librt_oracle.so"))
ctx = rt.Context(0)
worlds = {name: rt.World2D(name, seed=1) for name in ("simple_world", "cornell_box", "sample_world")}
scenes = {name: rt.Scene2D(ctx, w) for name, w in worlds.items()}
t0 = time.time()
bad = resampled = 0
for k in range(n_passes):
    rng = np.random.default_rng(20_000 + k)
    name = list(worlds)[k % 3]
    W, H = int(rng.integers(3, 80)), int(rng.integers(3, 60))
    merged = rng.random((H, W, 3)).astype(np.float32) * 2.0
    loop = rng.random((H, W, 3)).astype(np.float32)
    if k % 2:
        merged, loop = np.round(merged * 4) / 4, np.round(loop * 4) / 4
    if k % 5 == 0:
        loop[:] = merged * 0.5                                          # a flat noise image: every key ties
    weight = (rng.random((H, W)) < 0.3).astype(np.float32) * 3.0
    top = float(rng.choice([0.5 / (W * H), 1.0 / (W * H), 0.01, 0.05, 0.3, 1.0, 2.0]))
    mask, over = int(rng.integers(0, 4)), float(rng.choice([1.0, 2.5, 0.4]))
    p = params2d(W, H, int(rng.integers(1, 4)), int(rng.choice([1, 2, 3])), int(rng.choice([1, 3, 10])), seed=int(rng.integers(0, 1 << 30)),
                 sample_offset=int(rng.choice([0, 8])))
    want_rgb, want_w, key, want_n, want_rays = oracle.denoise2d(worlds[name], p, top, mask, over, merged, weight, loop)
    ctx.reset_stats()
    got_rgb, got_w, n = scenes[name].denoise(p, top, mask, over, merged, weight, loop)
    rays = ctx.stats()["rays"]
    same = (n == want_n and rays == want_rays and np.array_equal(got_w.reshape(H, W), want_w.reshape(H, W))
            and np.allclose(got_rgb.reshape(H, W, 3), want_rgb.reshape(H, W, 3), rtol=2e-4, atol=2e-5))
    resampled += n
    if not same:
        bad += 1
        print(f"pass {k} ({name} {W}x{H} top {top:.4g} mask {mask} over {over}): MISMATCH recomputed {n} vs {want_n}, rays {rays} vs {want_rays}, "
              f"weights {int((got_w.reshape(H, W) != want_w.reshape(H, W)).sum())}", flush=True)
    if k % 100 == 99:
        print(f"{k + 1} passes, {bad} with a mismatch, {resampled} pixels re-sampled, {time.time() - t0:.0f} s", flush=True)
for s in scenes.values():
    s.close()
ctx.close()
print(f"done: {n_passes} passes, {resampled} pixels re-sampled, {bad} with a mismatch")
sys.exit(1 if bad else 0)
