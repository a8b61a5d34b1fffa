#!/usr/bin/env python3
"""One scene of tools/soak3d.py again, with the rays on which the BVH kernel, the linear scan and the checker differ."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from tests import oracle_ffi
w = int(sys.argv[1]); n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
oracle = oracle_ffi.Oracle(os.path.join(ROOT, "oracle", "build", "librt_oracle.so"))
ctx = rt.Context(0)
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
    r[5] = 1e-9
if n >= 40 and w % 5 == 0:
    c[8:40, 1] = c[8, 1]
    c[8:40, 2] = c[8, 2]
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
want = oracle.trace_batch3d(scenes[True], o, d, 10, seed=w)
print("n", n, "scale", scale, "off", off)
print("linear vs checker: ids", (a[1] != want[1]).sum(), "t", (a[2] != want[2]).sum(), "bounces", (a[3] != want[3]).sum())
print("bvh    vs checker: ids", (b[1] != want[1]).sum(), "t", (b[2] != want[2]).sum(), "bounces", (b[3] != want[3]).sum())
diff = np.nonzero((b[1] != want[1]) | (b[2] != want[2]) | (b[3] != want[3]) | (a[1] != want[1]) | (a[3] != want[3]))[0]
for i in diff[:12]:
    print(f"ray {i}: aimed at sphere {k[i]} (r {r[k[i]]:.3g}); first hit lin {a[1][i]} t {a[2][i]!r} nb {a[3][i]} | bvh {b[1][i]} t {b[2][i]!r} nb {b[3][i]} | checker {want[1][i]} t {want[2][i]!r} nb {want[3][i]}")
print("rgb: NaN lin", int(np.isnan(a[0]).any(1).sum()), "bvh", int(np.isnan(b[0]).any(1).sum()), "checker", int(np.isnan(want[0]).any(1).sum()),
      "| lin == bvh bitwise (NaN == NaN)", bool(np.array_equal(a[0], b[0], equal_nan=True)),
      "| bvh ~ checker", bool(np.allclose(b[0], want[0], rtol=1e-5, atol=1e-30, equal_nan=True)))
