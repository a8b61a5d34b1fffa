#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ with the CPU restatement
(oracle/).  The reference itself (Rust) cannot be run here and ships no test
vectors, so these pin the restatement's behaviour over time and give the GPU
tests a checker-independent target; they are NOT outputs of the reference.

    python tests/golden/make_golden.py        # from the repo root
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import raytracer_amd as rt  # noqa: E402
from raytracer_amd.api import params2d, params3d  # noqa: E402
from tests.oracle_ffi import Oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rays2d(n, seed):
    rng = np.random.default_rng(seed)
    ang = rng.random(n) * 2 * np.pi
    return rng.random(n) * 800, rng.random(n) * 600, np.cos(ang), np.sin(ang), rng.integers(0, 95, n).astype(np.int32)


def main():
    oracle = Oracle(os.path.join(ROOT, "oracle", "build", "librt_oracle.so"))
    out = {}
    for k, name in enumerate(rt.World2D.NAMES):
        world = rt.World2D(name, seed=1)
        ox, oy, dx, dy, li = rays2d(1500, 100 + k)
        rad, fp, ft, nb = oracle.trace_batch2d(world, ox, oy, dx, dy, li, 10, seed=42)
        out.update({f"{name}/ox": ox, f"{name}/oy": oy, f"{name}/dx": dx, f"{name}/dy": dy, f"{name}/li": li,
                    f"{name}/radiance": rad, f"{name}/first_prim": fp, f"{name}/first_t": ft, f"{name}/bounces": nb})
        p = params2d(24, 18, 3, 2, 10, seed=7)
        _, _, spec, rays = oracle.render2d(world, p, threads=4)
        out[f"{name}/render_spectrum"] = spec
        out[f"{name}/render_rays"] = np.array([rays], np.uint64)
    np.savez_compressed(os.path.join(HERE, "golden2d.npz"), **out)

    s3 = rt.Standalone3D(32, 24, use_bvh=False)
    rng = np.random.default_rng(5)
    o = np.tile([0.0, 2.0, 0.0], (3000, 1)) + rng.normal(0, 0.5, (3000, 3))
    d = rng.normal(0, 1, (3000, 3)) * [1.0, 0.5, 1.0] + [0.0, -0.1, -1.5]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rgb, fp, ft, nb = oracle.trace_batch3d(s3, o, d, 10, seed=9)
    img, rays = oracle.render3d(s3, params3d(3, 10, seed=4), threads=4)
    np.savez_compressed(os.path.join(HERE, "golden3d.npz"), o=o, d=d, rgb=rgb, first_prim=fp, first_t=ft, bounces=nb,
                        render=img, render_rays=np.array([rays], np.uint64))

    # the 200-circle stress-test scene drawn from Philox seed 1 (the reference draws it from an unseeded generator)
    np.savez_compressed(os.path.join(HERE, "complex_world_seed1.npz"), **rt.World2D("complex_world", seed=1).arrays())
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
