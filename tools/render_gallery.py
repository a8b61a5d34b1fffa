#!/usr/bin/env python3
"""Converged renders of the worlds behind the reference's gallery images (2d/README.md:20-53), at the gallery's
size, written as PNGs through the Reinhard kernel: python tools/render_gallery.py <out_dir> [spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import raytracer_amd as rt  # noqa: E402
from raytracer_amd.api import params2d  # noqa: E402


def main():
    out = sys.argv[1]
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    os.makedirs(out, exist_ok=True)
    ctx = rt.Context(0)
    for name, ls in (("sample_world", 2), ("cornell_box", 16), ("cornell_box_absorption", 16), ("spectrum_world", 16), ("complex_world", 2)):
        scene = rt.Scene2D(ctx, rt.World2D(name, seed=1))
        t0 = time.time()
        rgb, _ = scene.global_render(params2d(800, 600, spp, ls, 10, seed=0))
        dt = time.time() - t0
        rgba = ctx.tonemap_reinhard(rgb, 1.0, 255).reshape(600, 800, 4)
        rt.write_png(os.path.join(out, name + ".png"), rgba, 800, 600)
        print(name, "spp", spp, "lambda_samples", ls, "%.2f s" % dt, "mean", float(np.mean(rgb)), flush=True)
        scene.close()
    ctx.close()


if __name__ == "__main__":
    main()
