#!/usr/bin/env python3
"""Time the device-resident progressive loop (rt2d_progressive_step) at full size.

    python tools/time_progressive.py [--width 1920 --height 1080 --spp 64 --steps 6 --top 0.01 --mask 2 --oversampling 1 --passes 1]

Prints one JSON line: wall time per step with and without the adaptive passes, the share of the trace kernel,
and the cost of handing a RenderProgress image to the host every step.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(scene, ctx, p, denoiser, steps, with_image):
    import torch
    prog = scene.progressive(p, denoiser=denoiser)
    prog.step()                                   # warm-up: first pass, workspace growth
    if denoiser:
        prog.step()
    torch.cuda.synchronize()
    ctx.reset_stats()
    rays = recomputed = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        s = prog.step()
        rays += s["rays"]
        recomputed += s["recomputed"]
        if with_image:
            prog.image(1.0)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    prog.close()
    return {"ms_per_step": dt / steps * 1e3, "trace_kernel_ms_per_step": st["kernel_ms"] / steps, "rays_per_step": rays / steps,
            "recomputed_per_step": recomputed / steps, "Mrays_per_s": rays / dt / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="complex_world")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--lambda-samples", type=int, default=2)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--top", type=float, default=0.01)
    ap.add_argument("--mask", type=int, default=2)
    ap.add_argument("--oversampling", type=float, default=1.0)
    ap.add_argument("--passes", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()

    import torch
    import raytracer_amd as rt
    from raytracer_amd.api import params2d

    world = rt.World2D(a.scene, seed=1)
    ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    scene = rt.Scene2D(ctx, world)
    p = params2d(a.width, a.height, a.spp, a.lambda_samples, a.depth, seed=0)
    dn = (a.top, a.mask, a.oversampling, a.passes)
    out = {"workload": f"{a.scene} {a.width}x{a.height} spp {a.spp}/pass, denoiser top {a.top} +-{a.mask} x{a.oversampling}, {a.passes} pass(es)",
           "plain": run(scene, ctx, p, None, a.steps, False),
           "plain_with_image_readback": run(scene, ctx, p, None, a.steps, True),
           "adaptive": run(scene, ctx, p, dn, a.steps, False)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
