#!/usr/bin/env python3
"""Time one adaptive re-sampling pass (rt2d_denoise_device) at full size.

    python tools/time_denoise.py [--width 1920 --height 1080 --spp 64 --top 0.01 --mask 2 --oversampling 10 --reps 5]

Two independent passes of the stress-test scene give `merged` (their sum, as the reference's `merged + loop`
does) and `loop`; the pass is then timed with both RawImage planes resident in HBM.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
    ap.add_argument("--oversampling", type=float, default=10.0)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()

    import torch
    import raytracer_amd as rt
    from raytracer_amd.api import params2d

    W, H = a.width, a.height
    world = rt.World2D(a.scene, seed=1)
    ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    scene = rt.Scene2D(ctx, world)
    spec = torch.zeros(W * H * rt.NBINS, dtype=torch.float32, device="cuda")
    rgb = [torch.empty(W * H * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
    for k in range(2):
        spec.zero_()
        scene.accumulate(params2d(W, H, a.spp, a.lambda_samples, a.depth, seed=0, sample_offset=k * a.spp), spec.data_ptr())
        scene.resolve(spec.data_ptr(), W, H, a.spp, d_rgb_ptr=rgb[k].data_ptr(), to_host=False)
    loop = rgb[1]
    merged0 = rgb[0] + rgb[1]
    weight0 = torch.ones(W * H, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    times, rays, n_px = [], 0, 0
    for r in range(a.reps + 1):
        merged, weight = merged0.clone(), weight0.clone()
        torch.cuda.synchronize()
        ctx.reset_stats()
        t0 = time.perf_counter()
        n_px = scene.denoise_device(params2d(W, H, a.spp, a.lambda_samples, a.depth, seed=0, sample_offset=2 * a.spp), a.top, a.mask,
                                    a.oversampling, merged.data_ptr(), weight.data_ptr(), loop.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        if r:  # first call is the warm-up
            times.append(dt)
            rays = st["rays"]
            kernel_ms = st["kernel_ms"]
    best = min(times)
    print(json.dumps({"workload": f"{a.scene} {W}x{H} spp {a.spp} x{a.oversampling} on top {a.top} +-{a.mask}", "recomputed_pixels": n_px,
                      "fraction_of_image": n_px / (W * H), "rays": rays, "pass_ms_best": best * 1e3,
                      "pass_ms_mean": sum(times) / len(times) * 1e3, "trace_kernel_ms": kernel_ms,
                      "selection_ms": best * 1e3 - kernel_ms, "Mrays_per_s_whole_pass": rays / best / 1e6}))


if __name__ == "__main__":
    main()
