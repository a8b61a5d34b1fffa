#!/usr/bin/env python3
"""Time the forward light tracer (rt2d_forward_accumulate) on device planes.

    python tools/time_forward.py [--scene complex_world --width 800 --height 600 --rays 1000 --reps 5]

Prints one JSON line: the reference's pass (1000 rays per emitter, depth 5) and a large pass."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="complex_world")
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    import raytracer_amd as rt

    world = rt.World2D(a.scene, seed=1)
    ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    scene = rt.Scene2D(ctx, world)
    rgb = torch.zeros(a.width * a.height * 3, device="cuda")
    weight = torch.ones(a.width * a.height, device="cuda")
    out = {"workload": f"{a.scene} {a.width}x{a.height}, depth 5"}
    for rays in (1000, 1_000_000):
        p = scene.forward_params(a.width, a.height, rays, 5, seed=0)
        lines = scene.forward_accumulate(p, rgb.data_ptr(), weight.data_ptr(), count_lines=True)   # warm-up
        times = []
        for r in range(a.reps):
            p = scene.forward_params(a.width, a.height, rays, 5, seed=0, sample_offset=(r + 1) * rays)
            torch.cuda.synchronize()
            ctx.reset_stats()
            t0 = time.perf_counter()
            scene.forward_accumulate(p, rgb.data_ptr(), weight.data_ptr())
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            st = ctx.stats()
        best = min(times)
        out[f"rays_per_emitter_{rays}"] = {"paths": st["samples"], "hit_searches": st["rays"], "lines_drawn_first_pass": lines,
                                            "pass_ms": best * 1e3, "trace_kernel_ms": st["kernel_ms"],
                                            "Mpaths_per_s": st["samples"] / best / 1e6, "Mrays_per_s": st["rays"] / best / 1e6}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
