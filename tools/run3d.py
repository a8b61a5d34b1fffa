#!/usr/bin/env python3
"""One timed 3d render of the standalone scene (for rocprofv3 passes): python tools/run3d.py [--linear] [--spp N]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--linear", action="store_true")
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
args = ap.parse_args()
import torch
import raytracer_amd as rt
from raytracer_amd.api import params3d

ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
scene = rt.Scene3D(ctx, rt.Standalone3D(args.width, args.height, use_bvh=not args.linear))
buf = torch.zeros(args.width * args.height * 3, dtype=torch.float32, device="cuda")
scene.accumulate(params3d(1, 10, seed=0), buf.data_ptr())
torch.cuda.synchronize()
ctx.reset_stats()
scene.accumulate(params3d(args.spp, 10, seed=0), buf.data_ptr())
st = ctx.stats()
print(json.dumps({"bvh": not args.linear, "kernel_ms": st["kernel_ms"], "rays": st["rays"],
                  "Mrays_per_s": st["rays"] / st["kernel_ms"] / 1e3,
                  "node_visits_per_ray": st["bvh_node_visits"] / st["rays"], "sphere_tests_per_ray": st["sphere_tests"] / st["rays"]}))
scene.close()
