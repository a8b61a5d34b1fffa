#!/usr/bin/env python3
"""Exercises the per-pixel kernels that follow the trace kernel (resolve, XYZ projection, the progressive merge and
the two `Blending::Add` kernels of the adaptive passes) a few times each, for a rocprofv3 --kernel-trace --stats run:

    rocprofv3 --kernel-trace --stats -d gpurun_out/k5 -- python3 tools/run_k5.py [--width 1920 --height 1080]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
import torch
import raytracer_amd as rt
from raytracer_amd.api import params2d

W, H = a.width, a.height
ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
scene = rt.Scene2D(ctx, rt.World2D("complex_world", seed=1))
p = params2d(W, H, 4, 2, 10, seed=0)
buf = torch.zeros(W * H * rt.NBINS, dtype=torch.float32, device="cuda")
rgb = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
xyz = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
scene.accumulate(p, buf.data_ptr())
for _ in range(a.reps):
    scene.resolve(buf.data_ptr(), W, H, 4, d_rgb_ptr=rgb.data_ptr(), to_host=False)
    scene.project_xyz(buf.data_ptr(), W, H, xyz.data_ptr())
    scene.resolve_xyz(xyz.data_ptr(), W, H, 4, d_rgb_ptr=rgb.data_ptr(), to_host=False)
# adaptive pass on RawImage planes (denoise_add_kernel) and the progressive loop (progressive_merge / progressive_add)
merged = rgb.clone()
weight = torch.full((W * H,), 4.0, dtype=torch.float32, device="cuda")
scene.accumulate(params2d(W, H, 4, 2, 10, seed=0, sample_offset=4), buf.data_ptr())
scene.resolve(buf.data_ptr(), W, H, 8, d_rgb_ptr=rgb.data_ptr(), to_host=False)
for _ in range(a.reps):
    scene.denoise_device(p, 0.05, 2, 1.0, merged.data_ptr(), weight.data_ptr(), rgb.data_ptr())
prog = scene.progressive(p, denoiser=(0.05, 2, 1.0, 1))
for _ in range(a.reps + 1):
    prog.step()
ctx.synchronize()
prog.close()
scene.close()
print("ok")
