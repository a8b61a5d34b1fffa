#!/usr/bin/env python3
"""Section shares of the 2d render kernel from the -DRT_PROF diagnostic build (tools/build_prof.sh):
RT_AMD_LIBNAME=librt_amd_prof.so python tools/prof_shares.py [scene width height spp lambda_samples]..."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd.api import params2d

ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
args = sys.argv[1:] or ["cornell_box", "1920", "1080", "16", "16"]
for k in range(0, len(args), 5):
    name, W, H, spp, ls = args[k], int(args[k + 1]), int(args[k + 2]), int(args[k + 3]), int(args[k + 4])
    scene = rt.Scene2D(ctx, rt.World2D(name, seed=1))
    buf = torch.zeros(W * H * rt.NBINS, dtype=torch.float32, device="cuda")
    print(f"== {name} {W}x{H} spp {spp} lambda_samples {ls}", file=sys.stderr, flush=True)
    scene.accumulate(params2d(W, H, spp, ls, 10, seed=0), buf.data_ptr())
    ctx.synchronize()
    scene.close()
