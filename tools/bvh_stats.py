#!/usr/bin/env python3
"""Diagnostic (needs a library built with the traversal counters, see DESIGN): loop counts of the 3d BVH traversal."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RT_AMD_LIBNAME", "librt_amd_stats.so")
import torch
import raytracer_amd as rt
from raytracer_amd import _ffi
from raytracer_amd.api import params3d
ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
scene = rt.Scene3D(ctx, rt.Standalone3D(1920, 1080, use_bvh=True))
buf = torch.zeros(1920 * 1080 * 3, device="cuda")
scene.accumulate(params3d(32, 10, seed=0), buf.data_ptr())
ctx.synchronize()
lib = C.CDLL(os.path.join(os.path.dirname(_ffi.__file__), "lib", os.environ["RT_AMD_LIBNAME"]))
out = (C.c_ulonglong * 8)()
print(lib.rt_debug_bvh_stats(out), list(out), ctx.stats())
c, inner, inner_l, leaf, leaf_l, act = out[0], out[1], out[2], out[3], out[4], out[5]
print(f"per wave-step: active lanes {act/c:.1f}, inner iterations {inner/c:.1f} at {inner_l/max(inner,1):.1f} lanes, leaf rounds {leaf/c:.1f} at {leaf_l/max(leaf,1):.1f} lanes")
print(f"per ray: inner node visits {inner_l/act:.1f}, leaf visits {leaf_l/act:.1f}")
