"""Time the host-pointer entry rt2d_render (= global_render) on the headline workload."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd.api import params2d
ctx = rt.Context(0)
world = rt.World2D("complex_world", seed=1)
scene = rt.Scene2D(ctx, world)
p = params2d(1920, 1080, 64, 2, 10, seed=0)
scene.global_render(p)
ctx.reset_stats()
t0 = time.perf_counter()
for i in range(3):
    scene.global_render(p)
dt = (time.perf_counter() - t0) / 3
st = ctx.stats()
print("rt2d_render host-to-host: %.1f ms per call, %.1f Mrays/s (kernel %.1f ms)" % (dt * 1e3, st["rays"] / 3 / dt / 1e6, st["kernel_ms"] / 3))
