#!/usr/bin/env python3
"""Timings of the BASELINE.json configs other than the headline one (bench.py covers configs[1]):
cfg 2' the 2d stress-test scene at 3840x2160 (the north-star target's size; bench.py runs it at 1920x1080)
cfg 3  2d dispersion / absorption scenes at 3840x2160, many wavelengths
cfg 4  3d standalone scene at 1920x1080, spp 512, linear scan and BVH
One JSON line per run (rays and kernel time come from the library's device counter and HIP events)."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="quarter-size workloads")
    ap.add_argument("--only", default="", help="comma-separated subset: cornell, absorption, sample, stress4k, 3dlinear, 3dbvh")
    args = ap.parse_args()
    import torch
    import raytracer_amd as rt
    from raytracer_amd.api import params2d, params3d

    ctx = rt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    q = 4 if args.quick else 1

    def run2d(name, W, H, spp, ls, passes):
        world = rt.World2D(name, seed=1)
        scene = rt.Scene2D(ctx, world)
        buf = torch.zeros(W * H * rt.NBINS, dtype=torch.float32, device="cuda")
        scene.accumulate(params2d(W, H, 1, ls, 10, seed=0), buf.data_ptr())   # warm-up
        torch.cuda.synchronize()
        buf.zero_()
        ctx.reset_stats()
        for i in range(passes):
            scene.accumulate(params2d(W, H, spp, ls, 10, seed=0, sample_offset=i * spp), buf.data_ptr())
        st = ctx.stats()
        P = world.n_prims
        print(json.dumps({"config": f"2d {name} {W}x{H} spp {spp * passes} lambda_samples {ls} depth 10", "primitives": P,
                          "kernel_ms": st["kernel_ms"], "rays": st["rays"], "samples": st["samples"], "searches": st["searches"],
                          "Mrays_per_s": st["rays"] / st["kernel_ms"] / 1e3, "Msamples_per_s": st["samples"] / st["kernel_ms"] / 1e3,
                          "algorithmic_TFLOPs": st["rays"] * (15.0 * P + 60.0) / st["kernel_ms"] / 1e9,
                          # both roofline fractions: algorithmic FLOP against the 78.6 TFLOP/s f64 vector peak, and the
                          # read-modify-write of the spectral buffer (2 x 380 B per pixel and launch) against 8 TB/s
                          "frac_f64_vector_peak": st["rays"] * (15.0 * P + 60.0) / st["kernel_ms"] / 1e9 / 78.6,
                          "frac_hbm_peak": passes * 2.0 * W * H * rt.NBINS * 4 / (st["kernel_ms"] * 1e-3) / 8e12}), flush=True)
        scene.close()

    def run3d(use_bvh, W, H, spp):
        sc = rt.Standalone3D(W, H, use_bvh=use_bvh)
        scene = rt.Scene3D(ctx, sc)
        buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
        scene.accumulate(params3d(1, 10, seed=0), buf.data_ptr())
        torch.cuda.synchronize()
        buf.zero_()
        ctx.reset_stats()
        scene.accumulate(params3d(spp, 10, seed=0), buf.data_ptr())
        st = ctx.stats()
        print(json.dumps({"config": f"3d standalone scene (218 spheres) {W}x{H} spp {spp} bounce 10, {'BVH' if use_bvh else 'linear scan'}",
                          "kernel_ms": st["kernel_ms"], "rays": st["rays"], "samples": st["samples"],
                          "node_visits_per_ray": st["bvh_node_visits"] / st["rays"], "sphere_tests_per_ray": st["sphere_tests"] / st["rays"],
                          "Mrays_per_s": st["rays"] / st["kernel_ms"] / 1e3, "Msamples_per_s": st["samples"] / st["kernel_ms"] / 1e3,
                          "algorithmic_TFLOPs_linear_model": st["rays"] * (23.0 * 218 + 60.0) / st["kernel_ms"] / 1e9,
                          # the linear scan's work is what the reference does; the BVH kernel returns the same hits with
                          # less of it, so its fraction says how far it is ahead of a perfect linear scan
                          "frac_f64_vector_peak_linear_model": st["rays"] * (23.0 * 218 + 60.0) / st["kernel_ms"] / 1e9 / 78.6,
                          "frac_hbm_peak": 2.0 * W * H * 3 * 4 / (st["kernel_ms"] * 1e-3) / 8e12}), flush=True)
        scene.close()

    only = set(filter(None, args.only.split(",")))
    def want(k):
        return not only or k in only
    if want("cornell"): run2d("cornell_box", 3840 // q, 2160 // q, 16, 16, 4 // min(q, 4) or 1)
    if want("absorption"): run2d("cornell_box_absorption", 3840 // q, 2160 // q, 16, 16, 4 // min(q, 4) or 1)
    if want("sample"): run2d("sample_world", 1920 // q, 1080 // q, 64, 2, 1)
    if want("stress4k"): run2d("complex_world", 3840 // q, 2160 // q, 64, 2, 1)
    if want("3dlinear"): run3d(False, 1920 // q, 1080 // q, 512 // q)
    if want("3dbvh"): run3d(True, 1920 // q, 1080 // q, 512 // q)

if __name__ == "__main__":
    main()
