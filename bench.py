#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the per-ray hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload at N = 1 (BASELINE.json configs[1]): the 2d stress-test scene
(`complex_world`, 200 circles, Philox seed 1) at 1920x1080, lambda_samples 2,
recursion limit 10, f64 geometry / f32 radiance.  One step = one pass of
`--spp` (64) samples per pixel over the whole image (rt2d_accumulate), the
spectral radiance buffer staying resident in HBM.  The timed job is K steps,
then the resolve to sRGB; for N > 1 every rank projects its spectral sums to XYZ (a linear map), one
RCCL all-reduce adds the 12-byte-per-pixel XYZ sums, and the encode to sRGB follows.  With N GPUs every rank renders its own sample range of every
pixel (weak scaling: per-GPU work is fixed, total samples grow with N).

1 ray = one trace_ray invocation with depth > 0 (one nearest-hit query);
1 sample = one top-level trace_ray call.  Ray counts are data dependent and are
read back from the device counter after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X public spec: 256 CU x 4 SIMD x 16 f64 lanes x 2 (FMA) x 2.4 GHz
HBM_PEAK_GBS = 8000.0
FLOP_PER_TEST_2D = 15.0         # ray-circle / ray-segment test
FLOP_PER_SHADE = 60.0


def cpu_baseline(world, p_full, threads, spp_sample):
    """The CPU restatement of the reference's render path (serial over columns, thread pool over
    rows) timed on the host cores on a bounded sample: the same image at `spp_sample` samples."""
    from tests.oracle_ffi import Oracle
    from raytracer_amd.api import params2d
    so = os.path.join(ROOT, "oracle", "build", "librt_oracle.so")
    oracle = Oracle(so)
    p = params2d(p_full.width, p_full.height, spp_sample, p_full.lambda_samples, p_full.recursion_limit,
                 seed=p_full.seed, px_to_world=p_full.px_to_world)
    t0 = time.perf_counter()
    _, _, _, rays = oracle.render2d(world, p, threads=threads, want_spectrum=False)
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"same scene and image at spp {spp_sample} ({rays} rays in {dt:.1f} s), "
                      f"C restatement of the reference CPU path, {threads} threads",
            "samples_per_s": p.width * p.height * spp_sample * p.lambda_samples / dt}


def measured_traffic(workload_is_default):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled
    as gfx950 needs, + WRITE_SIZE; separate --pmc runs, tools/profile_r1.sh): only valid for the default workload."""
    if not workload_is_default:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_*.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
            if d.get("kernel") == "render2d_kernel" and "hbm_bytes_per_launch_corrected" in d["per_dispatch"]:
                return d["per_dispatch"]["hbm_bytes_per_launch_corrected"]
        except Exception:
            continue
    return None


def measured_valu_busy(workload_is_default):
    """Fraction of the SIMDs' cycles in which a vector instruction was executing, from the same committed PMC pass:
    SQ_ACTIVE_INST_VALU x 4 over GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs.  The counter sums per-wave execution cycles,
    which overlap in the pipeline, so it can overstate (it reads 1.04 on an 11-primitive scene): an indicator that the
    vector pipe is saturated, not a precise fraction.  Only valid for the default workload."""
    if not workload_is_default:
        return None
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_*.json")))):
        try:
            d = json.load(open(f))["per_dispatch"]
            return d["SQ_ACTIVE_INST_VALU"] * 4.0 / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        except Exception:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="complex_world")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel per step")
    ap.add_argument("--lambda-samples", type=int, default=2)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--cpu-spp", type=int, default=40, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--merge", choices=("xyz", "spectral"), default="xyz",
                    help="N > 1: what the ranks all-reduce - their XYZ projections (12 B/pixel, default) or the spectral "
                         "buffers themselves (380 B/pixel); the image is the same up to f32 rounding order")
    args = ap.parse_args()

    if os.environ.get("RT_BENCH_WATCHDOG"):   # seconds: dump every thread's stack and exit if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RT_BENCH_WATCHDOG"]), exit=True)

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    # RT_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if world_size > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    import raytracer_amd as rt
    from raytracer_amd.api import params2d
    from raytracer_amd.sharding import sample_offset, total_spp

    world = rt.World2D(args.scene, seed=1)
    stream = torch.cuda.current_stream()
    ctx = rt.Context(device, stream=stream.cuda_stream)
    scene = rt.Scene2D(ctx, world)
    W, H = args.width, args.height
    spectrum = torch.zeros(W * H * rt.NBINS, dtype=torch.float32, device="cuda")
    rgb = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")

    def step(i):
        # global sample index: step-major, rank-minor, so no two ranks or steps share a sample
        off = sample_offset(i, rank, world_size, args.spp)
        scene.accumulate(params2d(W, H, args.spp, args.lambda_samples, args.depth, seed=0, sample_offset=off),
                         spectrum.data_ptr())

    xyz = torch.empty(W * H * 3, dtype=torch.float32, device="cuda") if world_size > 1 else None

    def reduce_sum(t):
        if backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:                      # rehearsal backends reduce on the host
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)

    def finish(n_steps):
        n = total_spp(n_steps, world_size, args.spp)
        if world_size > 1 and args.merge == "xyz":
            # the projection on the colour matching functions is linear: ranks exchange 12 bytes per pixel, not 380
            scene.project_xyz(spectrum.data_ptr(), W, H, xyz.data_ptr())
            reduce_sum(xyz)
            scene.resolve_xyz(xyz.data_ptr(), W, H, n, d_rgb_ptr=rgb.data_ptr(), to_host=False)
        else:
            if world_size > 1:
                reduce_sum(spectrum)   # every rank now holds the total (the buffer is zeroed between the warm-up job and the timed one)
            scene.resolve(spectrum.data_ptr(), W, H, n, d_rgb_ptr=rgb.data_ptr(), to_host=False)

    def barrier():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if args.warmup:
        finish(args.warmup)
    barrier()
    spectrum.zero_()
    ctx.reset_stats()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    finish(args.steps)
    barrier()
    dt = time.perf_counter() - t0

    st = ctx.stats()
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    counts = torch.tensor([st["rays"], st["samples"]], dtype=torch.float64, device="cuda")
    if world_size > 1 and backend == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    elif world_size > 1:
        t, counts = t.cpu(), counts.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    total_rays, total_samples = float(counts[0].item()), float(counts[1].item())

    if rank == 0:
        P = world.n_prims
        kernel_ms = st["kernel_ms"] / max(1, st["launches"])           # this rank's average launch duration
        rays_per_launch = st["rays"] / max(1, st["launches"])
        flop_per_launch = rays_per_launch * (FLOP_PER_TEST_2D * P + FLOP_PER_SHADE)
        achieved = flop_per_launch / (kernel_ms * 1e-3) / 1e12
        # algorithmic HBM bytes of one launch: read-modify-write of the spectral buffer (380 B per pixel each way)
        hbm_bytes = 2.0 * W * H * rt.NBINS * 4
        out = {
            "metric": "Mrays/s", "value": total_rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"2d stress_test ({args.scene}, {P} primitives, seed 1) {W}x{H}, "
                                   f"{args.spp} spp/step, lambda_samples {args.lambda_samples}, depth {args.depth}; "
                                   f"job = {args.steps} steps + " + (("XYZ projection + all-reduce + encode" if args.merge == "xyz" else
                                                                      "spectral all-reduce + resolve") if world_size > 1 else "resolve"),
                       "parallelism": f"sample-batch x{world_size}", "mode": "f64 geometry / f32 radiance (parity mode)"},
            "samples_per_s": total_samples / dt,
            "rays_per_sample": total_rays / max(1.0, total_samples),
            "roofline": {"bound": "valu", "achieved": achieved, "peak": F64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F64_VECTOR_PEAK_TFLOPS,
                         "traffic": measured_traffic((args.scene, W, H, args.spp, args.lambda_samples, args.depth)
                                                     == ("complex_world", 1920, 1080, 64, 2, 10)),
                         "valu_busy": measured_valu_busy((args.scene, W, H, args.spp, args.lambda_samples, args.depth)
                                                         == ("complex_world", 1920, 1080, 64, 2, 10)),
                         "kernel": "render2d_kernel", "kernel_ms": kernel_ms,
                         "flop_model": f"algorithmic: rays x (15 x {P} + 60) FLOP, the reference's f64 work per ray, over "
                                       "the measured launch time, against the f64 vector FMA peak; the kernel's own "
                                       "instruction count is lower (f32 sign-bit filter ahead of the exact f64 tests)",
                         "hbm": {"achieved": hbm_bytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": hbm_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
        if args.cpu_spp > 0 and world_size == 1:
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            p_full = params2d(W, H, args.spp, args.lambda_samples, args.depth, seed=0)
            out["cpu_baseline"] = cpu_baseline(world, p_full, threads, args.cpu_spp)
        print(json.dumps(out), flush=True)

    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
