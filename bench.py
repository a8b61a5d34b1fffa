#!/usr/bin/env python3
"""Benchmark of the per-ray hot path on N MI355X: Mrays/s and samples/s.

    python bench.py --gpus N --steps K --warmup W              # weak scaling (default)
    python bench.py --gpus N --mode strong [--workload ...]    # fixed 4K / 4096-spp job split over the ranks

Headline workload (BASELINE.json configs[1]): the 2d stress-test scene (`complex_world`, 200 circles, Philox
seed 1) at 1920x1080, lambda_samples 2, recursion limit 10, f64 geometry / f32 radiance.  One step = one pass of
`--spp` (64) samples per pixel over the whole image (rt2d_accumulate), the spectral radiance buffer staying resident
in HBM.  The timed job is K steps, then the resolve to sRGB; for N > 1 every rank renders its own sample range of
every pixel, projects its spectral sums to XYZ (a linear map), one RCCL all-reduce adds the 12-byte-per-pixel XYZ
sums, and the encode to sRGB follows.  Weak scaling: per-GPU work is fixed, total samples grow with N.

After the timed headline job, at N = 1, the same process times the other configurations BASELINE.json names
(`configs` in the output line): the stress scene at 3840x2160, the dispersion and absorption scenes at 3840x2160
with 16 lambda-samples, the animated sample scene, and the 3d scene with the linear scan and with the BVH.

--mode strong is BASELINE.json configs[4]: a fixed job - 3840x2160, 4096 samples per pixel in total - whose
global sample range is cut into G contiguous parts, one per rank (raytracer_amd.sharding.strong_range); every rank
accumulates its part in passes of --spp samples, then the partial sums meet in one all-reduce (XYZ or spectral for
2d, the f32[W*H][3] image for 3d) and the image is resolved.  A step is the whole job.

1 ray = one trace_ray invocation with depth > 0; 1 sample = one top-level trace_ray call (pixel, sample,
lambda-sample).  The lambda-samples of one (pixel, sample) trace the same primary ray; the kernel searches its
nearest hit once and reuses it (`searches` <= `rays` in the output).  Counts are data dependent and are read back
from the device counters after the timed region.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X vector peaks: 256 CU x 4 SIMD x 16 lanes x 2 (FMA) x 2.4 GHz = 78.6 TFLOP/s for f64 and for unpacked f32;
# packed f32 (v_pk_fma_f32, two per lane and cycle) doubles that.  BASELINE.json's metric names the FP32 VALU roofline.
F32_PACKED_PEAK_TFLOPS = 157.3
F64_VECTOR_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0
FLOP_PER_TEST_2D = 15.0     # ray-circle / ray-segment test
FLOP_PER_SHADE = 60.0
FLOP_PER_SPHERE_TEST = 23.0
FLOP_PER_BOX_TEST = 30.0    # 3d slab test of one box


def profile_counters(tag):
    """Executed-instruction evidence for one configuration from the committed rocprofv3 PMC passes
    (profiles/r*/pmc_<tag>.json, written by tools/collect_profiles.py from separate --pmc runs):
      valu_issue_frac       the share of the SIMDs' cycles in which a vector instruction was executing
                            (SQ_ACTIVE_INST_VALU x 4 over GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs; the counter sums per-wave
                            execution cycles, which overlap in the pipeline, so a saturated pipe reads 0.95 to 1.15);
      hbm_bytes_per_launch  FETCH_SIZE doubled as gfx950 needs, + WRITE_SIZE;
      flop_f32 / flop_f64   the hardware's own FLOP count by type (SQ_INSTS_VALU_FLOPS_FP32 / _FP64: per wave
                            instruction, FMA = 2) x 64 lanes, and the share of lanes active while vector instructions ran.
    These are replayed from the profile files, not measured in this run: `source` says which file, and `kernel_ms`
    of the profiled launches turns the counts into rates."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_{tag}.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))["per_dispatch"]
        out = {"source": os.path.relpath(files[-1], ROOT)}
        if "SQ_ACTIVE_INST_VALU" in d and "GRBM_GUI_ACTIVE" in d:
            out["valu_issue_frac"] = d["SQ_ACTIVE_INST_VALU"] * 4.0 / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "SQ_INSTS_VALU" in d:
            out["valu_insts_per_launch"] = d["SQ_INSTS_VALU"]
        if "hbm_bytes_per_launch_corrected" in d:
            out["hbm_bytes_per_launch"] = d["hbm_bytes_per_launch_corrected"]
        for k in ("flop_f32_per_launch_full_waves", "flop_f64_per_launch_full_waves", "valu_active_lane_share"):
            if k in d:
                out[k] = d[k]
        return out
    except Exception:
        return None


def roofline(kind, st, kernel, n_prims, hbm_model_bytes_per_launch, tag):
    """The roofline object of one measured configuration.  `st` = the context's counters over the timed launches.

    frac (= achieved / peak) is an EQUIVALENT rate: the reference's algorithmic work per launch over the kernel's
    average launch duration (HIP events on the launch stream, folded into rt_stats.kernel_ms): 2d and the 3d linear
    scan = rays x (FLOP per test x primitives + 60), the work trace_ray does per invocation; 3d BVH = counted work
    (inner-node visits x 2 boxes x 30 + exact sphere tests x 23 + 60 per ray).  The kernels decide most tests with a
    packed-f32 sign filter in 2.5 to 5 instruction slots and search the shared primary hit of a pair's lambda-samples
    once, so they execute fewer operations than the model counts.  peak = the packed-f32 vector peak, the roofline
    BASELINE.json's metric names.  Beside it:
      frac_searches_executed  the same model priced on the searches the kernel ran (2d), over the same peak;
      valu_issue_frac         how busy the vector pipes were (PMC);
      executed_f32 / _f64     the FLOP the hardware counted, by type, over the peak of that type (PMC);
      hbm                     HBM bytes per launch from the PMC counters (or, for a workload nobody profiled, the
                              algorithmic bytes of the accumulation atomics as a stand-in, labelled) over 8 TB/s."""
    launches = max(1, st["launches"])
    kernel_ms = st["kernel_ms"] / launches
    rays = st["rays"] / launches
    searched = None
    if kind == "2d":
        flop = rays * (FLOP_PER_TEST_2D * n_prims + FLOP_PER_SHADE)
        searched = (st["searches"] / launches) * FLOP_PER_TEST_2D * n_prims + rays * FLOP_PER_SHADE
        model = f"rays x (15 x {n_prims} + 60) FLOP"
    elif kind == "3d_linear":
        flop = rays * (FLOP_PER_SPHERE_TEST * n_prims + FLOP_PER_SHADE)
        model = f"rays x (23 x {n_prims} + 60) FLOP"
    else:
        flop = (st["bvh_node_visits"] * 2.0 * FLOP_PER_BOX_TEST + st["sphere_tests"] * FLOP_PER_SPHERE_TEST) / launches + rays * FLOP_PER_SHADE
        model = "counted: node visits x 2 x 30 + exact sphere tests x 23 + rays x 60 FLOP"
    sec = kernel_ms * 1e-3
    achieved = flop / sec / 1e12
    out = {"bound": "valu", "achieved": achieved, "peak": F32_PACKED_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": achieved / F32_PACKED_PEAK_TFLOPS, "kind": "equivalent (reference work per launch / launch time)",
           "kernel": kernel, "kernel_ms": kernel_ms, "flop_model": model, "traffic": None}
    if searched is not None:
        out["frac_searches_executed"] = searched / sec / 1e12 / F32_PACKED_PEAK_TFLOPS
    if kind == "3d_bvh":
        # for comparison with the linear-scan line only: the work the reference's scan does for the same rays, over
        # this kernel's time.  The tree skips most of it, so this is no utilisation and can exceed any peak.
        scan = rays * (FLOP_PER_SPHERE_TEST * n_prims + FLOP_PER_SHADE)
        out["reference_scan_equivalent"] = {"achieved": scan / sec / 1e12, "unit": "TFLOP/s",
                                            "model": f"rays x (23 x {n_prims} + 60) FLOP, the linear scan's work for the same rays; not a fraction of a peak"}
    prof = profile_counters(tag)
    hbm_bytes, hbm_model = hbm_model_bytes_per_launch, "stand-in (no PMC profile of this workload): algorithmic bytes of the accumulation atomics and the shared primary hits"
    if prof:
        out["counters_from"] = prof["source"]
        if "valu_issue_frac" in prof:
            out["valu_issue_frac"] = prof["valu_issue_frac"]
        if "valu_insts_per_launch" in prof:
            out["valu_insts_per_ray"] = prof["valu_insts_per_launch"] * 64.0 / max(1.0, rays)
        if "hbm_bytes_per_launch" in prof:
            out["traffic"] = hbm_bytes = prof["hbm_bytes_per_launch"]
            hbm_model = "PMC: 2 x FETCH_SIZE + WRITE_SIZE per launch, over this run's launch time"
        for name, key, peak in (("executed_f32", "flop_f32_per_launch_full_waves", F32_PACKED_PEAK_TFLOPS),
                                ("executed_f64", "flop_f64_per_launch_full_waves", F64_VECTOR_PEAK_TFLOPS)):
            if key in prof:
                rate = prof[key] / sec / 1e12
                out[name] = {"achieved": rate, "peak": peak, "unit": "TFLOP/s", "frac": rate / peak,
                             "model": "PMC: hardware FLOP count of this type per launch (per wave instruction, FMA = 2) x 64 lanes, over this run's launch time"}
        if "valu_active_lane_share" in prof:
            out["valu_active_lane_share"] = prof["valu_active_lane_share"]
    out["hbm"] = {"achieved": hbm_bytes / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": hbm_bytes / sec / 1e9 / HBM_PEAK_GBS, "model": hbm_model}
    return out


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


class Work2D:
    """One 2d configuration: scene, image size, samples per pass; accumulate passes, then merge and resolve."""
    kind = "2d"

    def __init__(self, rt, torch, ctx, scene_name, W, H, spp, ls, depth, merge="xyz", tag=None):
        from raytracer_amd.api import params2d
        self.rt, self.torch, self.ctx, self.params2d = rt, torch, ctx, params2d
        self.name, self.W, self.H, self.spp, self.ls, self.depth, self.merge = scene_name, W, H, spp, ls, depth, merge
        self.world = rt.World2D(scene_name, seed=1)
        self.scene = rt.Scene2D(ctx, self.world)
        self.n_prims = self.world.n_prims
        self.buf = torch.zeros(W * H * rt.NBINS, dtype=torch.float32, device="cuda")
        self.rgb = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
        self.xyz = None
        self.kernel = "render2d_kernel"
        self.tag = tag or f"{scene_name}_{W}x{H}_s{spp}_l{ls}"
        self.samples_per_pass = W * H * spp * ls
        # algorithmic bytes when no counters exist for the workload: one 32-byte memory-side atomic per sample (an upper
        # bound: samples that return 0 add nothing) + the 44 bytes a (pixel, sample) pair's shared primary hit takes
        self.hbm_bytes_per_launch = 32.0 * self.samples_per_pass + 44.0 * W * H * spp

    def describe(self):
        return (f"2d {self.name} ({self.n_prims} primitives, seed 1) {self.W}x{self.H}, {self.spp} spp/pass, "
                f"lambda_samples {self.ls}, depth {self.depth}")

    def clear(self):
        self.buf.zero_()

    def render_pass(self, first_sample, spp=None):
        self.scene.accumulate(self.params2d(self.W, self.H, self.spp if spp is None else spp, self.ls, self.depth, seed=0, sample_offset=first_sample),
                              self.buf.data_ptr())

    def finish(self, total_spp, reduce_sum):
        """Merge over ranks (reduce_sum = None on one rank) and resolve to sRGB on the device."""
        if reduce_sum is not None and self.merge == "xyz":
            if self.xyz is None:
                self.xyz = self.torch.empty(self.W * self.H * 3, dtype=self.torch.float32, device="cuda")
            # the projection on the colour matching functions is linear: ranks exchange 12 bytes per pixel, not 380
            self.scene.project_xyz(self.buf.data_ptr(), self.W, self.H, self.xyz.data_ptr())
            reduce_sum(self.xyz)
            self.scene.resolve_xyz(self.xyz.data_ptr(), self.W, self.H, total_spp, d_rgb_ptr=self.rgb.data_ptr(), to_host=False)
        else:
            if reduce_sum is not None:
                reduce_sum(self.buf)
            self.scene.resolve(self.buf.data_ptr(), self.W, self.H, total_spp, d_rgb_ptr=self.rgb.data_ptr(), to_host=False)

    def close(self):
        self.scene.close()
        self.buf = self.rgb = self.xyz = None


class Work3D:
    """The 3d crate's standalone scene (218 spheres): accumulate passes into f32[W*H][3], sum over ranks."""

    def __init__(self, rt, torch, ctx, W, H, spp, bounce, use_bvh, tag=None):
        from raytracer_amd.api import params3d
        self.rt, self.torch, self.ctx, self.params3d = rt, torch, ctx, params3d
        self.W, self.H, self.spp, self.depth, self.use_bvh = W, H, spp, bounce, use_bvh
        self.kind = "3d_bvh" if use_bvh else "3d_linear"
        self.host_scene = rt.Standalone3D(W, H, use_bvh=use_bvh)
        self.scene = rt.Scene3D(ctx, self.host_scene)
        self.n_prims = self.host_scene.desc.n_spheres
        self.buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
        self.kernel = "render3d_kernel<%s>" % ("true" if use_bvh else "false")
        self.tag = tag or f"3d_{'bvh' if use_bvh else 'linear'}_{W}x{H}_s{spp}"
        self.samples_per_pass = W * H * spp
        # algorithmic bytes when no counters exist: the wave accumulators' flushes, 3 atomics of 32 bytes per pixel and
        # work unit of 16 samples
        self.hbm_bytes_per_launch = 96.0 * W * H * max(1.0, spp / 16.0)

    def describe(self):
        return (f"3d standalone scene ({self.n_prims} spheres) {self.W}x{self.H}, {self.spp} spp/pass, bounce limit {self.depth}, "
                + ("BVH" if self.use_bvh else "linear scan"))

    def clear(self):
        self.buf.zero_()

    def render_pass(self, first_sample, spp=None):
        self.scene.accumulate(self.params3d(self.spp if spp is None else spp, self.depth, seed=0, sample_offset=first_sample), self.buf.data_ptr())

    def finish(self, total_spp, reduce_sum):
        if reduce_sum is not None:
            reduce_sum(self.buf)      # World::compute_pixel sums over spp (3d/src/rt.rs:239-246): the image IS the sum

    def close(self):
        self.scene.close()
        self.buf = None


def time_config(torch, ctx, work, steps, warmup):
    """Warm-up passes, then `steps` timed passes + resolve on one GPU; returns the entry for `configs`."""
    for i in range(warmup):
        work.render_pass(i * work.spp)
    work.finish(max(1, warmup) * work.spp, None)
    torch.cuda.synchronize()
    work.clear()
    ctx.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        work.render_pass(i * work.spp)
    work.finish(steps * work.spp, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    return {"workload": work.describe(), "steps": steps, "ms_per_step": dt / steps * 1e3,
            "Mrays_per_s": st["rays"] / dt / 1e6, "samples_per_s": st["samples"] / dt,
            "searches_per_s": st["searches"] / dt, "rays_per_sample": st["rays"] / max(1, st["samples"]),
            "roofline": roofline(work.kind, st, work.kernel, work.n_prims, work.hbm_bytes_per_launch, work.tag)}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of this one (same
    interpreter, same arguments, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run sets them),
    wait for all of them and return the first non-zero exit code.  This parent never imports torch and never touches
    HIP, and nothing is exec'ed from a process that has: the children are ordinary subprocesses.  Rank 0's one JSON
    line goes to this process's stdout; the other ranks print nothing there."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(os.environ.get("MASTER_PORT", port)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        # a rank that dies leaves the others waiting in a collective: once one has failed, the rest get a grace
        # period and are then terminated
        pending = list(procs)
        deadline = None
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is not None:
                    pending.remove(p)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 1
                        deadline = time.monotonic() + 30.0
            if deadline is not None and time.monotonic() > deadline:
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=("weak", "strong"), default="weak")
    ap.add_argument("--workload", choices=("stress", "cornell", "absorption", "3d"), default="stress",
                    help="--mode strong: which 4K / 4096-spp job (3d: 1920x1080 BVH, 4096 spp)")
    ap.add_argument("--total-spp", type=int, default=4096, help="--mode strong: samples per pixel of the whole job")
    ap.add_argument("--scene", default="complex_world", help="headline scene; `standalone3d` = the 3d scene")
    ap.add_argument("--bvh", type=int, default=1, help="--scene standalone3d: 1 = BVH, 0 = linear scan")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel per pass")
    ap.add_argument("--lambda-samples", type=int, default=2)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--cpu-spp", type=int, default=40, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--configs", choices=("all", "none"), default="all",
                    help="N = 1, weak mode: also time the other BASELINE.json configurations after the headline job")
    ap.add_argument("--merge", choices=("xyz", "spectral"), default="xyz",
                    help="N > 1: what the ranks all-reduce - their XYZ projections (12 B/pixel, default) or the spectral "
                         "buffers themselves (380 B/pixel); the image is the same up to f32 rounding order")
    args = ap.parse_args()

    if os.environ.get("RT_BENCH_WATCHDOG"):   # seconds: dump every thread's stack and exit if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RT_BENCH_WATCHDOG"]), exit=True)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked plainly (no launcher): this process becomes the launcher, before anything here touches torch or HIP
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: give both the same number of ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    # RT_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    # RT_BENCH_FORCE_DIST=1 takes the N > 1 road with one rank too: the process group is created and the merge step
    # goes through its all-reduce (RCCL with backend nccl), which a one-GPU box can then exercise
    use_dist = world_size > 1 or bool(os.environ.get("RT_BENCH_FORCE_DIST"))
    if use_dist:
        if world_size == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    import raytracer_amd as rt
    from raytracer_amd.api import params2d
    from raytracer_amd.sharding import sample_offset, strong_range, total_spp

    stream = torch.cuda.current_stream()
    ctx = rt.Context(device, stream=stream.cuda_stream)

    def reduce_sum(t):
        if backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:                      # rehearsal backends reduce on the host
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    strong = args.mode == "strong"
    if strong:
        if args.workload == "3d":
            work = Work3D(rt, torch, ctx, 1920, 1080, 256, args.depth, True)
        else:
            name, ls, spp = {"stress": ("complex_world", 2, 64), "cornell": ("cornell_box", 16, 16),
                             "absorption": ("cornell_box_absorption", 16, 16)}[args.workload]
            work = Work2D(rt, torch, ctx, name, 3840, 2160, spp, ls, args.depth, merge=args.merge)
    elif args.scene == "standalone3d":
        work = Work3D(rt, torch, ctx, args.width, args.height, args.spp, args.depth, bool(args.bvh))
    else:
        work = Work2D(rt, torch, ctx, args.scene, args.width, args.height, args.spp, args.lambda_samples, args.depth, merge=args.merge)
    merge_fn = reduce_sum if use_dist else None

    if strong:
        lo, hi = strong_range(rank, world_size, args.total_spp)

        def job():
            # this rank's contiguous part of the global sample range, in passes of at most work.spp samples
            at = lo
            while at < hi:
                n = min(work.spp, hi - at)
                work.render_pass(at, n)
                at += n
            work.finish(args.total_spp, merge_fn)

        def run(n_jobs):
            for _ in range(n_jobs):
                work.clear()
                job()

        def warm():
            # a rank whose part of the job is empty (more ranks than samples) renders nothing but still joins the merge
            if hi > lo:
                work.render_pass(lo, min(work.spp, hi - lo))
            work.finish(args.total_spp, merge_fn)
    else:
        def run(n_steps):
            for i in range(n_steps):
                # global sample index: step-major, rank-minor, so no two ranks or steps share a sample
                work.render_pass(sample_offset(i, rank, world_size, work.spp))
            work.finish(total_spp(n_steps, world_size, work.spp), merge_fn)
        warm = lambda: run(args.warmup)  # noqa: E731

    if use_dist:
        # communicator set-up (the first collective of a process group builds its rings) stays outside the timed job
        # whatever --warmup says: one reduction of the size the job's merge moves
        n_merge = work.W * work.H * 3 if (work.kind != "2d" or args.merge == "xyz") else work.W * work.H * rt.NBINS
        scratch = torch.zeros(n_merge, dtype=torch.float32, device="cuda")
        reduce_sum(scratch)
        del scratch
    if args.warmup:
        if strong:
            for _ in range(args.warmup):
                warm()
        else:
            warm()
    barrier()
    work.clear()
    ctx.reset_stats()

    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0

    st = ctx.stats()
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    counts = torch.tensor([st["rays"], st["samples"], st["searches"]], dtype=torch.float64, device="cuda")
    if use_dist and backend == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    elif use_dist:
        t, counts = t.cpu(), counts.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    total_rays, total_samples, total_searches = (float(c) for c in counts.tolist())

    if rank == 0:
        if not use_dist:
            merge = "resolve" if work.kind == "2d" else "no merge step"
        elif work.kind != "2d":
            merge = "all-reduce of the f32[W*H][3] image"
        else:
            merge = "XYZ projection + all-reduce + encode" if args.merge == "xyz" else "spectral all-reduce + resolve"
        if strong:
            workload = (f"{work.describe().replace(' spp/pass', ' spp/pass at most')}; fixed job of {args.total_spp} spp split into {world_size} contiguous sample ranges "
                        f"(BASELINE configs[4]); step = the whole job + {merge}")
        else:
            workload = f"{work.describe()}; job = {args.steps} steps + {merge}"
        out = {
            "metric": "Mrays/s", "value": total_rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world_size, "ranks": dist.get_world_size() if use_dist else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "parallelism": f"sample-batch x{world_size}",
                       "mode": "f64 geometry / f32 radiance (parity mode)"},
            "samples_per_s": total_samples / dt,
            "searches_per_s": total_searches / dt,
            "rays_per_sample": total_rays / max(1.0, total_samples),
            "ray_definition": "trace_ray invocations with depth > 0; the lambda-samples of one (pixel, sample) share the "
                              "nearest-hit search of their common primary ray, so searches_per_s < value",
            "roofline": roofline(work.kind, st, work.kernel, work.n_prims, work.hbm_bytes_per_launch, work.tag),
        }
        if world_size == 1 and not strong and args.configs == "all":
            # the other BASELINE.json configurations, timed in this same process on the same GPU
            work.close()
            configs = []
            plan = [
                lambda: Work2D(rt, torch, ctx, "complex_world", 3840, 2160, 64, 2, 10),
                lambda: Work2D(rt, torch, ctx, "cornell_box", 3840, 2160, 16, 16, 10),
                lambda: Work2D(rt, torch, ctx, "cornell_box_absorption", 3840, 2160, 16, 16, 10),
                lambda: Work2D(rt, torch, ctx, "sample_world", 1920, 1080, 64, 2, 10),
                lambda: Work3D(rt, torch, ctx, 1920, 1080, 128, 10, False),
                lambda: Work3D(rt, torch, ctx, 1920, 1080, 128, 10, True),
            ]
            for make in plan:
                w = make()
                configs.append(time_config(torch, ctx, w, 3, 1))
                w.close()
                del w
                torch.cuda.empty_cache()
            out["configs"] = configs
        if args.cpu_spp > 0 and world_size == 1 and not strong and args.scene != "standalone3d":
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            p_full = params2d(args.width, args.height, args.spp, args.lambda_samples, args.depth, seed=0)
            out["cpu_baseline"] = cpu_baseline(rt.World2D(args.scene, seed=1), p_full, threads, args.cpu_spp)
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
