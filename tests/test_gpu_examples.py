"""The C++ front-ends (examples/*.cpp over raytracer_amd/host/*.hpp, the host-side mirror of the reference's types:
ReverseRenderer, ForwardRenderer, 3d World, worlds::*) run as programs and their PNGs are compared with the same renders
made through the Python binding of the C ABI.  Both sides end in the same kernels; what this covers is the C++ layer
between: scene construction and flattening, parameter mapping, RawImage assembly, tone mapping and the PNG writer.
Sums of f32 atomics may differ in the last bit between two runs, which can move a tone-mapped byte by one."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import raytracer_amd as rt
from raytracer_amd.api import params2d, params3d

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "bin")


def run(name, *args):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.skip(exe + " is missing: __graft_entry__.build() compiles the front-ends")
    subprocess.run([exe] + [str(a) for a in args], check=True, timeout=120, stdout=subprocess.DEVNULL)


def annotated(rgba, camera_size=(400.0, 300.0)):
    """annotate() of the reference (2d/src/rt_common.rs:23-41) on a tone-mapped image: Vec3::X under Reinhard{1.} is
    (127, 0, 0); ticks of 5 pixels at every 100th column / row as far as camera.size reaches, those outside dropped."""
    out = rgba.copy()
    h, w = out.shape[:2]
    for i in range(int(camera_size[0]) // 100):
        for j in range(5):
            if i * 100 < w and j < h:
                out[j, i * 100, :3] = (127, 0, 0)
    for j in range(int(camera_size[1]) // 100):
        for i in range(5):
            if i < w and j * 100 < h:
                out[j * 100, i, :3] = (127, 0, 0)
    return out


def close_bytes(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return d.max() <= 1 and (d != 0).mean() < 0.01


@pytest.mark.parametrize("world_name", ["cornell_box", "sample_world"])
def test_rt2d_reverse_writes_the_image_of_global_render(world_name, gpu_ctx, tmp_path):
    run("rt2d_reverse", 1, 12, world_name, 160, str(tmp_path / "cpp_"))
    png = np.asarray(Image.open(tmp_path / "cpp_0000.png"))
    assert png.shape == (120, 160, 4)
    scene = rt.Scene2D(gpu_ctx, rt.World2D(world_name, seed=1))
    rgb, _ = scene.global_render(params2d(160, 120, 12, 2, 10, seed=0))
    scene.close()
    want = annotated(gpu_ctx.tonemap_reinhard(rgb, 1.0, 255).reshape(120, 160, 4))
    assert png[..., :3].any()
    assert tuple(png[0, 100, :3]) == (127, 0, 0) and tuple(png[100, 0, :3]) == (127, 0, 0)  # annotate()'s ticks
    assert close_bytes(png, want)


def test_rt2d_reverse_renders_every_frame_of_the_animation(gpu_ctx, tmp_path):
    """The reference's shipped use: a loop-local world per frame, sample_world(t = idx / frames)
    (2d/src/bin/2d_reverse.rs:43-56).  Every frame must show ITS world: the worlds of successive frames sit at the
    same address with the same object count."""
    run("rt2d_reverse", 2, 8, "sample_world", 160, str(tmp_path / "anim_"))
    frames = [np.asarray(Image.open(tmp_path / f"anim_{k:04d}.png")) for k in range(2)]
    for k, png in enumerate(frames):
        scene = rt.Scene2D(gpu_ctx, rt.World2D("sample_world", t=k / 2, idx=k, seed=1))
        rgb, _ = scene.global_render(params2d(160, 120, 8, 2, 10, seed=0))
        scene.close()
        want = annotated(gpu_ctx.tonemap_reinhard(rgb, 1.0, 255).reshape(120, 160, 4))
        assert close_bytes(png, want), f"frame {k} is not the render of sample_world(t = {k / 2})"
    assert not close_bytes(frames[0], frames[1])


def test_host_mirror_call_semantics():
    """trace_ray draws a fresh random stream per call and trace_ray_at reproduces one; a renderer given another world
    in the same variable renders that world (examples/host_mirror_check.cpp)."""
    exe = os.path.join(BIN, "host_mirror_check")
    if not os.path.exists(exe):
        pytest.skip(exe + " is missing: __graft_entry__.build() compiles it")
    out = subprocess.run([exe], check=True, timeout=120, capture_output=True, text=True).stdout
    kv = {ln.split()[0]: ln.split()[1:] for ln in out.splitlines() if ln.strip()}
    n = int(kv["trace_ray2d_rays"][0])
    assert int(kv["trace_ray2d_at_repeats"][0]) == n and int(kv["trace_ray2d_first_call_is_stream0"][0]) == n
    # rays that meet a diffuse circle take a random bounce: many of the 64 directions must see distinct values
    assert int(kv["trace_ray2d_rays_with_distinct_calls"][0]) >= 8
    assert kv["world_slot_0"][2] == kv["world_slot_1"][2], "the check is about worlds that look alike from outside"
    assert kv["rebind_frame1_matches_fresh_renderer"] == ["1"] and kv["rebind_frames_differ"] == ["1"]
    assert int(kv["trace_ray3d_distinct_of_8"][0]) >= 5 and kv["trace_ray3d_at_repeats"] == ["1"]


def test_rt2d_forward_writes_the_image_of_forward_render(gpu_ctx, tmp_path):
    run("rt2d_forward", "complex_world", str(tmp_path / "fwd.png"))
    png = np.asarray(Image.open(tmp_path / "fwd.png"))
    assert png.shape == (600, 800, 4)
    scene = rt.Scene2D(gpu_ctx, rt.World2D("complex_world", seed=1))
    rgb, _, lines = scene.forward_render(scene.forward_params(800, 600, 1000, 5))  # ForwardRenderer: 1000 rays per emitter, depth 5
    scene.close()
    want = gpu_ctx.tonemap_reinhard(rgb, 1.0, 255).reshape(600, 800, 4)
    assert lines > 0 and png[..., :3].any()
    assert close_bytes(png, want)


@pytest.mark.parametrize("use_bvh", [0, 1])
def test_rt3d_standalone_writes_the_image_of_render(use_bvh, gpu_ctx, tmp_path):
    run("rt3d_standalone", 6, 64, 48, use_bvh, str(tmp_path / "s3d.png"))
    png = np.asarray(Image.open(tmp_path / "s3d.png"))
    assert png.shape == (48, 64, 4)
    s = rt.Standalone3D(64, 48, use_bvh=bool(use_bvh))
    scene = rt.Scene3D(gpu_ctx, s)
    img = scene.render(params3d(6, 10, seed=0))
    want = gpu_ctx.tonemap_reinhard(img, 1.0, 254).reshape(48, 64, 4)
    assert png[..., :3].any()
    assert close_bytes(png, want)


@pytest.mark.parametrize("device_loop", [0, 1])
def test_rt2d_reverse_progressive_loops_with_adaptive_passes(device_loop, tmp_path):
    """endless_render (host merge as the reference does it) and the device-resident loop, two loops with the
    re-sampling pass: the programs finish and their images show the same scene (the two merges differ by design:
    encoded colours against linear sums, so the comparison is the rank order of block luminance)."""
    from scipy.stats import spearmanr
    run("rt2d_reverse", 1, 32, "cornell_box_absorption", 160, str(tmp_path / "loop_"), 2, 1, device_loop)
    png = np.asarray(Image.open(tmp_path / "loop_0000.png")).astype(np.float64)
    run("rt2d_reverse", 1, 64, "cornell_box_absorption", 160, str(tmp_path / "one_"))
    one = np.asarray(Image.open(tmp_path / "one_0000.png")).astype(np.float64)
    assert png.shape == one.shape == (120, 160, 4)
    a = png[..., :3].reshape(15, 8, 20, 8, 3).mean(axis=(1, 3, 4))
    b = one[..., :3].reshape(15, 8, 20, 8, 3).mean(axis=(1, 3, 4))
    lit = b > 2.0
    assert lit.sum() > 50
    assert spearmanr(a[lit], b[lit])[0] > 0.95
