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
    want = gpu_ctx.tonemap_reinhard(rgb, 1.0, 255).reshape(120, 160, 4)
    assert png[..., :3].any()
    assert close_bytes(png, want)


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
