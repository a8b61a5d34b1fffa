"""A world that went through the reference's checkpoint format (2d/src/bin/gui.rs:24-45, MessagePack as rmp-serde
writes it) traces on the GPU exactly like the world it was saved from: decode -> flatten -> rt2d_scene_create ->
rt2d_trace_batch, compared bit for bit with the built world's batch and with the CPU checker.  The reference ships no
checkpoint file, so the bytes come from this repository's writer (pinned against the format rules in
tests/test_checkpoint.py): parity with the reference's own files stays unpinned."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d

pytestmark = pytest.mark.gpu


def batch(n, seed):
    rng = np.random.default_rng(seed)
    ang = rng.random(n) * 2 * np.pi
    return rng.random(n) * 800, rng.random(n) * 600, np.cos(ang), np.sin(ang), rng.integers(0, 95, n)


@pytest.mark.parametrize("name", ["sample_world", "complex_world", "cornell_box", "cornell_box_absorption", "spectrum_world"])
def test_decoded_world_traces_like_the_built_world(name, gpu_ctx, oracle):
    built = rt.World2D(name, seed=1)
    loaded = rt.World2D.from_bytes(built.to_bytes())
    assert loaded.n_prims == built.n_prims
    ox, oy, dx, dy, li = batch(20000, 11)
    a_scene, b_scene = rt.Scene2D(gpu_ctx, built), rt.Scene2D(gpu_ctx, loaded)
    a = a_scene.trace_batch(ox, oy, dx, dy, li, 10, seed=3)
    b = b_scene.trace_batch(ox, oy, dx, dy, li, 10, seed=3)
    a_scene.close()
    b_scene.close()
    for x, y, what in zip(a, b, ("radiance", "first primitive", "first distance", "bounces")):
        assert np.array_equal(x, y), f"{what} differs between the built world and its decoded checkpoint"
    want = oracle.trace_batch2d(loaded, ox, oy, dx, dy, li, 10, seed=3)
    assert np.array_equal(b[1], want[1]) and np.array_equal(b[3], want[3]) and np.array_equal(b[2], want[2])
    assert np.allclose(b[0], want[0], rtol=1e-5, atol=1e-12)


def test_checkpoint_file_with_image_renders_on_the_gpu(gpu_ctx, tmp_path):
    """save_checkpoint(world, RawImage of a GPU render) -> load_checkpoint: the image planes come back bit for bit and
    the loaded world renders the same image again (same seed, same samples; f32 atomics may move a last bit)."""
    built = rt.World2D("cornell_box", seed=1)
    scene = rt.Scene2D(gpu_ctx, built)
    p = params2d(64, 48, 8, 2, 10, seed=5)
    rgb, weight = scene.global_render(p)
    scene.close()
    path = tmp_path / "checkpoint.data"
    built.save_checkpoint(path, rgb, weight)
    loaded = rt.World2D.load_checkpoint(path)
    lrgb, lweight = loaded.image
    assert np.array_equal(lrgb, rgb) and np.array_equal(lweight, weight)
    scene2 = rt.Scene2D(gpu_ctx, loaded)
    rgb2, weight2 = scene2.global_render(p)
    scene2.close()
    assert np.array_equal(weight2, weight)
    assert np.allclose(rgb2, rgb, rtol=2e-4, atol=2e-5)
