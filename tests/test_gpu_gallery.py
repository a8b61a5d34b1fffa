"""GPU renders against the reference's own gallery images (2d/README.md:20-53), the only outputs of the reference
that exist.  They came out of its GUI with an unseeded generator, so the comparison is statistical (rank order of
block luminance, dominant colour channel; tests/gallery_stats.py), on the same worlds and camera at 800x600 with
enough samples that noise does not blur the order.  sample_scene.png was taken from the animated world at an unknown
time, hence its lower bound."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d
from tests.gallery_stats import compare, compare_bytes, gallery

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world_name,min_rho,min_hue,max_leak", [
    ("sample_world", 0.90, None, None),
    ("cornell_box", 0.97, None, 0.02),
    ("cornell_box_absorption", 0.98, 0.97, 0.02),
    ("spectrum_world", 0.97, 0.95, 0.02),
])
def test_render_resembles_gallery_image(world_name, min_rho, min_hue, max_leak, gpu_ctx):
    world = rt.World2D(world_name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    p = params2d(800, 600, 512, 4, 10, seed=5)
    rgb, weight = scene.global_render(p, want_spectrum=False)
    scene.close()
    assert rgb.shape == (600, 800, 3) and np.all(weight == 512)
    rho, n_colour, hue, leak = compare(rgb, gallery(world_name))
    print(world_name, "rank correlation %.3f, colourful blocks %d, hue agreement %.3f, leak %.4f" % (rho, n_colour, hue, leak))
    assert rho >= min_rho, rho
    if min_hue is not None:
        assert n_colour > 20 and hue >= min_hue, (n_colour, hue)
    if max_leak is not None:
        assert leak < max_leak, leak


@pytest.mark.parametrize("world_name,max_mae,min_corr", [
    ("cornell_box_absorption", 0.006, 0.999),   # measured 0.0021 (half a byte level) and 0.9999
    ("spectrum_world", 0.012, 0.997),           # 0.0056, 0.9990
    ("cornell_box", 0.04, 0.98),                # 0.024, 0.990: the lit floor is brighter in the gallery image
    ("sample_world", 0.06, 0.80),               # 0.034, 0.87: animated world, time unknown
])
def test_tone_mapped_render_lies_over_the_gallery_image(world_name, max_mae, min_corr, gpu_ctx):
    """The reference's own output, byte for byte at low resolution: a converged render through `to_dvec3` (clamped sRGB)
    and Reinhard(1), against the gallery image of the same world."""
    world = rt.World2D(world_name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    rgb, _ = scene.global_render(params2d(800, 600, 2048, 8, 10, seed=6))
    scene.close()
    assert rgb.min() >= 0.0 and rgb.max() <= 1.0          # palette's `into_color` clamps
    rgba = gpu_ctx.tonemap_reinhard(rgb, 1.0, 255).reshape(600, 800, 4)
    mae, corr, top = compare_bytes(rgba, gallery(world_name))
    print(world_name, "block MAE %.4f of full scale, correlation %.4f, largest byte %d" % (mae, corr, top))
    assert top == 127                                      # as in every gallery image
    assert mae < max_mae, mae
    assert corr > min_corr, corr
