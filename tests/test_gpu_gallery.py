"""GPU renders against the reference's own gallery images (2d/README.md:20-53), the only outputs of the reference
that exist.  They came out of its GUI with an unseeded generator, so the comparison is statistical (rank order of
block luminance, dominant colour channel; tests/gallery_stats.py), on the same worlds and camera at 800x600 with
enough samples that noise does not blur the order.  sample_scene.png was taken from the animated world at an unknown
time, hence its lower bound."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d
from tests.gallery_stats import compare, gallery

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
