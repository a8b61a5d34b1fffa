"""The CPU restatement against the reference's own gallery images (the only outputs of the reference there are):
same worlds, same camera, 800x600.  Statistical, not bit-level - see tests/gallery_stats.py - and at the sample
counts a CPU test can afford; tests/test_gpu_gallery.py repeats it on the GPU with enough samples for tight bounds."""
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d
from tests.gallery_stats import compare, gallery


@pytest.mark.parametrize("world_name,spp,min_rho,min_hue", [
    ("cornell_box_absorption", 48, 0.88, 0.85),
    ("spectrum_world", 32, 0.93, 0.85),
])
def test_checker_render_resembles_gallery_image(world_name, spp, min_rho, min_hue, oracle):
    world = rt.World2D(world_name, seed=1)
    p = params2d(800, 600, spp, 4, 10, seed=3)
    rgb, _, _, _ = oracle.render2d(world, p, threads=8, want_spectrum=False)
    rho, n_colour, hue, leak = compare(rgb, gallery(world_name))
    assert rho >= min_rho, rho
    assert n_colour > 20 and hue >= min_hue, (n_colour, hue)
    assert leak < 0.05, leak
