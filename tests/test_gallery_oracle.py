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


def test_checker_denoiser_key_plane_has_the_structure_of_the_reference_image(oracle):
    """denoise_laplace.png / denoise_mask.png (2d/README.md:30-36) are the reference's own picture of the sort key and of
    the pixels it marked, for an earlier cornell_box with one lens.  On the image half the worlds share, the checker's
    key plane ranks the blocks as the picture does, and the blocks the reference marked rank high in it."""
    import numpy as np
    from tests.gallery_stats import denoiser_structure, laplace_key
    world = rt.World2D("cornell_box", seed=1)
    spp = 4
    first = oracle.render2d(world, params2d(800, 600, spp, 2, 10, seed=9, sample_offset=0), threads=8, want_spectrum=False)[0]
    loop = oracle.render2d(world, params2d(800, 600, spp, 2, 10, seed=9, sample_offset=spp), threads=8, want_spectrum=False)[0]
    merged = (first + loop).astype(np.float32)
    p = params2d(800, 600, 1, 1, 1, seed=9, sample_offset=2 * spp)           # the re-sampling itself is not the point here
    _, w, key, n, _ = oracle.denoise2d(world, p, 0.01, 2, 10.0, merged, np.zeros((600, 800), np.float32), loop)
    assert np.allclose(key, laplace_key(merged, loop), rtol=1e-5, atol=1e-6)   # the numpy statement the GPU test uses
    s = denoiser_structure(key_plane=key, selected=w > 0)
    assert s["key_rank_corr"] > 0.85 and s["mask_auc"] > 0.8 and s["selected_vs_key_rank_corr"] > 0.8, s


def test_checker_stress_scene_has_the_colour_and_black_share_of_the_gallery_image(oracle):
    """stress_test.png against complex_world(seed 1) on the checker at 48 spp (what a CPU test can afford): the share of
    black pixels and the mean colour ratios; the GPU test repeats it at the shipped 500 spp with tighter bounds."""
    import numpy as np
    from tests.gallery_stats import fixture, stress_statistics
    world = rt.World2D("complex_world", seed=1)
    rgb = oracle.render2d(world, params2d(800, 600, 48, 2, 10, seed=3), threads=8, want_spectrum=False)[0]
    st = stress_statistics(oracle.reinhard(rgb.reshape(-1, 3), 1.0, 255).reshape(600, 800, 4))
    ref_rgb = fixture("stress_mean_rgb")
    assert st["max_byte"] == 127
    assert 0.7 < st["black_fraction"] / float(fixture("stress_black_fraction")) < 1.6, st
    m = st["mean_rgb"]
    assert abs(m[0] / m[1] - ref_rgb[0] / ref_rgb[1]) < 0.04 and abs(m[2] / m[1] - ref_rgb[2] / ref_rgb[1]) < 0.04, st
