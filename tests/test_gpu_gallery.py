"""GPU renders against the reference's own gallery images (2d/README.md:20-53), the only outputs of the reference
that exist.  They came out of its GUI with an unseeded generator, so the comparison is statistical (rank order of
block luminance, dominant colour channel; tests/gallery_stats.py), on the same worlds and camera at 800x600 with
enough samples that noise does not blur the order.  sample_scene.png was taken from the animated world at an unknown
time, hence its lower bound."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d
from tests.gallery_stats import compare, compare_bytes, denoiser_structure, fixture, gallery, laplace_key, stress_statistics

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


@pytest.mark.parametrize("world_name,spp,ls,max_mae,min_corr", [
    ("cornell_box_absorption", 2048, 8, 0.005, 0.999),  # measured 0.0014 (a third of a byte level), 0.9999
    ("spectrum_world", 500, 2, 0.008, 0.998),           # 0.0027, 0.9996 (converged: 0.0051)
    ("cornell_box", 500, 2, 0.018, 0.995),              # 0.0088, 0.9977 (converged: 0.023)
    ("sample_world", 500, 2, 0.06, 0.80),               # 0.033, 0.87: the image predates the world the code builds
])
def test_tone_mapped_render_lies_over_the_gallery_image(world_name, spp, ls, max_mae, min_corr, gpu_ctx):
    """The reference's own output, byte for byte at low resolution: a render through `to_dvec3` (clamped sRGB) and
    Reinhard(1) against the gallery image of the same world.  The gallery images are single passes (their largest byte
    is 127; `merged + loop` would exceed it) whose sample counts differ: `to_dvec3` clamps each pixel's noisy channels
    at 0, which brightens dark areas by an amount that depends on spp.  absorption.png is a converged image; the other
    two agree best at the shipped binary's 500 spp x 2 wavelengths (bin/2d_reverse.rs:16-20) - the sampling noise of
    the estimator is reproduced along with its mean."""
    world = rt.World2D(world_name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    rgb, _ = scene.global_render(params2d(800, 600, spp, ls, 10, seed=6))
    scene.close()
    assert rgb.min() >= 0.0 and rgb.max() <= 1.0          # palette's `into_color` clamps
    rgba = gpu_ctx.tonemap_reinhard(rgb, 1.0, 255).reshape(600, 800, 4)
    mae, corr, top = compare_bytes(rgba, gallery(world_name))
    print(world_name, "block MAE %.4f of full scale, correlation %.4f, largest byte %d" % (mae, corr, top))
    assert top == 127                                      # as in every gallery image
    assert mae < max_mae, mae
    assert corr > min_corr, corr


def test_stress_scene_has_the_statistics_of_the_gallery_image(gpu_ctx):
    """stress_test.png (2d/README.md:52-53) came from complex_world with an unseeded generator: no world reproduces it.
    What every member of that class shares is compared: the share of black pixels (the inside of the ~100 mirror and
    diffuse circles of 200, librt2d_reverse.rs:146-207: measured 0.0139 against the image's 0.0138), the spread of the lit
    pixels' luminance around its median, and the mean colour ratios (flat-spectrum emitters through to_dvec3 and
    Reinhard).  The image is about a tenth brighter overall than this seed's world (its number of emitters is random)."""
    scene = rt.Scene2D(gpu_ctx, rt.World2D("complex_world", seed=1))
    rgb, _ = scene.global_render(params2d(800, 600, 500, 2, 10, seed=3))       # the shipped binary's sample counts
    scene.close()
    st = stress_statistics(gpu_ctx.tonemap_reinhard(rgb, 1.0, 255).reshape(600, 800, 4))
    ref_q, ref_rgb = fixture("stress_lit_quantiles_5_25_50_75_95"), fixture("stress_mean_rgb")
    print("stress:", st, "reference:", float(fixture("stress_black_fraction")), ref_q, ref_rgb)
    assert st["max_byte"] == 127 == int(fixture("stress_max_byte"))
    assert 0.7 < st["black_fraction"] / float(fixture("stress_black_fraction")) < 1.4
    q = st["quantiles"]
    assert abs(q[2] / ref_q[2] - 1.0) < 0.15                                   # overall level
    assert abs(q[0] / q[2] - ref_q[0] / ref_q[2]) < 0.08 and abs(q[4] / q[2] - ref_q[4] / ref_q[2]) < 0.04
    m = st["mean_rgb"]
    assert abs(m[0] / m[1] - ref_rgb[0] / ref_rgb[1]) < 0.03 and abs(m[2] / m[1] - ref_rgb[2] / ref_rgb[1]) < 0.03


def test_denoiser_selects_where_the_reference_images_show_noise(gpu_ctx):
    """The only outputs of ReverseRenderer::denoise the reference holds (2d/README.md:30-39): the key plane it sorts by
    (denoise_laplace.png), the pixels it marks (denoise_mask.png) and the accumulated weights (denoise_heatmap.png), of
    an earlier cornell_box with one lens.  On the image half the two worlds share: the key plane computed from two GPU
    passes ranks the blocks as the reference's does, the blocks the reference marked rank high in it, the pixels
    rt2d_denoise re-samples concentrate where the reference's key is large, and the sample counts of the
    device-resident loop pile up where the reference's heat map is hot."""
    scene = rt.Scene2D(gpu_ctx, rt.World2D("cornell_box", seed=1))
    spp = 8
    first = scene.global_render(params2d(800, 600, spp, 2, 10, seed=9, sample_offset=0))
    loop = scene.global_render(params2d(800, 600, spp, 2, 10, seed=9, sample_offset=spp))
    merged = (first[0] + loop[0]).astype(np.float32)            # `merged_image + loop_image` (librt2d_reverse.rs:467)
    key = laplace_key(merged, loop[0])
    _, w, n = scene.denoise(params2d(800, 600, spp, 2, 10, seed=9, sample_offset=2 * spp), 0.01, 2, 10.0, merged,
                            np.zeros((600, 800), np.float32), loop[0])   # the shipped Denoiser (bin/2d_reverse.rs:22-27)
    g = scene.progressive(params2d(800, 600, spp, 2, 10, seed=4), denoiser=(0.01, 2, 10.0, 1))
    for _ in range(8):
        g.step()
    count = g.read(sum=False, merged=False, loop=False)[1]
    g.close()
    scene.close()
    s = denoiser_structure(key_plane=key, selected=w > 0, count=count)
    print("denoiser structure:", s, "recomputed", n)
    assert s["key_rank_corr"] > 0.88          # measured 0.94 .. 0.96
    assert s["mask_auc"] > 0.8
    assert s["selected_vs_key_rank_corr"] > 0.8   # measured 0.92
    assert s["count_vs_heat_rank_corr"] > 0.6 and s["hot_ratio"] > 1.4   # measured 0.77, 1.95
