"""Low-resolution copies of the gallery images the reference holds (2d/README.md:20-53), the only outputs of
the reference that exist: 800x600 RGBA8 PNGs rendered through its GUI with an unseeded generator and unknown
loop counts.  Each is reduced to 30x40 block means of its 8-bit RGB values / 255 (20x20-pixel blocks) and stored in
gallery_lowres.npz beside the name of the world that produced it (2d/src/worlds.rs); the denoiser's three images and
the stress-test image, which no world of today's code reproduces, are kept as the statistics that can still be compared.  Run in the build container,
where /root/reference exists:  python tests/golden/make_gallery.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/2d"
# world constructor (2d/src/worlds.rs) -> the gallery image it produced (2d/README.md)
IMAGES = {
    "sample_world": "sample_scene.png",
    "cornell_box": "diffration_dispersion.png",
    "cornell_box_absorption": "absorption.png",
    "spectrum_world": "visible_spectrum2.png",
}


def block_means(img, rows=30, cols=40):
    h, w = img.shape[:2]
    return img[: h // rows * rows, : w // cols * cols].reshape(rows, h // rows, cols, w // cols, -1).mean(axis=(1, 3))


LUMA = np.array([0.2126, 0.7152, 0.0722], np.float32)


def load(png):
    img = np.asarray(Image.open(os.path.join(REF, png)).convert("RGB"), np.float32)
    assert img.shape == (600, 800, 3), img.shape
    return img


def denoiser_images():
    """The three images the reference's adaptive re-sampling pass leaves behind (2d/README.md:30-39, written by
    librt2d_reverse.rs:355-360, :400-405, :492) - its only outputs of that path.  They show an earlier cornell_box
    (one lens, the Cauchy one at the lower left; today's world has a second lens at the lower right), so comparisons
    use the left half of the image, where the two worlds agree.
      denoise_laplace.png  Reinhard(1) of the sort key, splat on three channels: key = byte / (255 - byte)
      denoise_mask.png     the merged image with the pixels to recompute painted (1, 0, 1) -> bytes (127, 0, 127)
      denoise_heatmap.png  the per-pixel weight through colorgrad's inferno map, which rises monotonically in luminance"""
    lap = load("denoise_laplace.png")[..., 0]
    key = lap / (255.0 - lap)
    mask = load("denoise_mask.png")
    magenta = (np.abs(mask[..., 0] - 127) <= 1) & (mask[..., 1] <= 1) & (np.abs(mask[..., 2] - 127) <= 1)
    heat = load("denoise_heatmap.png") @ LUMA
    return {"denoise_key_blocks": block_means(key[..., None])[..., 0].astype(np.float32),
            "denoise_mask_blocks": block_means(magenta[..., None].astype(np.float32))[..., 0].astype(np.float32),
            "denoise_heat_blocks": block_means(heat[..., None])[..., 0].astype(np.float32)}


def stress_statistics():
    """stress_test.png (2d/README.md:52-53) is complex_world (2d/src/worlds.rs:319-343) with circles and materials drawn
    from an unseeded generator: no world reproduces it, so what is kept are statistics of the tone-mapped bytes that any
    member of that class of worlds shares - the share of black pixels (the inside of a mirror or a diffuse circle
    returns 0, librt2d_reverse.rs:146-207), the luminance quantiles of the rest relative to their median, the mean
    colour - and the block means for the record."""
    img = load("stress_test.png")
    lum = img @ LUMA
    lit = lum[lum >= 8.0]
    q = np.percentile(lit, [5, 25, 50, 75, 95])
    return {"stress_blocks": block_means(img / 255.0).astype(np.float32),
            "stress_black_fraction": np.float32((lum < 8.0).mean()),
            "stress_lit_quantiles_5_25_50_75_95": q.astype(np.float32),
            "stress_mean_rgb": img.mean(axis=(0, 1)).astype(np.float32),
            "stress_max_byte": np.float32(img.max())}


def main():
    out = {}
    for world, png in IMAGES.items():
        img = load(png) / 255.0
        out[world] = block_means(img).astype(np.float32)
    out["spectrum_world_first_version"] = block_means(load("visible_spectrum.png") / 255.0).astype(np.float32)
    out.update(denoiser_images())
    out.update(stress_statistics())
    np.savez_compressed(os.path.join(HERE, "gallery_lowres.npz"), **out)
    print({k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
