"""Statistics that compare a render with one of the reference's gallery images (tests/golden/gallery_lowres.npz, made
by tests/golden/make_gallery.py).  The gallery images went through the reference's GUI: unseeded generator, unknown
loop count.  Two comparisons on 30x40 block means:
 * compare(): ORDER and HUE, which survive any monotone tone map - the rank correlation of luminance over the blocks
   the reference shows lit, and, where the reference is colourful, which of R, G, B dominates the block;
 * compare_bytes(): the tone-mapped bytes themselves.  Every gallery image has 127 as its largest byte, i.e. channel
   values capped at 1 (palette's clamped `into_color`) under Reinhard with parameter 1, so a converged render encoded
   the same way can be laid over the reference's image directly."""
import os

import numpy as np
from scipy.stats import spearmanr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gallery_lowres.npz")
LUMA = np.array([0.2126, 0.7152, 0.0722])


def block_means(img, rows=30, cols=40):
    h, w = img.shape[:2]
    return img[: h // rows * rows, : w // cols * cols].reshape(rows, h // rows, cols, w // cols, -1).mean(axis=(1, 3))


def gallery(world):
    with np.load(GOLDEN) as z:
        return z[world]


def compare(linear_rgb, ref_blocks):
    """linear_rgb: [600][800][3] linear sRGB means of a render (before tone mapping).  Returns (rank correlation of
    luminance over the reference's lit blocks, number of colourful blocks, fraction of them whose dominant channel
    agrees, largest |luminance| among the blocks the reference shows black relative to the brightest block)."""
    ours = block_means(np.asarray(linear_rgb, np.float64))
    lum_o, lum_r = ours @ LUMA, ref_blocks.astype(np.float64) @ LUMA
    lit = lum_r > 0.03
    rho = float(spearmanr(lum_o[lit], lum_r[lit])[0])
    colourful = (ref_blocks.max(axis=2) - ref_blocks.min(axis=2)) > 0.15
    agree = float((ours.argmax(axis=2) == ref_blocks.argmax(axis=2))[colourful].mean()) if colourful.any() else 1.0
    black = lum_r < 0.004
    leak = float(np.abs(lum_o[black]).max() / lum_o.max()) if black.any() else 0.0
    return rho, int(colourful.sum()), agree, leak


def compare_bytes(rgba8, ref_blocks):
    """rgba8: [600][800][4] bytes of a render after Reinhard(1).  Returns (mean |difference| of the block means in
    units of full scale, their Pearson correlation, the largest colour byte)."""
    img = np.asarray(rgba8)[..., :3]
    ours = block_means(img.astype(np.float64) / 255.0)
    ref = ref_blocks.astype(np.float64)
    return float(np.abs(ours - ref).mean()), float(np.corrcoef(ours.ravel(), ref.ravel())[0, 1]), int(img.max())
