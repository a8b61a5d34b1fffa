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


# ---- the denoiser's images and the stress-test image (tests/golden/make_gallery.py says what each fixture holds) ----
LEFT = slice(0, 20)   # block columns of the image half where today's cornell_box and the pictured one agree


def fixture(name):
    with np.load(GOLDEN) as z:
        return z[name]


def laplace_key(merged_rgb, loop_rgb):
    """The reference's sort key as an image (librt2d_reverse.rs:346-354, img.rs:167-185), restated in numpy for planes
    [H][W][3]: |merged - loop|, the un-centred 3x3 Laplace window over x < W - 3, y < H - 3, length of the splat length."""
    d = np.abs(np.asarray(merged_rgb, np.float32) - np.asarray(loop_rgb, np.float32))
    h, w = d.shape[:2]
    k = np.zeros_like(d)
    for i in range(3):
        for j in range(3):
            k[:h - 3, :w - 3] += d[j:j + h - 3, i:i + w - 3] * np.float32(8.0 if (i == 1 and j == 1) else -1.0)
    return np.sqrt(np.float32(3.0)) * np.linalg.norm(k, axis=2)


def denoiser_structure(key_plane=None, selected=None, count=None):
    """Agreement in STRUCTURE with the reference's denoiser images on the left image half (its absolute scale depends on
    unknown sample counts and thresholds): rank correlation of our key plane's block means with denoise_laplace.png's;
    how well our key ranks the blocks denoise_mask.png marks against those it does not (area under the ROC curve); rank
    correlation of a per-pixel sample count with denoise_heatmap.png, and the mean count over the reference's hottest
    tenth of the blocks relative to the rest."""
    out = {}
    ref_key, ref_mask, ref_heat = (fixture(n)[:, LEFT] for n in ("denoise_key_blocks", "denoise_mask_blocks", "denoise_heat_blocks"))
    if key_plane is not None:
        kb = block_means(np.asarray(key_plane, np.float64)[..., None])[..., 0][:, LEFT]
        out["key_rank_corr"] = float(spearmanr(kb.ravel(), ref_key.ravel())[0])
        marked, rest = kb[ref_mask > 0], kb[ref_mask == 0]
        out["mask_auc"] = float((marked[:, None] > rest[None, :]).mean() + 0.5 * (marked[:, None] == rest[None, :]).mean())
    if selected is not None:
        sb = block_means(np.asarray(selected, np.float64)[..., None])[..., 0][:, LEFT]
        out["selected_vs_key_rank_corr"] = float(spearmanr(sb.ravel(), ref_key.ravel())[0])
    if count is not None:
        cb = block_means(np.asarray(count, np.float64)[..., None])[..., 0][:, LEFT]
        out["count_vs_heat_rank_corr"] = float(spearmanr(cb.ravel(), ref_heat.ravel())[0])
        hot = ref_heat > np.percentile(ref_heat, 90)
        out["hot_ratio"] = float(cb[hot].mean() / cb[~hot].mean())
    return out


def stress_statistics(rgba8):
    """The statistics make_gallery.py keeps of stress_test.png, of a render's tone-mapped bytes [600][800][4]."""
    img = np.asarray(rgba8)[..., :3].astype(np.float64)
    lum = img @ LUMA
    lit = lum[lum >= 8.0]
    return {"black_fraction": float((lum < 8.0).mean()), "quantiles": np.percentile(lit, [5, 25, 50, 75, 95]),
            "mean_rgb": img.mean(axis=(0, 1)), "max_byte": int(img.max())}
