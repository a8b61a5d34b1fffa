"""Known answers for the CPU checker's restatement of ReverseRenderer::denoise
(2d/src/librt2d_reverse.rs:337-422, 2d/src/img.rs:167-185): the Laplace sort key with the reference's
un-centred 3x3 window and empty border, the top-fraction count, tie order, mask dilation and Blending::Add."""
import numpy as np

import raytracer_amd as rt
from raytracer_amd.api import params2d


def world():
    return rt.World2D("simple_world")


def test_laplace_key_of_a_single_bright_pixel(oracle):
    W, H = 12, 9
    merged = np.zeros((H, W, 3), np.float32)
    loop = np.zeros_like(merged)
    merged[4, 5] = (3.0, 0.0, 4.0)       # |diff| = (3, 0, 4)
    p = params2d(W, H, 1, 1, 1, seed=0)
    _, _, key, n, _ = oracle.denoise2d(world(), p, 0.0, 0, 1.0, merged, np.zeros((H, W), np.float32), loop)
    assert n == 0
    # output (x, y) sums in(x + i, y + j) k[i][j], i, j in 0..3: the bright pixel (5, 4) is seen by x in 3..5, y in 2..4,
    # with weight 8 from (4, 3) (where it sits at i = j = 1) and -1 elsewhere
    want = np.zeros((H, W), np.float32)
    for x in range(3, 6):
        for y in range(2, 5):
            k = 8.0 if (x, y) == (4, 3) else -1.0
            v = np.abs(np.float32(k) * np.array([3.0, 0.0, 4.0], np.float32))
            length = np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
            want[y, x] = np.sqrt(np.float32(length * length + length * length) + length * length)
    assert np.array_equal(key, want)
    assert key[3, 4] == np.float32(np.sqrt(np.float32(3 * 40.0 * 40.0)))


def test_border_of_three_pixels_is_never_written(oracle):
    W, H = 10, 8
    rng = np.random.default_rng(1)
    merged = rng.random((H, W, 3)).astype(np.float32)
    loop = rng.random((H, W, 3)).astype(np.float32)
    p = params2d(W, H, 1, 1, 1, seed=0)
    _, _, key, _, _ = oracle.denoise2d(world(), p, 0.0, 0, 1.0, merged, np.zeros((H, W), np.float32), loop)
    assert not key[:, W - 3:].any() and not key[H - 3:, :].any()
    assert key[: H - 3, : W - 3].all()


def test_top_count_ties_and_dilation(oracle):
    W, H = 16, 12
    merged = np.zeros((H, W, 3), np.float32)
    loop = np.zeros_like(merged)
    weight = np.full((H, W), 5.0, np.float32)
    p = params2d(W, H, 2, 1, 3, seed=4)
    # every key is 0: the order is index descending, so the top two pixels are the last two of the image
    top = 2.5 / (W * H)                    # (192 as f32 * top) as usize = 2
    rgb, w, key, n, rays = oracle.denoise2d(world(), p, top, 1, 1.5, merged, weight, loop)
    assert not key.any()
    # pixels (15, 11) and (14, 11) with a +-1 mask: x in 13..16, y in 10..12 -> 4 x 3 = 12 pixels, 6 of them inside
    assert n == 12
    changed = np.argwhere(w != 5.0)
    assert sorted(map(tuple, changed)) == [(10, 13), (10, 14), (10, 15), (11, 13), (11, 14), (11, 15)]
    assert np.all(w[w != 5.0] == np.float32(5.0 + 2 * 1.5))          # weight += spp * oversampling_factor
    # value += compute_pixel(...).to_dvec3()
    for (y, x) in changed:
        spec, _ = oracle.compute_pixel2d(world(), p, int(x), int(y))
        assert np.allclose(rgb[y, x], oracle.resolve_pixel(spec), rtol=1e-6, atol=1e-7)
    assert rays > 0
    # a larger key beats the index order
    merged[2, 3] = 1.0
    _, w2, _, n2, _ = oracle.denoise2d(world(), p, 1.5 / (W * H), 0, 1.0, merged, weight, loop)
    assert n2 == 1 and np.argwhere(w2 != 5.0).tolist() == [[1, 2]]    # the 8x response sits at (x - 1, y - 1)
