"""Converged images: the GPU render and the CPU checker drawn from DIFFERENT random streams (different seeds, and far
more samples on the GPU) must agree as estimates of the same image.  RMSE of the linear RGB over small blocks of
pixels, relative to the image's mean; the thresholds are two to three times what the checker's own sampling noise
leaves at these sample counts (measured: 0.009 / 0.055 / 0.027 for the three 2d worlds, 0.023 in 3d; a second GPU
render from a third stream shows the GPU's own share is less than half of that), and a deliberately wrong render
(recursion cut short) misses them by a factor of two or more.  This is the statistical check beside
the same-stream tests (tests/test_gpu_render2d.py, tests/test_gpu_3d.py), which compare sample for sample."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import params2d, params3d

pytestmark = pytest.mark.gpu


def blocks(img, n):
    h, w = img.shape[:2]
    return np.asarray(img, np.float64).reshape(h // n, n, w // n, n, -1).mean(axis=(1, 3))


def rel_rmse(a, b, n):
    """RMSE over n x n-pixel block means (the estimators are heavy-tailed: single pixels converge slowly), relative
    to the mean of the second image."""
    a, b = blocks(a, n), blocks(b, n)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.mean(np.abs(b)))


@pytest.mark.parametrize("name,threshold", [("complex_world", 0.025), ("cornell_box_absorption", 0.12), ("sample_world", 0.06)])
def test_converged_2d_image_agrees_across_random_streams(name, threshold, gpu_ctx, oracle):
    world = rt.World2D(name, seed=1)
    scene = rt.Scene2D(gpu_ctx, world)
    w, h = 80, 60
    gpu_a, _ = scene.global_render(params2d(w, h, 16384, 4, 10, seed=101))
    gpu_b, _ = scene.global_render(params2d(w, h, 16384, 4, 10, seed=202))
    cpu, _, _, _ = oracle.render2d(world, params2d(w, h, 2048, 4, 10, seed=303), threads=8, want_spectrum=False)
    wrong, _ = scene.global_render(params2d(w, h, 16384, 4, 2, seed=101))
    scene.close()
    e_gpu, e_cpu, e_wrong = rel_rmse(gpu_a, gpu_b, 4), rel_rmse(gpu_a, cpu, 4), rel_rmse(wrong, cpu, 4)
    print(name, "relative RMSE: GPU vs GPU %.4f, GPU vs checker %.4f, depth-2 render vs checker %.4f" % (e_gpu, e_cpu, e_wrong))
    assert e_gpu < threshold / 2
    assert e_cpu < threshold
    assert e_wrong > 2 * threshold


@pytest.mark.parametrize("use_bvh", [False, True])
def test_converged_3d_image_agrees_across_random_streams(use_bvh, gpu_ctx, oracle):
    s = rt.Standalone3D(64, 48, use_bvh=use_bvh)
    scene = rt.Scene3D(gpu_ctx, s)
    # the 3d crate's compute_pixel SUMS its samples (3d/src/rt.rs:238-249): compare per-sample means
    gpu_a = scene.render(params3d(8192, 10, seed=11)) / 8192.0
    gpu_b = scene.render(params3d(8192, 10, seed=12)) / 8192.0
    wrong = scene.render(params3d(8192, 1, seed=11)) / 8192.0
    cpu = oracle.render3d(s, params3d(2048, 10, seed=13))[0] / 2048.0
    e_gpu, e_cpu, e_wrong = rel_rmse(gpu_a, gpu_b, 8), rel_rmse(gpu_a, cpu, 8), rel_rmse(wrong, cpu, 8)
    print("3d bvh=%s relative RMSE: GPU vs GPU %.4f, GPU vs checker %.4f, one-bounce render vs checker %.4f" % (use_bvh, e_gpu, e_cpu, e_wrong))
    assert e_gpu < 0.03
    assert e_cpu < 0.08
    assert e_wrong > 0.16
