"""Known answers for the CPU checker's restatement of the forward light tracer (2d/src/librt2d_forward.rs) and of
the line iterator it draws with (line_drawing::XiaolinWu<f64, i32>, crate line_drawing 1.0.1, not in the reference
tree: restated from its published source, so these answers pin the restatement, not the crate)."""
import numpy as np

import raytracer_amd as rt


def test_axis_parallel_lines(oracle):
    cov, hits = oracle.xiaolin_wu((1.0, 2.0), (5.0, 2.0), 8, 6)
    want = np.zeros((6, 8), np.float32)
    want[2, 1:6] = 1.0                                   # integer y: one pixel per column, full coverage
    assert np.array_equal(cov, want) and np.array_equal(hits, want)
    cov, hits = oracle.xiaolin_wu((0.0, 0.5), (3.0, 0.5), 8, 6)
    want = np.zeros((6, 8), np.float32)
    want[0, 0:4] = 0.5                                   # fractional y: the pixel above and the one below share it
    want[1, 0:4] = 0.5
    assert np.array_equal(cov, want) and np.array_equal(hits, (want > 0).astype(np.float32))
    # a vertical line is "steep": the roles of x and y swap
    cov, _ = oracle.xiaolin_wu((2.25, 1.0), (2.25, 4.0), 8, 6)
    want = np.zeros((6, 8), np.float32)
    want[1:5, 2] = 0.75
    want[1:5, 3] = 0.25
    assert np.array_equal(cov, want)


def test_diagonal_direction_and_clipping(oracle):
    a, _ = oracle.xiaolin_wu((0.0, 0.0), (4.0, 2.0), 8, 6)
    b, _ = oracle.xiaolin_wu((4.0, 2.0), (0.0, 0.0), 8, 6)          # drawn from the other end: same pixels
    assert np.array_equal(a, b)
    want = np.zeros((6, 8), np.float32)
    for x in range(5):                                   # y = x / 2: whole on even x, split on odd x
        y = x / 2
        if y == int(y):
            want[int(y), x] = 1.0
        else:
            want[int(y), x] = 0.5
            want[int(y) + 1, x] = 0.5
    assert np.array_equal(a, want)
    # pixels outside the image are dropped (draw_pixel's Err), the rest is unchanged
    c, _ = oracle.xiaolin_wu((-3.0, -1.5), (20.0, 10.0), 8, 6)
    assert c.sum() > 0 and c.shape == (6, 8)
    # each column of a shallow line carries total coverage 1
    d, _ = oracle.xiaolin_wu((0.0, 0.3), (7.0, 3.1), 8, 6)
    assert np.allclose(d.sum(axis=0), 1.0, atol=1e-6)


def test_forward_render_draws_only_escaping_last_rays(oracle):
    world = rt.World2D("simple_world")
    rc, rgb, weight, lines, n, rays = oracle.forward_render(world, 200, 150, rays_per_emitter=300, depth=5, seed=2)
    assert rc == 0
    esc = lines[lines[:, 0] == 1.0]
    assert len(esc) == n and 0 < n <= 300 and rays >= 300
    assert np.allclose(np.hypot(esc[:, 3], esc[:, 4]), 1.0, atol=1e-12)
    assert weight.min() == 1.0 and weight.max() > 1.0                # RawImage::new weights + one per drawn item
    assert np.array_equal(rgb[..., 0], rgb[..., 2]) and rgb.sum() > 0
    # passes with disjoint ray ranges are different rays
    rc2, rgb2, *_ = oracle.forward_render(world, 200, 150, rays_per_emitter=300, depth=5, seed=2, sample_offset=300)
    assert rc2 == 0 and not np.array_equal(rgb, rgb2)
    # depth 0: nothing is traced, nothing drawn
    rc0, rgb0, w0, _, n0, rays0 = oracle.forward_render(world, 200, 150, rays_per_emitter=10, depth=0, seed=2)
    assert rc0 == 0 and n0 == 0 and rays0 == 0 and not rgb0.any() and np.all(w0 == 1.0)


def segment_emitter_world():
    """A diffuse circle lit by an emissive SEGMENT: the reference's sample_out_ray is todo!() for it."""
    z = np.zeros(0)
    return rt.ArrayWorld2D(dict(
        circle_cx=[50.0], circle_cy=[40.0], circle_r=[5.0], circle_obj=[0], circle_pid=[0],
        seg_ax=[0.0], seg_ay=[0.0], seg_bx=[10.0], seg_by=[0.0], seg_nx=[0.0], seg_ny=[1.0], seg_obj=[1], seg_pid=[1],
        materials=[[2, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 1]],
        spectra=np.stack([np.full(95, 0.5, np.float32), np.ones(95, np.float32)])))


def test_emitter_that_is_not_a_circle_is_refused(oracle):
    rc, *_ = oracle.forward_render(segment_emitter_world(), 64, 48, rays_per_emitter=4)
    assert rc == -3
