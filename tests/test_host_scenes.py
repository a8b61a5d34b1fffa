"""Host-side mirror of the reference's scene builders (2d/src/worlds.rs, 3d standalone.rs) and the flattener."""
import os

import numpy as np
import pytest

import raytracer_amd as rt

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name,circles,segments,objects", [
    ("sample_world", 4, 115, 17),           # 99-segment bezier strip + 3-segment prism + 3-segment mirror strip + 10 segments
    ("complex_world", 200, 0, 200),
    ("cornell_box", 4, 7, 8),
    ("cornell_box_absorption", 5, 3, 8),
    ("cornell_box_sss", 1, 3, 4),
    ("spectrum_world", 18, 19, 37),
    ("simple_world", 1, 1, 2),
])
def test_scene_sizes(name, circles, segments, objects):
    w = rt.World2D(name)
    assert (w.desc.n_circles, w.desc.n_segments, w.desc.n_objects) == (circles, segments, objects)
    a = w.arrays()
    pids = np.sort(np.concatenate([a["circle_pid"], a["seg_pid"]]))
    assert np.array_equal(pids, np.arange(circles + segments))      # scan order is a permutation of 0..P-1
    # objects appear in order and a strip's segments are contiguous
    assert np.all(np.diff(a["seg_obj"]) >= 0) and np.all(np.diff(a["circle_obj"]) >= 0)
    ln = np.hypot(a["seg_nx"], a["seg_ny"])
    assert np.allclose(ln, 1.0, atol=1e-15)
    # Segment::new: n = perp(b - a).normalize()
    sx, sy = a["seg_bx"] - a["seg_ax"], a["seg_by"] - a["seg_ay"]
    assert np.allclose(a["seg_nx"] * sx + a["seg_ny"] * sy, 0.0, atol=1e-9)
    assert np.all(a["seg_nx"] * -sy + a["seg_ny"] * sx > 0)


def test_sample_world_details():
    a = rt.World2D("sample_world", t=0.0).arrays()
    assert np.allclose(a["circle_cx"], [600, 600, 600 + 50 * np.sin(2 * np.pi / 3), 50])
    assert a["circle_r"].tolist() == [20, 10, 10, 20]
    # Bezier::as_segments(100) covers t in [0, 0.99]: the strip starts at the first control point
    first = np.argmin(a["seg_pid"][a["seg_obj"] == 5])
    strip = np.where(a["seg_obj"] == 5)[0]
    assert len(strip) == 99 and a["seg_ax"][strip[first]] == 400.0 and a["seg_ay"][strip[first]] == 100.0
    assert np.array_equal(a["seg_bx"][strip[:-1]], a["seg_ax"][strip[1:]])       # chained
    mats = a["materials"]
    assert mats[5, 0] == 4 and mats[5, 1] == 1 and mats[5, 2] == 1.45 and mats[5, 3] == 0.05   # Cauchy glass
    # absorption_from_color(White) = White - White = 0
    assert not a["spectra"][int(mats[5, 6])].any()
    moved = rt.World2D("sample_world", t=0.25).arrays()
    assert not np.array_equal(moved["circle_cx"], a["circle_cx"])


def test_spectrum_bands():
    a = rt.World2D("cornell_box").arrays()
    red = a["spectra"][int(a["materials"][2, 6])]       # right wall: emission_from_color(Red) = 625..750 nm
    lam = 360 + 5 * np.arange(95)
    assert np.array_equal(red != 0, (lam >= 625) & (lam <= 750))
    light = a["spectra"][int(a["materials"][0, 6])]
    assert np.all(light == np.float32(0.5))


def test_complex_world_matches_committed_fixture():
    want = np.load(os.path.join(HERE, "golden", "complex_world_seed1.npz"))
    got = rt.World2D("complex_world", seed=1).arrays()
    for k in want.files:
        assert np.array_equal(got[k], want[k]), k
    assert np.all((got["circle_cx"] >= 0) & (got["circle_cx"] < 800) & (got["circle_cy"] >= 0) & (got["circle_cy"] < 600))
    kinds = np.bincount(got["materials"][:, 0].astype(int), minlength=5)
    assert kinds[1] == 0 and kinds[[0, 2, 3, 4]].min() >= 30        # four material classes, roughly uniform
    other = rt.World2D("complex_world", seed=2).arrays()
    assert not np.array_equal(other["circle_cx"], got["circle_cx"])


def test_standalone_3d_scene():
    s = rt.Standalone3D(640, 480)
    d = s.desc
    assert d.n_spheres == 218 and d.r[216] == 5.0 and d.r[217] == 10000.0 and d.cy[217] == -10000.0
    assert (d.cx[0], d.cy[0], d.cz[0], d.r[0]) == (-9.0, -9.0, -29.0, 0.8)
    assert abs(sum(v * v for v in d.cam_dir) - 1.0) < 1e-15 and list(d.cam_up) == [0.0, 1.0, 0.0]
    assert d.focal_length == 2.0 and list(d.cam_pos) == [0.0, 2.0, 0.0]


def test_unknown_world_is_an_error():
    with pytest.raises(ValueError):
        rt.World2D("no_such_world")


def test_png_writer_roundtrip(tmp_path):
    import zlib
    rgba = (np.arange(5 * 3 * 4) % 256).astype(np.uint8).reshape(3, 5, 4)
    path = tmp_path / "x.png"
    rt.write_png(path, rgba, 5, 3)
    raw = path.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    idat = raw[raw.index(b"IDAT") + 4: raw.index(b"IEND") - 8]
    pix = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(3, 1 + 5 * 4)
    assert not pix[:, 0].any() and np.array_equal(pix[:, 1:].reshape(3, 5, 4), rgba)
