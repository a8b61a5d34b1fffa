"""The reference's checkpoint format (2d/src/bin/gui.rs:24-45: CheckPoint {world, raw_image} through
rmp_serde::encode::to_vec) read and written by the C++ host mirror (raytracer_amd/host/checkpoint.hpp).

No checkpoint file ships with the reference and its own loader cannot run (Spectrum's Deserialize is todo!(),
2d/src/spectrum.rs:117-124), so the format is pinned here three ways: an encoder written in this file straight
from rmp-serde's rules for the serde data model (structs as arrays, enum variants as {"Name": payload}, unit
variants as "Name", f32/f64 as MessagePack floats), the independent msgpack-python decoder, and round trips."""
import struct

import numpy as np
import pytest

import raytracer_amd as rt

msgpack = pytest.importorskip("msgpack")


# ---- an independent statement of the encoding -----------------------------------
def arr(n):
    return bytes([0x90 | n]) if n < 16 else b"\xdc" + struct.pack(">H", n)


def f64(v):
    return b"\xcb" + struct.pack(">d", v)


def f32(v):
    return b"\xca" + struct.pack(">f", v)


def s(name):
    b = name.encode()
    assert len(b) < 32
    return bytes([0xa0 | len(b)]) + b


def variant(name, payload):
    return b"\x81" + s(name) + payload


def dvec2(x, y):
    return arr(2) + f64(x) + f64(y)


def spectrum(vals):
    assert len(vals) == 95
    return arr(95) + b"".join(f32(v) for v in vals)


def segment(a, b, n):
    return arr(3) + dvec2(*a) + dvec2(*b) + dvec2(*n)


EMPTY_QUADTREE = arr(6) + arr(0) + arr(2) + dvec2(400.0, 300.0) + dvec2(400.0, 300.0) + b"\xc0" * 4


def tiny_world_bytes():
    red = [0.0] * 95
    red[60] = 1.5
    grey = [0.25] * 95
    objects = [
        arr(2) + variant("Circle", arr(2) + dvec2(100.0, 50.5) + f64(7.25)) + variant("Emissive", arr(1) + spectrum(red)),
        arr(2) + variant("Segment", segment((0.0, 0.0), (10.0, 0.0), (-0.0, 1.0))) + s("Reflective"),
        arr(2) + variant("Strip", arr(1) + arr(2) + segment((1.0, 1.0), (2.0, 1.0), (0.0, 1.0)) + segment((2.0, 1.0), (2.0, 3.0), (-1.0, 0.0)))
        + variant("Dielectric", arr(2) + variant("Cauchy", arr(2) + f64(1.5) + f64(0.004)) + spectrum(grey)),
        arr(2) + variant("Circle", arr(2) + dvec2(-3.0, 4.0) + f64(1.0))
        + variant("DirectionalEmissive", arr(3) + spectrum(grey) + f64(0.9) + f64(20.0)),
        arr(2) + variant("Circle", arr(2) + dvec2(9.0, 9.0) + f64(2.0)) + variant("Diffuse", arr(1) + spectrum(grey)),
        arr(2) + variant("Circle", arr(2) + dvec2(19.0, 9.0) + f64(2.0))
        + variant("Dielectric", arr(2) + variant("Simple", f64(1.33)) + spectrum(red)),
    ]
    camera = arr(2) + dvec2(400.0, 300.0) + dvec2(400.0, 300.0)
    return arr(3) + camera + arr(len(objects)) + b"".join(objects) + EMPTY_QUADTREE


def test_decodes_bytes_written_from_the_format_rules():
    w = rt.World2D.from_bytes(tiny_world_bytes())
    a = w.arrays()
    assert w.camera == ((400.0, 300.0), (400.0, 300.0))
    assert list(a["circle_cx"]) == [100.0, -3.0, 9.0, 19.0] and list(a["circle_r"]) == [7.25, 1.0, 2.0, 2.0]
    assert list(a["circle_pid"]) == [0, 4, 5, 6] and list(a["circle_obj"]) == [0, 3, 4, 5]
    assert list(a["seg_pid"]) == [1, 2, 3] and list(a["seg_obj"]) == [1, 2, 2]
    assert list(a["seg_bx"]) == [10.0, 2.0, 2.0] and list(a["seg_by"]) == [0.0, 1.0, 3.0]
    assert list(a["seg_nx"]) == [-0.0, 0.0, -1.0] and np.signbit(a["seg_nx"][0])     # stored normals are kept as written
    kinds = a["materials"][:, 0]
    assert list(kinds) == [0, 3, 4, 1, 2, 4]                                           # Emissive, Reflective, Dielectric, ...
    assert tuple(a["materials"][2, 1:4]) == (1.0, 1.5, 0.004)                          # Cauchy a, b
    assert tuple(a["materials"][5, 1:3]) == (0.0, 1.33)                                # Simple
    assert tuple(a["materials"][3, 4:6]) == (0.9, 20.0)                                # angle, d
    assert a["spectra"][0][60] == 1.5 and a["spectra"][0].sum() == 1.5
    assert np.all(a["spectra"][2] == 0.25)
    # and the writer reproduces those bytes exactly
    assert w.to_bytes() == tiny_world_bytes()


@pytest.mark.parametrize("name", rt.World2D.NAMES)
def test_every_scene_round_trips(name):
    w = rt.World2D(name, seed=3)
    data = w.to_bytes()
    back = rt.World2D.from_bytes(data)
    a, b = w.arrays(), back.arrays()
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert back.to_bytes() == data
    # seen by an unrelated decoder: structs are arrays, variants are one-entry maps or bare names
    world = msgpack.unpackb(data, raw=False, strict_map_key=False)
    camera, objects, quadtree = world
    assert camera == [[400.0, 300.0], [400.0, 300.0]]
    assert quadtree == [[], [[400.0, 300.0], [400.0, 300.0]], None, None, None, None]
    assert len(objects) == w.desc.n_objects
    for shape, mat in objects:
        (kind, payload), = shape.items()
        assert kind in ("Circle", "Segment", "Strip")
        if kind == "Circle":
            assert len(payload) == 2 and len(payload[0]) == 2
        if isinstance(mat, str):
            assert mat == "Reflective"
        else:
            (mk, fields), = mat.items()
            assert mk in ("Emissive", "DirectionalEmissive", "Diffuse", "Dielectric")
            assert len(fields[-1 if mk == "Dielectric" else 0]) == 95
    # f32 stays f32 on the wire: one 0xca per spectrum sample at least
    assert data.count(b"\xca") >= 95 * len([m for _, m in objects if not isinstance(m, str)])


def test_checkpoint_with_image(tmp_path):
    w = rt.World2D("simple_world")
    rng = np.random.default_rng(0)
    rgb = rng.random((5, 7, 3)).astype(np.float32)
    weight = rng.random((5, 7)).astype(np.float32) * 100
    path = tmp_path / "checkpoint.dat"
    w.save_checkpoint(path, rgb, weight)
    back = rt.World2D.load_checkpoint(path)
    got_rgb, got_w = back.image
    assert np.array_equal(got_rgb, rgb) and np.array_equal(got_w, weight)
    assert np.array_equal(back.arrays()["circle_cx"], w.arrays()["circle_cx"])
    world, image = msgpack.unpackb(path.read_bytes(), raw=False, strict_map_key=False)
    assert image[0] == 7 and image[1] == 5 and len(image[2]) == 35
    assert image[2][3][0] == weight.reshape(-1)[3] and image[2][3][1] == list(rgb.reshape(-1, 3)[3])
    assert rt.World2D.from_bytes(path.read_bytes(), checkpoint=True).image[1].shape == (5, 7)
    with pytest.raises(ValueError):
        rt.World2D.load_checkpoint(tmp_path / "missing.dat")


def test_lenient_reading():
    base = tiny_world_bytes()
    # a populated quadtree (objects in the root, one child) is skipped whatever it holds
    child = arr(6) + arr(0) + arr(2) + dvec2(200.0, 150.0) + dvec2(200.0, 150.0) + b"\xc0" * 4
    obj = arr(2) + variant("Circle", arr(2) + dvec2(1.0, 2.0) + f64(3.0)) + s("Reflective")
    tree = arr(6) + arr(1) + obj + arr(2) + dvec2(400.0, 300.0) + dvec2(400.0, 300.0) + b"\xc0" + child + b"\xc0\xc0"
    assert base.endswith(EMPTY_QUADTREE)
    w = rt.World2D.from_bytes(base[: -len(EMPTY_QUADTREE)] + tree)
    assert w.desc.n_objects == 6
    # variants by index in a two-element array, integers where floats are expected, f64 spectrum samples
    spec = arr(95) + b"".join(f64(0.5) for _ in range(95))
    obj = arr(2) + arr(2) + b"\x00" + (arr(2) + arr(2) + b"\x05" + b"\xd0\xfd" + b"\x02") + arr(2) + b"\x02" + arr(1) + spec
    data = arr(3) + arr(2) + dvec2(0.0, 0.0) + dvec2(1.0, 1.0) + arr(1) + obj + EMPTY_QUADTREE
    a = rt.World2D.from_bytes(data).arrays()
    assert (a["circle_cx"][0], a["circle_cy"][0], a["circle_r"][0]) == (5.0, -3.0, 2.0)
    assert a["materials"][0, 0] == 2 and np.all(a["spectra"][0] == 0.5)


@pytest.mark.parametrize("mutate,needle", [
    (lambda b: b[:-1], "truncated"),
    (lambda b: b[:200], "truncated"),
    (lambda b: b + b"\xc0", "trailing"),
    (lambda b: b.replace(s("Reflective"), s("Refractive")), "unknown variant"),
    (lambda b: b"\x92" + b[1:], "World"),
    (lambda b: b.replace(arr(95), arr(94), 1), "Spectrum"),
])
def test_malformed_input_is_refused(mutate, needle):
    with pytest.raises(ValueError, match=needle):
        rt.World2D.from_bytes(mutate(tiny_world_bytes()))


@pytest.mark.gpu
def test_loaded_world_renders_like_the_built_one(gpu_ctx):
    from raytracer_amd.api import params2d
    built = rt.World2D("cornell_box", seed=1)
    loaded = rt.World2D.from_bytes(built.to_bytes())
    p = params2d(40, 30, 2, 3, 10, seed=6)
    a = rt.Scene2D(gpu_ctx, built)
    b = rt.Scene2D(gpu_ctx, loaded)
    gpu_ctx.reset_stats()
    ra = a.global_render(p, want_spectrum=True)[2]
    rays_a = gpu_ctx.stats()["rays"]
    gpu_ctx.reset_stats()
    rb = b.global_render(p, want_spectrum=True)[2]
    assert gpu_ctx.stats()["rays"] == rays_a
    assert np.allclose(ra, rb, rtol=1e-4, atol=1e-6)
    a.close()
    b.close()
