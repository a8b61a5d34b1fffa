"""The C-ABI library loads without a GPU and exports every function include/rt_abi.h declares."""
import ctypes
import os
import re

import pytest

from raytracer_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt[23]?d?_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("rt_ctx_create", "rt2d_scene_create", "rt2d_render", "rt2d_accumulate", "rt2d_resolve",
                 "rt2d_project_xyz", "rt2d_resolve_xyz", "rt2d_compute_pixels", "rt2d_denoise", "rt2d_denoise_device", "rt2d_forward_render", "rt2d_forward_accumulate", "rt2d_progressive_create", "rt2d_progressive_step",
                 "rt2d_progressive_pass", "rt2d_progressive_resample", "rt2d_progressive_image", "rt2d_progressive_read",
                 "rt2d_progressive_planes", "rt2d_progressive_destroy", "rt2d_trace_batch", "rt3d_scene_create", "rt3d_render", "rt3d_trace_batch"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd", "lib", "librt_amd.so"))
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in rt_abi.h but not exported"


def test_python_binding_covers_the_header():
    assert sorted(_ffi.ABI) == declared_functions()
    _ffi.lib()  # sets argtypes on every symbol


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(_ffi.Material2D) == 48
    assert ctypes.sizeof(_ffi.RenderParams2D) == 40
    assert ctypes.sizeof(_ffi.RenderParams3D) == 24
    assert ctypes.sizeof(_ffi.Stats) == 56
    assert ctypes.sizeof(_ffi.Scene2DDesc) == 16 + 15 * 8
    assert ctypes.sizeof(_ffi.Scene3DDesc) == 8 + 4 * 8 + 9 * 8 + 8 + 8


def test_no_gpu_means_a_clear_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = ctypes.c_void_p()
    rc = _ffi.lib().rt_ctx_create(0, ctypes.byref(out))
    assert rc == -2 and not out.value                      # RT_E_NO_DEVICE
    assert b"no CPU path" in _ffi.lib().rt_last_error() or b"no HIP device" in _ffi.lib().rt_last_error()


def test_product_does_not_touch_the_checker():
    """Nothing under raytracer_amd/ or include/ may reference oracle/."""
    for base in ("raytracer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp", ".inc")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "oracle" not in text.lower() or f == "__init__.py" and False, f"{f} mentions the oracle"
