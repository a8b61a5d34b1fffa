"""ctypes binding of the C ABI in include/rt_abi.h (librt_amd.so) and of the
host-side scene builders (librt_host.so).

Plumbing only: every call goes straight to the HIP library.  There is no
Python or CPU implementation of the hot path here; if the HIP library is
missing, loading fails with an error that says how to build it.
"""
import ctypes as C
import os

import numpy as np

NBINS = 95
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class Material2D(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ior_kind", C.c_int32), ("ior_a", C.c_double), ("ior_b", C.c_double),
                ("angle", C.c_double), ("d", C.c_double), ("spectrum", C.c_int32), ("pad_", C.c_int32)]


class Scene2DDesc(C.Structure):
    _fields_ = [("n_circles", C.c_int32), ("n_segments", C.c_int32), ("n_objects", C.c_int32), ("n_spectra", C.c_int32),
                ("circle_cx", c_double_p), ("circle_cy", c_double_p), ("circle_r", c_double_p),
                ("circle_obj", c_int32_p), ("circle_pid", c_int32_p),
                ("seg_ax", c_double_p), ("seg_ay", c_double_p), ("seg_bx", c_double_p), ("seg_by", c_double_p),
                ("seg_nx", c_double_p), ("seg_ny", c_double_p), ("seg_obj", c_int32_p), ("seg_pid", c_int32_p),
                ("materials", C.POINTER(Material2D)), ("spectra", c_float_p)]


class RenderParams2D(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("lambda_samples", C.c_int32),
                ("recursion_limit", C.c_int32), ("sample_offset", C.c_int32), ("seed", C.c_uint64),
                ("px_to_world", C.c_double)]


class Denoiser2D(C.Structure):
    _fields_ = [("top", C.c_float), ("mask_size", C.c_int32), ("oversampling_factor", C.c_float), ("passes", C.c_int32)]


class Progress2D(C.Structure):
    _fields_ = [("loops", C.c_int64), ("next_sample", C.c_int64), ("recomputed", C.c_int64), ("rays", C.c_uint64)]


class ForwardParams2D(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rays_per_emitter", C.c_int32), ("depth", C.c_int32),
                ("sample_offset", C.c_int32), ("pad_", C.c_int32), ("seed", C.c_uint64)]


class Scene3DDesc(C.Structure):
    _fields_ = [("n_spheres", C.c_int32), ("use_bvh", C.c_int32),
                ("cx", c_double_p), ("cy", c_double_p), ("cz", c_double_p), ("r", c_double_p),
                ("cam_pos", C.c_double * 3), ("cam_dir", C.c_double * 3), ("cam_up", C.c_double * 3),
                ("focal_length", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class RenderParams3D(C.Structure):
    _fields_ = [("spp", C.c_int32), ("bounce_limit", C.c_int32), ("sample_offset", C.c_int32), ("pad_", C.c_int32),
                ("seed", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("launches", C.c_uint64), ("kernel_ms", C.c_double),
                ("bvh_node_visits", C.c_uint64), ("sphere_tests", C.c_uint64), ("searches", C.c_uint64)]


ABI_VERSION = 3  # RT_ABI_VERSION of the include/rt_abi.h these bindings mirror

# every symbol include/rt_abi.h declares: name -> (restype, argtypes)
ABI = {
    "rt_abi_version": (C.c_int32, []),
    "rt_version": (C.c_char_p, []),
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int32, [c_int32_p]),
    "rt_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "rt_ctx_destroy": (C.c_int32, [C.c_void_p]),
    "rt_ctx_set_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "rt_ctx_use_own_stream": (C.c_int32, [C.c_void_p]),
    "rt_ctx_synchronize": (C.c_int32, [C.c_void_p]),
    "rt_ctx_get_stats": (C.c_int32, [C.c_void_p, C.POINTER(Stats)]),
    "rt_selftest_math": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rt_ctx_reset_stats": (C.c_int32, [C.c_void_p]),
    "rt2d_scene_create": (C.c_int32, [C.c_void_p, C.POINTER(Scene2DDesc), C.POINTER(C.c_void_p)]),
    "rt2d_scene_destroy": (C.c_int32, [C.c_void_p]),
    "rt2d_render": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(RenderParams2D), c_float_p, c_float_p, c_float_p]),
    "rt2d_accumulate": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(RenderParams2D), C.c_void_p]),
    "rt2d_resolve": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, c_float_p, c_float_p]),
    "rt2d_project_xyz": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "rt2d_resolve_xyz": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, c_float_p, c_float_p]),
    "rt2d_compute_pixels": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(RenderParams2D), C.c_int64, c_int32_p, c_int32_p, c_float_p]),
    "rt2d_denoise": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(RenderParams2D), C.c_float, C.c_int32, C.c_float, c_float_p,
                                 c_float_p, c_float_p, C.POINTER(C.c_int64)]),
    "rt2d_denoise_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(RenderParams2D), C.c_float, C.c_int32, C.c_float, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "rt2d_progressive_create": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(RenderParams2D), C.POINTER(Denoiser2D),
                                            C.POINTER(C.c_void_p)]),
    "rt2d_progressive_destroy": (C.c_int32, [C.c_void_p]),
    "rt2d_progressive_step": (C.c_int32, [C.c_void_p, C.POINTER(Progress2D)]),
    "rt2d_progressive_pass": (C.c_int32, [C.c_void_p]),
    "rt2d_progressive_resample": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64)]),
    "rt2d_progressive_image": (C.c_int32, [C.c_void_p, C.c_float, C.POINTER(C.c_uint8)]),
    "rt2d_progressive_read": (C.c_int32, [C.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p]),
    "rt2d_progressive_planes": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p)]),
    "rt2d_forward_accumulate": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(ForwardParams2D), C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_int64)]),
    "rt2d_forward_accumulate_static": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(ForwardParams2D), C.c_void_p, C.c_void_p,
                                                   C.c_void_p]),
    "rt2d_forward_render": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(ForwardParams2D), c_float_p, c_float_p,
                                        C.POINTER(C.c_int64)]),
    "rt2d_trace_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, c_double_p, c_double_p, c_double_p, c_double_p,
                                     c_int32_p, C.c_int32, C.c_uint64, c_float_p, c_int32_p, c_double_p, c_int32_p]),
    "rt3d_scene_create": (C.c_int32, [C.c_void_p, C.POINTER(Scene3DDesc), C.POINTER(C.c_void_p)]),
    "rt3d_scene_destroy": (C.c_int32, [C.c_void_p]),
    "rt3d_render": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(RenderParams3D), c_float_p]),
    "rt3d_accumulate": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(RenderParams3D), C.c_void_p]),
    "rt3d_trace_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, c_double_p, c_double_p, C.c_int32, C.c_uint64,
                                     c_float_p, c_int32_p, c_double_p, c_int32_p]),
    "rt_tonemap_reinhard": (C.c_int32, [C.c_void_p, c_float_p, C.c_int64, C.c_float, C.c_uint8, C.POINTER(C.c_uint8)]),
}

HOST_ABI = {
    "rth_last_error": (C.c_char_p, []),
    "rth_world2d_build": (C.c_void_p, [C.c_char_p, C.c_double, C.c_uint64, C.c_uint64, C.c_double, C.c_double]),
    "rth_world2d_desc": (C.c_int32, [C.c_void_p, C.POINTER(Scene2DDesc)]),
    "rth_world2d_free": (None, [C.c_void_p]),
    "rth_scene3d_standalone": (C.c_void_p, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(Scene3DDesc)]),
    "rth_scene3d_free": (None, [C.c_void_p]),
    "rth_world2d_decode": (C.c_void_p, [C.POINTER(C.c_uint8), C.c_uint64, C.c_int32]),
    "rth_checkpoint_load": (C.c_void_p, [C.c_char_p]),
    "rth_world2d_encode": (C.c_int64, [C.c_void_p, C.c_int32, c_float_p, c_float_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8),
                                       C.c_uint64]),
    "rth_world2d_image": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), c_float_p, c_float_p]),
    "rth_world2d_camera": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double)]),
    "rth_png_write": (C.c_int32, [C.c_char_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
}


def _load(name, table):
    path = os.path.join(_LIBDIR, name)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(hipcc --offload-arch=gfx950).  There is no fallback implementation.")
    lib = C.CDLL(path)
    for sym, (res, args) in table.items():
        fn = getattr(lib, sym)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None
_host = None


def lib():
    global _lib
    if _lib is None:
        # RT_AMD_LIBNAME selects a diagnostic build of the same HIP library (e.g. the -DRT_PROF one)
        _lib = _load(os.environ.get("RT_AMD_LIBNAME", "librt_amd.so"), ABI)
        if _lib.rt_abi_version() != ABI_VERSION:  # the ctypes structs below mirror ONE generation of include/rt_abi.h
            got, _lib = _lib.rt_abi_version(), None
            raise RuntimeError(f"librt_amd.so has struct layout generation {got}, this binding was written for {ABI_VERSION}")
    return _lib


def host():
    global _host
    if _host is None:
        _host = _load("librt_host.so", HOST_ABI)
    return _host


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt_abi error {code}: {msg}")
        self.code = code


def check(rc):
    if rc != 0:
        raise RtError(rc, lib().rt_last_error().decode(errors="replace"))


def ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)
