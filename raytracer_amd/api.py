"""Object wrappers over the C ABI (see _ffi.py).  Names follow the reference:
World, ReverseRenderParams fields, global_render / compute_pixel / trace_ray."""
import ctypes as C
import weakref

import numpy as np

from . import _ffi
from ._ffi import NBINS, RenderParams2D, RenderParams3D, RtError, Scene2DDesc, Scene3DDesc, Stats, check  # noqa: F401


class World2D:
    """A scene of 2d/src/worlds.rs built by the C++ host mirror, flattened to the SoA
    arrays of rt2d_scene_desc.  `desc` stays valid while this object lives."""

    NAMES = ("sample_world", "complex_world", "cornell_box", "cornell_box_absorption", "cornell_box_sss",
             "spectrum_world", "simple_world")

    def __init__(self, name, t=0.0, idx=0, seed=1, width=800.0, height=600.0):
        h = _ffi.host()
        self._h = h.rth_world2d_build(name.encode(), float(t), int(idx), int(seed), float(width), float(height))
        if not self._h:
            raise ValueError(h.rth_last_error().decode())
        self.desc = Scene2DDesc()
        if h.rth_world2d_desc(self._h, C.byref(self.desc)) != 0:
            raise ValueError(h.rth_last_error().decode())
        self.name = name

    # ---- the reference's checkpoint format (2d/src/bin/gui.rs:24-45): MessagePack as rmp-serde writes it
    @classmethod
    def _adopt(cls, handle, name):
        h = _ffi.host()
        if not handle:
            raise ValueError(h.rth_last_error().decode())
        self = cls.__new__(cls)
        self._h, self.name, self.desc = handle, name, Scene2DDesc()
        if h.rth_world2d_desc(self._h, C.byref(self.desc)) != 0:
            raise ValueError(h.rth_last_error().decode())
        return self

    @classmethod
    def from_bytes(cls, data, checkpoint=False):
        """A serialised World (or, with checkpoint=True, a CheckPoint {world, raw_image})."""
        buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
        return cls._adopt(_ffi.host().rth_world2d_decode(buf, len(data), 1 if checkpoint else 0), "bytes")

    @classmethod
    def load_checkpoint(cls, path):
        return cls._adopt(_ffi.host().rth_checkpoint_load(str(path).encode()), str(path))

    def to_bytes(self, rgb=None, weight=None):
        """The World's bytes; with RawImage planes (rgb [H, W, 3], weight [H, W]) a whole CheckPoint."""
        h = _ffi.host()
        with_image = rgb is not None
        if with_image:
            rgb = np.ascontiguousarray(rgb, np.float32)
            weight = np.ascontiguousarray(weight, np.float32)
            hh, ww = weight.shape
            assert rgb.shape == (hh, ww, 3)
        else:
            hh = ww = 0
        args = (self._h, 1 if with_image else 0, _ffi.ptr(rgb, C.c_float), _ffi.ptr(weight, C.c_float), ww, hh)
        n = h.rth_world2d_encode(*args, None, 0)
        if n < 0:
            raise ValueError(h.rth_last_error().decode())
        out = (C.c_uint8 * n)()
        if h.rth_world2d_encode(*args, out, n) != n:
            raise ValueError(h.rth_last_error().decode())
        return bytes(out)

    def save_checkpoint(self, path, rgb, weight):
        with open(path, "wb") as f:
            f.write(self.to_bytes(rgb, weight))

    @property
    def image(self):
        """(rgb [H, W, 3], weight [H, W]) of a loaded checkpoint; empty arrays for a built world."""
        h = _ffi.host()
        w, hh = C.c_int32(), C.c_int32()
        h.rth_world2d_image(self._h, C.byref(w), C.byref(hh), None, None)
        rgb = np.zeros((hh.value, w.value, 3), np.float32)
        weight = np.zeros((hh.value, w.value), np.float32)
        h.rth_world2d_image(self._h, C.byref(w), C.byref(hh), _ffi.ptr(rgb, C.c_float), _ffi.ptr(weight, C.c_float))
        return rgb, weight

    @property
    def camera(self):
        out = (C.c_double * 4)()
        _ffi.host().rth_world2d_camera(self._h, out)
        return (out[0], out[1]), (out[2], out[3])

    @property
    def n_prims(self):
        return self.desc.n_circles + self.desc.n_segments

    def arrays(self):
        """numpy copies of the SoA arrays (for fixtures and inspection)."""
        d = self.desc
        nc, ns = d.n_circles, d.n_segments

        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt).copy() if n else np.zeros(0, dt)

        out = {k: arr(getattr(d, k), nc, np.float64) for k in ("circle_cx", "circle_cy", "circle_r")}
        out.update({k: arr(getattr(d, k), nc, np.int32) for k in ("circle_obj", "circle_pid")})
        out.update({k: arr(getattr(d, k), ns, np.float64) for k in ("seg_ax", "seg_ay", "seg_bx", "seg_by", "seg_nx", "seg_ny")})
        out.update({k: arr(getattr(d, k), ns, np.int32) for k in ("seg_obj", "seg_pid")})
        mats = np.zeros((d.n_objects, 7), np.float64)
        for i in range(d.n_objects):
            m = d.materials[i]
            mats[i] = (m.kind, m.ior_kind, m.ior_a, m.ior_b, m.angle, m.d, m.spectrum)
        out["materials"] = mats
        out["spectra"] = arr(d.spectra, d.n_spectra * NBINS, np.float32).reshape(d.n_spectra, NBINS)
        return out

    def close(self):
        if getattr(self, "_h", None):
            _ffi.host().rth_world2d_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: the module globals may be gone
            pass


class ArrayWorld2D:
    """rt2d_scene_desc over caller-owned numpy arrays (keys as World2D.arrays())."""

    def __init__(self, arrays):
        a = self._a = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        d = self.desc = Scene2DDesc()
        d.n_circles, d.n_segments = len(a["circle_cx"]), len(a["seg_ax"])
        d.n_objects, d.n_spectra = len(a["materials"]), len(a["spectra"])
        for k in ("circle_cx", "circle_cy", "circle_r", "seg_ax", "seg_ay", "seg_bx", "seg_by", "seg_nx", "seg_ny"):
            a[k] = np.ascontiguousarray(a[k], np.float64)
            setattr(d, k, _ffi.ptr(a[k], C.c_double))
        for k in ("circle_obj", "circle_pid", "seg_obj", "seg_pid"):
            a[k] = np.ascontiguousarray(a[k], np.int32)
            setattr(d, k, _ffi.ptr(a[k], C.c_int32))
        self._mats = (_ffi.Material2D * max(1, d.n_objects))()
        for i, row in enumerate(np.asarray(a["materials"], np.float64)):
            m = self._mats[i]
            m.kind, m.ior_kind, m.ior_a, m.ior_b, m.angle, m.d, m.spectrum = (
                int(row[0]), int(row[1]), row[2], row[3], row[4], row[5], int(row[6]))
        d.materials = C.cast(self._mats, C.POINTER(_ffi.Material2D))
        a["spectra"] = np.ascontiguousarray(a["spectra"], np.float32)
        d.spectra = _ffi.ptr(a["spectra"], C.c_float)

    @property
    def n_prims(self):
        return self.desc.n_circles + self.desc.n_segments


class Standalone3D:
    """Scene + camera of 3d/src/bin/standalone.rs at a given image size."""

    def __init__(self, width=640, height=480, use_bvh=True):
        h = _ffi.host()
        self.desc = Scene3DDesc()
        self._h = h.rth_scene3d_standalone(int(width), int(height), 1 if use_bvh else 0, C.byref(self.desc))
        if not self._h:
            raise ValueError(h.rth_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            _ffi.host().rth_scene3d_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ArrayScene3D:
    """rt3d_scene_desc over caller-owned sphere arrays and a Camera::new argument set (3d/src/rt.rs:139-147)."""

    def __init__(self, cx, cy, cz, r, cam_pos, cam_dir, cam_up, focal_length, width, height, use_bvh=True):
        self._a = [np.ascontiguousarray(v, np.float64) for v in (cx, cy, cz, r)]
        d = self.desc = Scene3DDesc()
        d.n_spheres, d.use_bvh = len(self._a[0]), 1 if use_bvh else 0
        d.cx, d.cy, d.cz, d.r = (_ffi.ptr(v, C.c_double) for v in self._a)
        for k in range(3):
            d.cam_pos[k], d.cam_dir[k], d.cam_up[k] = float(cam_pos[k]), float(cam_dir[k]), float(cam_up[k])
        d.focal_length, d.width, d.height = float(focal_length), int(width), int(height)


class Context:
    """One GPU.  Work is enqueued on `stream` (a hipStream_t handle, e.g.
    torch.cuda.current_stream().cuda_stream) or on the context's own stream."""

    def __init__(self, device=0, stream=None):
        self._ctx = C.c_void_p()
        self._scenes = weakref.WeakSet()  # closed with the context: the library refuses to drop a context that owns scenes
        check(_ffi.lib().rt_ctx_create(int(device), C.byref(self._ctx)))
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        """stream: a hipStream_t handle as an int (0 = HIP's null stream), or None = the context's own stream."""
        if stream is None:
            check(_ffi.lib().rt_ctx_use_own_stream(self._ctx))
        else:
            check(_ffi.lib().rt_ctx_set_stream(self._ctx, C.c_void_p(int(stream))))

    def synchronize(self):
        check(_ffi.lib().rt_ctx_synchronize(self._ctx))

    def stats(self):
        s = Stats()
        check(_ffi.lib().rt_ctx_get_stats(self._ctx, C.byref(s)))
        return {"rays": s.rays, "samples": s.samples, "launches": s.launches, "kernel_ms": s.kernel_ms,
                "bvh_node_visits": s.bvh_node_visits, "sphere_tests": s.sphere_tests, "searches": s.searches}

    def reset_stats(self):
        check(_ffi.lib().rt_ctx_reset_stats(self._ctx))

    def selftest_math(self, log2_n=24, seed=0x1234):
        """Mismatches per argument class between the kernels' own sqrt / reciprocal-sqrt sequences and the IEEE ones."""
        out = (C.c_uint64 * 6)()
        check(_ffi.lib().rt_selftest_math(self._ctx, int(log2_n), int(seed), out))
        return [int(v) for v in out]

    def tonemap_reinhard(self, rgb, param=1.0, alpha=255):
        rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
        out = np.empty((rgb.shape[0], 4), np.uint8)
        check(_ffi.lib().rt_tonemap_reinhard(self._ctx, _ffi.ptr(rgb, C.c_float), rgb.shape[0], float(param), int(alpha),
                                             _ffi.ptr(out, C.c_uint8)))
        return out

    def close(self):
        if getattr(self, "_ctx", None):
            for sc in list(self._scenes):
                sc.close()
            check(_ffi.lib().rt_ctx_destroy(self._ctx))
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def params2d(width, height, spp, lambda_samples, recursion_limit, seed=0, sample_offset=0, px_to_world=None):
    p = RenderParams2D()
    p.width, p.height, p.spp, p.lambda_samples = int(width), int(height), int(spp), int(lambda_samples)
    p.recursion_limit, p.sample_offset, p.seed = int(recursion_limit), int(sample_offset), int(seed)
    p.px_to_world = float((800.0 / width if width > 0 else 1.0) if px_to_world is None else px_to_world)
    return p


class Progressive2D:
    """The loop of ReverseRenderer::endless_render (2d/src/librt2d_reverse.rs:424-510) with its state on the GPU:
    spectral sums and per-pixel sample counts.  `denoiser` = (top, mask_size, oversampling_factor, passes) or None."""

    def __init__(self, scene, p, denoiser=None):
        self.scene, self.width, self.height = scene, p.width, p.height
        self._g = C.c_void_p()
        dn = None
        if denoiser is not None:
            dn = _ffi.Denoiser2D(float(denoiser[0]), int(denoiser[1]), float(denoiser[2]), int(denoiser[3]))
        check(_ffi.lib().rt2d_progressive_create(scene.ctx._ctx, scene._sc, C.byref(p), C.byref(dn) if dn is not None else None,
                                                 C.byref(self._g)))
        if not hasattr(scene, "_progressives"):
            scene._progressives = weakref.WeakSet()
        scene._progressives.add(self)

    def step(self):
        """One loop iteration; returns dict(loops, next_sample, recomputed, rays)."""
        out = _ffi.Progress2D()
        check(_ffi.lib().rt2d_progressive_step(self._g, C.byref(out)))
        return {"loops": out.loops, "next_sample": out.next_sample, "recomputed": out.recomputed, "rays": out.rays}

    def render_pass(self):
        check(_ffi.lib().rt2d_progressive_pass(self._g))

    def resample(self):
        n = C.c_int64(0)
        check(_ffi.lib().rt2d_progressive_resample(self._g, C.byref(n)))
        return n.value

    def image(self, reinhard_param=1.0):
        out = np.empty((self.height, self.width, 4), np.uint8)
        check(_ffi.lib().rt2d_progressive_image(self._g, float(reinhard_param), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def read(self, sum=True, count=True, merged=True, loop=True):
        """Host copies (sum [H,W,95], count [H,W], merged_rgb [H,W,3], loop_rgb [H,W,3]); None where not asked."""
        H, W = self.height, self.width
        a = np.empty((H, W, NBINS), np.float32) if sum else None
        b = np.empty((H, W), np.float32) if count else None
        c = np.empty((H, W, 3), np.float32) if merged else None
        d = np.empty((H, W, 3), np.float32) if loop else None
        check(_ffi.lib().rt2d_progressive_read(self._g, _ffi.ptr(a, C.c_float), _ffi.ptr(b, C.c_float), _ffi.ptr(c, C.c_float),
                                               _ffi.ptr(d, C.c_float)))
        return a, b, c, d

    def planes(self):
        """Device addresses (sum, count, merged_rgb, loop_rgb)."""
        ptrs = [C.c_void_p() for _ in range(4)]
        check(_ffi.lib().rt2d_progressive_planes(self._g, *[C.byref(q) for q in ptrs]))
        return tuple(q.value for q in ptrs)

    def close(self):
        if getattr(self, "_g", None):
            _ffi.lib().rt2d_progressive_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene2D:
    """A World uploaded to one GPU; methods named after ReverseRenderer's."""

    def __init__(self, ctx, world):
        self.ctx = ctx
        self._world = world  # keeps the desc arrays alive during creation
        self._sc = C.c_void_p()
        check(_ffi.lib().rt2d_scene_create(ctx._ctx, C.byref(world.desc), C.byref(self._sc)))
        ctx._scenes.add(self)
        self.n_prims = world.n_prims

    def global_render(self, p, want_spectrum=False):
        n = p.width * p.height
        rgb = np.empty((p.height, p.width, 3), np.float32)
        weight = np.empty((p.height, p.width), np.float32)
        spec = np.empty((p.height, p.width, NBINS), np.float32) if want_spectrum else None
        check(_ffi.lib().rt2d_render(self.ctx._ctx, self._sc, C.byref(p), _ffi.ptr(rgb, C.c_float),
                                     _ffi.ptr(weight, C.c_float), _ffi.ptr(spec, C.c_float)))
        assert rgb.size == 3 * n
        return (rgb, weight, spec) if want_spectrum else (rgb, weight)

    def accumulate(self, p, d_spectrum_ptr):
        check(_ffi.lib().rt2d_accumulate(self.ctx._ctx, self._sc, C.byref(p), C.c_void_p(d_spectrum_ptr)))

    def resolve(self, d_spectrum_ptr, width, height, total_spp, d_rgb_ptr=None, to_host=True):
        rgb = np.empty((height, width, 3), np.float32) if to_host else None
        weight = np.empty((height, width), np.float32) if to_host else None
        check(_ffi.lib().rt2d_resolve(self.ctx._ctx, C.c_void_p(d_spectrum_ptr), int(width), int(height), int(total_spp),
                                      C.c_void_p(d_rgb_ptr) if d_rgb_ptr else None, _ffi.ptr(rgb, C.c_float),
                                      _ffi.ptr(weight, C.c_float)))
        return rgb, weight

    def project_xyz(self, d_spectrum_ptr, width, height, d_xyz_ptr):
        """d_xyz [H*W, 3] = the spectral sums projected on the colour matching functions (linear: sums over GPUs commute)."""
        check(_ffi.lib().rt2d_project_xyz(self.ctx._ctx, C.c_void_p(d_spectrum_ptr), int(width), int(height), C.c_void_p(d_xyz_ptr)))

    def resolve_xyz(self, d_xyz_ptr, width, height, total_spp, d_rgb_ptr=None, to_host=True):
        rgb = np.empty((height, width, 3), np.float32) if to_host else None
        weight = np.empty((height, width), np.float32) if to_host else None
        check(_ffi.lib().rt2d_resolve_xyz(self.ctx._ctx, C.c_void_p(d_xyz_ptr), int(width), int(height), int(total_spp),
                                          C.c_void_p(d_rgb_ptr) if d_rgb_ptr else None, _ffi.ptr(rgb, C.c_float),
                                          _ffi.ptr(weight, C.c_float)))
        return rgb, weight

    def compute_pixels(self, p, px, py):
        px, py = _ffi.as_i32(px), _ffi.as_i32(py)
        out = np.empty((len(px), NBINS), np.float32)
        check(_ffi.lib().rt2d_compute_pixels(self.ctx._ctx, self._sc, C.byref(p), len(px), _ffi.ptr(px, C.c_int32),
                                             _ffi.ptr(py, C.c_int32), _ffi.ptr(out, C.c_float)))
        return out

    def denoise_device(self, p, top, mask_size, oversampling_factor, d_merged_rgb_ptr, d_merged_weight_ptr, d_loop_rgb_ptr):
        """One pass of ReverseRenderer::denoise on device RawImage planes; returns the number of recomputed pixels."""
        n = C.c_int64(0)
        check(_ffi.lib().rt2d_denoise_device(self.ctx._ctx, self._sc, C.byref(p), float(top), int(mask_size),
                                             float(oversampling_factor), C.c_void_p(d_merged_rgb_ptr),
                                             C.c_void_p(d_merged_weight_ptr), C.c_void_p(d_loop_rgb_ptr), C.byref(n)))
        return n.value

    def denoise(self, p, top, mask_size, oversampling_factor, merged_rgb, merged_weight, loop_rgb):
        """The same on host planes (numpy): returns (merged_rgb, merged_weight, recomputed); inputs are not modified."""
        rgb = np.ascontiguousarray(merged_rgb, np.float32).copy()
        w = np.ascontiguousarray(merged_weight, np.float32).copy()
        lp = np.ascontiguousarray(loop_rgb, np.float32)
        assert rgb.size == 3 * p.width * p.height and w.size == p.width * p.height and lp.size == rgb.size
        n = C.c_int64(0)
        check(_ffi.lib().rt2d_denoise(self.ctx._ctx, self._sc, C.byref(p), float(top), int(mask_size), float(oversampling_factor),
                                      _ffi.ptr(rgb, C.c_float), _ffi.ptr(w, C.c_float), _ffi.ptr(lp, C.c_float), C.byref(n)))
        return rgb, w, n.value

    def render_column(self, p, i):
        return self.compute_pixels(p, np.full(p.height, i, np.int32), np.arange(p.height, dtype=np.int32))

    def compute_pixel(self, p, x, y):
        return self.compute_pixels(p, [x], [y])[0]

    def trace_batch(self, ox, oy, dx, dy, lambda_idx, depth, seed=0):
        ox, oy, dx, dy = (_ffi.as_f64(a) for a in (ox, oy, dx, dy))
        li = _ffi.as_i32(lambda_idx)
        n = len(ox)
        rad = np.empty(n, np.float32)
        fp = np.empty(n, np.int32)
        ft = np.empty(n, np.float64)
        nb = np.empty(n, np.int32)
        check(_ffi.lib().rt2d_trace_batch(self.ctx._ctx, self._sc, n, _ffi.ptr(ox, C.c_double), _ffi.ptr(oy, C.c_double),
                                          _ffi.ptr(dx, C.c_double), _ffi.ptr(dy, C.c_double), _ffi.ptr(li, C.c_int32),
                                          int(depth), int(seed), _ffi.ptr(rad, C.c_float), _ffi.ptr(fp, C.c_int32),
                                          _ffi.ptr(ft, C.c_double), _ffi.ptr(nb, C.c_int32)))
        return rad, fp, ft, nb

    # ---- forward light tracer (ForwardRenderer, 2d/src/librt2d_forward.rs)
    @staticmethod
    def forward_params(width, height, rays_per_emitter=1000, depth=5, seed=0, sample_offset=0):
        return _ffi.ForwardParams2D(int(width), int(height), int(rays_per_emitter), int(depth), int(sample_offset), 0, int(seed))

    def forward_render(self, p):
        """ForwardRenderer::global_render: (rgb [H, W, 3], weight [H, W], lines drawn)."""
        rgb = np.empty((p.height, p.width, 3), np.float32)
        weight = np.empty((p.height, p.width), np.float32)
        n = C.c_int64(0)
        check(_ffi.lib().rt2d_forward_render(self.ctx._ctx, self._sc, C.byref(p), _ffi.ptr(rgb, C.c_float),
                                             _ffi.ptr(weight, C.c_float), C.byref(n)))
        return rgb, weight, n.value

    def forward_accumulate(self, p, d_rgb_ptr, d_weight_ptr, count_lines=False):
        """Adds one forward pass onto device RawImage planes; returns the number of lines drawn if asked (synchronises)."""
        n = C.c_int64(0)
        check(_ffi.lib().rt2d_forward_accumulate(self.ctx._ctx, self._sc, C.byref(p), C.c_void_p(d_rgb_ptr), C.c_void_p(d_weight_ptr),
                                                 C.byref(n) if count_lines else None))
        return n.value if count_lines else None

    def forward_accumulate_static(self, p, d_rgb_ptr, d_weight_ptr, d_lines_ptr=None):
        """forward_accumulate without any allocation (scratch inside the library, images up to 2^20 pixels);
        d_lines_ptr: device uint64 that receives the number of lines drawn.  Never synchronises."""
        check(_ffi.lib().rt2d_forward_accumulate_static(self.ctx._ctx, self._sc, C.byref(p), C.c_void_p(d_rgb_ptr),
                                                        C.c_void_p(d_weight_ptr), C.c_void_p(d_lines_ptr) if d_lines_ptr else None))

    def progressive(self, p, denoiser=None):
        """Device-resident endless_render loop: see Progressive2D."""
        return Progressive2D(self, p, denoiser)

    def close(self):
        if getattr(self, "_sc", None):
            for g in list(getattr(self, "_progressives", ())):
                g.close()
            _ffi.lib().rt2d_scene_destroy(self._sc)
            self._sc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def params3d(spp, bounce_limit, seed=0, sample_offset=0):
    p = RenderParams3D()
    p.spp, p.bounce_limit, p.sample_offset, p.seed = int(spp), int(bounce_limit), int(sample_offset), int(seed)
    return p


class Scene3D:
    """Spheres + camera uploaded to one GPU; methods named after the 3d crate's World."""

    def __init__(self, ctx, scene):
        self.ctx = ctx
        self._scene = scene
        self.width, self.height = scene.desc.width, scene.desc.height
        self._sc = C.c_void_p()
        check(_ffi.lib().rt3d_scene_create(ctx._ctx, C.byref(scene.desc), C.byref(self._sc)))
        ctx._scenes.add(self)

    def render(self, p):
        out = np.empty((self.height, self.width, 3), np.float32)
        check(_ffi.lib().rt3d_render(self.ctx._ctx, self._sc, C.byref(p), _ffi.ptr(out, C.c_float)))
        return out

    def accumulate(self, p, d_rgb_ptr):
        check(_ffi.lib().rt3d_accumulate(self.ctx._ctx, self._sc, C.byref(p), C.c_void_p(d_rgb_ptr)))

    def trace_batch(self, o, d, bounce_limit, seed=0):
        o, d = _ffi.as_f64(o).reshape(-1, 3), _ffi.as_f64(d).reshape(-1, 3)
        n = o.shape[0]
        rgb = np.empty((n, 3), np.float32)
        fp = np.empty(n, np.int32)
        ft = np.empty(n, np.float64)
        nb = np.empty(n, np.int32)
        check(_ffi.lib().rt3d_trace_batch(self.ctx._ctx, self._sc, n, _ffi.ptr(o, C.c_double), _ffi.ptr(d, C.c_double),
                                          int(bounce_limit), int(seed), _ffi.ptr(rgb, C.c_float), _ffi.ptr(fp, C.c_int32),
                                          _ffi.ptr(ft, C.c_double), _ffi.ptr(nb, C.c_int32)))
        return rgb, fp, ft, nb

    def close(self):
        if getattr(self, "_sc", None):
            _ffi.lib().rt3d_scene_destroy(self._sc)
            self._sc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_png(path, rgba8, width, height):
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    rc = _ffi.host().rth_png_write(str(path).encode(), _ffi.ptr(rgba8, C.c_uint8), int(width), int(height))
    if rc != 0:
        raise OSError(_ffi.host().rth_last_error().decode())
