"""ctypes binding of oracle/build/librt_oracle.so — the CPU restatement used as
the checker.  Test infrastructure: nothing under raytracer_amd/ imports this."""
import ctypes as C

import numpy as np

from raytracer_amd._ffi import Scene2DDesc, Scene3DDesc, RenderParams3D

NBINS = 95
dp = C.POINTER(C.c_double)
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)


class OParams2D(C.Structure):
    # rto2d_params: note the field order differs from the product's rt2d_render_params
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("lambda_samples", C.c_int32),
                ("recursion_limit", C.c_int32), ("sample_offset", C.c_int32), ("seed", C.c_uint64),
                ("px_to_world", C.c_double)]


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, path):
        L = self.L = C.CDLL(path)
        L.rto_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.rto_sincos2pi.argtypes = [C.c_double, dp, dp]
        L.rto_expf.restype = C.c_float
        L.rto_expf.argtypes = [C.c_float]
        L.rto_resolve_pixel.argtypes = [fp, fp]
        L.rto_project_xyz.argtypes = [fp, fp]
        L.rto_encode_xyz.argtypes = [fp, C.c_float, fp]
        L.rto_cmf_table.restype = fp
        L.rto_reinhard_rgba8.argtypes = [fp, C.c_int64, C.c_float, C.c_uint8, C.POINTER(C.c_uint8)]
        L.rto2d_circle_hit.argtypes = [C.c_double] * 7 + [dp] * 5 + [C.POINTER(C.c_int)]
        L.rto2d_segment_hit.argtypes = [C.c_double] * 10 + [dp] * 3 + [C.POINTER(C.c_int)]
        L.rto3d_sphere_hit.argtypes = [dp, dp, dp, C.c_double, dp, dp, dp, C.POINTER(C.c_int)]
        L.rto2d_trace_batch.argtypes = [C.POINTER(Scene2DDesc), C.c_int64, dp, dp, dp, dp, ip, C.c_int32, C.c_uint64,
                                        fp, ip, dp, ip]
        L.rto2d_compute_pixel.argtypes = [C.POINTER(Scene2DDesc), C.POINTER(OParams2D), C.c_int32, C.c_int32, fp,
                                          C.POINTER(C.c_uint64)]
        L.rto2d_denoise.argtypes = [C.POINTER(Scene2DDesc), C.POINTER(OParams2D), C.c_float, C.c_int32, C.c_float, C.c_int32, fp, fp, fp,
                                    fp, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
        L.rto2d_render.argtypes = [C.POINTER(Scene2DDesc), C.POINTER(OParams2D), C.c_int32, fp, fp, fp,
                                   C.POINTER(C.c_uint64)]
        L.rto_xiaolin_wu_add.argtypes = [C.c_double] * 4 + [C.c_int32, C.c_int32, fp, fp]
        L.rto2d_forward_render.argtypes = [C.POINTER(Scene2DDesc), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64,
                                           fp, fp, dp, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
        L.rto2d_progressive_pass.argtypes = [C.POINTER(Scene2DDesc), C.POINTER(OParams2D), C.c_int32, fp, fp, fp, fp,
                                             C.POINTER(C.c_uint64)]
        L.rto2d_progressive_resample.argtypes = [C.POINTER(Scene2DDesc), C.POINTER(OParams2D), C.c_float, C.c_int32, C.c_int32,
                                                 fp, fp, fp, fp, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
        L.rto3d_trace_batch.argtypes = [C.POINTER(Scene3DDesc), C.c_int64, dp, dp, C.c_int32, C.c_uint64, fp, ip, dp, ip]
        L.rto3d_camera_rays.argtypes = [C.POINTER(Scene3DDesc), C.POINTER(RenderParams3D), C.c_int64, ip, ip, ip, dp, dp]
        L.rto3d_render.argtypes = [C.POINTER(Scene3DDesc), C.POINTER(RenderParams3D), C.c_int32, C.c_int32, fp,
                                   C.POINTER(C.c_uint64)]

    # ---- building blocks
    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.L.rto_philox4x32_10(c, k, o)
        return list(o)

    def sincos2pi(self, u):
        s, c = C.c_double(), C.c_double()
        self.L.rto_sincos2pi(float(u), C.byref(s), C.byref(c))
        return s.value, c.value

    def expf(self, x):
        return self.L.rto_expf(float(x))

    def cmf(self):
        return np.ctypeslib.as_array(self.L.rto_cmf_table(), shape=(NBINS, 3)).copy()

    def resolve_pixel(self, spectrum):
        s = np.ascontiguousarray(spectrum, np.float32)
        out = np.zeros(3, np.float32)
        self.L.rto_resolve_pixel(_p(s, C.c_float), _p(out, C.c_float))
        return out

    def project_xyz(self, spectrum_sums):
        """[n, 95] spectral sums -> [n, 3] XYZ sums."""
        s = np.ascontiguousarray(spectrum_sums, np.float32).reshape(-1, NBINS)
        out = np.zeros((s.shape[0], 3), np.float32)
        for i in range(s.shape[0]):
            self.L.rto_project_xyz(_p(s[i], C.c_float), _p(out[i], C.c_float))
        return out

    def encode_xyz(self, xyz, total_spp):
        x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        out = np.zeros_like(x)
        for i in range(x.shape[0]):
            self.L.rto_encode_xyz(_p(x[i], C.c_float), float(total_spp), _p(out[i], C.c_float))
        return out

    def reinhard(self, rgb, param=1.0, alpha=255):
        rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
        out = np.zeros((rgb.shape[0], 4), np.uint8)
        self.L.rto_reinhard_rgba8(_p(rgb, C.c_float), rgb.shape[0], float(param), int(alpha), _p(out, C.c_uint8))
        return out

    def circle_hit(self, o, d, c, r):
        t, px, py, nx, ny = (C.c_double() for _ in range(5))
        side = C.c_int()
        ok = self.L.rto2d_circle_hit(o[0], o[1], d[0], d[1], c[0], c[1], r, C.byref(t), C.byref(px), C.byref(py),
                                     C.byref(nx), C.byref(ny), C.byref(side))
        return None if not ok else dict(t=t.value, p=(px.value, py.value), n=(nx.value, ny.value), side=side.value)

    def segment_hit(self, o, d, a, b, n):
        t, px, py = (C.c_double() for _ in range(3))
        side = C.c_int()
        ok = self.L.rto2d_segment_hit(o[0], o[1], d[0], d[1], a[0], a[1], b[0], b[1], n[0], n[1], C.byref(t),
                                      C.byref(px), C.byref(py), C.byref(side))
        return None if not ok else dict(t=t.value, p=(px.value, py.value), side=side.value)

    def sphere_hit(self, o, d, c, r):
        o, d, c = (np.asarray(v, np.float64) for v in (o, d, c))
        t = C.c_double()
        p = np.zeros(3)
        n = np.zeros(3)
        side = C.c_int()
        ok = self.L.rto3d_sphere_hit(_p(o, C.c_double), _p(d, C.c_double), _p(c, C.c_double), float(r), C.byref(t),
                                     _p(p, C.c_double), _p(n, C.c_double), C.byref(side))
        return None if not ok else dict(t=t.value, p=p, n=n, side=side.value)

    # ---- 2d
    @staticmethod
    def params2d(p):
        o = OParams2D()
        for f, _ in OParams2D._fields_:
            setattr(o, f, getattr(p, f))
        return o

    def trace_batch2d(self, world, ox, oy, dx, dy, li, depth, seed=0):
        ox, oy, dx, dy = (np.ascontiguousarray(a, np.float64) for a in (ox, oy, dx, dy))
        li = np.ascontiguousarray(li, np.int32)
        n = len(ox)
        rad = np.zeros(n, np.float32)
        fpr = np.zeros(n, np.int32)
        ft = np.zeros(n, np.float64)
        nb = np.zeros(n, np.int32)
        rc = self.L.rto2d_trace_batch(C.byref(world.desc), n, _p(ox, C.c_double), _p(oy, C.c_double), _p(dx, C.c_double),
                                      _p(dy, C.c_double), _p(li, C.c_int32), int(depth), int(seed), _p(rad, C.c_float),
                                      _p(fpr, C.c_int32), _p(ft, C.c_double), _p(nb, C.c_int32))
        assert rc == 0, rc
        return rad, fpr, ft, nb

    def compute_pixel2d(self, world, p, x, y):
        out = np.zeros(NBINS, np.float32)
        rays = C.c_uint64()
        op = self.params2d(p)
        rc = self.L.rto2d_compute_pixel(C.byref(world.desc), C.byref(op), int(x), int(y), _p(out, C.c_float), C.byref(rays))
        assert rc == 0, rc
        return out, rays.value

    def denoise2d(self, world, p, top, mask_size, oversampling, merged_rgb, merged_weight, loop_rgb, threads=8):
        """Returns (merged_rgb, merged_weight, key, recomputed, rays); inputs are not modified."""
        rgb = np.ascontiguousarray(merged_rgb, np.float32).copy()
        w = np.ascontiguousarray(merged_weight, np.float32).copy()
        lp = np.ascontiguousarray(loop_rgb, np.float32)
        key = np.zeros(p.width * p.height, np.float32)
        n, rays = C.c_int64(), C.c_uint64()
        op = self.params2d(p)
        rc = self.L.rto2d_denoise(C.byref(world.desc), C.byref(op), float(top), int(mask_size), float(oversampling), int(threads),
                                  _p(rgb, C.c_float), _p(w, C.c_float), _p(lp, C.c_float), _p(key, C.c_float), C.byref(n),
                                  C.byref(rays))
        assert rc == 0, rc
        return rgb, w, key.reshape(p.height, p.width), n.value, rays.value

    def xiaolin_wu(self, start, end, width, height):
        """Coverage image [H, W] and hit-count image of one line drawn onto empty planes."""
        rgb = np.zeros((height, width, 3), np.float32)
        weight = np.zeros((height, width), np.float32)
        self.L.rto_xiaolin_wu_add(float(start[0]), float(start[1]), float(end[0]), float(end[1]), int(width), int(height),
                                  _p(rgb, C.c_float), _p(weight, C.c_float))
        assert np.array_equal(rgb[..., 0], rgb[..., 1]) and np.array_equal(rgb[..., 0], rgb[..., 2])
        return rgb[..., 0], weight

    def forward_render(self, world, width, height, rays_per_emitter=1000, depth=5, seed=0, sample_offset=0, n_emitters_max=64):
        """ForwardRenderer::global_render from RawImage::new: (rc, rgb, weight, lines [n, 5], n_lines, rays)."""
        rgb = np.zeros((height, width, 3), np.float32)
        weight = np.ones((height, width), np.float32)
        lines = np.zeros((n_emitters_max * rays_per_emitter, 5), np.float64)
        n, rays = C.c_int64(), C.c_uint64()
        rc = self.L.rto2d_forward_render(C.byref(world.desc), int(width), int(height), int(rays_per_emitter), int(depth),
                                         int(sample_offset), int(seed), _p(rgb, C.c_float), _p(weight, C.c_float),
                                         _p(lines, C.c_double), C.byref(n), C.byref(rays))
        return rc, rgb, weight, lines, n.value, rays.value

    def progressive_pass(self, world, p, sum_, count, threads=8):
        """In place on sum_ [H,W,95] and count [H,W]; returns (merged_rgb, loop_rgb, rays)."""
        merged = np.zeros((p.height, p.width, 3), np.float32)
        loop = np.zeros((p.height, p.width, 3), np.float32)
        rays = C.c_uint64()
        op = self.params2d(p)
        rc = self.L.rto2d_progressive_pass(C.byref(world.desc), C.byref(op), int(threads), _p(sum_, C.c_float), _p(count, C.c_float),
                                           _p(merged, C.c_float), _p(loop, C.c_float), C.byref(rays))
        assert rc == 0, rc
        return merged, loop, rays.value

    def progressive_resample(self, world, p, top, mask_size, sum_, count, merged_rgb, loop_rgb, threads=8):
        """In place on sum_, count and merged_rgb; p.spp = samples per selected pixel; returns (recomputed, rays)."""
        n, rays = C.c_int64(), C.c_uint64()
        op = self.params2d(p)
        lp = np.ascontiguousarray(loop_rgb, np.float32)
        rc = self.L.rto2d_progressive_resample(C.byref(world.desc), C.byref(op), float(top), int(mask_size), int(threads),
                                               _p(sum_, C.c_float), _p(count, C.c_float), _p(merged_rgb, C.c_float), _p(lp, C.c_float),
                                               C.byref(n), C.byref(rays))
        assert rc == 0, rc
        return n.value, rays.value

    def render2d(self, world, p, threads=8, want_spectrum=True):
        n = p.width * p.height
        rgb = np.zeros((p.height, p.width, 3), np.float32)
        w = np.zeros((p.height, p.width), np.float32)
        spec = np.zeros((p.height, p.width, NBINS), np.float32) if want_spectrum else None
        rays = C.c_uint64()
        op = self.params2d(p)
        rc = self.L.rto2d_render(C.byref(world.desc), C.byref(op), int(threads), _p(rgb, C.c_float), _p(w, C.c_float),
                                 _p(spec, C.c_float), C.byref(rays))
        assert rc == 0 and rgb.size == 3 * n, rc
        return rgb, w, spec, rays.value

    # ---- 3d
    def trace_batch3d(self, scene, o, d, bounce_limit, seed=0):
        o = np.ascontiguousarray(o, np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float64).reshape(-1, 3)
        n = o.shape[0]
        rgb = np.zeros((n, 3), np.float32)
        fpr = np.zeros(n, np.int32)
        ft = np.zeros(n, np.float64)
        nb = np.zeros(n, np.int32)
        rc = self.L.rto3d_trace_batch(C.byref(scene.desc), n, _p(o, C.c_double), _p(d, C.c_double), int(bounce_limit),
                                      int(seed), _p(rgb, C.c_float), _p(fpr, C.c_int32), _p(ft, C.c_double),
                                      _p(nb, C.c_int32))
        assert rc == 0, rc
        return rgb, fpr, ft, nb

    def camera_rays3d(self, scene, p, pi, pj, sample):
        pi, pj, sample = (np.ascontiguousarray(a, np.int32) for a in (pi, pj, sample))
        n = len(pi)
        o = np.zeros((n, 3))
        d = np.zeros((n, 3))
        rc = self.L.rto3d_camera_rays(C.byref(scene.desc), C.byref(p), n, _p(pi, C.c_int32), _p(pj, C.c_int32),
                                      _p(sample, C.c_int32), _p(o, C.c_double), _p(d, C.c_double))
        assert rc == 0, rc
        return o, d

    def render3d(self, scene, p, threads=8, threads_over_spp=False):
        h, w = scene.desc.height, scene.desc.width
        rgb = np.zeros((h, w, 3), np.float32)
        rays = C.c_uint64()
        rc = self.L.rto3d_render(C.byref(scene.desc), C.byref(p), int(threads), 1 if threads_over_spp else 0,
                                 _p(rgb, C.c_float), C.byref(rays))
        assert rc == 0, rc
        return rgb, rays.value
