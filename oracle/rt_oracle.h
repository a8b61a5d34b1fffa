/*
 * rt_oracle.h — CPU restatement of tatref/raytracer's per-ray hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (raytracer_amd/, include/)
 * may include, link or load this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the CPU timing.
 *
 * PARITY UNPINNED: the reference (pure Rust) ships no tests, golden vectors or
 * fixtures for this path and cannot be compiled in this environment (no Rust
 * toolchain, no vendored crates).  This restatement follows the reference
 * source line by line (citations on every function, paths relative to the
 * reference root) and is pinned only by hand-derived known-answer tests, the
 * published Philox4x32-10 known-answer vectors and the CIE table.  The glam /
 * palette formulas it relies on (normalize = v * (1/len), reflect, refract,
 * XYZ->sRGB matrix) are restated from those crates' published definitions.
 * The only outputs of the reference that exist - its gallery images - anchor
 * the rendered worlds statistically (tests/test_gallery_oracle.py), not bit
 * for bit; they also show that palette's into_color() clamps to [0, 1]
 * (largest byte 127 in every image).
 *
 * Arithmetic contract: f64 geometry, f32 radiance, no FMA contraction
 * (-ffp-contract=off), IEEE division and sqrt.  The reference draws randomness
 * from an unseeded thread-local generator; this restatement injects a
 * counter-based Philox4x32-10 stream instead (see "RNG streams" below) so that
 * the device path can consume exactly the same numbers.
 *
 * RNG streams (key = {seed_lo, seed_hi}); counter words {c0,c1,c2,c3}:
 *   2d render : c0 = pixel index (W*y+x), c1 = global sample index
 *       c2 = 0xFFFFFFFF, c3 = 0 : jitter.x = f64(w0,w1), jitter.y = f64(w2,w3)
 *       c2 = 0xFFFFFFFF, c3 = 1 + (l+2)/4 : word (l+2)%4 gives the wavelength
 *            index of lambda-sample l, idx = (w * 95) >> 32; the c3 = 1 call's
 *            words 0,1 give the ray angle
 *       c2 = l, c3 = (bounce << 20) | attempt : the two f64 of one
 *            rand_hemisphere rejection attempt at that bounce
 *   2d batch  : c0 = ray index (low 32 bits), c1 = 0x80000000 | (ray index >> 32),
 *       c2 = 0, c3 as above
 *   3d render : c0 = pixel index (W*j+i), c1 = sample index,
 *       c2 = 0xFFFFFFFF, c3 = 0 : jitter x,y
 *       c2 = 0, c3 = (bounce << 20) | k, k = 0: hemisphere x,y; k = 1: z
 *   3d batch  : c0/c1 as 2d batch, c2 = 0, c3 as 3d render
 *   f64(lo,hi) = ((hi << 32 | lo) >> 11) * 2^-53
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTO_NBINS 95

/* material kinds: 2d/src/rt_common.rs:403-431 */
enum { RTO_EMISSIVE = 0, RTO_DIRECTIONAL = 1, RTO_DIFFUSE = 2, RTO_REFLECTIVE = 3, RTO_DIELECTRIC = 4 };

typedef struct {
    int32_t kind;
    int32_t ior_kind;      /* 0 = Ior::Simple(a), 1 = Ior::Cauchy{a,b}  (rt_common.rs:433-450) */
    double ior_a, ior_b;
    double angle, d;       /* DirectionalEmissive */
    int32_t spectrum;      /* row of the spectra table: emission / albedo / absorption */
    int32_t pad_;
} rto2d_material;

/* Flat scene: the object list of rt_common.rs:541-546 with every Strip
 * expanded into its segments.  pid = position of the primitive in the scan
 * order (object order, then segment order inside a strip). */
typedef struct {
    int32_t n_circles, n_segments, n_objects, n_spectra;
    const double *ccx, *ccy, *cr;
    const int32_t *cobj, *cpid;
    const double *sax, *say, *sbx, *sby, *snx, *sny;
    const int32_t *sobj, *spid;
    const rto2d_material *materials;
    const float *spectra;  /* [n_spectra][95] */
} rto2d_scene;

typedef struct {
    int32_t width, height;
    int32_t spp;             /* samples rendered by this call */
    int32_t lambda_samples;
    int32_t recursion_limit;
    int32_t sample_offset;   /* global index of this call's first sample */
    uint64_t seed;
    double px_to_world;      /* 1.0 reproduces the reference (pixel == world) */
} rto2d_params;

typedef struct {
    int32_t n_spheres;
    const double *cx, *cy, *cz, *r;
    double cam_pos[3], cam_dir[3], cam_up[3];
    double focal_length;
    int32_t width, height;
} rto3d_scene;

typedef struct {
    int32_t spp, bounce_limit, sample_offset, pad_;
    uint64_t seed;
} rto3d_params;

/* building blocks (exposed for known-answer tests) */
void rto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void rto_sincos2pi(double u, double *s, double *c);
float rto_expf(float x);
void rto_resolve_pixel(const float spectrum[RTO_NBINS], float rgb[3]);
void rto_project_xyz(const float spectrum_sum[RTO_NBINS], float xyz[3]);
void rto_encode_xyz(const float xyz[3], float total_spp, float rgb[3]);
const float *rto_cmf_table(void); /* [95][3] */
void rto_reinhard_rgba8(const float *rgb, int64_t n_pixels, float param, uint8_t alpha, uint8_t *out);

/* single primitive tests: t<0 => miss; side 0 = Inside, 1 = Outside */
int rto2d_circle_hit(double ox, double oy, double dx, double dy, double cx, double cy, double r,
                     double *t, double *px, double *py, double *nx, double *ny, int *side);
int rto2d_segment_hit(double ox, double oy, double dx, double dy, double ax, double ay, double bx,
                      double by, double nx, double ny, double *t, double *px, double *py, int *side);
int rto3d_sphere_hit(const double o[3], const double d[3], const double c[3], double r, double *t,
                     double p[3], double n[3], int *side);

/* 2d */
int rto2d_trace_batch(const rto2d_scene *sc, int64_t n, const double *ox, const double *oy,
                      const double *dx, const double *dy, const int32_t *lambda_idx, int32_t depth,
                      uint64_t seed, float *radiance, int32_t *first_prim, double *first_t,
                      int32_t *bounces);
int rto2d_compute_pixel(const rto2d_scene *sc, const rto2d_params *p, int32_t px, int32_t py,
                        float *spectrum95, uint64_t *rays);
/* ReverseRenderer::denoise (2d/src/librt2d_reverse.rs:337-422) on RawImage planes rgb[W*H][3], weight[W*H]:
 * merged is updated in place; out_key (nullable) [W*H] = the value the reference sorts pixels by;
 * out_recomputed = size of the pixel set (pixels outside the image included, as in the reference). */
int rto2d_denoise(const rto2d_scene *sc, const rto2d_params *p, float top, int32_t mask_size,
                  float oversampling_factor, int32_t n_threads, float *merged_rgb, float *merged_weight,
                  const float *loop_rgb, float *out_key, int64_t *out_recomputed, uint64_t *rays);
/* whole image, reference decomposition: serial over columns, threads over rows */
/* Progressive accumulation in linear light (spectral sums + per-pixel sample counts); states the merge the GPU
 * library's rt2d_progressive_* entry points perform, see rt_oracle.c. */
int rto2d_progressive_pass(const rto2d_scene *sc, const rto2d_params *p, int32_t n_threads, float *sum,
                           float *count, float *merged_rgb, float *loop_rgb, uint64_t *rays);
int rto2d_progressive_resample(const rto2d_scene *sc, const rto2d_params *p, float top, int32_t mask_size,
                               int32_t n_threads, float *sum, float *count, float *merged_rgb,
                               const float *loop_rgb, int64_t *out_recomputed, uint64_t *rays);
/* One line through line_drawing's XiaolinWu iterator, each item added as draw_pixel(Add, ONE * v, weight 1). */
void rto_xiaolin_wu_add(double sx, double sy, double ex, double ey, int32_t W, int32_t H, float *rgb, float *weight);
/* ForwardRenderer::global_render (2d/src/librt2d_forward.rs:198-249) added onto RawImage planes; see rt_oracle.c. */
int rto2d_forward_render(const rto2d_scene *sc, int32_t width, int32_t height, int32_t rays_per_emitter, int32_t depth,
                         int32_t sample_offset, uint64_t seed, float *rgb, float *weight, double *out_lines,
                         int64_t *out_n_lines, uint64_t *rays);
int rto2d_render(const rto2d_scene *sc, const rto2d_params *p, int32_t n_threads, float *out_rgb,
                 float *out_weight, float *out_spectrum, uint64_t *rays);

/* 3d */
int rto3d_trace_batch(const rto3d_scene *sc, int64_t n, const double *o, const double *d,
                      int32_t bounce_limit, uint64_t seed, float *rgb, int32_t *first_prim,
                      double *first_t, int32_t *bounces);
int rto3d_camera_rays(const rto3d_scene *sc, const rto3d_params *p, int64_t n, const int32_t *pi,
                      const int32_t *pj, const int32_t *sample, double *o, double *d);
/* serial over pixels, threads over the samples of one pixel when
 * threads_over_spp != 0 (the reference's decomposition), else threads over rows */
int rto3d_render(const rto3d_scene *sc, const rto3d_params *p, int32_t n_threads,
                 int32_t threads_over_spp, float *out_rgb, uint64_t *rays);

#ifdef __cplusplus
}
#endif
#endif
