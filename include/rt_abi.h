/*
 * rt_abi.h — C ABI of the MI355X-native per-ray hot path for tatref/raytracer.
 *
 * The reference is pure Rust and has no FFI of its own; these entry points are
 * what a Rust `extern "C"` block (bindgen output of this header) binds so that
 * the 2d crate's `ReverseRenderer::{global_render, render_column, compute_pixel,
 * trace_ray}` (2d/src/librt2d_reverse.rs:80-302) and the 3d crate's
 * `World::{render, compute_pixel, trace_ray}` (3d/src/rt.rs:216-273) keep their
 * Rust signatures and run on the GPU - and, one step out from that path, its
 * callers: `ReverseRenderer::denoise` and `endless_render`
 * (librt2d_reverse.rs:337-510) and `ForwardRenderer::global_render`
 * (librt2d_forward.rs:198-249).  Every declaration cites the reference code it
 * replaces (paths relative to the reference root).
 *
 * Conventions
 *  - plain pointers and sizes only; every function returns 0 on success or a
 *    negative RT_E_* code; no exception or abort crosses the boundary;
 *    `rt_last_error()` returns the calling thread's last message.
 *  - inputs are borrowed for the duration of the call and copied where they
 *    must outlive it (scene creation); outputs are caller-allocated; handles
 *    are freed only by the matching *_destroy.
 *  - a context is bound to one GPU; it may be used from any host thread but
 *    not from two threads at once (the reference calls global_render from a
 *    worker thread, 2d/src/bin/gui.rs:171-187, one render at a time).
 *  - arithmetic: f64 geometry, f32 radiance, no fused multiply-add — the
 *    reference's arithmetic (2d/src/rt_common.rs:61-64, spectrum.rs:113-115).
 *  - randomness: counter-based Philox4x32-10 keyed by `seed`; the reference's
 *    thread-local unseeded generator (librt2d_reverse.rs:89, rt_common.rs:370,
 *    379-380, spectrum.rs:186, 3d/src/rt.rs:41,181) has no reproducible
 *    stream to match.
 *  - there is no CPU fallback: every compute entry point fails with
 *    RT_E_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_NBINS 95 /* SPECTRUM_SAMPLES, 2d/src/spectrum.rs:106 */

/* The layout generation of the structs in this header.  A caller compiled against an older header passes structs of
 * another size (rt_stats was 32 bytes in generation 1 and is 56 since generation 2; rt_ctx_get_stats writes all of
 * it): compare rt_abi_version() with the RT_ABI_VERSION it was compiled with before the first call that takes a
 * struct.  No counterpart in the reference, which has no binary boundary. */
#define RT_ABI_VERSION 3

enum {
    RT_OK = 0,
    RT_E_INVALID = -1,   /* bad argument / malformed scene */
    RT_E_NO_DEVICE = -2, /* no usable GPU; there is no CPU path */
    RT_E_HIP = -3,       /* a HIP runtime call failed */
    RT_E_OOM = -4,
    RT_E_LIMIT = -5      /* size beyond what the kernels support */
};

/* Material variants, 2d/src/rt_common.rs:403-431 */
enum {
    RT_MAT_EMISSIVE = 0,
    RT_MAT_DIRECTIONAL_EMISSIVE = 1,
    RT_MAT_DIFFUSE = 2,
    RT_MAT_REFLECTIVE = 3,
    RT_MAT_DIELECTRIC = 4
};

/* One entry per reference `Object` (rt_common.rs:486-490): its Material with the
 * Spectrum moved into the shared spectra table.  Ior: rt_common.rs:433-450.   */
typedef struct rt2d_material {
    int32_t kind;
    int32_t ior_kind; /* 0 = Ior::Simple(ior_a), 1 = Ior::Cauchy{a: ior_a, b: ior_b} */
    double ior_a, ior_b;
    double angle, d;  /* DirectionalEmissive */
    int32_t spectrum; /* row of `spectra`: emission / Diffuse albedo / Dielectric absorption */
    int32_t pad_;
} rt2d_material;

/* Flat SoA form of `World.objects` (rt_common.rs:541-546): circles
 * (rt_common.rs:60-64) and segments (rt_common.rs:201-206, every Strip expanded).
 * `*_obj` = index of the owning object; `*_pid` = position of the primitive in
 * the reference's scan order (object order, then segment order inside a strip):
 * nearest-hit ties go to the lowest pid, as `min_by` / `Strip::hit` do
 * (librt2d_reverse.rs:135-139, rt_common.rs:295-305).  pids are 0..P-1, unique. */
typedef struct rt2d_scene_desc {
    int32_t n_circles, n_segments, n_objects, n_spectra;
    const double *circle_cx, *circle_cy, *circle_r;
    const int32_t *circle_obj, *circle_pid;
    const double *seg_ax, *seg_ay, *seg_bx, *seg_by, *seg_nx, *seg_ny;
    const int32_t *seg_obj, *seg_pid;
    const rt2d_material *materials; /* [n_objects] */
    const float *spectra;           /* [n_spectra][RT_NBINS] */
} rt2d_scene_desc;

/* ReverseRenderParams, librt2d_reverse.rs:43-52, plus what the reference leaves
 * implicit: the RNG seed, the global index of the first sample (sample-batch
 * sharding / progressive passes) and the pixel->world scale (the reference
 * renders 800x600 with pixel == world, librt2d_reverse.rs:89; 1.0 reproduces it,
 * 800.0/width keeps its scenes framed at other resolutions).                  */
typedef struct rt2d_render_params {
    int32_t width, height;
    int32_t spp;
    int32_t lambda_samples;
    int32_t recursion_limit;
    int32_t sample_offset;
    uint64_t seed;
    double px_to_world;
} rt2d_render_params;

/* 3d: `World.objects` (all Object::Sphere, 3d/src/rt.rs:47-69) and the arguments
 * of Camera::new (3d/src/rt.rs:139-147).                                      */
typedef struct rt3d_scene_desc {
    int32_t n_spheres;
    int32_t use_bvh; /* 0 = linear scan as the reference (rt.rs:221-224), 1 = BVH over sphere AABBs */
    const double *cx, *cy, *cz, *r;
    double cam_pos[3], cam_dir[3], cam_up[3];
    double focal_length;
    int32_t width, height;
} rt3d_scene_desc;

/* RenderParams, 3d/src/rt.rs:194-197 */
typedef struct rt3d_render_params {
    int32_t spp, bounce_limit, sample_offset, pad_;
    uint64_t seed;
} rt3d_render_params;

/* Counters since context creation or the last rt_ctx_reset_stats. */
typedef struct rt_stats {
    uint64_t rays;      /* trace_ray invocations with depth > 0 (nearest-hit queries) */
    uint64_t samples;   /* top-level trace_ray calls */
    uint64_t launches;  /* trace-kernel launches */
    double kernel_ms;   /* summed duration of those launches, HIP events on the context's stream */
    /* rt3d_render / rt3d_accumulate only - what the tree walk did, for pricing it: inner BVH nodes visited (each
     * tests its two child boxes) and exact Sphere::hit evaluations (3d/src/rt.rs:71-118) after the filters */
    uint64_t bvh_node_visits;
    uint64_t sphere_tests;
    /* nearest-hit searches actually run.  <= rays: the reference traces ONE ray per (pixel, sample) once per
     * lambda-sample (2d/src/librt2d_reverse.rs:90-101), and those calls all begin with the same search on the same
     * ray; rt2d_render / rt2d_accumulate / rt2d_compute_pixels run it once per pair and reuse the hit. */
    uint64_t searches;
} rt_stats;

typedef struct rt_ctx rt_ctx;
typedef struct rt2d_scene rt2d_scene;
typedef struct rt3d_scene rt3d_scene;

/* ---- library / context --------------------------------------------------- */
/* The library's RT_ABI_VERSION (see above): compare it with the one the caller was compiled with. */
int32_t rt_abi_version(void);
const char *rt_version(void);
const char *rt_last_error(void);
int32_t rt_device_count(int32_t *count);
int32_t rt_ctx_create(int32_t device_id, rt_ctx **out);
/* Fails with RT_E_INVALID, and leaves the context intact, while scenes created on it are still alive. */
int32_t rt_ctx_destroy(rt_ctx *ctx);
/* A new context enqueues its work on a non-blocking stream of its own.
 * rt_ctx_set_stream switches it to the caller's `hip_stream` (a hipStream_t; NULL is
 * HIP's null stream, as everywhere in HIP); rt_ctx_use_own_stream switches back. */
int32_t rt_ctx_set_stream(rt_ctx *ctx, void *hip_stream);
int32_t rt_ctx_use_own_stream(rt_ctx *ctx);
int32_t rt_ctx_synchronize(rt_ctx *ctx);
/* Both synchronize the stream first. */
int32_t rt_ctx_get_stats(rt_ctx *ctx, rt_stats *out);
int32_t rt_ctx_reset_stats(rt_ctx *ctx);
/* Self-check, no counterpart in the reference: the kernels compute sqrt(x) and 1 / sqrt(x) (glam's `normalize`,
 * the discriminants of Circle::hit and Sphere::hit) and x / 100 (the sky colour of the 3d tracer) with their own
 * instruction sequences, shorter than the compiler's generic ones.  This runs both forms on the GPU - the square roots
 * over 2^log2_n arguments of each of five classes (the tracer's range, random bit patterns with every special value,
 * significands of mostly ones or zeros, exact squares and neighbours, square roots next to a power of two), the
 * division, as class 5, over every float it is used on (all 2^32 patterns but the quotients that would be denormal,
 * -0 and the infinities) - and returns how many results differ in any bit: all six must be 0. */
int32_t rt_selftest_math(rt_ctx *ctx, int32_t log2_n, uint64_t seed, uint64_t *out_mismatches); /* [6] */

/* ---- 2d: scene ----------------------------------------------------------- */
/* Replaces borrowing `&World` (librt2d_reverse.rs:82,114): the arrays are copied. */
int32_t rt2d_scene_create(rt_ctx *ctx, const rt2d_scene_desc *desc, rt2d_scene **out);
int32_t rt2d_scene_destroy(rt2d_scene *scene);

/* ---- 2d: whole image ------------------------------------------------------
 * ReverseRenderer::global_render (librt2d_reverse.rs:282-302) for one pass of
 * `spp` samples: out_rgb[W*y+x][3] = Spectrum::to_dvec3 of the pixel's mean
 * spectrum (spectrum.rs:242-278: sRGB-encoded, each channel in [0, 1] as
 * palette's clamped into_color leaves it), out_weight[W*y+x] = spp (librt2d_reverse.rs:
 * 291-294, img.rs:20-24,93-99); out_spectrum (nullable) [W*y+x][95] = the value
 * compute_pixel returns.  Host pointers.                                      */
int32_t rt2d_render(rt_ctx *ctx, const rt2d_scene *scene, const rt2d_render_params *params,
                    float *out_rgb, float *out_weight, float *out_spectrum);

/* The two halves of rt2d_render on device memory, for sample-batch sharding over
 * GPUs and progressive passes (the role of `merged_image + loop_image`,
 * librt2d_reverse.rs:463-467, done on linear spectra):
 * rt2d_accumulate ADDS this call's samples into d_spectrum (device f32
 * [W*H][95]; per-sample term = value / lambda_samples * 95, librt2d_reverse.rs:
 * 102-107); rt2d_resolve divides by total_spp (:109), converts to sRGB
 * (spectrum.rs:242-278) and writes host out_rgb / out_weight (either nullable)
 * and, if d_rgb is non-NULL, device f32 [W*H][3].
 * d_spectrum must be ordinary device memory of this GPU (hipMalloc, a torch
 * tensor): samples are added with hardware f32 atomics, which host-pinned or
 * fine-grained memory does not support.                                       */
int32_t rt2d_accumulate(rt_ctx *ctx, const rt2d_scene *scene, const rt2d_render_params *params,
                        void *d_spectrum);
int32_t rt2d_resolve(rt_ctx *ctx, const void *d_spectrum, int32_t width, int32_t height,
                     int32_t total_spp, void *d_rgb, float *out_rgb, float *out_weight);

/* The same resolve in two steps, for a sum over GPUs in between: Spectrum::to_xyz
 * (spectrum.rs:242-262) is linear, so processes holding partial sums of one image
 * (sample-batch sharding) exchange their XYZ sums - device f32 [W*H][3], 12 bytes
 * per pixel - instead of their spectra.  rt2d_project_xyz writes
 * d_xyz[pix] = sum_b d_spectrum[pix][b] * cmf[b] (no division); after the
 * all-reduce rt2d_resolve_xyz divides by the total sample count and encodes
 * (to_dvec3, spectrum.rs:264-278), with rt2d_resolve's outputs.  Against
 * rt2d_resolve on the summed spectra only the order of f32 roundings differs.     */
int32_t rt2d_project_xyz(rt_ctx *ctx, const void *d_spectrum, int32_t width, int32_t height,
                         void *d_xyz);
int32_t rt2d_resolve_xyz(rt_ctx *ctx, const void *d_xyz, int32_t width, int32_t height,
                         int32_t total_spp, void *d_rgb, float *out_rgb, float *out_weight);

/* ---- 2d: pixel lists ------------------------------------------------------
 * ReverseRenderer::compute_pixel (librt2d_reverse.rs:80-110) for n pixels at
 * once; out_spectrum[k][95] is the returned Spectrum of pixel (px[k], py[k]).
 * render_column (librt2d_reverse.rs:264-279) is the list {(i, j) : j in 0..H};
 * the denoiser's re-sampling pass (librt2d_reverse.rs:407-413) is an arbitrary
 * list.  Pixels outside the image are legal (the reference computes them too). */
int32_t rt2d_compute_pixels(rt_ctx *ctx, const rt2d_scene *scene, const rt2d_render_params *params,
                            int64_t n, const int32_t *px, const int32_t *py, float *out_spectrum);

/* ---- 2d: adaptive re-sampling ----------------------------------------------
 * ReverseRenderer::denoise (librt2d_reverse.rs:337-422), one pass, on RawImage
 * planes: merged_rgb f32[W*H][3] and merged_weight f32[W*H] (updated in place),
 * loop_rgb f32[W*H][3] (read), idx = W*y + x (img.rs:93-99).
 * |merged - loop| -> 3x3 Laplace convolution (img.rs:167-185, its un-centred
 * window and empty 3-pixel border included) -> the `top` fraction of the pixels
 * by value, ties in the reference's sort order -> their +-mask_size
 * neighbourhoods -> compute_pixel on that set with `params` (spp, seed and
 * sample_offset as the caller chooses: the reference draws fresh random numbers)
 * -> Blending::Add of (to_dvec3, spp * oversampling_factor).  out_recomputed
 * (nullable) = size of the set, pixels outside the image included.
 * rt2d_denoise takes host planes (as rt2d_render returns them); rt2d_denoise_device
 * takes planes in this GPU's memory and leaves its work on the context's stream. */
int32_t rt2d_denoise(rt_ctx *ctx, const rt2d_scene *scene, const rt2d_render_params *params, float top,
                     int32_t mask_size, float oversampling_factor, float *merged_rgb,
                     float *merged_weight, const float *loop_rgb, int64_t *out_recomputed);
int32_t rt2d_denoise_device(rt_ctx *ctx, const rt2d_scene *scene, const rt2d_render_params *params,
                            float top, int32_t mask_size, float oversampling_factor,
                            void *d_merged_rgb, void *d_merged_weight, const void *d_loop_rgb,
                            int64_t *out_recomputed);

/* ---- 2d: progressive rendering ---------------------------------------------
 * The loop of ReverseRenderer::endless_render (librt2d_reverse.rs:424-510) with
 * its state resident on the GPU.  Where the reference merges passes as
 * `merged_image + loop_image` on encoded colours and drops the weights
 * (img.rs:257-274), this object keeps per pixel the spectral SUM of every sample
 * taken (f32[W*H][95]) and the COUNT of those samples (f32[W*H]); the image is
 * sRGB(sum / count).  A step is one iteration of the reference's loop:
 *   pass:      global_render with the next `spp` samples of every pixel
 *              (:436 the first time, :463 after) -> loop image = sRGB(pass/spp);
 *              sum += pass, count += spp; merged image = sRGB(sum / count);
 *   resample:  from the second step on, `denoiser.passes` times
 *              ReverseRenderer::denoise (:337-422) on (merged, loop): the same
 *              selection as rt2d_denoise; the selected pixels then take
 *              floor(spp * oversampling_factor) (at least 1) new samples each,
 *              which join sum and count - the reference books exactly that
 *              weight (:417) while tracing only spp samples.
 * Every pass draws the next unused global sample indices, starting at
 * params->sample_offset.  The scene must outlive the object; calls are ordered
 * on the context's stream.                                                      */
typedef struct rt2d_progressive rt2d_progressive;

typedef struct rt2d_denoiser { /* Denoiser, librt2d_reverse.rs:55-60 */
    float top;
    int32_t mask_size;
    float oversampling_factor;
    int32_t passes;
} rt2d_denoiser;

typedef struct rt2d_progress { /* RenderProgress (rt_common.rs:563-566) without the image */
    int64_t loops;       /* whole-image passes merged so far */
    int64_t next_sample; /* first unused global sample index */
    int64_t recomputed;  /* pixels re-sampled by the last step (all its passes, out-of-image ones included) */
    uint64_t rays;       /* rays traced by the last step */
} rt2d_progress;

/* denoiser may be NULL (`denoiser: None`). */
int32_t rt2d_progressive_create(rt_ctx *ctx, const rt2d_scene *scene, const rt2d_render_params *params,
                                const rt2d_denoiser *denoiser, rt2d_progressive **out);
int32_t rt2d_progressive_destroy(rt2d_progressive *prog);
/* One iteration of the loop; out (nullable) describes the state after it. */
int32_t rt2d_progressive_step(rt2d_progressive *prog, rt2d_progress *out);
/* The two halves of a step, for callers that schedule them themselves:
 * _pass is the whole-image pass and merge; _resample is ONE adaptive pass with
 * the object's denoiser settings (RT_E_INVALID without a denoiser or before the
 * first pass); out_recomputed nullable. */
int32_t rt2d_progressive_pass(rt2d_progressive *prog);
int32_t rt2d_progressive_resample(rt2d_progressive *prog, int64_t *out_recomputed);
/* RenderProgress hand-off (librt2d_reverse.rs:450-456, gui.rs:449-475): the merged image after
 * convert_to_image(Reinhard{param}) as host RGBA8 [W*H][4], alpha 255 (img.rs:187-218).        */
int32_t rt2d_progressive_image(rt2d_progressive *prog, float reinhard_param, uint8_t *out_rgba8);
/* Host copies of the state; every pointer nullable: sum f32[W*H][95], count f32[W*H],
 * merged_rgb and loop_rgb f32[W*H][3].                                                         */
int32_t rt2d_progressive_read(rt2d_progressive *prog, float *out_sum, float *out_count, float *out_merged_rgb,
                              float *out_loop_rgb);
/* The same four planes as device pointers (valid until destroy; nullable outputs). */
int32_t rt2d_progressive_planes(rt2d_progressive *prog, void **d_sum, void **d_count, void **d_merged_rgb,
                                void **d_loop_rgb);

/* ---- 2d: forward light tracer ----------------------------------------------
 * ForwardRenderer::global_render (librt2d_forward.rs:198-249): every emissive
 * object shoots rays_per_emitter rays (the reference: 1000) from its rim
 * (Circle::sample_out_ray, rt_common.rs:113-120), each traced for at most
 * `depth` hit searches (the reference: 5) by ForwardRenderer::trace_ray
 * (:66-195) at wavelength index 0 / lambda = 1.; the reference's trace_ray
 * returns what its innermost call returns, so a path yields its last ray when
 * that ray leaves the scene, and that ray is drawn from its origin to
 * origin + 1000 dir with line_drawing's XiaolinWu iterator (pixel == world
 * unit), each item added as draw_pixel(Add, value = coverage on all three
 * channels, weight 1).  An emitter that is not a circle is RT_E_INVALID (the
 * reference panics: sample_out_ray is todo!() for segments and strips).
 * rt2d_forward_accumulate ADDS onto device RawImage planes (rgb f32[W*H][3],
 * weight f32[W*H]) - one term of `merged + loop` in ForwardRenderer's
 * endless_render (:263-296); rt2d_forward_render returns host planes that start
 * from RawImage::new (value 0, weight 1).  out_lines (nullable) = rays drawn. */
typedef struct rt2d_forward_params {
    int32_t width, height;
    int32_t rays_per_emitter;
    int32_t depth;
    int32_t sample_offset; /* index of the first ray of every emitter: passes that add up use disjoint ranges */
    int32_t pad_;
    uint64_t seed;
} rt2d_forward_params;

int32_t rt2d_forward_accumulate(rt_ctx *ctx, const rt2d_scene *scene, const rt2d_forward_params *params,
                                void *d_rgb, void *d_weight, int64_t *out_lines);
int32_t rt2d_forward_render(rt_ctx *ctx, const rt2d_scene *scene, const rt2d_forward_params *params,
                            float *out_rgb, float *out_weight, int64_t *out_lines);
/* rt2d_forward_accumulate for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same inputs, same
 * result on d_rgb / d_weight; what differs: its scratch (the buffer of last rays between the trace and the
 * drawing kernel, the coverage plane, the line counter) is a fixed array inside the library's code object, 11 MB per
 * GPU, instead of the context's grow-only workspace, and the call makes no hipMalloc / hipFree, creates no event
 * and never synchronises.  Limits: width * height <= 2^20 pixels (RT_E_LIMIT beyond; the reference draws 800 x 600).
 * d_lines (nullable) is DEVICE memory, one uint64: it receives the number of lines drawn, on the context's stream.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two contexts or two streams)
 * race; calls ordered on one stream do not.  In rt_stats such a call shows in `rays` and `samples` only: with no
 * event it adds nothing to `kernel_ms` and `launches`, so rays / kernel_ms of a context that mixes the two forward
 * entry points overstates the kernels' rate.                                                                    */
int32_t rt2d_forward_accumulate_static(rt_ctx *ctx, const rt2d_scene *scene, const rt2d_forward_params *params,
                                       void *d_rgb, void *d_weight, void *d_lines);

/* ---- 2d: fed ray batches --------------------------------------------------
 * ReverseRenderer::trace_ray (librt2d_reverse.rs:112-262) for n rays:
 * radiance[k] = trace_ray(world, Ray2d{(ox,oy),(dx,dy)}, lambda_idx[k],
 * 5*lambda_idx[k]+360, depth).  Also returns what the reference computes but
 * does not return: pid of the first nearest hit (-1 = miss), its distance
 * (-1.0 = miss) and the number of trace_ray invocations with depth > 0.
 * Optional outputs may be NULL.  Host pointers.                               */
int32_t rt2d_trace_batch(rt_ctx *ctx, const rt2d_scene *scene, int64_t n, const double *ox,
                         const double *oy, const double *dx, const double *dy,
                         const int32_t *lambda_idx, int32_t depth, uint64_t seed, float *radiance,
                         int32_t *first_prim, double *first_t, int32_t *bounces);

/* ---- 3d ------------------------------------------------------------------- */
int32_t rt3d_scene_create(rt_ctx *ctx, const rt3d_scene_desc *desc, rt3d_scene **out);
int32_t rt3d_scene_destroy(rt3d_scene *scene);
/* World::render (3d/src/rt.rs:252-273): out_rgb[W*(H-1-j)+i][3] = sum over spp of
 * trace_ray (no mean, rt.rs:239-246); the 10x10 corner i<10 && j<10 stays 0.  */
int32_t rt3d_render(rt_ctx *ctx, const rt3d_scene *scene, const rt3d_render_params *params,
                    float *out_rgb);
/* Device-memory form: ADDS this call's samples into d_rgb (device f32 [W*H][3]). */
int32_t rt3d_accumulate(rt_ctx *ctx, const rt3d_scene *scene, const rt3d_render_params *params,
                        void *d_rgb);
/* World::trace_ray (3d/src/rt.rs:216-236) for n rays; o, d are [n][3]. */
int32_t rt3d_trace_batch(rt_ctx *ctx, const rt3d_scene *scene, int64_t n, const double *o,
                         const double *d, int32_t bounce_limit, uint64_t seed, float *rgb,
                         int32_t *first_prim, double *first_t, int32_t *bounces);

/* ---- image output ---------------------------------------------------------
 * RawImage::convert_to_image(Reinhard{param}) (2d/src/img.rs:187-218, alpha 255;
 * 3d/src/img.rs:126-157, param 1, alpha 254): rgba8 = (v / (param + v) * 255) as u8.
 * Host pointers; runs on the GPU like everything else here.                   */
int32_t rt_tonemap_reinhard(rt_ctx *ctx, const float *rgb, int64_t n_pixels, float param,
                            uint8_t alpha, uint8_t *out_rgba8);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
