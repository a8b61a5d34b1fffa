// __global__ kernels of the per-ray hot path (gfx950, wave64).
//
// Render kernels are tile-persistent: one workgroup owns a tile of 64 pixels (or
// 64 entries of a pixel list), keeps the tile's accumulator in LDS, and its
// lanes pull (pixel, sample, lambda-sample) work items from an LDS counter.  A
// lane whose path ended is refilled in place (ballot + mbcnt rank, one LDS
// atomic per wave), so the brute-force intersection loop always runs on full
// waves; rays never leave registers, and HBM sees the scene (L2-resident) and one
// read-modify-write of the tile's accumulator per launch.
#pragma once

#include "rt2d_device.hpp"
#include "rt3d_device.hpp"

namespace rtk {
using namespace rtd;

constexpr int TILE_PIXELS = 64; // 3d: 8 x 8
// 2d: a work unit is 8 x 6 pixels (x all samples and lambda-samples) and belongs to ONE wave, so every
// spectral bin it touches (48 x 380 B) is written by that wave only and stays in its XCD's L2.
constexpr int TILE2D_W = 8, TILE2D_H = 6, TILE2D_PIXELS = TILE2D_W * TILE2D_H;
constexpr int BLOCK = 256;

struct Render2dArgs {
    rt2d::SceneDev S;
    int32_t width, height, spp, lambda_samples, recursion_limit, sample_offset;
    uint32_t k0, k1;
    double px_to_world;
    int32_t list_mode; // 0: 8x6 image tiles, out[W*y+x][95] += sum; 1: pixel list, out[k][95] += sum (zeroed and scaled by the host side)
    int32_t tiles_x;
    int32_t n_tiles;
    // floor(x / lambda_samples) for x < 2^24 as (mulhi(x, ls_magic) + x) >> ls_shift: ls_shift = ceil(log2 lambda_samples),
    // ls_magic = ceil(2^(32 + ls_shift) / lambda_samples) - 2^32 (exact: the error term x e / 2^(32 + shift) stays below 1 / lambda_samples)
    uint32_t ls_magic;
    int32_t ls_shift;
    int32_t chunk, n_units; // a tile's items are handed out in runs of `chunk`; unit u = run u / n_tiles of tile u % n_tiles
    const int32_t *list_px, *list_py;
    int64_t n_list;
    float *out;
    unsigned long long *ray_counter;
    int32_t *tile_counter; // next unit to hand out, zeroed before the launch
    int32_t refill_min; // refill dead lanes once at least this many are dead (1..64)
    unsigned long long *prof_out; // diagnostic builds only
};

// ReverseRenderer::compute_pixel for every pixel of the image or list
// (2d/src/librt2d_reverse.rs:80-110), fused with trace_ray (:112-262).
//
// Waves are persistent and independent: each pulls work units (a run of one tile's (pixel, sample,
// lambda-sample) items) from a global counter, and its lanes pull items from a wave-private cursor.  A lane
// whose path ended is refilled in place, from the next unit when the current one runs out, so lanes idle
// only once, at the end of the launch; units are small enough (about 1 k items) that the waves finish within
// a few percent of each other, and they are numbered run-major so that the waves working at the same time are
// on different tiles.  A finished sample is added to its pixel's bin in HBM with a hardware f32 atomic
// (the tile's 18 KB stay in L2 while a unit lasts); nothing but the candidate queues lives in LDS.
__global__ __launch_bounds__(BLOCK, 6) void render2d_kernel(const Render2dArgs A)
{
    __shared__ rt2d::CoopLds coop[BLOCK / 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    const int per_pixel = A.spp * A.lambda_samples;
    const int total = TILE2D_PIXELS * per_pixel;
    const float inv_scale_l = (float)A.lambda_samples;

    rt2d::Path p;
    p.ox = p.oy = p.dx = p.dy = 0.0;
    p.thr = 0.0f; p.li = 0; p.depth_left = 0; p.c0 = p.c1 = p.c2 = 0;
    bool alive = false;
    bool exhausted = false; // wave-uniform: no tile left
    int tile = 0, cursor = 0, cursor_end = 0, tx = 0, ty = 0; // wave-uniform
    int64_t opix = 0; // row of `out` this lane's path adds to
    unsigned long long nrays = 0;
    Prof prof{};
#ifdef RT_PROF
    prof.last = __builtin_amdgcn_s_memtime();
#endif

    while (true) {
        RT_STAMP(prof, 4);
        const uint64_t dead = __ballot(!alive);
        const int ndead = __popcll(dead);
        if (!exhausted && (ndead >= A.refill_min || ndead == 64)) {
            if (cursor >= cursor_end) {
                int t = 0;
                if (lane == 0) t = atomicAdd(A.tile_counter, 1);
                const int unit = __builtin_amdgcn_readfirstlane(t);
                if (unit >= A.n_units) {
                    exhausted = true;
                } else {
                    const int run = unit / A.n_tiles;
                    tile = unit - run * A.n_tiles;
                    cursor = run * A.chunk;
                    cursor_end = (cursor + A.chunk < total) ? cursor + A.chunk : total;
                    tx = A.list_mode ? 0 : (tile % A.tiles_x) * TILE2D_W;
                    ty = A.list_mode ? 0 : (tile / A.tiles_x) * TILE2D_H;
                }
            }
            if (!exhausted) {
                if (!alive) {
                    const int item = cursor + rank_in_mask(dead);
                    if (item < cursor_end) {
                        const int rest = item / TILE2D_PIXELS;
                        const int lp = item - rest * TILE2D_PIXELS;
                        const int s = (int)((__umulhi((uint32_t)rest, A.ls_magic) + (uint32_t)rest) >> A.ls_shift);
                        const int l = rest - s * A.lambda_samples;
                        int px, py;
                        bool valid;
                        int64_t row;
                        if (A.list_mode) {
                            const int64_t k = (int64_t)tile * TILE2D_PIXELS + lp;
                            valid = k < A.n_list;
                            px = valid ? A.list_px[k] : 0;
                            py = valid ? A.list_py[k] : 0;
                            row = k;
                        } else {
                            px = tx + lp % TILE2D_W;
                            py = ty + lp / TILE2D_W;
                            valid = px < A.width && py < A.height;
                            row = (int64_t)A.width * py + px;
                        }
                        if (valid) {
                            rt2d::generate(p, px, py, A.width, (uint32_t)(A.sample_offset + s), (uint32_t)l, A.px_to_world,
                                           A.recursion_limit, A.k0, A.k1);
                            opix = row;
                            alive = true;
                        }
                    }
                }
                cursor += ndead; // items past the unit's end are not handed out: those lanes refill from the next unit
            }
        }
        if (!__any(alive)) {
            if (exhausted) break;
            continue;
        }
        const bool run = alive && p.depth_left > 0;
        RT_STAMP(prof, 0);
#ifdef RT_PROF
        prof.t[5] += 1;
        prof.t[6] += (unsigned long long)__popcll(__ballot(run));
#endif
        const rt2d::Hit h = rt2d::nearest_hit(A.S, p.ox, p.oy, p.dx, p.dy, run, coop[tid >> 6], prof);
        float value = 0.0f;
        double dnx = 0.0, dny = 0.0;
        rt2d::ShadeResult r = rt2d::SHADE_DONE;
        const uint32_t bounce_idx = (uint32_t)(A.recursion_limit - p.depth_left);
        if (run) {
            ++nrays;
            r = rt2d::shade(A.S, p, h, value, dnx, dny, prof);
        }
        RT_STAMP(prof, 12);
        rt2d::rand_hemisphere_wave(run && r == rt2d::SHADE_DIFFUSE, dnx, dny, p.c0, p.c1, p.c2, bounce_idx, A.k0, A.k1, p.dx,
                                   p.dy);
        RT_STAMP(prof, 13);
        if (alive) {
            if (r == rt2d::SHADE_DONE) {
                if (value != 0.0f) {
                    // per-sample term of `spectrum_spp / lambda_samples * 95` (librt2d_reverse.rs:104-105)
                    const float v = p.thr * value;
                    const float c = v / inv_scale_l * (float)NBINS;
                    unsafeAtomicAdd(A.out + opix * NBINS + p.li, c);
                }
                alive = false;
            } else if (p.depth_left == 0) {
                alive = false; // the recursive call returns 0 at depth 0 (:120-122)
            }
        }
        RT_STAMP(prof, 3);
    }
#ifdef RT_PROF
    if (lane == 0 && A.prof_out)
        for (int i = 0; i < 16; ++i) atomicAdd(A.prof_out + i, prof.t[i]);
#endif

    // rays traced by this wave -> one global atomic
    for (int off = 32; off > 0; off >>= 1) nrays += __shfl_down(nrays, off);
    if (lane == 0 && nrays != 0) atomicAdd(A.ray_counter, nrays);
}

// out[i] = out[i] / spp: `total_spectrum / spp` of compute_pixel (2d/src/librt2d_reverse.rs:109) for the pixel-list entry.
__global__ __launch_bounds__(BLOCK) void divide_kernel(float *__restrict__ out, int64_t n, float spp)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = out[i] / spp;
}

// ReverseRenderer::trace_ray for a fed batch (2d/src/librt2d_reverse.rs:112-262),
// one lane per ray, rays read coalesced from SoA arrays.
__global__ __launch_bounds__(BLOCK) void trace_batch2d_kernel(const rt2d::SceneDev S, int64_t n,
                                                             const double *__restrict__ ox, const double *__restrict__ oy,
                                                             const double *__restrict__ dx, const double *__restrict__ dy,
                                                             const int32_t *__restrict__ lambda_idx, int32_t depth,
                                                             uint32_t k0, uint32_t k1, float *__restrict__ radiance,
                                                             int32_t *__restrict__ first_prim, double *__restrict__ first_t,
                                                             int32_t *__restrict__ bounces, unsigned long long *ray_counter)
{
    __shared__ rt2d::CoopLds coop[BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool in_range = i < n;
    rt2d::Path p;
    p.ox = in_range ? ox[i] : 0.0; p.oy = in_range ? oy[i] : 0.0;
    p.dx = in_range ? dx[i] : 1.0; p.dy = in_range ? dy[i] : 0.0;
    p.li = in_range ? lambda_idx[i] : 0;
    p.thr = 1.0f;
    p.depth_left = depth;
    p.c0 = (uint32_t)i;
    p.c1 = 0x80000000u | (uint32_t)((uint64_t)i >> 32);
    p.c2 = 0u;
    bool alive = in_range;
    float out = 0.0f;
    int nb = 0, fp = -1;
    double ft = -1.0;
    while (__any(alive)) {
        const bool run = alive && p.depth_left > 0;
        Prof prof{};
        const rt2d::Hit h = rt2d::nearest_hit(S, p.ox, p.oy, p.dx, p.dy, run, coop[threadIdx.x >> 6], prof);
        float value = 0.0f;
        double dnx = 0.0, dny = 0.0;
        rt2d::ShadeResult r = rt2d::SHADE_DONE;
        const uint32_t bounce_idx = (uint32_t)(depth - p.depth_left);
        if (run) {
            if (nb == 0 && h.slot >= 0) { fp = h.pid; ft = h.t; }
            ++nb;
            r = rt2d::shade(S, p, h, value, dnx, dny, prof);
        }
        rt2d::rand_hemisphere_wave(run && r == rt2d::SHADE_DIFFUSE, dnx, dny, p.c0, p.c1, p.c2, bounce_idx, k0, k1, p.dx, p.dy);
        if (alive) {
            if (!run) {
                alive = false;
            } else if (r == rt2d::SHADE_DONE) {
                out = p.thr * value;
                alive = false;
            }
        }
    }
    if (in_range) {
        radiance[i] = out;
        if (first_prim) first_prim[i] = fp;
        if (first_t) first_t[i] = ft;
        if (bounces) bounces[i] = nb;
    }
    unsigned long long nr = (unsigned long long)nb;
    for (int off = 32; off > 0; off >>= 1) nr += __shfl_down(nr, off);
    if ((threadIdx.x & 63) == 0 && nr != 0) atomicAdd(ray_counter, nr);
}

// CIE 1931 2-degree colour matching functions, 360..830 nm step 5 nm
// (data table of 2d/src/spectrum.rs:8-104), [95][3].
__constant__ float CMF[NBINS * 3] = {
#include "cie1931_table.inc"
};

// Spectrum::to_xyz + to_dvec3 (2d/src/spectrum.rs:242-278) of accumulated / total_spp;
// the XYZ(D65) -> sRGB step is palette's: linear matrix, then the sRGB transfer
// function, unclamped.  One wave per 64 pixels: coalesced load of the 64 spectra
// into LDS, then one lane per pixel sums its 95 bins in index order.
__global__ __launch_bounds__(64) void resolve2d_kernel(const float *__restrict__ spectrum, int64_t n_pixels,
                                                      float total_spp, float *__restrict__ rgb,
                                                      float *__restrict__ weight)
{
    __shared__ float sp[64 * NBINS];
    const int64_t first = (int64_t)blockIdx.x * 64;
    const int64_t cnt = (n_pixels - first < 64) ? (n_pixels - first) : 64;
    const float *src = spectrum + first * NBINS;
    for (int i = threadIdx.x; i < cnt * NBINS; i += 64) sp[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x;
    if (lane >= cnt) return;
    float x = 0.0f, y = 0.0f, z = 0.0f;
    for (int b = 0; b < NBINS; ++b) {
        const float v = sp[lane * NBINS + b] / total_spp;
        x = x + v * CMF[3 * b + 0];
        y = y + v * CMF[3 * b + 1];
        z = z + v * CMF[3 * b + 2];
    }
    float lin[3];
    lin[0] = 3.2404542f * x + -1.5371385f * y + -0.4985314f * z;
    lin[1] = -0.9692660f * x + 1.8760108f * y + 0.0415560f * z;
    lin[2] = 0.0556434f * x + -0.2040259f * y + 1.0572252f * z;
    const int64_t pix = first + lane;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float c = lin[k];
        rgb[pix * 3 + k] = (c <= 0.0031308f) ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    }
    if (weight) weight[pix] = total_spp;
}

// RawImage::convert_to_image(Reinhard) (2d/src/img.rs:187-218, 3d/src/img.rs:126-157):
// `(v / (param + v) * 255.) as u8`, the cast saturating and NaN -> 0.
__global__ void reinhard_kernel(const float *__restrict__ rgb, int64_t n_pixels, float param, uint32_t alpha,
                                uint32_t *__restrict__ rgba)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    uint32_t px = alpha << 24;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = rgb[3 * i + k];
        const float m = v / (param + v) * 255.0f;
        uint32_t b = 0;
        if (m == m) b = (m <= 0.0f) ? 0u : (m >= 255.0f) ? 255u : (uint32_t)m;
        px |= b << (8 * k);
    }
    rgba[i] = px;
}

// ---------------------------------------------------------------------------
struct Render3dArgs {
    rt3d::SceneDev S;
    int32_t spp, bounce_limit, sample_offset, tiles_x;
    uint32_t k0, k1;
    float *out; // [W*H][3], accumulated
    unsigned long long *ray_counter;
    int32_t refill_min;
};

// World::render + compute_pixel (3d/src/rt.rs:238-273) over 8x8 pixel tiles.
__global__ __launch_bounds__(BLOCK) void render3d_kernel(const Render3dArgs A)
{
    __shared__ float acc[TILE_PIXELS * 3];
    __shared__ int next_item;
    __shared__ int32_t stack[rt3d::BVH_STACK * BLOCK];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    if (tid < TILE_PIXELS * 3) acc[tid] = 0.0f;
    if (tid == 0) next_item = 0;
    __syncthreads();
    const int total = TILE_PIXELS * A.spp;
    const int tx = (tile % A.tiles_x) * 8, ty = (tile / A.tiles_x) * 8;

    rt3d::Path p;
    p.ox = p.oy = p.oz = 0.0; p.dx = 1.0; p.dy = p.dz = 0.0;
    p.thr = 0.0f; p.left = 0; p.c0 = p.c1 = 0;
    bool alive = false, wave_done = false;
    int lpix = 0;
    unsigned long long nrays = 0;
    while (true) {
        const uint64_t dead = __ballot(!alive);
        const int ndead = __popcll(dead);
        if (!wave_done && (ndead >= A.refill_min || ndead == 64)) {
            int base = 0;
            if (rank_in_mask(dead) == 0 && !alive) base = atomicAdd(&next_item, ndead);
            const int src = __ffsll((long long)dead) - 1;
            base = __shfl(base, src);
            if (!alive) {
                const int item = base + rank_in_mask(dead);
                if (item < total) {
                    const int lp = item & (TILE_PIXELS - 1);
                    const int s = item >> 6;
                    const int i = tx + (lp & 7), j = ty + (lp >> 3);
                    // `if i < 10 && j < 10 { continue; }` (rt.rs:257-259)
                    if (i < A.S.width && j < A.S.height && !(i < 10 && j < 10) && A.bounce_limit > 0) {
                        rt3d::generate(A.S, p, i, j, (uint32_t)(A.sample_offset + s), A.bounce_limit, A.k0, A.k1);
                        lpix = lp;
                        alive = true;
                    }
                }
            }
            if (base + ndead >= total) wave_done = true;
        }
        if (!__any(alive)) {
            if (wave_done) break;
            continue;
        }
        const rt3d::Hit h = rt3d::nearest_hit(A.S, p.ox, p.oy, p.oz, p.dx, p.dy, p.dz, alive, stack + tid, BLOCK);
        if (alive) {
            ++nrays;
            float rgb[3];
            if (rt3d::shade(A.S, p, h, A.bounce_limit, A.k0, A.k1, rgb)) {
                if (rgb[0] != 0.0f || rgb[1] != 0.0f || rgb[2] != 0.0f) {
                    atomicAdd(&acc[lpix * 3 + 0], rgb[0]);
                    atomicAdd(&acc[lpix * 3 + 1], rgb[1]);
                    atomicAdd(&acc[lpix * 3 + 2], rgb[2]);
                }
                alive = false;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) nrays += __shfl_down(nrays, off);
    if ((tid & 63) == 0 && nrays != 0) atomicAdd(A.ray_counter, nrays);
    __syncthreads();
    if (tid < TILE_PIXELS * 3) {
        const int lp = tid / 3, k = tid - lp * 3;
        const int i = tx + (lp & 7), j = ty + (lp >> 3);
        if (i < A.S.width && j < A.S.height) {
            // draw_pixel at (i, H-1-j) (rt.rs:263-268)
            float *dst = A.out + ((int64_t)A.S.width * (A.S.height - j - 1) + i) * 3 + k;
            *dst = *dst + acc[tid];
        }
    }
}

// World::trace_ray for a fed batch (3d/src/rt.rs:216-236); o, d are [n][3].
__global__ __launch_bounds__(BLOCK) void trace_batch3d_kernel(const rt3d::SceneDev S, int64_t n,
                                                             const double *__restrict__ o, const double *__restrict__ d,
                                                             int32_t bounce_limit, uint32_t k0, uint32_t k1,
                                                             float *__restrict__ rgb_out, int32_t *__restrict__ first_prim,
                                                             double *__restrict__ first_t, int32_t *__restrict__ bounces,
                                                             unsigned long long *ray_counter)
{
    __shared__ int32_t stack[rt3d::BVH_STACK * BLOCK];
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool in_range = i < n;
    rt3d::Path p;
    p.ox = in_range ? o[3 * i] : 0.0; p.oy = in_range ? o[3 * i + 1] : 0.0; p.oz = in_range ? o[3 * i + 2] : 0.0;
    p.dx = in_range ? d[3 * i] : 1.0; p.dy = in_range ? d[3 * i + 1] : 0.0; p.dz = in_range ? d[3 * i + 2] : 0.0;
    p.thr = 1.0f;
    p.left = bounce_limit;
    p.c0 = (uint32_t)i;
    p.c1 = 0x80000000u | (uint32_t)((uint64_t)i >> 32);
    bool alive = in_range && bounce_limit > 0;
    float out[3] = { 0.0f, 0.0f, 0.0f };
    int nb = 0, fp = -1;
    double ft = -1.0;
    while (__any(alive)) {
        const rt3d::Hit h = rt3d::nearest_hit(S, p.ox, p.oy, p.oz, p.dx, p.dy, p.dz, alive, stack + threadIdx.x, BLOCK);
        if (alive) {
            if (nb == 0 && h.slot >= 0) { fp = h.sid; ft = h.t; }
            ++nb;
            float rgb[3];
            if (rt3d::shade(S, p, h, bounce_limit, k0, k1, rgb)) {
                out[0] = rgb[0]; out[1] = rgb[1]; out[2] = rgb[2];
                alive = false;
            }
        }
    }
    if (in_range) {
        rgb_out[3 * i] = out[0]; rgb_out[3 * i + 1] = out[1]; rgb_out[3 * i + 2] = out[2];
        if (first_prim) first_prim[i] = fp;
        if (first_t) first_t[i] = ft;
        if (bounces) bounces[i] = nb;
    }
    unsigned long long nr = (unsigned long long)nb;
    for (int off = 32; off > 0; off >>= 1) nr += __shfl_down(nr, off);
    if ((threadIdx.x & 63) == 0 && nr != 0) atomicAdd(ray_counter, nr);
}

} // namespace rtk
