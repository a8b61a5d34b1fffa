// __global__ kernels of the per-ray hot path (gfx950, wave64).
//
// The render kernels run persistent, independent waves: the grid is what fits on the chip, every wave pulls work
// units (runs of (pixel, sample, lambda-sample) items of one small image tile) from a global counter, a lane owns one
// path in registers, and a lane whose path ended takes the next primary ray from its wave's pool in LDS (ballot +
// mbcnt rank), so the brute-force intersection loop always runs on full waves.  Rays never leave the chip; HBM sees
// the scene (a few KB, scalar-cache / L2 resident) and one hardware f32 atomic per finished sample into the
// caller's accumulation buffer.  LDS holds only the per-wave candidate queues and ray pools.
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
    // 1 / lambda_samples when that is a power of two (then x * inv == x / lambda_samples for every float x: both are
    // the exactly scaled value, rounded once), else 0: the kernel divides
    float inv_lambda_pow2;
    // a tile's (pixel, sample) pairs - 48 x spp of them, pixel fastest - are handed out in runs of `chunk` pairs;
    // unit u = run u / n_tiles of tile u % n_tiles; every pair stands for lambda_samples items
    int32_t chunk, n_units;
    const int32_t *list_px, *list_py;
    int64_t n_list;
    float *out;
    unsigned long long *ray_counter;
    int32_t *tile_counter; // next unit to hand out, zeroed before the launch
    int32_t refill_min; // refill dead lanes once at least this many are dead (1..64)
    struct PairSlab *pairs;              // one per resident wave
    unsigned long long *shared_counter;  // rays that reused the nearest hit of their pair's first lambda-sample
    unsigned long long *prof_out; // diagnostic builds only
};

// The circle filter's padded squared radii for a launch whose frame has padding `pad` (see rt2d::filter_word):
// r^2 (1 + 2^-22) + pad, rounded up to f32.  The slot that pads an odd circle count keeps its -inf.
__global__ __launch_bounds__(BLOCK) void pad_radii_kernel(const rt2d::CircleX *__restrict__ cex, int n_real, double pad, rt2d::CirclePair *__restrict__ c32)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_real) return;
    const double v = cex[k].r2 * (1.0 + 0x1.0p-22) + pad;
    float f = (float)v;
    if ((double)f < v) f = __uint_as_float(__float_as_uint(f) + 1u); // v > 0: next float up
    c32[k >> 1].e[k & 1] = f;
}

// Per-wave LDS of render2d_kernel, 14 dwords per lane.  Between steps it is the pool of rays that have taken their
// first bounce and wait for a free lane (rows 0..12: origin and direction as four doubles, collected factors,
// wavelength bin | lambda-sample << 8, output row, Philox counter words); during an intake step, while it is empty,
// the lanes park their own paths in it (all 14 rows; the last holds depth_left (< 2^12) | lambda-sample << 12).
// Together with the candidate queue that is 5 632 bytes per wave: seven workgroups fill 157.7 of a CU's 160 KB.
// WIDE (the kernels without a candidate queue, which have the room): depth_left and the lambda-sample in rows of
// their own, two instructions less per intake step.
template <bool WIDE>
struct ItemPool {
    double ox[64], oy[64], dx[64], dy[64];
    float thr[64];
    uint32_t meta[64], row[64], c0[64], c1[64];
    uint32_t park[64];
    uint32_t park_c2[WIDE ? 64 : 1];
};

// The (pixel, sample) pairs a wave is working through, in global memory (one slab per resident wave, written once
// per 64 pairs and read once per further lambda-sample, coalesced): the primary ray and ITS nearest hit
// (slot >= 0, or -1 for a miss).  44 bytes per pair.
struct PairSlab {
    double ox[64], oy[64], dx[64], dy[64], t[64];
    int32_t slot[64];
};

// ReverseRenderer::compute_pixel for every pixel of the image or list
// (2d/src/librt2d_reverse.rs:80-110), fused with trace_ray (:112-262).
//
// Waves are persistent and independent: each pulls work units (a run of one tile's (pixel, sample) pairs) from a
// global counter.  A lane owns one path in registers; a lane whose path ended takes a waiting ray from the wave's
// pool, so lanes idle only once, at the end of the launch.  A step of the wave - nearest hit, shading, the diffuse
// direction, the bookkeeping of finished samples - advances either the lanes' own paths or, when the pool is empty
// and a lane is free, takes in new work (an INTAKE step): the 64 lanes park their paths in LDS and run the same
// step on 64 pairs at once, one lambda-sample of each, and the rays that go on after that first bounce fill the
// pool.
// The reference draws ONE ray per (pixel, sample) and traces it once per lambda-sample (:90-101): all those
// trace_ray calls start with the same nearest-hit search on the same ray.  The intake step of a pair's first
// lambda-sample runs that search (at full lane use) and keeps ray and hit in the wave's slab; the intake steps of
// its other lambda-samples read them back and go straight to the shading, which is where the wavelength first
// matters.  Same arithmetic on the same operands, so the hit is the one every call would have found; the ray
// counter still counts every trace_ray invocation, the search counter what was actually searched.
// Units are small enough (about 1 k items) that the waves finish within a few percent of each other, and they are
// numbered run-major so that the waves working at the same time are on different tiles.  A finished sample is added
// to its pixel's bin in HBM with a hardware f32 atomic (the tile's 18 KB stay in L2 while a unit lasts).
template <bool SEGW, bool SMALL = false, bool NOSEG = false>
__global__ __launch_bounds__(BLOCK, 7) void render2d_kernel(const Render2dArgs A)
{
    __shared__ rt2d::CoopLds coop[SMALL ? 1 : BLOCK / 64];
    __shared__ ItemPool<SMALL> pools[BLOCK / 64];
    // the arguments are read in place (kernel-argument segment), see rt2d::SceneK
    typedef const Render2dArgs __attribute__((address_space(4))) *ArgsK;
    ArgsK Ak = (ArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A;
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    const int n_lambda = Ak->lambda_samples;
    const int total = TILE2D_PIXELS * Ak->spp; // pairs of a tile

    rt2d::Path p;
    p.ox = p.oy = p.dx = p.dy = 0.0;
    p.thr = 0.0f; p.li = 0; p.depth_left = 0; p.c0 = p.c1 = p.c2 = 0;
    bool alive = false;
    bool exhausted = false; // wave-uniform: no unit left
    int tile = 0, cursor = 0, cursor_end = 0, tx = 0, ty = 0; // wave-uniform; the cursor counts pairs
    int pool_pos = 0, pool_items = 0;                         // wave-uniform
    int set_lambda = n_lambda; // wave-uniform: next lambda-sample of the pair set in the slab to take in (n_lambda: none left)
    // the wave's index as a scalar: the addresses of its pool and slab rows are then scalar base + lane offset
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ItemPool<SMALL> &pool = pools[wave];
    PairSlab &slab = Ak->pairs[blockIdx.x * (BLOCK / 64) + wave];
    uint32_t orow = 0;     // row of `out` this lane's path adds to
    uint32_t set_bins = 0; // wavelength bins of four consecutive lambda-samples of this lane's pair of the set, one byte each
    uint32_t nrays = 0, nsearch = 0;
    Prof prof{};
#ifdef RT_PROF
    prof.last = __builtin_amdgcn_s_memtime();
#endif

    while (true) {
        asm volatile("" : "+s"(Ak)); // arguments are re-read where a step needs them, not held in registers across it
        RT_STAMP(prof, 4);
        // ---- free lanes take waiting rays; if none wait, this step takes in new work
        bool intake = false; // wave-uniform
        {
            const uint64_t dead = __ballot(!alive);
            const int ndead = __popcll(dead);
            if (ndead >= Ak->refill_min || ndead == 64) {
                const int avail = pool_items - pool_pos;
                const int take = ndead < avail ? ndead : avail;
                const int r = rank_in_mask(dead);
                if (!alive && r < take) {
                    const int i = pool_pos + r;
                    p.ox = pool.ox[i]; p.oy = pool.oy[i]; p.dx = pool.dx[i]; p.dy = pool.dy[i];
                    p.thr = pool.thr[i];
                    const uint32_t m = pool.meta[i];
                    p.li = (int)(m & 0xFFu);
                    p.c2 = m >> 8;
                    orow = pool.row[i];
                    p.c0 = pool.c0[i]; p.c1 = pool.c1[i];
                    p.depth_left = Ak->recursion_limit - 1;
                    alive = true;
                }
                pool_pos += take;
                intake = take < ndead && (set_lambda < n_lambda || !exhausted);
            }
        }
        if (intake && set_lambda >= n_lambda) {
            // the next 64 pairs: of this unit, or of the next one
            if (cursor >= cursor_end) {
                int t = 0;
                if (lane == 0) t = atomicAdd(Ak->tile_counter, 1);
                const int unit = __builtin_amdgcn_readfirstlane(t);
                if (unit >= Ak->n_units) {
                    exhausted = true;
                    intake = false;
                } else {
                    const int run = unit / Ak->n_tiles;
                    tile = unit - run * Ak->n_tiles;
                    cursor = run * Ak->chunk;
                    cursor_end = (cursor + Ak->chunk < total) ? cursor + Ak->chunk : total;
                    tx = Ak->list_mode ? 0 : (tile % Ak->tiles_x) * TILE2D_W;
                    ty = Ak->list_mode ? 0 : (tile / Ak->tiles_x) * TILE2D_H;
                }
            }
            if (intake) set_lambda = 0;
        }
        if (!intake && !__any(alive)) {
            if (exhausted && set_lambda >= n_lambda && pool_pos >= pool_items) break;
            continue;
        }

        // ---- an intake step works on lambda-sample `set_lambda` of the set's 64 pairs instead of the lanes' own paths
        bool act = alive;        // lanes this step advances
        bool have_hit = false;   // wave-uniform: the nearest hit comes from the slab
        rt2d::Hit h;
        h.t = 0.0; h.slot = -1; h.pid = 0x7fffffff;
        if (intake) {
            __builtin_amdgcn_wave_barrier();
            pool.ox[lane] = p.ox; pool.oy[lane] = p.oy; pool.dx[lane] = p.dx; pool.dy[lane] = p.dy;
            pool.thr[lane] = p.thr; pool.meta[lane] = (uint32_t)p.li; pool.row[lane] = orow;
            pool.c0[lane] = p.c0; pool.c1[lane] = p.c1;
            if (SMALL) {
                pool.park[lane] = (uint32_t)p.depth_left; pool.park_c2[lane] = p.c2;
            } else {
                pool.park[lane] = (uint32_t)p.depth_left | (p.c2 << 12);
            }
            asm volatile("" ::: "memory");
            // which pair this lane stands for: the same for every lambda-sample of the set, so it is recomputed
            // rather than kept in the slab
            const int pair = (set_lambda == 0 ? cursor : cursor - 64) + lane;
            if (set_lambda == 0) cursor += 64;
            bool valid = pair < cursor_end;
            int px = 0, py = 0;
            int64_t row = 0;
            const int s = pair / TILE2D_PIXELS;
            const int lp = pair - s * TILE2D_PIXELS;
            if (Ak->list_mode) {
                const int64_t k = (int64_t)tile * TILE2D_PIXELS + lp;
                valid = valid && k < Ak->n_list;
                px = valid ? Ak->list_px[k] : 0;
                py = valid ? Ak->list_py[k] : 0;
                row = k;
            } else {
                px = tx + lp % TILE2D_W;
                py = ty + lp / TILE2D_W;
                valid = valid && px < Ak->width && py < Ak->height;
                row = (int64_t)Ak->width * py + px;
            }
            valid = valid && Ak->recursion_limit > 0; // depth 0: trace_ray returns 0 before anything else (:120-122)
            const uint32_t gs = (uint32_t)(Ak->sample_offset + s);
            orow = (uint32_t)row;
            p.c0 = (uint32_t)(Ak->width * py + px);
            p.c1 = gs;
            if (set_lambda == 0) {
                rt2d::generate_primary(p.ox, p.oy, p.dx, p.dy, set_bins, px, py, Ak->width, gs, Ak->px_to_world, Ak->k0, Ak->k1);
            } else {
                have_hit = true;
                p.ox = slab.ox[lane]; p.oy = slab.oy[lane]; p.dx = slab.dx[lane]; p.dy = slab.dy[lane];
                h.t = slab.t[lane];
                h.slot = slab.slot[lane];
                // lambda-sample l draws word (l + 2) % 4 of Philox call 1 + (l + 2) / 4 of the pair's stream
                if (((set_lambda + 2) & 3) == 0) set_bins = rt2d::lambda_bins4(p.c0, p.c1, (uint32_t)(set_lambda + 2) >> 2, Ak->k0, Ak->k1);
            }
            p.thr = 1.0f;
            p.li = (int)((set_bins >> (8 * ((set_lambda + 2) & 3))) & 0xFFu);
            p.c2 = (uint32_t)set_lambda;
            p.depth_left = Ak->recursion_limit;
            act = valid;
        }
        const bool run = act && p.depth_left > 0;
        RT_STAMP(prof, 0);
#ifdef RT_PROF
        prof.t[5] += 1;
        prof.t[6] += (unsigned long long)__popcll(__ballot(run));
#endif
        if (!have_hit) {
            h = rt2d::nearest_hit<SEGW, SMALL, NOSEG>((rt2d::SceneK)&Ak->S, p.ox, p.oy, p.dx, p.dy, run, coop[SMALL ? 0 : (tid >> 6)], prof);
            if (run) ++nsearch;
            if (intake) {
                slab.ox[lane] = p.ox; slab.oy[lane] = p.oy; slab.dx[lane] = p.dx; slab.dy[lane] = p.dy;
                slab.t[lane] = h.t;
                slab.slot[lane] = h.slot;
            }
        }
        float value = 0.0f;
        double dnx = 0.0, dny = 0.0;
        rt2d::ShadeResult r = rt2d::SHADE_DONE;
        const uint32_t bounce_idx = (uint32_t)(Ak->recursion_limit - p.depth_left);
        if (run) {
            ++nrays;
            r = rt2d::shade((rt2d::SceneK)&Ak->S, p, h, value, dnx, dny, prof);
        }
        RT_STAMP(prof, 12);
        rt2d::rand_hemisphere_wave(run && r == rt2d::SHADE_DIFFUSE, dnx, dny, p.c0, p.c1, p.c2, bounce_idx, Ak->k0, Ak->k1, p.dx,
                                   p.dy, prof);
        RT_STAMP(prof, 13);
        if (act) {
            if (r == rt2d::SHADE_DONE) {
                if (value != 0.0f) {
                    // per-sample term of `spectrum_spp / lambda_samples * 95` (librt2d_reverse.rs:104-105)
                    const float v = p.thr * value;
                    const float inv = Ak->inv_lambda_pow2; // wave-uniform: the division is ten instructions, every step
                    const float c = (inv != 0.0f ? v * inv : v / (float)n_lambda) * (float)NBINS;
                    unsafeAtomicAdd(Ak->out + (int64_t)orow * NBINS + p.li, c);
                }
                act = false;
            } else if (p.depth_left == 0) {
                act = false; // the recursive call returns 0 at depth 0 (:120-122)
            }
        }
        if (intake) {
            // the rays that go on wait in the pool; the lanes take their own paths back
            const double nox = p.ox, noy = p.oy, ndx = p.dx, ndy = p.dy;
            const float nthr = p.thr;
            const uint32_t nmeta = (uint32_t)p.li | (p.c2 << 8), nrow = orow, nc0 = p.c0, nc1 = p.c1;
            asm volatile("" ::: "memory");
            p.ox = pool.ox[lane]; p.oy = pool.oy[lane]; p.dx = pool.dx[lane]; p.dy = pool.dy[lane];
            p.thr = pool.thr[lane]; p.li = (int)pool.meta[lane]; orow = pool.row[lane];
            p.c0 = pool.c0[lane]; p.c1 = pool.c1[lane];
            if (SMALL) {
                p.depth_left = (int)pool.park[lane]; p.c2 = pool.park_c2[lane];
            } else {
                const uint32_t pk = pool.park[lane];
                p.depth_left = (int)(pk & 0xFFFu);
                p.c2 = pk >> 12;
            }
            const uint64_t cont = __ballot(act);
            __builtin_amdgcn_wave_barrier();
            if (act) {
                const int j = rank_in_mask(cont);
                pool.ox[j] = nox; pool.oy[j] = noy; pool.dx[j] = ndx; pool.dy[j] = ndy;
                pool.thr[j] = nthr; pool.meta[j] = nmeta; pool.row[j] = nrow;
                pool.c0[j] = nc0; pool.c1[j] = nc1;
            }
            __builtin_amdgcn_wave_barrier();
            pool_pos = 0;
            pool_items = __popcll(cont);
            set_lambda += 1;
        } else {
            alive = act;
        }
        RT_STAMP(prof, 3);
    }
#ifdef RT_PROF
    if (lane == 0 && Ak->prof_out)
        for (int i = 0; i < 20; ++i) atomicAdd(Ak->prof_out + i, prof.t[i]);
#endif

    // rays traced by this wave, and how many of them skipped the search -> one global atomic each
    unsigned long long nr = nrays, ns = nsearch;
    for (int off = 32; off > 0; off >>= 1) {
        nr += __shfl_down(nr, off);
        ns += __shfl_down(ns, off);
    }
    if (lane == 0 && nr != 0) {
        atomicAdd(Ak->ray_counter, nr);
        if (nr != ns) atomicAdd(Ak->shared_counter, nr - ns);
    }
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
    const rt2d::SceneK Sk = (rt2d::SceneK)__builtin_amdgcn_kernarg_segment_ptr(); // S is the first argument
    (void)S;
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
        const rt2d::Hit h = rt2d::nearest_hit(Sk, p.ox, p.oy, p.dx, p.dy, run, coop[threadIdx.x >> 6], prof);
        float value = 0.0f;
        double dnx = 0.0, dny = 0.0;
        rt2d::ShadeResult r = rt2d::SHADE_DONE;
        const uint32_t bounce_idx = (uint32_t)(depth - p.depth_left);
        if (run) {
            if (nb == 0 && h.slot >= 0) { fp = h.pid; ft = h.t; }
            ++nb;
            r = rt2d::shade(Sk, p, h, value, dnx, dny, prof);
        }
        rt2d::rand_hemisphere_wave(run && r == rt2d::SHADE_DIFFUSE, dnx, dny, p.c0, p.c1, p.c2, bounce_idx, k0, k1, p.dx, p.dy, prof);
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

// ---------------------------------------------------------------------------
// Forward light tracer (2d/src/librt2d_forward.rs).  An emitter is a circle (Circle::sample_out_ray,
// rt_common.rs:113-120; the reference has no sample_out_ray for segments and strips).
struct Emitter2d { double cx, cy, r; int32_t obj, pad_; };

// ForwardRenderer::trace_ray (:66-195) for rays_per_emitter rays of every emitter, one lane per ray: the same hit
// search and material steps as the gather tracer at wavelength index 0.  The reference's trace_ray returns what
// its innermost call returns, so a path yields its LAST ray when that ray leaves the scene, and nothing otherwise:
// line[i] = (escaped, origin, direction).
__global__ __launch_bounds__(BLOCK) void forward_trace_kernel(const rt2d::SceneDev S, const Emitter2d *__restrict__ emitters,
                                                             int64_t first, int64_t n, int32_t rays_per_emitter, int32_t depth,
                                                             int32_t sample_offset, uint32_t k0, uint32_t k1,
                                                             double *__restrict__ lines, unsigned long long *ray_counter)
{
    __shared__ rt2d::CoopLds coop[BLOCK / 64];
    const rt2d::SceneK Sk = (rt2d::SceneK)__builtin_amdgcn_kernarg_segment_ptr(); // S is the first argument
    (void)S;
    const int64_t local = (int64_t)blockIdx.x * BLOCK + threadIdx.x; // this launch traces rays [first, first + n)
    const int64_t i = first + local;
    const bool in_range = local < n;
    rt2d::Path p;
    p.ox = p.oy = 0.0; p.dx = 1.0; p.dy = 0.0;
    p.li = 0;          // sample_emission: lambda_idx 0, power 1 (:211-217)
    p.thr = 1.0f;
    p.depth_left = depth;
    p.c0 = p.c1 = 0u;
    p.c2 = 0u;
    if (in_range) {
        const int e = (int)(i / rays_per_emitter);
        const int k = (int)(i - (int64_t)e * rays_per_emitter);
        const Emitter2d em = emitters[e];
        p.c0 = 0x40000000u | (uint32_t)em.obj;
        p.c1 = (uint32_t)(sample_offset + k);
        // Circle::sample_out_ray: n = from_angle(random * 2 pi), origin = center + (r + 0.0001) n
        const U4 w = philox4x32_10(p.c0, p.c1, 0xFFFFFFFFu, 0u, k0, k1);
        double sn, cs;
        sincos2pi(u01(w.x, w.y), sn, cs);
        p.dx = cs; p.dy = sn;
        p.ox = em.cx + (em.r + 0.0001) * cs;
        p.oy = em.cy + (em.r + 0.0001) * sn;
    }
    bool alive = in_range;
    bool escaped = false;
    int nb = 0;
    while (__any(alive)) {
        const bool run = alive && p.depth_left > 0;
        Prof prof{};
        const rt2d::Hit h = rt2d::nearest_hit(Sk, p.ox, p.oy, p.dx, p.dy, run, coop[threadIdx.x >> 6], prof);
        float value = 0.0f;
        double dnx = 0.0, dny = 0.0;
        rt2d::ShadeResult r = rt2d::SHADE_DONE;
        const uint32_t bounce_idx = (uint32_t)(depth - p.depth_left);
        if (run) {
            ++nb;
            r = rt2d::shade<true>(Sk, p, h, value, dnx, dny, prof);
            if (r == rt2d::SHADE_DONE && h.slot < 0) escaped = true;            // no hit: this ray is the path's entry (:85-87)
            if (r == rt2d::SHADE_DIFFUSE && !(p.thr > 0.0f)) r = rt2d::SHADE_DONE; // `if new_power > 0.` (:136)
        }
        rt2d::rand_hemisphere_wave(run && r == rt2d::SHADE_DIFFUSE, dnx, dny, p.c0, p.c1, p.c2, bounce_idx, k0, k1, p.dx, p.dy, prof);
        if (alive && (!run || r == rt2d::SHADE_DONE)) alive = false;
    }
    if (in_range) {
        double *q = lines + 5 * local;
        q[0] = escaped ? 1.0 : 0.0;
        q[1] = escaped ? p.ox : 0.0; q[2] = escaped ? p.oy : 0.0; q[3] = escaped ? p.dx : 0.0; q[4] = escaped ? p.dy : 0.0;
    }
    unsigned long long nr = (unsigned long long)nb;
    for (int off = 32; off > 0; off >>= 1) nr += __shfl_down(nr, off);
    if ((threadIdx.x & 63) == 0 && nr != 0) atomicAdd(ray_counter, nr);
}

// The drawing half of ForwardRenderer::global_render (:219-236): every escaped ray becomes the line
// origin .. origin + 1000 dir through line_drawing::XiaolinWu<f64, i32> (crate line_drawing 1.0.1, restated from its
// published source) and each item is draw_pixel((x, y), {value: ONE * v as f32, weight: 1}, Blending::Add).
// One lane per line, in the iterator's own sequential order (its y advances by repeated addition); pixels of
// different lines meet in f32 atomics on a scratch plane cov[W*H][2] = (coverage sum, item count) - the splat is
// bound by the atomic rate, and the three colour channels receive the same value, so they are filled from the
// scratch plane afterwards (forward_merge_kernel) instead of costing an atomic each.  n_lines counts the escaped rays.
__global__ __launch_bounds__(BLOCK) void forward_splat_kernel(const double *__restrict__ lines, int64_t n, int W, int H,
                                                             float *__restrict__ cov, unsigned long long *n_lines)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool on = i < n && lines[5 * i] != 0.0;
    const uint64_t m = __ballot(on);
    if (m != 0 && rtd::lane_id() == (int)__builtin_ctzll(m)) atomicAdd(n_lines, (unsigned long long)__popcll(m));
    if (!on) return;
    const double ox = lines[5 * i + 1], oy = lines[5 * i + 2], dx = lines[5 * i + 3], dy = lines[5 * i + 4];
    double sx = ox, sy = oy, ex = ox + 1000.0 * dx, ey = oy + 1000.0 * dy;
    const bool steep = __builtin_fabs(ey - sy) > __builtin_fabs(ex - sx);
    if (steep) { double t = sx; sx = sy; sy = t; t = ex; ex = ey; ey = t; }
    if (sx > ex) { double t = sx; sx = ex; ex = t; t = sy; sy = ey; ey = t; }
    const double ddx = ex - sx;
    const double gradient = (ddx == 0.0) ? 1.0 : (ey - sy) / ddx;
    // round(): half away from zero; the casts saturate where the reference's NumCast would panic (|v| >= 2^31)
    int32_t x = (int32_t)__builtin_round(sx);
    const int32_t end_x = (int32_t)__builtin_round(ex);
    // only the columns that can touch the image matter; stepping there keeps y's repeated additions in order
    double y = sy;
    bool lower = false;
    const int32_t limit = steep ? H : W; // x runs over rows of a steep line
    while (x <= end_x) {
        if (x >= limit) break; // x only grows: nothing further can land inside
        const double fpart = y - __builtin_floor(y);
        int32_t yi = (int32_t)y;
        if (lower) yi += 1;
        const int px = steep ? yi : x, py = steep ? x : yi;
        double v;
        if (lower) {
            lower = false;
            x += 1;
            y += gradient;
            v = fpart;
        } else {
            if (fpart > 0.0) lower = true;
            else { x += 1; y += gradient; }
            v = 1.0 - fpart;
        }
        if (px >= 0 && px < W && py >= 0 && py < H) {
            const int64_t idx = (int64_t)W * py + px;
            unsafeAtomicAdd(cov + 2 * idx, (float)v);
            unsafeAtomicAdd(cov + 2 * idx + 1, 1.0f);
        }
    }
}

// Blending::Add of the scratch plane onto the RawImage planes: value += ONE * coverage, weight += items.
__global__ __launch_bounds__(BLOCK) void forward_merge_kernel(const float *__restrict__ cov, int64_t n_pixels, float *__restrict__ rgb,
                                                             float *__restrict__ weight)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pixels) return;
    const float c = cov[2 * i], k = cov[2 * i + 1];
    if (k == 0.0f) return;
    rgb[3 * i] = rgb[3 * i] + c;
    rgb[3 * i + 1] = rgb[3 * i + 1] + c;
    rgb[3 * i + 2] = rgb[3 * i + 2] + c;
    weight[i] = weight[i] + k;
}

// CIE 1931 2-degree colour matching functions, 360..830 nm step 5 nm
// (data table of 2d/src/spectrum.rs:8-104), [95][3].
__constant__ float CMF[NBINS * 3] = {
#include "cie1931_table.inc"
};

// One channel of palette 0.7.7's `Xyz<D65, f32>::into_color() -> Srgb<f32>` after the matrix (2d/src/spectrum.rs:258-259):
// the sRGB transfer function, then the clamp to [0, 1] that `into_color` - unlike `into_color_unclamped` - applies.
// (Every image in the reference's gallery has 127 = (1 / (1 + 1) * 255) as u8 as its largest byte.)
RT_DEV float srgb_channel(float c)
{
    const float e = (c <= 0.0031308f) ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    return (e < 0.0f) ? 0.0f : (e > 1.0f) ? 1.0f : e;
}

// The per-pixel kernels below read whole spectra (95 floats, 380 B per pixel) and are bound by HBM: a wave takes
// STAGE_PIXELS = 32 consecutive pixels - 12 160 contiguous bytes, 16-byte aligned when the plane is - through LDS with
// 16-byte loads, several in flight per lane, and one lane per pixel then sums its 95 bins in index order (the
// reference's order).  12 KB of LDS per one-wave workgroup leaves 13 of them resident per CU; the first version
// (64 pixels and 24 KB per wave, 4-byte loads) had 6 and reached 1.4 TB/s.
constexpr int STAGE_PIXELS = 32;

// Bins of `cnt` <= STAGE_PIXELS consecutive pixels from `src` into `sp`, every value passed through `f`.
template <class F>
RT_DEV void stage_spectra(const float *__restrict__ src, int cnt, float *sp, F f)
{
    const int nf = cnt * NBINS;
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int nv = nf / 4;
#pragma unroll 4
        for (int i = threadIdx.x; i < nv; i += 64) {
            float4 a = reinterpret_cast<const float4 *>(src)[i];
            a.x = f(a.x); a.y = f(a.y); a.z = f(a.z); a.w = f(a.w);
            reinterpret_cast<float4 *>(sp)[i] = a;
        }
        done = nv * 4;
    }
    for (int i = done + threadIdx.x; i < nf; i += 64) sp[i] = f(src[i]);
}

// Spectrum::to_xyz (2d/src/spectrum.rs:242-254) of one staged pixel: sums in index order, unfused.
RT_DEV void staged_to_xyz(const float *sp, float &x, float &y, float &z)
{
    x = 0.0f; y = 0.0f; z = 0.0f;
    for (int b = 0; b < NBINS; ++b) {
        const float v = sp[b];
        x = x + v * CMF[3 * b + 0];
        y = y + v * CMF[3 * b + 1];
        z = z + v * CMF[3 * b + 2];
    }
}

// to_dvec3 after to_xyz (2d/src/spectrum.rs:256-278): palette's XYZ(D65) -> sRGB matrix, transfer function, clamp.
RT_DEV void xyz_to_srgb(float x, float y, float z, float out[3])
{
    float lin[3];
    lin[0] = 3.2404542f * x + -1.5371385f * y + -0.4985314f * z;
    lin[1] = -0.9692660f * x + 1.8760108f * y + 0.0415560f * z;
    lin[2] = 0.0556434f * x + -0.2040259f * y + 1.0572252f * z;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = srgb_channel(lin[k]);
}

// Spectrum::to_xyz + to_dvec3 (2d/src/spectrum.rs:242-278) of accumulated / total_spp.
__global__ __launch_bounds__(64) void resolve2d_kernel(const float *__restrict__ spectrum, int64_t n_pixels,
                                                      float total_spp, float *__restrict__ rgb,
                                                      float *__restrict__ weight)
{
    __shared__ __align__(16) float sp[STAGE_PIXELS * NBINS];
    const int64_t first = (int64_t)blockIdx.x * STAGE_PIXELS;
    const int cnt = (int)((n_pixels - first < STAGE_PIXELS) ? (n_pixels - first) : STAGE_PIXELS);
    stage_spectra(spectrum + first * NBINS, cnt, sp, [=](float v) { return v / total_spp; });
    __syncthreads();
    const int lane = threadIdx.x;
    if (lane >= cnt) return;
    float x, y, z, out[3];
    staged_to_xyz(sp + lane * NBINS, x, y, z);
    xyz_to_srgb(x, y, z, out);
    const int64_t pix = first + lane;
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[pix * 3 + k] = out[k];
    if (weight) weight[pix] = total_spp;
}

// The two halves of resolve2d_kernel for a reduction between them: Spectrum::to_xyz (2d/src/spectrum.rs:242-262) is
// linear, so GPUs that hold partial sums of the same image exchange 12 bytes per pixel (their XYZ sums) instead of
// 380 (their spectra).  project: xyz[pix] = sum over bins, in index order, of spectrum[pix][b] * CMF[b] (no division).
__global__ __launch_bounds__(64) void project_xyz_kernel(const float *__restrict__ spectrum, int64_t n_pixels, float *__restrict__ xyz)
{
    __shared__ __align__(16) float sp[STAGE_PIXELS * NBINS];
    const int64_t first = (int64_t)blockIdx.x * STAGE_PIXELS;
    const int cnt = (int)((n_pixels - first < STAGE_PIXELS) ? (n_pixels - first) : STAGE_PIXELS);
    stage_spectra(spectrum + first * NBINS, cnt, sp, [](float v) { return v; });
    __syncthreads();
    const int lane = threadIdx.x;
    if (lane >= cnt) return;
    float x, y, z;
    staged_to_xyz(sp + lane * NBINS, x, y, z);
    const int64_t pix = first + lane;
    xyz[pix * 3 + 0] = x; xyz[pix * 3 + 1] = y; xyz[pix * 3 + 2] = z;
}

// encode: (xyz / total_spp) -> linear sRGB -> transfer function (to_dvec3, spectrum.rs:264-278), as resolve2d_kernel.
__global__ __launch_bounds__(BLOCK) void encode_xyz_kernel(const float *__restrict__ xyz, int64_t n_pixels, float total_spp,
                                                          float *__restrict__ rgb, float *__restrict__ weight)
{
    const int64_t pix = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (pix >= n_pixels) return;
    const float x = xyz[pix * 3 + 0] / total_spp, y = xyz[pix * 3 + 1] / total_spp, z = xyz[pix * 3 + 2] / total_spp;
    float out[3];
    xyz_to_srgb(x, y, z, out);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[pix * 3 + k] = out[k];
    if (weight) weight[pix] = total_spp;
}

// ---------------------------------------------------------------------------
// ReverseRenderer::denoise (2d/src/librt2d_reverse.rs:337-422) on RawImage planes rgb[W*H][3], weight[W*H].

// Sort key of every pixel: |merged - loop| (img.rs:144-155, :279-300), RawImage::convolution with the 3x3
// Laplace kernel (img.rs:167-185: output (x, y) = sum over i, j in 0..3, i outer, of in(x + i, y + j) k[i][j],
// only for x < W - 3 and y < H - 3; elsewhere 0), value = splat(length) and key = value.length()
// (librt2d_reverse.rs:353-358, :368), all in f32 in the reference's order.  Written as key bits << 32 | idx:
// sorting these descending IS the reference's order (stable ascending sort by key, then reversed: among equal
// keys the higher index comes first); keys are >= 0, so their bit patterns order like the floats.
__global__ __launch_bounds__(BLOCK) void denoise_keys_kernel(const float *__restrict__ merged, const float *__restrict__ loop_img,
                                                            int W, int H, uint64_t *__restrict__ keys)
{
    const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= (int64_t)W * H) return;
    const int x = (int)(idx % W), y = (int)(idx / W);
    float key = 0.0f;
    if (x < W - 3 && y < H - 3) {
        float v[3] = { 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float k = (i == 1 && j == 1) ? 8.0f : -1.0f;
                const int64_t at = 3 * ((int64_t)W * (y + j) + (x + i));
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = v[c] + __builtin_fabsf(merged[at + c] - loop_img[at + c]) * k;
            }
        const float len = __builtin_sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        key = __builtin_sqrtf((len * len + len * len) + len * len);
    }
    keys[idx] = ((uint64_t)__float_as_uint(key) << 32) | (uint64_t)(uint32_t)idx;
}

// The `top` pixels (librt2d_reverse.rs:368-378 sorts all keys and takes the first n_top) are the pixels whose key is
// >= the n_top-th largest key; keys are distinct (they end in the pixel index), so that key alone names the set and
// no order is needed.  It is found one byte at a time from the top: a pass counts, per value of the current byte,
// the keys that agree with the bytes fixed so far (select_count_kernel: a histogram per workgroup in LDS, added to
// the global one), then one workgroup fixes the byte whose bucket holds the wanted rank and lowers the rank by the
// buckets above it (select_pick_kernel).  Eight passes over the 8 n bytes of keys; nothing returns to the host.
struct SelectState {
    uint64_t prefix; // the bytes fixed so far (the rest zero); after the last pass the n_top-th largest key itself
    uint64_t want;   // rank, from the top and 1-based, of that key among the keys that agree with `prefix`
};

__global__ void select_start_kernel(SelectState *st, SelectState start) { *st = start; }

__global__ __launch_bounds__(BLOCK) void select_count_kernel(const uint64_t *__restrict__ keys, int64_t n, int shift,
                                                            const SelectState *__restrict__ st, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t prefix = st->prefix;
    const uint64_t fixed = shift == 56 ? 0ull : ~0ull << (shift + 8);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t k = keys[i];
        if ((k & fixed) == prefix) atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// One workgroup of 256: thread t owns bucket t; clears the histogram for the next pass.
__global__ __launch_bounds__(BLOCK) void select_pick_kernel(uint32_t *__restrict__ hist, SelectState *__restrict__ st, int shift)
{
    __shared__ uint32_t h[256];
    const int t = threadIdx.x;
    h[t] = hist[t];
    hist[t] = 0;
    const uint64_t want = st->want, prefix = st->prefix; // read by every thread before the one write below
    __syncthreads();
    uint64_t above = 0;
    for (int u = t + 1; u < 256; ++u) above += h[u];
    if (above < want && want <= above + h[t]) {
        st->prefix = prefix | ((uint64_t)t << shift);
        st->want = want - above;
    }
}

// The set of pixels to recompute (librt2d_reverse.rs:380-392): every top pixel marks its (2m+1)^2 neighbourhood
// in a byte map over the image grown by m on every side (pixels outside the image are recomputed too, as in the
// reference, and dropped when added).
__global__ __launch_bounds__(BLOCK) void denoise_mark_kernel(const uint64_t *__restrict__ keys, int64_t n, const SelectState *__restrict__ st,
                                                            int W, int m, uint8_t *__restrict__ mask)
{
    const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n || keys[idx] < st->prefix) return;
    const int x = (int)((uint32_t)idx % (uint32_t)W), y = (int)((uint32_t)idx / (uint32_t)W);
    const int EW = W + 2 * m;
    for (int j = -m; j <= m; ++j)
        for (int i = -m; i <= m; ++i) mask[(int64_t)EW * (y + j + m) + (x + i + m)] = 1;
}

// Mask -> pixel list.  A wave scans LIST_CELLS_PER_WAVE cells, four mask bytes per lane and load, counts its marked
// cells, reserves that many list slots with ONE atomic (an atomic on a single address costs about 10 ns wherever it
// is issued from, so a wave per 64 cells would spend 0.35 ms on a 1080p mask) and writes them in cell order.
// `mask` is readable up to the next multiple of 4 bytes past n_cells, the excess being zero.
constexpr int LIST_CELLS_PER_WAVE = 2048;
__global__ __launch_bounds__(64) void denoise_list_kernel(const uint8_t *__restrict__ mask, int EW, int64_t n_cells, int m,
                                                         int32_t *__restrict__ px, int32_t *__restrict__ py,
                                                         unsigned long long *__restrict__ counter)
{
    constexpr int ROUNDS = LIST_CELLS_PER_WAVE / 256;
    const int lane = threadIdx.x;
    const int64_t first = (int64_t)blockIdx.x * LIST_CELLS_PER_WAVE;
    uint32_t word[ROUNDS];
    int mine = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t c = first + r * 256 + lane * 4;
        word[r] = c < n_cells ? *reinterpret_cast<const uint32_t *>(mask + c) : 0u;
        mine += (word[r] & 0xffu ? 1 : 0) + (word[r] & 0xff00u ? 1 : 0) + (word[r] & 0xff0000u ? 1 : 0) + (word[r] >> 24 ? 1 : 0);
    }
    // inclusive prefix of the lanes' counts
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    const int total = __shfl(incl, 63);
    if (total == 0) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned long long)total);
    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    // a lane's cells of round r come after every lane's cells of earlier rounds: order inside the wave is
    // (lane, round, byte), any fixed order will do
    unsigned long long at = base + (unsigned)(incl - mine);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t c0 = first + r * 256 + lane * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((word[r] >> (8 * k)) & 0xffu) {
                const int64_t c = c0 + k;
                px[at] = (int)(c % EW) - m;
                py[at] = (int)(c / EW) - m;
                ++at;
            }
    }
}

// merged.draw_pixel(pixel, {value: spectrum.to_dvec3(), weight: spp * oversampling_factor}, Blending::Add)
// (librt2d_reverse.rs:410-418, img.rs:124-128) for a list of recomputed pixels; to_dvec3 as in resolve2d_kernel.
__global__ __launch_bounds__(64) void denoise_add_kernel(const float *__restrict__ spectrum, const int32_t *__restrict__ px,
                                                        const int32_t *__restrict__ py, int64_t n, int W, int H, float spp,
                                                        float wadd, float *__restrict__ rgb, float *__restrict__ weight)
{
    __shared__ __align__(16) float sp[STAGE_PIXELS * NBINS];
    const int64_t first = (int64_t)blockIdx.x * STAGE_PIXELS;
    const int cnt = (int)((n - first < STAGE_PIXELS) ? (n - first) : STAGE_PIXELS);
    stage_spectra(spectrum + first * NBINS, cnt, sp, [=](float v) { return v / spp; }); // `spectrum / spp`, librt2d_reverse.rs:109
    __syncthreads();
    const int lane = threadIdx.x;
    if (lane >= cnt) return;
    const int x = px[first + lane], y = py[first + lane];
    if (x < 0 || y < 0 || x >= W || y >= H) return; // pixel_to_idx -> Err, ignored by the reference
    float X, Y, Z, out[3];
    staged_to_xyz(sp + lane * NBINS, X, Y, Z);
    xyz_to_srgb(X, Y, Z, out);
    const int64_t pix = (int64_t)W * y + x; // every pixel appears once in the list: no atomics needed
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[pix * 3 + k] = rgb[pix * 3 + k] + out[k];
    weight[pix] = weight[pix] + wadd;
}

// ---------------------------------------------------------------------------
// Progressive accumulation in linear light (the loop of ReverseRenderer::endless_render,
// 2d/src/librt2d_reverse.rs:424-510, with the pass merge done on spectral sums and per-pixel sample counts
// instead of `merged_image + loop_image` on encoded colours, img.rs:257-274).

// sRGB of (spectrum / div): Spectrum::to_xyz + to_dvec3 exactly as resolve2d_kernel does them.
RT_DEV void spectrum_to_srgb(const float *sp, float div, float out[3])
{
    float x = 0.0f, y = 0.0f, z = 0.0f;
    for (int b = 0; b < NBINS; ++b) {
        const float v = sp[b] / div;
        x = x + v * CMF[3 * b + 0];
        y = y + v * CMF[3 * b + 1];
        z = z + v * CMF[3 * b + 2];
    }
    xyz_to_srgb(x, y, z, out);
}

// One whole-image pass of `spp` samples per pixel joins the running sums: sum += pass, count += spp, and both
// images the noise estimate compares are refreshed: loop_rgb = sRGB(pass / spp) (what global_render returned for
// this pass), merged_rgb = sRGB(sum / count).  One wave per MERGE_PIXELS pixels, both spectra staged through LDS.
// (32 pixels: every lane encodes one image of one pixel, and the 2 x 95 divisions per pixel are what the kernel's time
// goes to beside its three HBM streams; with 16 pixels per wave - 13 resident workgroups per CU instead of 6, half the
// lanes encoding - it ran at 0.92 ms per 1080p pass against 0.79.)
constexpr int MERGE_PIXELS = 32;
__global__ __launch_bounds__(64) void progressive_merge_kernel(const float *__restrict__ pass, int64_t n_pixels, float spp,
                                                              float *__restrict__ sum, float *__restrict__ count,
                                                              float *__restrict__ merged_rgb, float *__restrict__ loop_rgb)
{
    __shared__ __align__(16) float sp[MERGE_PIXELS * NBINS], sm[MERGE_PIXELS * NBINS];
    const int64_t first = (int64_t)blockIdx.x * MERGE_PIXELS;
    const int cnt = (int)((n_pixels - first < MERGE_PIXELS) ? (n_pixels - first) : MERGE_PIXELS);
    // a block's spectra are MERGE_PIXELS x 380 contiguous bytes, 16-byte aligned when the planes are: 16-byte loads and stores
    const float *src = pass + first * NBINS;
    float *dst = sum + first * NBINS;
    const int nf = cnt * NBINS, nv = nf / 4;
    for (int i = threadIdx.x; i < nv; i += 64) {
        const float4 a = reinterpret_cast<const float4 *>(src)[i];
        float4 t = reinterpret_cast<const float4 *>(dst)[i];
        t.x = t.x + a.x; t.y = t.y + a.y; t.z = t.z + a.z; t.w = t.w + a.w;
        reinterpret_cast<float4 *>(dst)[i] = t;
        reinterpret_cast<float4 *>(sp)[i] = a;
        reinterpret_cast<float4 *>(sm)[i] = t;
    }
    for (int i = nv * 4 + threadIdx.x; i < nf; i += 64) {
        const float a = src[i];
        const float t = dst[i] + a;
        dst[i] = t;
        sp[i] = a;
        sm[i] = t;
    }
    __syncthreads();
    // lanes 0 .. MERGE_PIXELS-1 encode the pass image, the next MERGE_PIXELS lanes the merged image of the same pixels
    const int lane = threadIdx.x & (MERGE_PIXELS - 1);
    if (lane >= cnt || threadIdx.x >= 2 * MERGE_PIXELS) return;
    const int64_t pix = first + lane;
    float out[3];
    if (threadIdx.x < MERGE_PIXELS) {
        spectrum_to_srgb(sp + lane * NBINS, spp, out);
        for (int k = 0; k < 3; ++k) loop_rgb[pix * 3 + k] = out[k];
    } else {
        const float c = count[pix] + spp;
        count[pix] = c;
        spectrum_to_srgb(sm + lane * NBINS, c, out);
        for (int k = 0; k < 3; ++k) merged_rgb[pix * 3 + k] = out[k];
    }
}

// The re-sampled pixels of an adaptive pass join the running sums: for every listed pixel inside the image
// sum += its `spp` new samples, count += spp, merged_rgb = sRGB(sum / count).  A pixel appears once in the list.
__global__ __launch_bounds__(64) void progressive_add_kernel(const float *__restrict__ spectrum, const int32_t *__restrict__ px,
                                                            const int32_t *__restrict__ py, int64_t n, int W, int H, float spp,
                                                            float *__restrict__ sum, float *__restrict__ count,
                                                            float *__restrict__ merged_rgb)
{
    __shared__ float sp[STAGE_PIXELS * NBINS];
    __shared__ int64_t pix_of[STAGE_PIXELS];
    const int64_t first = (int64_t)blockIdx.x * STAGE_PIXELS;
    const int cnt = (int)((n - first < STAGE_PIXELS) ? (n - first) : STAGE_PIXELS);
    const int lane = threadIdx.x;
    if (lane < cnt) {
        const int x = px[first + lane], y = py[first + lane];
        pix_of[lane] = (x < 0 || y < 0 || x >= W || y >= H) ? -1 : (int64_t)W * y + x; // outside: traced, then dropped
    }
    __syncthreads();
    const float *src = spectrum + first * NBINS;
    for (int i = lane; i < cnt * NBINS; i += 64) {
        const int e = i / NBINS, b = i - e * NBINS;
        const int64_t pix = pix_of[e];
        if (pix < 0) continue;
        const float t = sum[pix * NBINS + b] + src[i];
        sum[pix * NBINS + b] = t;
        sp[i] = t;
    }
    __syncthreads();
    if (lane >= cnt || pix_of[lane] < 0) return;
    const int64_t pix = pix_of[lane];
    const float c = count[pix] + spp;
    count[pix] = c;
    float out[3];
    spectrum_to_srgb(sp + lane * NBINS, c, out);
    for (int k = 0; k < 3; ++k) merged_rgb[pix * 3 + k] = out[k];
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
// Self-check of rtd::sqrt_exact / rtd::rsqrt_exact against the IEEE operations they stand for (sqrt(x), 1.0 / sqrt(x) as
// the compiler expands them), bit for bit, over n arguments of one class:
//   0  random significand, exponent uniform in 2^-60 .. 2^60 (the tracer's range)
//   1  random bits over the whole double range: zero, denormals, infinities, NaN and negatives among them
//   2  significands of mostly ones or mostly zeros around every exponent
//   3  exact squares and their neighbours
//   4  arguments whose square root has a significand of mostly ones (what the fast path hands to the generic form)
// and, class 5, of rtd::div_by_100 against x / 100.0f over ALL floats with an exponent field of 3 .. 254, and +0
// (n is ignored: the class is exhaustive).
__device__ inline double math_check_arg(int cls, uint64_t i, uint32_t seed)
{
    const rtd::U4 w = rtd::philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)cls, 0u, seed, 0x5eedu);
    const uint64_t r = ((uint64_t)w.x << 32) | w.y, r2 = ((uint64_t)w.z << 32) | w.w;
    const uint64_t frac = r & 0x000FFFFFFFFFFFFFull;
    if (cls == 0) return __longlong_as_double((long long)(((uint64_t)(1023 - 60 + (r2 % 121)) << 52) | frac));
    if (cls == 1) return __longlong_as_double((long long)r);
    if (cls == 2) {
        const int k = (int)(r2 % 53); // k low bits random, the rest all ones or all zeros
        const uint64_t low = k ? (frac & ((1ull << k) - 1ull)) : 0ull;
        const uint64_t high = ((r2 >> 8) & 1ull) ? (0x000FFFFFFFFFFFFFull & ~((1ull << k) - 1ull)) : 0ull;
        return __longlong_as_double((long long)(((uint64_t)(1 + ((r2 >> 16) % 2046)) << 52) | high | low));
    }
    if (cls == 3) {
        const double a = __longlong_as_double((long long)(((uint64_t)(1023 - 30 + (r2 % 61)) << 52) | (frac & 0x000FFFFFFC000000ull))); // 26 bits
        return __longlong_as_double(__double_as_longlong(a * a) + ((int)((r2 >> 8) % 5) - 2)); // a * a is exact
    }
    const int k = (int)(r2 % 33);
    const uint64_t sfrac = 0x000FFFFFFFFFFFFFull & ~(frac & ((1ull << k) - 1ull));
    const double s = __longlong_as_double((long long)(((uint64_t)(1023 - 40 + ((r2 >> 8) % 81)) << 52) | sfrac));
    return __longlong_as_double(__double_as_longlong(s * s) + ((int)((r2 >> 20) % 9) - 4));
}

__global__ __launch_bounds__(BLOCK) void math_check_kernel(int cls, uint64_t n, uint32_t seed, unsigned long long *mismatches)
{
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    unsigned long long mine = 0;
    if (cls == 5) {
        for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += stride) {
            const uint32_t ex = ((uint32_t)i >> 23) & 255u;
            if (!((ex >= 3u && ex <= 254u) || i == 0)) continue;
            const float x = __uint_as_float((uint32_t)i);
            mine += (__float_as_uint(x / 100.0f) == __float_as_uint(rtd::div_by_100(x))) ? 0u : 1u;
        }
        if (mine) atomicAdd(mismatches + cls, mine);
        return;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const double x = math_check_arg(cls, i, seed);
        const double want = 1.0 / __builtin_sqrt(x), want_s = __builtin_sqrt(x);
        const double got = rtd::rsqrt_exact(x), got_s = rtd::sqrt_exact(x);
        const bool same = (__double_as_longlong(want) == __double_as_longlong(got) || (want != want && got != got)) &&
                          (__double_as_longlong(want_s) == __double_as_longlong(got_s) || (want_s != want_s && got_s != got_s));
        mine += same ? 0u : 1u;
    }
    if (mine) atomicAdd(mismatches + cls, mine);
}

// ---------------------------------------------------------------------------
struct Render3dArgs {
    rt3d::SceneDev S;
    int32_t spp, bounce_limit, sample_offset, tiles_x;
    int32_t n_tiles;
    // a tile's (pixel, sample) items - 64 x spp of them, pixel fastest - are handed out in runs of `chunk` items
    // (a multiple of 64: whole samples of the tile); unit u = run u / n_tiles of tile u % n_tiles
    int32_t chunk, n_units;
    uint32_t k0, k1;
    float *out; // [W*H][3], accumulated
    unsigned long long *ray_counter;
    int32_t *tile_counter;            // next unit to hand out, zeroed before the launch
    unsigned long long *work_counter; // [0] inner-node visits, [1] exact sphere tests
    int32_t refill_min;
    unsigned long long *prof_out; // diagnostic builds only
};

// Generated primary rays of one wave (the origin is the camera position for all of them).
struct Ray3Pool {
    double dx[64], dy[64], dz[64];
    uint32_t c0[64], c1[64], lpix[64];
};

// World::render + compute_pixel (3d/src/rt.rs:238-273).
//
// Same shape as render2d_kernel: persistent, independent waves; each pulls work units (runs of about 1 k
// (pixel, sample) items of one 8 x 8 pixel tile) from a global counter, a lane owns one path in registers, dead
// lanes pop primary rays from the wave's pool in LDS, and lanes idle only at the very end of the launch (one tile per
// workgroup with a block-wide accumulator idled 8-10 % of the lane-slots in every tile's ramp-down and needed two
// block barriers per tile).  A finished path's colour goes to the wave's own 64-pixel accumulator in LDS while the
// wave is still on that path's unit; the accumulator is added to the image (hardware f32 atomics, 3 per pixel) when
// the wave moves on, and the few paths still in flight from the unit left behind add their colour to the image
// directly.  One atomic per finished path and channel would be 3.2 G atomics for a 1080p / 512 spp image: more than
// the memory-side atomic units retire in the kernel's whole run time.
template <bool BVH>
__global__ __launch_bounds__(BLOCK, 5) void render3d_kernel(const Render3dArgs A)
{
    __shared__ float accs[BLOCK / 64][TILE_PIXELS * 3];
    __shared__ Ray3Pool pools[BLOCK / 64];
    extern __shared__ int32_t stack[]; // [levels of this scene's tree][2][BLOCK], sized at launch
    typedef const Render3dArgs __attribute__((address_space(4))) *ArgsK;
    ArgsK Ak = (ArgsK)__builtin_amdgcn_kernarg_segment_ptr(); // arguments are read in place, see rt3d::SceneK
    (void)A;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    float *acc = accs[tid >> 6];
    Ray3Pool &pool = pools[tid >> 6];
    acc[lane] = 0.0f; acc[lane + 64] = 0.0f; acc[lane + 128] = 0.0f;

    rt3d::Path p;
    p.ox = p.oy = p.oz = 0.0; p.dx = 1.0; p.dy = p.dz = 0.0;
    p.thr = 0.0f; p.left = 0; p.c0 = p.c1 = 0;
    bool alive = false;
    bool exhausted = false;                     // wave-uniform: no unit left
    int cursor = 0, cursor_end = 0;             // wave-uniform: items of the current unit
    int unit_no = 0;                            // wave-uniform: how many units this wave has started
    int tx = 0, ty = 0, s0 = 0;                 // wave-uniform: the unit's tile origin and first sample
    int pool_pos = 0, pool_items = 0;           // wave-uniform
    int lpix = 0, my_unit = -1;                 // this lane's path: pixel within its tile, and which unit it came from
    uint32_t gpix = 0;                          // ... and its pixel in the image, (i, H-1-j) (rt.rs:263-268)
    unsigned long long nrays = 0;
    rt3d::Work work{ 0u, 0u };
    const int total = TILE_PIXELS * Ak->spp;
    rtd::Prof prof{};
#ifdef RT_PROF
    prof.last = __builtin_amdgcn_s_memtime();
    prof.octants = Ak->prof_out ? Ak->prof_out + 20 : nullptr;
#endif
    // adds the wave's accumulator to the image and clears it
    auto flush = [&](int ftx, int fty) {
        __builtin_amdgcn_wave_barrier();
        const int i = ftx + (lane & 7), j = fty + (lane >> 3);
        if (i < Ak->S.width && j < Ak->S.height) {
            float *dst = Ak->out + ((int64_t)Ak->S.width * (Ak->S.height - j - 1) + i) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float v = acc[lane * 3 + k];
                if (v != 0.0f) unsafeAtomicAdd(dst + k, v);
            }
        }
        __builtin_amdgcn_wave_barrier();
        acc[lane] = 0.0f; acc[lane + 64] = 0.0f; acc[lane + 128] = 0.0f;
        __builtin_amdgcn_wave_barrier();
    };
    // the colour a finished path returns goes to its pixel
    auto deposit = [&](const float rgb[3]) {
        if (rgb[0] != 0.0f || rgb[1] != 0.0f || rgb[2] != 0.0f) {
            if (my_unit == unit_no) {
                atomicAdd(&acc[lpix * 3 + 0], rgb[0]);
                atomicAdd(&acc[lpix * 3 + 1], rgb[1]);
                atomicAdd(&acc[lpix * 3 + 2], rgb[2]);
            } else {
                float *dst = Ak->out + (int64_t)gpix * 3;
                unsafeAtomicAdd(dst + 0, rgb[0]);
                unsafeAtomicAdd(dst + 1, rgb[1]);
                unsafeAtomicAdd(dst + 2, rgb[2]);
            }
        }
    };
    // the body of trace_ray after the search, for the lanes in `run`
    auto shade_step = [&](bool run, const rt3d::Hit &h) {
        if (run) {
            ++nrays;
            float rgb[3];
            if (rt3d::shade((rt3d::SceneK)&Ak->S, p, h, Ak->bounce_limit, Ak->k0, Ak->k1, rgb)) {
                deposit(rgb);
                alive = false;
            }
        }
    };
    while (true) {
        asm volatile("" : "+s"(Ak)); // arguments are re-read where a step needs them, not held in registers across it
        const uint64_t dead = __ballot(!alive);
        const int ndead = __popcll(dead);
        if ((!exhausted || pool_pos < pool_items) && (ndead >= Ak->refill_min || ndead == 64)) {
            const int r = rank_in_mask(dead);
            int given = 0;
            for (int pass = 0; pass < 2; ++pass) {
                const int avail = pool_items - pool_pos;
                const int take = (ndead - given < avail) ? ndead - given : avail;
                if (!alive && r >= given && r < given + take) {
                    const int j = pool_pos + (r - given);
                    p.ox = Ak->S.pos[0]; p.oy = Ak->S.pos[1]; p.oz = Ak->S.pos[2];
                    p.dx = pool.dx[j]; p.dy = pool.dy[j]; p.dz = pool.dz[j];
                    p.c0 = pool.c0[j]; p.c1 = pool.c1[j];
                    lpix = (int)pool.lpix[j];
                    const int pi = tx + (lpix & 7), pj = ty + (lpix >> 3);
                    gpix = (uint32_t)(Ak->S.width * (Ak->S.height - pj - 1) + pi);
                    my_unit = unit_no;
                    p.thr = 1.0f;
                    p.left = Ak->bounce_limit;
                }
                pool_pos += take;
                given += take;
                if (pass == 1 || given == ndead || exhausted) break;
                // the pool ran dry: the unit's next 64 items, one per lane
                if (cursor >= cursor_end) {
                    if (unit_no > 0) flush(tx, ty); // paths of the unit left behind (my_unit < unit_no) now add directly
                    int t = 0;
                    if (lane == 0) t = atomicAdd(Ak->tile_counter, 1);
                    const int unit = __builtin_amdgcn_readfirstlane(t);
                    if (unit >= Ak->n_units) {
                        exhausted = true;
                        unit_no += 1;
                        break;
                    }
                    const int run = unit / Ak->n_tiles;
                    const int tile = unit - run * Ak->n_tiles;
                    cursor = run * Ak->chunk;
                    cursor_end = (cursor + Ak->chunk < total) ? cursor + Ak->chunk : total;
                    tx = (tile % Ak->tiles_x) * 8;
                    ty = (tile / Ak->tiles_x) * 8;
                    s0 = Ak->sample_offset;
                    unit_no += 1;
                }
                const int item = cursor + lane;
                cursor += 64;
                const int lp = item & (TILE_PIXELS - 1);
                const int s = item >> 6;
                const int i = tx + (lp & 7), j = ty + (lp >> 3);
                // `if i < 10 && j < 10 { continue; }` (rt.rs:257-259)
                const bool valid = item < cursor_end && i < Ak->S.width && j < Ak->S.height && !(i < 10 && j < 10) && Ak->bounce_limit > 0;
                rt3d::Path g;
                rt3d::generate((rt3d::SceneK)&Ak->S, g, i, j, (uint32_t)(s0 + s), Ak->bounce_limit, Ak->k0, Ak->k1);
                const uint64_t vm = __ballot(valid);
                __builtin_amdgcn_wave_barrier();
                if (valid) {
                    const int k = rank_in_mask(vm);
                    pool.dx[k] = g.dx; pool.dy[k] = g.dy; pool.dz[k] = g.dz;
                    pool.c0[k] = g.c0; pool.c1[k] = g.c1;
                    pool.lpix[k] = (uint32_t)lp;
                }
                __builtin_amdgcn_wave_barrier();
                pool_pos = 0;
                pool_items = __popcll(vm);
            }
            alive = alive || (r < given && !alive);
        }
        if (!__any(alive)) {
            if (exhausted && pool_pos >= pool_items) break;
            continue;
        }
        RT_STAMP(prof, 0);
#ifdef RT_PROF
        prof.t[5] += 1;
        prof.t[6] += (unsigned long long)__popcll(__ballot(alive));
#endif
        const rt3d::Hit h = rt3d::nearest_hit<BVH>((rt3d::SceneK)&Ak->S, p.ox, p.oy, p.oz, p.dx, p.dy, p.dz, alive, stack + tid, BLOCK, work, prof);
        RT_STAMP(prof, 3);
        shade_step(alive, h);
        RT_STAMP(prof, 4);
    }
#ifdef RT_PROF
    if (lane == 0 && Ak->prof_out)
        for (int i = 0; i < 20; ++i) atomicAdd(Ak->prof_out + i, prof.t[i]);
#endif
    // the last unit's accumulator was flushed when the wave found no further unit
    for (int off = 32; off > 0; off >>= 1) {
        nrays += __shfl_down(nrays, off);
        work.node_visits += __shfl_down(work.node_visits, off);
        work.sphere_tests += __shfl_down(work.sphere_tests, off);
    }
    if (lane == 0 && nrays != 0) {
        atomicAdd(Ak->ray_counter, nrays);
        atomicAdd(Ak->work_counter, (unsigned long long)work.node_visits);
        atomicAdd(Ak->work_counter + 1, (unsigned long long)work.sphere_tests);
    }
}

// World::trace_ray for a fed batch (3d/src/rt.rs:216-236); o, d are [n][3].
template <bool BVH>
__global__ __launch_bounds__(BLOCK) void trace_batch3d_kernel(const rt3d::SceneDev S, int64_t n,
                                                             const double *__restrict__ o, const double *__restrict__ d,
                                                             int32_t bounce_limit, uint32_t k0, uint32_t k1,
                                                             float *__restrict__ rgb_out, int32_t *__restrict__ first_prim,
                                                             double *__restrict__ first_t, int32_t *__restrict__ bounces,
                                                             unsigned long long *ray_counter)
{
    extern __shared__ int32_t stack[]; // [levels of this scene's tree][2][BLOCK], sized at launch
    const rt3d::SceneK Sk = (rt3d::SceneK)__builtin_amdgcn_kernarg_segment_ptr(); // S is the first argument
    (void)S;
    rt3d::Work work{ 0u, 0u };
    rtd::Prof prof{};
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
        const rt3d::Hit h = rt3d::nearest_hit<BVH>(Sk, p.ox, p.oy, p.oz, p.dx, p.dy, p.dz, alive, stack + threadIdx.x, BLOCK, work, prof);
        if (alive) {
            if (nb == 0 && h.slot >= 0) { fp = h.sid; ft = h.t; }
            ++nb;
            float rgb[3];
            if (rt3d::shade(Sk, p, h, bounce_limit, k0, k1, rgb)) {
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
