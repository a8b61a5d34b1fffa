// 3d sphere path tracer: device-side nearest hit (linear scan or BVH), camera
// rays and the diffuse bounce.
//
// Reference semantics (paths relative to the reference root):
//   Sphere::hit            3d/src/rt.rs:71-118
//   World::trace_ray       3d/src/rt.rs:216-236   (nearest of a linear scan, x0.5 per bounce)
//   random_on_hemisphere   3d/src/rt.rs:40-45     (cube-normalised, not uniform)
//   Camera::get_ray        3d/src/rt.rs:180-191
//   get_background_color   3d/src/rt.rs:275-287
// The reference has no acceleration structure in 3d; the BVH here must return
// exactly the linear scan's nearest hit (same t bits, lowest index on ties).
// Its box test follows the slab method of 2d/src/aabb.rs:39-66.
#pragma once

#include "rt_device_math.hpp"

namespace rt3d {
using namespace rtd;

// Host-side build node (median split): box of the node, inner: count == 0 and children at `left`,
// `left + 1`; leaf: `count` spheres starting at slot `left`.
struct BvhNode {
    double lo[3], hi[3];
    int32_t left, count;
    int32_t axis;
    int32_t pad_;
};

// Traversal node, 64 B, one per inner build node: the boxes of BOTH children in f32, component c of every field
// belonging to child c (so one packed f32 instruction works on both boxes), and what each child is.
// child >= 0: index of an inner node; child < 0: a leaf, ~child = first slot << 3 | count (count <= LEAF_MAX).
// Slots are BVH leaf order; `sid` maps a slot back to the caller's sphere index.  Boxes are rounded
// outward to f32 from boxes already padded by 1e-7 x scene scale; the traversal adds a per-ray pad and a
// relative slack (see box_intervals), so a sphere hit at distance t always lies inside [tn, tf] of every
// ancestor box as computed here.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct BvhNode2 {
    f32x2 plane[6]; // lo x, hi x, lo y, hi y, lo z, hi z, each holding both children's value
    int32_t child[2];
    int32_t pad_[2];
};

// Most spheres a leaf holds.  A leaf is not a box test followed by exact tests: the walk runs the f32 sphere filter
// (leaf_filter) on all of a leaf's spheres at once, which costs less per sphere than a box test and passes far fewer
// misses on to the exact f64 test than the sphere's box would.
#ifndef RT_LEAF_MAX
#define RT_LEAF_MAX 4
#endif
constexpr int LEAF_MAX = RT_LEAF_MAX;
static_assert(LEAF_MAX >= 1 && LEAF_MAX <= 7, "the count shares three bits of a child reference");

// One sphere as a leaf's filter reads it (one 16-byte load per lane), in slot order; same values as SpherePair's.
struct SphereF32 { float vx, vy, vz, e; };

// Two spheres as the f32 filter reads them, each field holding the value of both (one scalar load and one address
// per pair: with four separate arrays the loop's scalar side weighs as much as its vector side).
struct SpherePair { f32x2 vx, vy, vz, e; };

struct SceneDev {
    int32_t n, n_always, n_nodes, width, height, use_bvh;
    // slot order: the `n_always` spheres tested on every ray first (those too
    // large for the tree), then BVH leaf order
    const double *cx, *cy, *cz, *r2;
    const int32_t *sid;
    const BvhNode2 *nodes;
    // spheres as the f32 filter sees them (see filter_word32), one record per pair of slots: v = centre - frame centre
    // and e = |v|^2 - r^2 (1 + 2^-22) rounded down (-inf for a sphere the frame does not hold: always a candidate;
    // +inf in an odd count's spare half: never one)
    const SpherePair *f32p;
    // BVH mode: the same per slot, followed by LEAF_MAX spare records (a leaf always reads LEAF_MAX of them)
    const SphereF32 *f32s;
    double filt_m[3], filt_B;
    double box_scale; // largest |coordinate| of a tree box
    double pos[3], p00[3], du[3], dv[3];
};

// The kernels read the scene where it arrives - in the kernel-argument segment, constant address space - field by
// field at the point of use: copied into a by-value struct every field is loaded into a scalar register at kernel
// entry and stays live over the whole trace loop, and there are more of them than scalar registers (the by-value form
// of these kernels spilled 49 (BVH) and 121 (linear scan) scalar registers, each costing a v_readlane per use).
typedef const SceneDev __attribute__((address_space(4))) *SceneK;
typedef const SpherePair __attribute__((address_space(4))) *K_SpherePair;

// Cuts the compiler's view of where a scene pointer came from, so the loads behind it are issued after this point
// (per phase of the loop) instead of being hoisted to the top and kept in registers across the phases.
RT_DEV SceneK fresh(SceneK s)
{
    asm volatile("" : "+s"(s));
    return s;
}

struct Hit {
    double t;
    int32_t slot; // < 0: miss
    int32_t sid;
};

// Work a launch did beyond counting rays, for pricing the tree walk: inner-node visits (each tests two child
// boxes) and exact Sphere::hit evaluations.  Per lane; the kernels sum them over the wave at the end.
struct Work {
    uint32_t node_visits, sphere_tests;
};

// Exact Sphere::hit distance in the reference's operation order.
RT_DEV bool sphere_t(double ox, double oy, double oz, double dx, double dy, double dz, double cx, double cy,
                     double cz, double r2, double &t)
{
    const double ocx = cx - ox, ocy = cy - oy, ocz = cz - oz;
    const double a = (dx * dx + dy * dy) + dz * dz;
    const double h = (dx * ocx + dy * ocy) + dz * ocz;
    const double c = ((ocx * ocx + ocy * ocy) + ocz * ocz) - r2;
    const double delta = h * h - a * c;
    if (delta < 0.0) return false;
    const double sq = sqrt_exact(delta);
    const double t0 = (h - sq) / a;
    if (t0 >= 0.0) { t = t0; return true; }
    const double t1 = (h + sq) / a;
    if (t1 >= 0.0) { t = t1; return true; }
    return false;
}

RT_DEV void consider(SceneK S, int slot, double ox, double oy, double oz, double dx, double dy, double dz,
                     Hit &best, int64_t &best_key, Work &work)
{
    double t;
    work.sphere_tests += 1;
    if (sphere_t(ox, oy, oz, dx, dy, dz, S->cx[slot], S->cy[slot], S->cz[slot], S->r2[slot], t)) {
        const int64_t key = hit_key(t);
        const int sid = S->sid[slot];
        if (key < best_key || (key == best_key && sid < best.sid)) {
            best_key = key;
            best.t = t;
            best.slot = slot;
            best.sid = sid;
        }
    }
}

// Phase-1 filter word over slots [lo, hi): the reference's products up to the
// discriminant, then keep iff delta >= 0 (sign bit).  Never drops a hit.
RT_DEV uint32_t filter_word(SceneK S, int lo, int hi, double ox, double oy, double oz, double dx, double dy,
                            double dz, double a)
{
    uint32_t acc = 0;
#pragma unroll 2
    for (int s = lo; s < hi; ++s) {
        const double ocx = S->cx[s] - ox, ocy = S->cy[s] - oy, ocz = S->cz[s] - oz;
        const double h = (dx * ocx + dy * ocy) + dz * ocz;
        const double c = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S->r2[s];
        const double delta = h * h - a * c;
        acc = __builtin_amdgcn_alignbit(acc, (uint32_t)~hi32(delta), 31);
    }
    return acc << (32 - (hi - lo));
}

// The same decision in f32, two spheres per packed instruction, for rays with | |d|^2 - 1 | <= 2^-40: the
// reference's discriminant delta = h^2 - a c itself, expanded around the frame centre m so that the per-ray and the
// per-sphere parts are computed once, in f64.  With u = o - m (U = |u|_inf, any size) and v = centre - m (|v_i| <= B
// for every sphere the frame holds), oc = v - u and
//   h = d . oc = d . v - (d . u),                     three fused steps on the per-ray -(d . u);
//   c = |oc|^2 - r^2 = (|v|^2 - r^2) - 2 u . v + |u|^2,   three fused steps on (q + k), q per sphere, k = |u|^2 per ray;
//   g = h^2 - c                                        one fused step:
// 8 packed operations per pair + one v_alignbit per sphere (the cross-product form this replaces took 10, the f64 form
// above 15 f64 operations per sphere).  Errors against the true delta (eps = 2^-24, |d_i| <= 1): h carries the rounded
// d and v (4 eps B), the rounded per-ray term (2 eps U) and three steps (6 eps (B + U)): within 12 eps (B + U) of a value
// of size <= 1.8 (B + U), so h^2 is within 42 eps (B + U)^2; c carries the centre's rounding (6 eps B (B + U)), the
// rounded u in the cross term (6 eps B U), the rounded q and k (3 eps (B^2 + U^2)) and four steps of size <= 3 (B + U)^2:
// within 27 eps (B + U)^2; the last step adds 6 eps (B + U)^2.  Together 75 eps (B + U)^2; a = 1 +- 2^-40 moves a c by
// 2^-40 x 3 (B + U)^2 and the reference's own f64 rounding of delta is many orders below.  With the stored
// q = |v|^2 - r^2 (1 + 2^-22) rounded DOWN and k = |u|^2 - 128 eps (B + U)^2 rounded DOWN,
//   g >= 0   whenever the reference's delta >= 0.
// A ray near the scene gets a thin pad; one that starts far away (a bounce off the ground plane's far side)
// gets a wide one and more candidates, but never the f64 loop.
struct FilterTerms {
    float dx, dy, dz;       // direction
    float ndu;              // -(d . u)
    float m2ux, m2uy, m2uz; // -2 u
    float k;                // |u|^2 - pad, rounded down
};

RT_DEV FilterTerms filter_terms(SceneK S, double ox, double oy, double oz, double dx, double dy, double dz, bool &in_range)
{
    const double ux = ox - S->filt_m[0], uy = oy - S->filt_m[1], uz = oz - S->filt_m[2];
    in_range = __builtin_fabs(((dx * dx + dy * dy) + dz * dz) - 1.0) <= 0x1.0p-40;
    const double reach = S->filt_B + __builtin_fmax(__builtin_fmax(__builtin_fabs(ux), __builtin_fabs(uy)), __builtin_fabs(uz));
    const double kd = ((ux * ux + uy * uy) + uz * uz) - reach * reach * (128.0 * 0x1.0p-24);
    float kf = (float)kd;
    if ((double)kf > kd) kf -= __builtin_fmaxf(__builtin_fabsf(kf) * 0x1.0p-23f, 0x1.0p-126f); // at least one ulp down
    FilterTerms f;
    f.dx = (float)dx; f.dy = (float)dy; f.dz = (float)dz;
    f.ndu = -(float)((dx * ux + dy * uy) + dz * uz);
    f.m2ux = -2.0f * (float)ux; f.m2uy = -2.0f * (float)uy; f.m2uz = -2.0f * (float)uz;
    f.k = kf;
    return f;
}

struct FilterRay3 {
    f32x2 dx, dy, dz;       // the terms above, each duplicated into both halves
    f32x2 ndu;
    f32x2 m2ux, m2uy, m2uz;
    f32x2 k;
};

RT_DEV FilterRay3 filter_ray(SceneK S, double ox, double oy, double oz, double dx, double dy, double dz, bool &in_range)
{
    const FilterTerms t = filter_terms(S, ox, oy, oz, dx, dy, dz, in_range);
    FilterRay3 f;
    f.dx = f32x2{ t.dx, t.dx }; f.dy = f32x2{ t.dy, t.dy }; f.dz = f32x2{ t.dz, t.dz };
    f.ndu = f32x2{ t.ndu, t.ndu };
    f.m2ux = f32x2{ t.m2ux, t.m2ux }; f.m2uy = f32x2{ t.m2uy, t.m2uy }; f.m2uz = f32x2{ t.m2uz, t.m2uz };
    f.k = f32x2{ t.k, t.k };
    return f;
}

// The same filter for the LEAF_MAX spheres of one BVH leaf, each lane on its own leaf: plain f32 steps on per-lane
// 16-byte records (8 operations and one v_alignbit per sphere; a box test is 18 and passes about twice the misses).
// A ray whose direction is not a unit vector carries k = -inf, which makes every sphere a candidate (c = -inf,
// g = +inf; every other operand is finite, so no NaN arises for a real sphere).  Returns the candidate bits,
// bit j = sphere first + j.
RT_DEV FilterTerms leaf_ray(SceneK S, double ox, double oy, double oz, double dx, double dy, double dz)
{
    bool in_range;
    FilterTerms t = filter_terms(S, ox, oy, oz, dx, dy, dz, in_range);
    if (!in_range) t.k = -__builtin_inff();
    return t;
}

RT_DEV uint32_t leaf_filter(const SphereF32 *q, int cnt, const FilterTerms &f)
{
    uint32_t acc = 0; // collects DROP bits (the sign of g), the last sphere first
#pragma unroll
    for (int j = LEAF_MAX - 1; j >= 0; --j) {
        const SphereF32 s = q[j];
        const float h = __builtin_fmaf(f.dx, s.vx, __builtin_fmaf(f.dy, s.vy, __builtin_fmaf(f.dz, s.vz, f.ndu)));
        const float c = __builtin_fmaf(f.m2ux, s.vx, __builtin_fmaf(f.m2uy, s.vy, __builtin_fmaf(f.m2uz, s.vz, s.e + f.k)));
        const float g = __builtin_fmaf(h, h, -c);
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g), 31);
    }
    return ~acc & ((1u << cnt) - 1u);
}

RT_DEV uint32_t filter_word32(SceneK S, int lo, int hi, const FilterRay3 &f)
{
    uint32_t acc = 0; // collects DROP bits (the sign of g)
    auto pair = [&](K_SpherePair q) {
        const f32x2 vx = q->vx, vy = q->vy, vz = q->vz, e = q->e;
        const f32x2 h = __builtin_elementwise_fma(f.dx, vx, __builtin_elementwise_fma(f.dy, vy, __builtin_elementwise_fma(f.dz, vz, f.ndu)));
        const f32x2 c = __builtin_elementwise_fma(f.m2ux, vx, __builtin_elementwise_fma(f.m2uy, vy, __builtin_elementwise_fma(f.m2uz, vz, e + f.k)));
        return __builtin_elementwise_fma(h, h, -c);
    };
    int s = lo;
    K_SpherePair q = (K_SpherePair)S->f32p + (lo >> 1); // lo is a multiple of 32
#pragma unroll 4
    for (; s + 2 <= hi; s += 2, ++q) {
        const f32x2 g = pair(q);
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.x), 31);
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.y), 31);
    }
    if (s < hi) { // odd count: the pair's second half is the padding entry
        const f32x2 g = pair(q);
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.x), 31);
    }
    return ~acc << (32 - (hi - lo));
}

// Brute force over slots [first, last): same two-phase scheme as the 2d tracer.
template <bool F32>
RT_DEV void scan_range(SceneK S, int first, int last, double ox, double oy, double oz, double dx, double dy,
                       double dz, bool active, Hit &best, int64_t &best_key, Work &work)
{
    const double a = (dx * dx + dy * dy) + dz * dz;
    // the f32 filter serves the wave when the range is worth it and every active lane's direction is a unit vector
    bool in_range = false;
    FilterRay3 fr{};
    const bool wide = F32 && last - first >= 16; // wave-uniform
    if (wide) fr = filter_ray(S, ox, oy, oz, dx, dy, dz, in_range);
    const bool all_in = wide && !__any(active && !in_range);
    for (int base = first; base < last; base += 256) {
        uint32_t m[8];
        if (F32 && all_in) {
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int lo = base + 32 * w;
                const int hi = (lo + 32 < last) ? lo + 32 : last;
                m[w] = (lo < last) ? filter_word32(S, lo, hi, fr) : 0u;
            }
        } else {
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int lo = base + 32 * w;
                const int hi = (lo + 32 < last) ? lo + 32 : last;
                m[w] = (lo < last) ? filter_word(S, lo, hi, ox, oy, oz, dx, dy, dz, a) : 0u;
            }
        }
        uint64_t m0 = ((uint64_t)m[0] << 32) | m[1], m1 = ((uint64_t)m[2] << 32) | m[3];
        uint64_t m2 = ((uint64_t)m[4] << 32) | m[5], m3 = ((uint64_t)m[6] << 32) | m[7];
        if (!active) { m0 = m1 = m2 = m3 = 0; }
        while ((m0 | m1 | m2 | m3) != 0) {
            const int w = m0 ? 0 : m1 ? 1 : m2 ? 2 : 3;
            const uint64_t cur = m0 ? m0 : m1 ? m1 : m2 ? m2 : m3;
            const int b = __clzll((long long)cur);
            const uint64_t rest = cur & ~(0x8000000000000000ULL >> b);
            m0 = (w == 0) ? rest : m0;
            m1 = (w == 1) ? rest : m1;
            m2 = (w == 2) ? rest : m2;
            m3 = (w == 3) ? rest : m3;
            consider(S, base + 64 * w + b, ox, oy, oz, dx, dy, dz, best, best_key, work);
        }
    }
}

// Slab intervals of a ray against the two child boxes of a node (method of 2d/src/aabb.rs:39-66, three axes) in f32,
// both boxes per packed instruction: with idf = 1 / (float) d, t = plane * idf + c in one fused step, where c =
// -(float) o * idf -+ pad |idf| per ray and axis, minus for the plane the ray meets first (the low one when d >= 0).
// Conservative in three steps: (1) the boxes are rounded outward on the host; (2) c's own error - o rounded to f32
// and the fused step's rounding, together < 2.1 * 2^-24 |o| |idf| - is below a quarter of the pad
// 2^-21 (|o|_inf + scene scale) |idf|, which moves the near bound down and the far bound up; (3) idf carries a relative
// error <= 2^-23 (v_rcp_f32 is good to one unit in the last place) and the step that forms t another 2^-24: the bounds
// are moved apart by 2^-20 |t| (one fused step each, a single rounding, well inside the slack).
// A direction component below 2^-60 in size (zero included) is taken as +-2^-60: idf stays finite, so no step ever
// meets 0 * inf, and the decision is the one the true component gives - an origin inside the padded slab sees t of
// opposite signs and astronomic size on its two planes (the axis constrains nothing), an origin outside sees both
// beyond 2^39 (|o|_inf + scale) on one side, farther than any sphere of the tree can be hit, while with the true
// component the slab is reached later still.
// hit[c] = the padded interval of child c reaches t >= 0 and starts at or before `bt`; tn[c] = its lower end.
// (Fetching near and far planes by per-lane offsets instead - no min / max per axis - cuts another 12 instructions and
// was slower, 27.5 against 32.4 Grays/s: seven divergent loads per visit instead of four.)
struct BoxRay {
    float idf[3], c_lo[3], c_hi[3];
};

RT_DEV BoxRay box_ray(double ox, double oy, double oz, double dx, double dy, double dz, float scale)
{
    const float of[3] = { (float)ox, (float)oy, (float)oz };
    const float df[3] = { (float)dx, (float)dy, (float)dz };
    const float pad = (__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(of[0]), __builtin_fabsf(of[1])), __builtin_fabsf(of[2])) + scale) * 0x1.0p-21f;
    BoxRay r;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        r.idf[a] = __builtin_amdgcn_rcpf(__builtin_copysignf(__builtin_fmaxf(__builtin_fabsf(df[a]), 0x1.0p-60f), df[a]));
        // pad * idf has the direction's sign: subtracting it lowers t on the low plane when that one is the near plane
        // (d >= 0) and raises it when it is the far one
        const float padt = pad * r.idf[a];
        r.c_lo[a] = __builtin_fmaf(-of[a], r.idf[a], -padt);
        r.c_hi[a] = __builtin_fmaf(-of[a], r.idf[a], padt);
    }
    return r;
}

RT_DEV void box_intervals(const BvhNode2 &nd, const BoxRay &r, float bt, f32x2 &tn, bool hit[2])
{
    // plain f32 steps, not packed ones: a packed operand wants the ray's constants duplicated into register pairs (18
    // registers instead of 9, which this kernel does not have), and two v_fma_f32 issue in about the time of one
    // v_pk_fma_f32
    float t1[3][2], t2[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            t1[a][c] = __builtin_fmaf(nd.plane[2 * a][c], r.idf[a], r.c_lo[a]);
            t2[a][c] = __builtin_fmaf(nd.plane[2 * a + 1][c], r.idf[a], r.c_hi[a]);
        }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1[0][c], t2[0][c]), __builtin_fminf(t1[1][c], t2[1][c])), __builtin_fminf(t1[2][c], t2[2][c]));
        const float hi = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1[0][c], t2[0][c]), __builtin_fmaxf(t1[1][c], t2[1][c])), __builtin_fmaxf(t1[2][c], t2[2][c]));
        tn[c] = __builtin_fmaf(-__builtin_fabsf(lo), 0x1.0p-20f, lo);
        const float tf = __builtin_fmaf(__builtin_fabsf(hi), 0x1.0p-20f, hi);
        // tf >= max(tn, 0) and not tn > bt (bt >= 0, or +inf)
        hit[c] = __builtin_fminf(tf, bt) >= __builtin_fmaxf(tn[c], 0.0f);
    }
}

RT_DEV Hit no_hit()
{
    Hit best;
    best.t = 0.0;
    best.slot = -1;
    best.sid = 0x7fffffff;
    return best;
}

// A hit as an f32 at or above it: a box whose interval starts beyond THAT starts beyond the hit (equal t may
// still win on index, so only strictly later boxes are dropped); +inf while there is no hit.
RT_DEV float hit_bound(const Hit &best)
{
    if (best.slot < 0) return __builtin_inff();
    float f = (float)best.t;
    if ((double)f < best.t) f = __uint_as_float(__float_as_uint(f) + 1u); // best.t >= 0 here: next float up
    return f;
}

// The spheres kept out of the tree (BVH mode): exact tests on every ray.  They are one or two (the ground): the
// exact test on every lane costs less than the filter pass over eight mostly empty words and its mask bookkeeping
// (standalone scene: 31.2 -> 33.5 Grays/s).
RT_DEV void always_hits(SceneK S, double ox, double oy, double oz, double dx, double dy, double dz, bool active,
                        Hit &best, int64_t &best_key, Work &work)
{
    if (S->n_always <= 4) {
        if (active)
            for (int s = 0; s < S->n_always; ++s) consider(S, s, ox, oy, oz, dx, dy, dz, best, best_key, work);
    } else {
        scan_range<false>(S, 0, S->n_always, ox, oy, oz, dx, dy, dz, active, best, best_key, work);
    }
}

constexpr int BVH_STACK = 32;

// The tree walk, from the root, for a ray that may already hold a hit (`best`, `best_key`).
// `stack` points at this lane's column of an LDS array laid out [level][2][lane]: child index and the
// lower end of its interval (float bits), so that a popped subtree can be dropped without a visit when a
// nearer hit has been found since it was pushed.
RT_DEV void walk_tree(SceneK S, double ox, double oy, double oz, double dx, double dy, double dz, bool active,
                      const BoxRay &br, int32_t *stack, int stack_stride, Hit &best, int64_t &best_key, Work &work, Prof &prof)
{
    (void)prof;
    const FilterTerms lr = leaf_ray(S, ox, oy, oz, dx, dy, dz);
    // "while-while" traversal: every lane walks inner nodes until it stands on a leaf (or has nothing left); only
    // then does the wave run the leaves' sphere filter and exact tests, all leaf-holding lanes together.  Running
    // them inside the inner loop would charge every iteration of the wave with the f64 sphere tests of whichever lane
    // happens to be at a leaf (27.5 -> 29.7 Grays/s on the standalone scene; leaving the inner loop early, once fewer than
    // 8..32 lanes still walk, was slower: 26.8..27.7).  A child reference is an inner node (>= 0) or a leaf
    // (< 0, see BvhNode2); NONE = finished.
    constexpr int NONE = (int)0x80000000;
    int sp = 0;
    int node = active ? 0 : NONE;
    float bt = hit_bound(best);
    // next subtree from the stack whose interval still starts before the best hit
    auto pop = [&]() {
        while (sp > 0) {
            --sp;
            const int cand = stack[(2 * sp) * stack_stride];
            const float tc = __int_as_float(stack[(2 * sp + 1) * stack_stride]);
            if (!(tc > bt)) return cand;
        }
        return NONE;
    };
    RT_STAMP(prof, 1);
#ifdef RT_PROF
    const uint32_t visits_before = work.node_visits;
#endif
    while (__any(node != NONE)) {
        while (__any(node >= 0)) {
#ifdef RT_PROF
            prof.t[7] += 1;
#endif
            if (node >= 0) {
                work.node_visits += 1;
                const BvhNode2 nd = S->nodes[node];
                const int child0 = nd.child[0], child1 = nd.child[1];
                f32x2 tn;
                bool h[2];
                box_intervals(nd, br, bt, tn, h);
                const bool h0 = h[0], h1 = h[1];
                if (h0 && h1) {
                    const bool first0 = tn.x <= tn.y;
                    node = first0 ? child0 : child1;
                    stack[(2 * sp) * stack_stride] = first0 ? child1 : child0; // depth is validated <= BVH_STACK on the host
                    stack[(2 * sp + 1) * stack_stride] = __float_as_int(first0 ? tn.y : tn.x);
                    ++sp;
                } else if (h0 || h1) {
                    node = h0 ? child0 : child1;
                } else {
                    node = pop();
                }
            }
        }
        RT_STAMP(prof, 2);
#ifdef RT_PROF
        prof.t[8] += 1;
        prof.t[10] += (unsigned long long)__popcll(__ballot(node != NONE));
#endif
        if (node != NONE) { // a leaf: the f32 filter on its spheres, the exact test on what it leaves
            const int enc = ~node;
            const int first = enc >> 3;
            uint32_t cand = leaf_filter(S->f32s + first, enc & 7, lr);
#ifdef RT_PROF
            { uint32_t any_c = cand; while (__any(any_c != 0)) { prof.t[9] += 1; any_c &= any_c - 1; } }
#endif
            while (cand != 0) {
                const int j = __builtin_ctz(cand);
                cand &= cand - 1;
                consider(S, first + j, ox, oy, oz, dx, dy, dz, best, best_key, work);
            }
            bt = hit_bound(best);
            node = pop();
        }
        RT_STAMP(prof, 3);
    }
#ifdef RT_PROF
    if (active && prof.octants) { // does the direction's octant say how long a ray walks?
        const unsigned long long nv = work.node_visits - visits_before;
        unsigned long long *o = prof.octants + 4 * ((dx < 0.0 ? 1 : 0) | (dy < 0.0 ? 2 : 0) | (dz < 0.0 ? 4 : 0));
        atomicAdd(o, 1ull);
        if (nv > 1) {
            atomicAdd(o + 1, 1ull);
            atomicAdd(o + 2, nv);
            atomicAdd(o + 3, nv * nv);
        }
    }
#endif
}

// BVH = false: the reference's linear scan over every sphere (with the f32 filter ahead of the exact tests).
// BVH = true: the spheres kept out of the tree, then the tree.  A kernel is compiled for each so that neither
// carries the other's registers.
template <bool BVH>
RT_DEV Hit nearest_hit(SceneK S, double ox, double oy, double oz, double dx, double dy, double dz,
                       bool active, int32_t *stack, int stack_stride, Work &work, Prof &prof)
{
    S = fresh(S);
    Hit best = no_hit();
    int64_t best_key = 0x7fffffffffffffffLL;
    if (!BVH) {
        scan_range<true>(S, 0, S->n, ox, oy, oz, dx, dy, dz, active, best, best_key, work);
        return best;
    }
    always_hits(S, ox, oy, oz, dx, dy, dz, active, best, best_key, work);
    if (S->n_nodes == 0) return best;
    const BoxRay br = box_ray(ox, oy, oz, dx, dy, dz, (float)S->box_scale);
    walk_tree(S, ox, oy, oz, dx, dy, dz, active, br, stack, stack_stride, best, best_key, work, prof);
    return best;
}

struct Path {
    double ox, oy, oz, dx, dy, dz;
    float thr;           // 0.5^bounces
    int32_t left;        // the reference's `bounce_limit` argument
    uint32_t c0, c1;
};

// Camera::get_ray for pixel (i, j), global sample index `gs`.
RT_DEV void generate(SceneK S, Path &p, int i, int j, uint32_t gs, int bounce_limit, uint32_t k0, uint32_t k1)
{
    const uint32_t pix = (uint32_t)(S->width * j + i);
    const U4 w = philox4x32_10(pix, gs, 0xFFFFFFFFu, 0u, k0, k1);
    const double rx = u01(w.x, w.y) - 1.0 / 2.0, ry = u01(w.z, w.w) - 1.0 / 2.0;
    const double a = (double)i + rx, b = (double)j + ry;
    const double vx = ((S->p00[0] + a * S->du[0]) + b * S->dv[0]) - S->pos[0];
    const double vy = ((S->p00[1] + a * S->du[1]) + b * S->dv[1]) - S->pos[1];
    const double vz = ((S->p00[2] + a * S->du[2]) + b * S->dv[2]) - S->pos[2];
    const double rl = rlen3(vx, vy, vz);
    p.dx = vx * rl; p.dy = vy * rl; p.dz = vz * rl;
    p.ox = S->pos[0]; p.oy = S->pos[1]; p.oz = S->pos[2];
    p.thr = 1.0f;
    p.left = bounce_limit;
    p.c0 = pix;
    p.c1 = gs;
}

// Body of World::trace_ray after the nearest hit is known.  Returns true when the
// path ended, with rgb = the colour the innermost call returned times 0.5^bounces.
RT_DEV bool shade(SceneK S, Path &p, const Hit &h, int bounce_limit, uint32_t k0, uint32_t k1, float rgb[3])
{
    if (h.slot < 0) {
        const float a = 0.5f * ((float)p.dy + 1.0f);
        float r = (1.0f - a) * 1.0f + a * 0.0f;
        float g = (1.0f - a) * 1.0f + a * 0.7f;
        float b = (1.0f - a) * 1.0f + a * 1.0f;
        // DVec3::new(0.3, 1., 0.).normalize() (rt.rs:283): constants, folded by the compiler with IEEE operations
        const double sl = 1.0 / __builtin_sqrt((0.3 * 0.3 + 1.0 * 1.0) + 0.0 * 0.0);
        const double sx = 0.3 * sl, sy = 1.0 * sl, sz = 0.0 * sl;
        if ((p.dx * sx + p.dy * sy) + p.dz * sz > 0.99) { r = 240.0f; g = 255.0f; b = 170.0f; }
        // r, g, b are +0 or in [2^-24, 255]: the range div_by_100 is exact on
        rgb[0] = p.thr * div_by_100(r);
        rgb[1] = p.thr * div_by_100(g);
        rgb[2] = p.thr * div_by_100(b);
        return true;
    }
    const int s = h.slot;
    const double hx = p.ox + p.dx * h.t, hy = p.oy + p.dy * h.t, hz = p.oz + p.dz * h.t;
    double nx = hx - S->cx[s], ny = hy - S->cy[s], nz = hz - S->cz[s];
    const double nl = rlen3(nx, ny, nz);
    nx = nx * nl; ny = ny * nl; nz = nz * nl;
    const uint32_t bounce = (uint32_t)(bounce_limit - p.left);
    const U4 w0 = philox4x32_10(p.c0, p.c1, 0u, (bounce << 20) | 0u, k0, k1);
    const U4 w1 = philox4x32_10(p.c0, p.c1, 0u, (bounce << 20) | 1u, k0, k1);
    double vx = u01(w0.x, w0.y) - 0.5 * 1.0, vy = u01(w0.z, w0.w) - 0.5 * 1.0, vz = u01(w1.x, w1.y) - 0.5 * 1.0;
    const double vl = rlen3(vx, vy, vz);
    vx = vx * vl; vy = vy * vl; vz = vz * vl;
    if (!((nx * vx + ny * vy) + nz * vz > 0.0)) { vx = -vx; vy = -vy; vz = -vz; }
    double bx = vx + nx, by = vy + ny, bz = vz + nz;
    const double bl = rlen3(bx, by, bz);
    bx = bx * bl; by = by * bl; bz = bz * bl;
    const double off = 100.0 * EPS64;
    p.ox = hx + nx * off; p.oy = hy + ny * off; p.oz = hz + nz * off;
    p.dx = bx; p.dy = by; p.dz = bz;
    p.thr = p.thr * 0.5f;
    p.left -= 1;
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    return p.left <= 0;
}

} // namespace rt3d
