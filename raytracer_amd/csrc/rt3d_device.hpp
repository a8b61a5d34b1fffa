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

// BVH node, 64 B.  Inner node: count == 0, children at `left` and `left + 1`.
// Leaf: `count` spheres starting at slot `left` (slots are BVH order; `sid`
// maps a slot back to the caller's sphere index).  Boxes are padded outward on
// the host by a relative margin far larger than the rounding error of the slab
// arithmetic, so a sphere hit at distance t always lies inside [tmin, tmax] of
// every ancestor box as computed here.
struct BvhNode {
    double lo[3], hi[3];
    int32_t left, count;
    int32_t axis; // split axis of an inner node: the child on the ray's near side is visited first
    int32_t pad_;
};

struct SceneDev {
    int32_t n, n_always, n_nodes, width, height, use_bvh;
    // slot order: the `n_always` spheres tested on every ray first (those too
    // large for the tree), then BVH leaf order
    const double *cx, *cy, *cz, *r2;
    const int32_t *sid;
    const BvhNode *nodes;
    double pos[3], p00[3], du[3], dv[3];
};

struct Hit {
    double t;
    int32_t slot; // < 0: miss
    int32_t sid;
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
    const double t0 = (h - __builtin_sqrt(delta)) / a;
    if (t0 >= 0.0) { t = t0; return true; }
    const double t1 = (h + __builtin_sqrt(delta)) / a;
    if (t1 >= 0.0) { t = t1; return true; }
    return false;
}

RT_DEV void consider(const SceneDev &S, int slot, double ox, double oy, double oz, double dx, double dy, double dz,
                     Hit &best, int64_t &best_key)
{
    double t;
    if (sphere_t(ox, oy, oz, dx, dy, dz, S.cx[slot], S.cy[slot], S.cz[slot], S.r2[slot], t)) {
        const int64_t key = total_key(t);
        const int sid = S.sid[slot];
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
RT_DEV uint32_t filter_word(const SceneDev &S, int lo, int hi, double ox, double oy, double oz, double dx, double dy,
                            double dz, double a)
{
    uint32_t acc = 0;
#pragma unroll 2
    for (int s = lo; s < hi; ++s) {
        const double ocx = S.cx[s] - ox, ocy = S.cy[s] - oy, ocz = S.cz[s] - oz;
        const double h = (dx * ocx + dy * ocy) + dz * ocz;
        const double c = ((ocx * ocx + ocy * ocy) + ocz * ocz) - S.r2[s];
        const double delta = h * h - a * c;
        acc = __builtin_amdgcn_alignbit(acc, (uint32_t)~hi32(delta), 31);
    }
    return acc << (32 - (hi - lo));
}

// Brute force over slots [first, last): same two-phase scheme as the 2d tracer.
RT_DEV void scan_range(const SceneDev &S, int first, int last, double ox, double oy, double oz, double dx, double dy,
                       double dz, bool active, Hit &best, int64_t &best_key)
{
    const double a = (dx * dx + dy * dy) + dz * dz;
    for (int base = first; base < last; base += 256) {
        uint32_t m[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int lo = base + 32 * w;
            const int hi = (lo + 32 < last) ? lo + 32 : last;
            m[w] = (lo < last) ? filter_word(S, lo, hi, ox, oy, oz, dx, dy, dz, a) : 0u;
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
            consider(S, base + 64 * w + b, ox, oy, oz, dx, dy, dz, best, best_key);
        }
    }
}

// Slab test against a padded box (method of 2d/src/aabb.rs:39-66, three axes):
// true iff the box may contain a hit with total_cmp key <= best_key.
RT_DEV bool box_may_hit(const BvhNode &nd, double ox, double oy, double oz, double ix, double iy, double iz,
                        double t_best, bool have_best)
{
    const double tx1 = (nd.lo[0] - ox) * ix, tx2 = (nd.hi[0] - ox) * ix;
    double tmin = __builtin_fmin(tx1, tx2), tmax = __builtin_fmax(tx1, tx2);
    const double ty1 = (nd.lo[1] - oy) * iy, ty2 = (nd.hi[1] - oy) * iy;
    tmin = __builtin_fmax(__builtin_fmin(ty1, ty2), tmin);
    tmax = __builtin_fmin(__builtin_fmax(ty1, ty2), tmax);
    const double tz1 = (nd.lo[2] - oz) * iz, tz2 = (nd.hi[2] - oz) * iz;
    tmin = __builtin_fmax(__builtin_fmin(tz1, tz2), tmin);
    tmax = __builtin_fmin(__builtin_fmax(tz1, tz2), tmax);
    // NaN (0 * inf when the origin lies on a slab plane of an axis-parallel ray)
    // is dropped by fmin/fmax, which keeps the other operand: conservative.
    if (!(tmax >= __builtin_fmax(tmin, 0.0))) return false;
    if (have_best && tmin > t_best) return false; // equal t may still win on index
    return true;
}

constexpr int BVH_STACK = 32;

// `stack` points at this lane's column of an LDS array laid out [level][lane].
RT_DEV Hit nearest_hit(const SceneDev &S, double ox, double oy, double oz, double dx, double dy, double dz,
                       bool active, int32_t *stack, int stack_stride)
{
    Hit best;
    best.t = 0.0;
    best.slot = -1;
    best.sid = 0x7fffffff;
    int64_t best_key = 0x7fffffffffffffffLL;
    if (!S.use_bvh) {
        scan_range(S, 0, S.n, ox, oy, oz, dx, dy, dz, active, best, best_key);
        return best;
    }
    scan_range(S, 0, S.n_always, ox, oy, oz, dx, dy, dz, active, best, best_key);
    if (S.n_nodes == 0) return best;
    const double ix = 1.0 / dx, iy = 1.0 / dy, iz = 1.0 / dz;
    int sp = 0;
    int node = active ? 0 : -1;
    while (__any(node >= 0)) {
        if (node >= 0) {
            const BvhNode nd = S.nodes[node];
            int next = -1;
            if (box_may_hit(nd, ox, oy, oz, ix, iy, iz, best.t, best.slot >= 0)) {
                if (nd.count > 0) {
                    for (int k = 0; k < nd.count; ++k)
                        consider(S, nd.left + k, ox, oy, oz, dx, dy, dz, best, best_key);
                } else {
                    const double da = (nd.axis == 0) ? dx : (nd.axis == 1) ? dy : dz;
                    const int near = (da < 0.0) ? 1 : 0;
                    next = nd.left + near;
                    stack[sp * stack_stride] = nd.left + (1 - near); // depth is validated <= BVH_STACK on the host
                    ++sp;
                }
            }
            if (next < 0 && sp > 0) { --sp; next = stack[sp * stack_stride]; }
            node = next;
        }
    }
    return best;
}

struct Path {
    double ox, oy, oz, dx, dy, dz;
    float thr;           // 0.5^bounces
    int32_t left;        // the reference's `bounce_limit` argument
    uint32_t c0, c1;
};

// Camera::get_ray for pixel (i, j), global sample index `gs`.
RT_DEV void generate(const SceneDev &S, Path &p, int i, int j, uint32_t gs, int bounce_limit, uint32_t k0, uint32_t k1)
{
    const uint32_t pix = (uint32_t)(S.width * j + i);
    const U4 w = philox4x32_10(pix, gs, 0xFFFFFFFFu, 0u, k0, k1);
    const double rx = u01(w.x, w.y) - 1.0 / 2.0, ry = u01(w.z, w.w) - 1.0 / 2.0;
    const double a = (double)i + rx, b = (double)j + ry;
    const double vx = ((S.p00[0] + a * S.du[0]) + b * S.dv[0]) - S.pos[0];
    const double vy = ((S.p00[1] + a * S.du[1]) + b * S.dv[1]) - S.pos[1];
    const double vz = ((S.p00[2] + a * S.du[2]) + b * S.dv[2]) - S.pos[2];
    const double rl = rlen3(vx, vy, vz);
    p.dx = vx * rl; p.dy = vy * rl; p.dz = vz * rl;
    p.ox = S.pos[0]; p.oy = S.pos[1]; p.oz = S.pos[2];
    p.thr = 1.0f;
    p.left = bounce_limit;
    p.c0 = pix;
    p.c1 = gs;
}

// Body of World::trace_ray after the nearest hit is known.  Returns true when the
// path ended, with rgb = the colour the innermost call returned times 0.5^bounces.
RT_DEV bool shade(const SceneDev &S, Path &p, const Hit &h, int bounce_limit, uint32_t k0, uint32_t k1, float rgb[3])
{
    if (h.slot < 0) {
        const float a = 0.5f * ((float)p.dy + 1.0f);
        float r = (1.0f - a) * 1.0f + a * 0.0f;
        float g = (1.0f - a) * 1.0f + a * 0.7f;
        float b = (1.0f - a) * 1.0f + a * 1.0f;
        const double sl = rlen3(0.3, 1.0, 0.0);
        const double sx = 0.3 * sl, sy = 1.0 * sl, sz = 0.0 * sl;
        if ((p.dx * sx + p.dy * sy) + p.dz * sz > 0.99) { r = 240.0f; g = 255.0f; b = 170.0f; }
        rgb[0] = p.thr * (r / 100.0f);
        rgb[1] = p.thr * (g / 100.0f);
        rgb[2] = p.thr * (b / 100.0f);
        return true;
    }
    const int s = h.slot;
    const double hx = p.ox + p.dx * h.t, hy = p.oy + p.dy * h.t, hz = p.oz + p.dz * h.t;
    double nx = hx - S.cx[s], ny = hy - S.cy[s], nz = hz - S.cz[s];
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
