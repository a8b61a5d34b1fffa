// 2d spectral gather tracer: device-side nearest-hit search, shading and ray
// generation.  One lane owns one path; primitives are wave-uniform operands.
//
// Reference semantics (paths relative to the reference root):
//   nearest hit    2d/src/librt2d_reverse.rs:135-139, rt_common.rs:79-111 (Circle::hit),
//                  :242-276 (Segment::hit), :295-305 (Strip::hit)
//   shading        2d/src/librt2d_reverse.rs:146-261
//   ray generation 2d/src/librt2d_reverse.rs:88-94, rt_common.rs:369-391, spectrum.rs:185-191
#pragma once

#include "rt_device_math.hpp"

namespace rt2d {
using namespace rtd;

enum { MAT_EMISSIVE = 0, MAT_DIRECTIONAL = 1, MAT_DIFFUSE = 2, MAT_REFLECTIVE = 3, MAT_DIELECTRIC = 4 };

struct MatDev {
    int32_t kind;
    int32_t ior_row;  // Dielectric: row of the per-wavelength ior tables
    double angle, d;  // DirectionalEmissive
    int32_t spectrum;
    int32_t black; // Diffuse whose spectrum is all zero (`absorption == Spectrum::default()`, librt2d_reverse.rs:212)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
// The scene is read-only for the whole launch.  Reading it through constant-address-space pointers lets the
// compiler keep wave-uniform reads on the scalar unit (s_load) no matter what else the loop does; through
// plain pointers it falls back to vector loads as soon as anything it cannot see through (a wave barrier, a
// timer read, volatile asm) sits between kernel entry and the load, which costs the filter loop 20 %.
typedef const f32x2 __attribute__((address_space(4))) *K_f32x2;
typedef const double __attribute__((address_space(4))) *K_f64;
typedef const float __attribute__((address_space(4))) *K_f32;
struct SegX;
typedef const SegX __attribute__((address_space(4))) *K_SegX;
// Two circles as the f32 filter reads them (see filter_word): centre - frame centre and the padded r^2 of both.
struct CirclePair { f32x2 vx, vy, e; };
typedef const CirclePair __attribute__((address_space(4))) *K_CirclePair;
// Two segments as the f32 filter reads them, each field holding the value of both: the end points relative to the
// frame centre, A = a - m and B = b - m.  One 32-byte record per pair keeps the loop's scalar side to one address and
// one load; separate arrays cost a load and two address instructions each, more than the pair's vector work.
struct SegPair { f32x2 ax, ay, bx, by; };
typedef const SegPair __attribute__((address_space(4))) *K_SegPair;
// Margin of the segment filter for one launch (see filter_word).
struct SegBounds { float eq, pad_; };

// Scene in HBM.  Slot order: circles by ascending pid, then segments by ascending pid.
// A circle as the exact test and the shading read it: one 32-byte record, two 16-byte loads from one address.
struct CircleX {
    double cx, cy, r2; // r2 = r*r, the value the reference recomputes in every hit test (rt_common.rs:83)
    int32_t pid, obj;
};

// A segment, 64 bytes: a, s = b - a (the value the reference recomputes in every hit test, rt_common.rs:244),
// the stored normal, ids.
struct SegX {
    double ax, ay, sx, sy, nx, ny;
    int32_t pid, obj;
    double pad_;
};

struct SceneDev {
    int32_t n_circ, n_seg, n_obj, pad_; // n_circ is even: an odd circle count is padded with a circle nothing can hit
    // circles as the f32 filter sees them (see filter_word): c32x/c32y = centre - frame centre (filt_mx, filt_my)
    // rounded to f32; c32e = r^2 (1 + 2^-22) + filt_M rounded up, rewritten by pad_radii_kernel when filt_M changes
    const CirclePair *c32;
    // filter frame, set per launch: the filter is proven for |origin - centre|_inf <= filt_B and | |dir|^2 - 1 | <= 2^-40;
    // filt_M = 64 * 2^-24 filt_B^2 pads the squared radius
    double filt_mx, filt_my, filt_B, filt_M;
    const double *ior, *inv_ior; // [n_ior][95]: Ior::ior(lambda_i) and its reciprocal (rt_common.rs:442-450)
    const double *ior_fwd, *inv_ior_fwd; // [n_ior]: the same at the forward tracer's `lambda = 1.` (librt2d_forward.rs:213)
    const CircleX *cex;
    const SegX *sex;
    // segments as the f32 filter reads them: one 40-byte record per PAIR of segments (the count padded to even)
    const SegPair *segp;
    SegBounds seg_bounds; // set per launch with the frame
    const MatDev *mats; // per SLOT (the material of the slot's object): a hit's material does not wait for its primitive's record
    const float *spectra;
};

// The kernels read the scene where it arrives - in the kernel-argument segment, constant address space - through
// this pointer, field by field at the point of use.  Copied into a by-value struct every field would be loaded
// into a scalar register at kernel entry and stay live over the whole trace loop; there are more of them than
// scalar registers, and each overflow costs a vector instruction (v_readlane) at every use.
typedef const SceneDev __attribute__((address_space(4))) *SceneK;

// Cuts the compiler's view of where a scene pointer came from, so the loads behind it are issued after this
// point (per phase of the loop) instead of being hoisted to the top and kept in registers across the phases.
RT_DEV SceneK fresh(SceneK s)
{
    asm volatile("" : "+s"(s));
    return s;
}

struct Hit {
    double t;
    int32_t slot; // < 0: miss
    int32_t pid;
};

// ---------------------------------------------------------------------------
// Exact (reference-order) single primitive tests, used on the few candidates
// that survive the sign filters below.
RT_DEV bool circle_t(double ox, double oy, double dx, double dy, double cx, double cy, double r2, double &t)
{
    const double ocx = ox - cx, ocy = oy - cy;
    const double a = 1.0;
    const double b = 2.0 * (dx * ocx + dy * ocy);
    const double c = (ocx * ocx + ocy * ocy) - r2;
    const double delta = b * b - 4.0 * a * c;
    if (delta < 0.0) return false;
    const double sq = sqrt_exact(delta);
    const double t0 = (-b - sq) / (2.0 * a);
    const double t1 = (-b + sq) / (2.0 * a);
    if (t0 > 0.0) { t = t0; return true; }
    if (t1 > 0.0) { t = t1; return true; }
    return false;
}

RT_DEV bool segment_t(double ox, double oy, double rx, double ry, double ax, double ay, double sx, double sy, double &t)
{
    const double denom = rx * sy - ry * sx;
    if (__builtin_fabs(denom) < EPS64) return false;
    const double qx = ax - ox, qy = ay - oy;
    // The reference divides twice (t and u) and asks 0 <= u <= 1.  u is only compared, and the rounded quotient
    // compares like the exact one.  Signs agree (or un = 0): un / denom rounds to something <= 1 iff |un| <= |denom|
    // (for |un| > |denom| the quotient is >= 1 + ulp(denom) / |denom| > 1 + 2^-53, past the midpoint above 1), and it
    // is >= 0.  Signs differ: the quotient is negative unless it underflows to -0 (which compares >= 0), i.e. unless
    // |un| / |denom| <= 2^-1075; scaling |un| by 2^1075 in two exact steps decides that (a larger |un| only overflows
    // towards "no": |denom| < 2^82 for coordinates within 2^40).
    const double un = qx * ry - qy * rx;
    const double aun = __builtin_fabs(un), aden = __builtin_fabs(denom);
    const bool same = (un < 0.0) == (denom < 0.0);
    const bool u_ok = (same ? aun : (aun * 0x1.0p+1000) * 0x1.0p+75) <= aden;
    if (!u_ok) return false;
    const double tt = (qx * sy - qy * sx) / denom;
    if (tt >= 0.0) { t = tt; return true; }
    return false;
}

// ---------------------------------------------------------------------------
// Nearest hit over the whole scene, brute force, in two phases per word of 32
// slots:
//  phase 1 (uniform loop, every lane x every primitive; primitives arrive as
//    scalar operands): a cheap test marks the lanes for which the primitive can
//    still be a hit.  It decides with SIGN BITS only (no sqrt, division or
//    compare) and it never drops a hit the reference would report:
//    circle, in f32 with fused ops, in coordinates relative to the frame centre m:
//      u = o - m, v = c - m (|u_i|, |v_i| <= B).  The reference reports no hit when its
//      delta / 4 = D = (d.oc)^2 - |oc|^2 + r^2 is negative.  For a unit direction
//      |oc|^2 - (d.oc)^2 is the squared cross product, so D = r^2 - X^2 with
//        X = d x (u - v) = (dx uy - dy ux) + dy vx - dx vy,
//      the first term per ray, and a circle pair costs three packed operations
//      (two fused steps for X, one for g = pad + r^2 - X^2) and one v_alignbit each.
//      With u, v, d rounded to f32 (eps = 2^-24, |d_i| <= 1): the per-ray term, the two
//      fused steps and the rounded inputs put the computed X within 10.7 eps B of the
//      true one, and |X| <= 2.83 B, so X_f^2 <= X^2 + 60.8 eps B^2.  A direction with
//      | |d|^2 - 1 | <= 2^-40 moves D by at most 2^-37 B^2, and the reference's own f64
//      rounding of delta by less than 2^-48 (B^2 + r^2).  Hence pad = 2^-22 r^2 + 64 eps B^2
//      makes g >= 0 whenever the reference's delta >= 0 (the final rounding keeps the
//      sign).  Circles behind the ray pass this test (about half of the candidates on
//      a dense scene); the exact test drops them.  Testing for them here would cost
//      three more packed operations per pair on EVERY circle, which is more than the
//      extra candidates cost in phase 2.
//      Lanes whose origin or direction lies outside the frame mark every slot.
//    segment, f32 with fused ops, same frame (U = o - m, A = a - m, B = b - m, all within the frame's half width in
//      every component): the reference accepts iff t >= 0 and 0 <= u <= 1, u the position of the crossing along the
//      segment.  0 <= u <= 1 says that the end points lie on opposite sides of the ray's line (or on it):
//        ca = (A - U) x d = A x d - (U x d),   cb = (B - U) x d = B x d - (U x d),   Q = ca cb <= 0
//      (ca is the reference's `un`, cb = un - den), with (U x d) per ray: two fused steps per end point and one for
//      g = eq - ca cb, whose sign bit is the drop bit.  Absolute error of a computed c (eps = 2^-24, |d_i| <= 1.0005):
//      the rounded end point and direction 4.01 eps B, the rounded per-ray term 1.42 eps B, the two fused steps
//      (intermediate <= 2.42 B, result <= 2.84 B) 5.26 eps B - together 10.7 eps B; with Eu = 12 eps B and |c| <=
//      2.83 B (|A - U| <= 2 sqrt2 B) the computed product exceeds the true one by at most eq = 2 x 2.83 B Eu + Eu^2
//      (+1 %), a constant of the launch: no absolute values, no per-segment margins, and g >= 0 whenever the true
//      Q <= 0.  The reference's own f64 rounding of un and den (and its -0.0 from an underflowing quotient) is many
//      orders of magnitude inside that margin.  The margin passes a sliver more (a band of about eq / den^2 of the
//      segment's length past each end); segments BEHIND the ray pass as well, as circles behind it do - the exact test
//      drops both.  (A second product for t >= 0, tn den >= -ep with tn = (A - U) x s, dropped them here at the price of
//      13 instead of 7 instructions per pair on every segment: 21.9 against 20.0 ms per launch on `sample_world`.)
//      5 packed operations per pair and a v_alignbit per segment.
//  phase 2 (marked slots only, shared out over the wave's lanes, see nearest_hit):
//    the exact test in the reference's operation order, then (total_cmp key, pid)
//    lexicographic minimum = "first minimum in scan order" (librt2d_reverse.rs:139).
// Inactive lanes run phase 1 with the wave and mark nothing.
// Per-ray operands of both filters, each duplicated into both halves of a packed register.
struct FilterRay {
    f32x2 ndx, dy; // -d.x, d.y
    f32x2 kc;      // d x u = -(u x d): the per-ray term of the circles' X and of the segments' ca, cb alike
};

RT_DEV FilterRay filter_ray(SceneK S, double ox, double oy, double dx, double dy, bool &in_range)
{
    const double ux = ox - S->filt_mx, uy = oy - S->filt_my;
    in_range = __builtin_fabs(ux) <= S->filt_B && __builtin_fabs(uy) <= S->filt_B &&
               __builtin_fabs((dx * dx + dy * dy) - 1.0) <= 0x1.0p-40;
    const float dxf = (float)dx, dyf = (float)dy;
    const float kc = (float)(dx * uy - dy * ux);
    FilterRay f;
    f.ndx = f32x2{ -dxf, -dxf };
    f.dy = f32x2{ dyf, dyf };
    f.kc = f32x2{ kc, kc };
    return f;
}

// A word whose 32 slots are all circles: the same test with no bounds to compute, 16 pairs straight through.
RT_DEV uint32_t filter_circle_word(K_CirclePair cp, const FilterRay &f, uint32_t force)
{
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const f32x2 vx = cp[k].vx, vy = cp[k].vy, e = cp[k].e;
        const f32x2 x = __builtin_elementwise_fma(f.dy, vx, __builtin_elementwise_fma(f.ndx, vy, f.kc));
        const f32x2 g = __builtin_elementwise_fma(x, -x, e);
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.x), 31);
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.y), 31);
    }
    return ~acc | force;
}

// A word whose 32 slots are all segments, likewise.
RT_DEV uint32_t filter_segment_word(K_SegPair gp, const SegBounds &E, const FilterRay &f, uint32_t force)
{
    uint32_t acc = 0;
    const f32x2 eq = { E.eq, E.eq };
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const f32x2 ax = gp[k].ax, ay = gp[k].ay, bx = gp[k].bx, by = gp[k].by;
        const f32x2 ca = __builtin_elementwise_fma(ax, f.dy, __builtin_elementwise_fma(ay, f.ndx, f.kc));
        const f32x2 cb = __builtin_elementwise_fma(bx, f.dy, __builtin_elementwise_fma(by, f.ndx, f.kc));
        const f32x2 g = __builtin_elementwise_fma(-ca, cb, eq);
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.x), 31);
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.y), 31);
    }
    return ~acc | force;
}

template <bool WITH_SEGMENTS = true>
RT_DEV uint32_t filter_word(int n_circ, K_CirclePair c32, K_SegPair segp, const SegBounds &E, int lo, int hi,
                            const FilterRay &f, uint32_t force)
{
    uint32_t acc = 0;
    const int c_end = hi < n_circ ? hi : n_circ;
    // two circles per packed f32 instruction (v_pk_fma_f32): the circle count is padded to even
    // and word boundaries are even, so a pair never straddles a word
    K_CirclePair cp = c32 + (lo >> 1);
#pragma unroll 4
    for (int s = lo; s < c_end; s += 2, ++cp) {
        const f32x2 vx = cp->vx, vy = cp->vy, e = cp->e;
        const f32x2 x = __builtin_elementwise_fma(f.dy, vx, __builtin_elementwise_fma(f.ndx, vy, f.kc)); // d x (u - v)
        const f32x2 g = __builtin_elementwise_fma(x, -x, e);                                             // padded r^2 - X^2
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.x), 31); // collects the sign of g: set = drop
        acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.y), 31);
    }
    // two segments per packed instruction as well (the segment arrays are padded to an even count; the first
    // segment slot, n_circ, is even).  Drop bits again: the sign of g.
    const int s_lo = lo > n_circ ? lo : n_circ;
    const f32x2 eq = { E.eq, E.eq };
    auto pair = [&](K_SegPair gp) {
        const f32x2 ax = gp->ax, ay = gp->ay, bx = gp->bx, by = gp->by;
        const f32x2 ca = __builtin_elementwise_fma(ax, f.dy, __builtin_elementwise_fma(ay, f.ndx, f.kc)); // (A - U) x d
        const f32x2 cb = __builtin_elementwise_fma(bx, f.dy, __builtin_elementwise_fma(by, f.ndx, f.kc)); // (B - U) x d
        return __builtin_elementwise_fma(-ca, cb, eq); // < 0: the end points surely lie on one side of the ray's line
    };
    if (WITH_SEGMENTS) {
        int s = s_lo;
        K_SegPair gp = segp + ((s_lo - n_circ) >> 1);
#pragma unroll 2
        for (; s + 2 <= hi; s += 2, ++gp) {
            const f32x2 g = pair(gp);
            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.x), 31);
            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.y), 31);
        }
        if (s < hi) { // an odd segment count ends on half a pair: the other half is the padding element
            const f32x2 g = pair(gp);
            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(g.x), 31);
        }
    }
    // keep = not drop.  The ones this puts above the collected bits are shifted out below: exactly hi - lo bits
    // have been appended and the word is shifted up by 32 - (hi - lo).
    acc = ~acc;
    return (acc << (32 - (hi - lo))) | force;
}

// Per-wave LDS of nearest_hit's second phase.
struct CoopLds {
    int64_t key[64];      // running minimum total_cmp key of each lane's ray
    uint64_t pidslot[64]; // (pid << 32 | slot) of the lowest pid among the hits that have that key
    // queue of candidate words: mask (MSB = the word's first slot) and owner lane | word index within the 256-slot tile << 6
    // (two arrays of one element size: an entry's two cells then share one address computation)
    uint32_t ring_bits[128];
    uint32_t ring_info[128];
};

// Phase 2 is shared by the wave.  A lane's own candidates are few (1.8 on average on the stress scene)
// but a per-lane loop runs for the MAXIMUM over the wave (6.3 there) at ~30 % lane use.  Instead every
// non-empty 32-slot mask word becomes an entry (owner lane, word, bits) in an LDS ring, appended with one
// ballot + rank; whenever 64 entries are queued (and once more at the end) each lane takes ONE entry,
// fetches the owner's ray by cross-lane permute, runs the exact test on the entry's first slot and puts
// the rest of the entry, if any, back in the queue.  Results meet in the owner's LDS cell: an atomic
// minimum on the total_cmp key, then - among the lanes holding that key - an atomic minimum on
// (pid, slot); a lane that lowers the key clears the pid cell first (the two steps are separated by the
// wave's in-order LDS execution).  The outcome is the minimum over all candidates of (key, pid): the
// order of evaluation does not matter.  The ring never overflows: fewer than 64 entries are queued when
// a word (<= 64 entries) is appended, and a round puts back at most what it took.
// SEGW: the scene has words made of segments only, which then take their own straight-line filter (a kernel
// compiled without it is ~1 % faster on scenes that have none).
// SMALL: the scene has at most 32 slots (and at least one).
// NOSEG: the scene has no segments at all; the kernel then carries neither their filter nor their exact test (the
// stress scene is circles only, and the segment code it never runs still costs it registers: 21.65 -> 21.3 ms).
template <bool SEGW = true, bool SMALL = false, bool NOSEG = false>
RT_DEV Hit nearest_hit(SceneK S, double ox, double oy, double dx, double dy, bool active, CoopLds &L, Prof &prof)
{
    S = fresh(S);
    // read once per call: inside the loops below the compiler would fetch them again for every word
    const int n_circ = S->n_circ;
    const int P = NOSEG ? n_circ : n_circ + S->n_seg;
    const K_CirclePair c32 = (K_CirclePair)S->c32;
    const CircleX *cex = S->cex;
    const SegX *sex = S->sex;
    const K_SegPair segp = (K_SegPair)S->segp;
    const SegBounds E = { S->seg_bounds.eq, 0.0f };
    const int lane = lane_id();
    // outside the frame the f32 filter is proven for, every slot becomes a candidate (exact tests decide)
    bool in_range;
    const FilterRay fr = filter_ray(S, ox, oy, dx, dy, in_range);
    const uint32_t force = in_range ? 0u : 0xFFFFFFFFu;
    if (SMALL) {
        // The whole scene is one word: every lane's candidates sit in its own mask, a shared queue has nothing to
        // balance (the wave would walk max-over-lanes entries either way) and its LDS traffic and permutes are pure
        // cost.  Each lane walks its own bits and keeps the minimum of (key, pid) in registers.
        uint32_t mw = filter_word<!NOSEG>(n_circ, c32, segp, E, 0, P, fr, force) & (0xFFFFFFFFu << (32 - P));
        if (!active) mw = 0u;
        RT_STAMP(prof, 1);
#ifdef RT_PROF
        {
            int cnt = __popc(mw);
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
            prof.t[10] += (unsigned long long)cnt;
        }
#endif
        int64_t bk = 0x7fffffffffffffffLL;
        uint64_t bps = ~0ull;
        auto keep = [&](bool got, double t, int pid, int s) {
            const int64_t key = hit_key(t);
            const uint64_t ps = ((uint64_t)(uint32_t)pid << 32) | (uint64_t)(uint32_t)s;
            // selects, not a branch: around a branch the compiler copies both 64-bit values back and forth every round
            const bool better = got & ((key < bk) | ((key == bk) & (ps < bps)));
            bk = better ? key : bk;
            bps = better ? ps : bps;
        };
        // circles first, then segments: a step of the wave then runs one kind of test, not both
        uint32_t mc = (n_circ >= 32) ? mw : (mw & ~(0xFFFFFFFFu >> n_circ));
        uint32_t ms = mw & ~mc;
        while (__any(mc != 0u)) {
            const int s = __clz((int)mc) & 31;
            const bool on = mc != 0u;
            mc &= ~(0x80000000u >> s);
            bool got = false;
            double t = 0.0;
            int pid = 0;
            if (on) {
                const CircleX c = cex[s];
                got = circle_t(ox, oy, dx, dy, c.cx, c.cy, c.r2, t);
                pid = c.pid;
            }
            keep(got, t, pid, s);
#ifdef RT_PROF
            prof.t[11] += 1;
#endif
        }
        while (!NOSEG && __any(ms != 0u)) {
            const int s = __clz((int)ms) & 31;
            const bool on = ms != 0u;
            ms &= ~(0x80000000u >> s);
            bool got = false;
            double t = 0.0;
            int pid = 0;
            if (on) {
                const SegX g = sex[s - n_circ];
                asm volatile("" ::"v"(g.ax), "v"(g.ay), "v"(g.sx), "v"(g.sy)); // one wait for the whole record, see below
                got = segment_t(ox, oy, dx, dy, g.ax, g.ay, g.sx, g.sy, t);
                pid = g.pid;
            }
            keep(got, t, pid, s);
#ifdef RT_PROF
            prof.t[11] += 1;
#endif
        }
        RT_STAMP(prof, 2);
        Hit best;
        const bool found = bk != 0x7fffffffffffffffLL;
        best.t = found ? key_hit(bk) : 0.0;
        best.slot = found ? (int32_t)(uint32_t)bps : -1;
        best.pid = found ? (int32_t)(bps >> 32) : 0x7fffffff;
        return best;
    }
    L.key[lane] = 0x7fffffffffffffffLL;
    L.pidslot[lane] = ~0ull;
    int head = 0, count = 0; // wave-uniform
    for (int lo = 0; lo < P; lo += 32) {
        const int hi = (lo + 32 < P) ? lo + 32 : P;
        // bits below the word's last slot are shifted-in zeros; `force` may set them, so mask to the valid slots
        const uint32_t valid = 0xFFFFFFFFu << (32 - (hi - lo));
        uint32_t mw;
        if (hi - lo == 32 && hi <= n_circ) mw = filter_circle_word(c32 + (lo >> 1), fr, force);
        else if (SEGW && hi - lo == 32 && lo >= n_circ) mw = filter_segment_word(segp + ((lo - n_circ) >> 1), E, fr, force);
        else mw = filter_word<!NOSEG>(n_circ, c32, segp, E, lo, hi, fr, force) & valid;
        if (!active) mw = 0u;
        RT_STAMP(prof, 1);
#ifdef RT_PROF
        {
            int cnt = __popc(mw);
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
            prof.t[10] += (unsigned long long)cnt;
        }
#endif
        const uint64_t fresh = __ballot(mw != 0u);
#ifdef RT_PROF
        prof.t[16] += (unsigned long long)__popcll(fresh);
        prof.t[17] += (unsigned long long)__popcll(__ballot((mw & (mw - 1u)) != 0u));
#endif
        if (fresh != 0) {
            if (mw != 0u) {
                const int at = (head + count + rank_in_mask(fresh)) & 127;
                L.ring_bits[at] = mw;
                L.ring_info[at] = (uint32_t)(lane | ((lo & 255) << 1)); // (lo & 255) / 32 << 6
            }
            count += __popcll(fresh);
        }
        const bool last = hi == P || (hi & 255) == 0; // the queue is drained at the end of every 256-slot tile
        const int base = lo & ~255;
        while (count >= 64 || (last && count > 0)) {
            const int n = count < 64 ? count : 64;
            __builtin_amdgcn_wave_barrier(); // compiler-only: LDS executes a wave's instructions in order
            const bool has = lane < n;
            const uint32_t ebits = L.ring_bits[(head + lane) & 127];
            const uint32_t info = L.ring_info[(head + lane) & 127];
            head = (head + n) & 127;
            count -= n;
            uint32_t bits = has ? ebits : 0u;
            const int owner = (int)(info & 63u);
            const double rox = __shfl(ox, owner), roy = __shfl(oy, owner), rdx = __shfl(dx, owner), rdy = __shfl(dy, owner);
            // the last round of a tile has idle lanes and nothing to ride along with: there a lane walks through all
            // the slots of its entry; elsewhere it tests one and queues the rest again
            const bool drain = last && count == 0;
            do {
                const int b = __clz((int)bits) & 31;
                const bool on = bits != 0u;
                bits &= ~(0x80000000u >> b);
                const int s = base + (int)((info & ~63u) >> 1) + b;
                bool got = false;
                double t = 0.0;
                int pid = 0;
                if (on) {
                    if (NOSEG || s < n_circ) {
                        const CircleX c = cex[s];
                        got = circle_t(rox, roy, rdx, rdy, c.cx, c.cy, c.r2, t);
                        pid = c.pid;
                    } else {
                        const int k = s - n_circ;
                        const SegX g = sex[k];
                        // both halves of the record are fetched before the test starts: left to itself the compiler
                        // fetches a only behind the test of |d x s|, and a round then waits for memory twice
                        asm volatile("" ::"v"(g.ax), "v"(g.ay), "v"(g.sx), "v"(g.sy));
                        got = segment_t(rox, roy, rdx, rdy, g.ax, g.ay, g.sx, g.sy, t);
                        pid = g.pid;
                    }
                }
                const int64_t key = hit_key(t);
                if (got) {
                    const int64_t old = __hip_atomic_fetch_min(&L.key[owner], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (key < old) L.pidslot[owner] = ~0ull;
                }
                __builtin_amdgcn_wave_barrier();
                if (got && L.key[owner] == key)
                    __hip_atomic_fetch_min(&L.pidslot[owner], ((uint64_t)(uint32_t)pid << 32) | (uint64_t)(uint32_t)s, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_wave_barrier();
            } while (drain && __any(bits != 0u));
            const uint64_t again = __ballot(bits != 0u);
            if (again != 0) {
                if (bits != 0u) {
                    const int at = (head + count + rank_in_mask(again)) & 127;
                    L.ring_bits[at] = bits;
                    L.ring_info[at] = info;
                }
                count += __popcll(again);
            }
#ifdef RT_PROF
            prof.t[11] += 1;
#endif
        }
        RT_STAMP(prof, 2);
    }
    __builtin_amdgcn_wave_barrier();
    Hit best;
    const int64_t bk = L.key[lane];
    const uint64_t ps = L.pidslot[lane];
    const bool found = bk != 0x7fffffffffffffffLL;
    best.t = found ? key_hit(bk) : 0.0;
    best.slot = found ? (int32_t)(uint32_t)ps : -1;
    best.pid = found ? (int32_t)(ps >> 32) : 0x7fffffff;
    return best;
}

// ---------------------------------------------------------------------------
// Path state of one lane.
struct Path {
    double ox, oy, dx, dy;
    float thr;          // product of the factors collected so far (albedo, Beer-Lambert)
    int32_t li;         // wavelength bin
    int32_t depth_left; // the reference's `depth` argument
    uint32_t c0, c1, c2; // Philox counter words naming this path
};

enum ShadeResult { SHADE_CONTINUE = 0, SHADE_DONE = 1, SHADE_DIFFUSE = 2 };

// One trace_ray body after the nearest hit is known (librt2d_reverse.rs:146-261).
// Returns SHADE_DONE with `value` = the radiance this invocation returns to its
// caller (to be multiplied by the path's collected factors), or SHADE_CONTINUE
// with the path advanced to the recursive call's ray.
//
// The reference's side x material switch is regrouped so that a wave runs each
// expensive piece once: (1) classify, (2) the circle normal (sqrt + divide) for
// every lane that needs one, (3) one reflect/refract body shared by Reflective,
// Dielectric entry and Dielectric exit (they differ only in the normal's sign, eta
// and the offset).  A Diffuse bounce returns SHADE_DIFFUSE with the path moved to
// its new origin and the surface normal in (dnx, dny); its direction is then drawn
// by rand_hemisphere_wave, which the whole wave enters together.  Every value is
// computed with the reference's operations in the reference's order.
// FORWARD: the forward tracer's trace_ray (librt2d_forward.rs:66-195) takes the same steps with the index of
// refraction evaluated at its fixed `lambda = 1.`.
template <bool FORWARD = false>
RT_DEV ShadeResult shade(SceneK S, Path &p, const Hit &h, float &value, double &dnx, double &dny, Prof &prof)
{
    value = 0.0f;
    if (h.slot < 0) return SHADE_DONE;
    S = fresh(S);
    const bool is_circle = h.slot < S->n_circ;
    const int k = is_circle ? h.slot : h.slot - S->n_circ;
    const double hx = p.ox + h.t * p.dx, hy = p.oy + h.t * p.dy; // ray.origin + t * ray.dir
    double nx = 0.0, ny = 0.0, cx = 0.0, cy = 0.0;
    bool outside;
    const MatDev m = S->mats[h.slot];
    if (is_circle) {
        const CircleX c = S->cex[k];
        cx = c.cx;
        cy = c.cy;
        const double ex = cx - p.ox, ey = cy - p.oy; // is_inside(ray.origin), rt_common.rs:75-77
        outside = !((ex * ex + ey * ey) < c.r2);
    } else {
        const SegX *g = S->sex + k;
        nx = g->nx;
        ny = g->ny;
        outside = (nx * p.dx + ny * p.dy) < 0.0; // rt_common.rs:262
    }
    const float sv = S->spectra[m.spectrum * NBINS + p.li];

    // (1) classify
    enum { TERM, DIRECTIONAL, DIFFUSE, BEND };
    int act = TERM;
    if (m.kind == MAT_EMISSIVE) value = sv;                       // either side (:148, :225)
    else if (m.kind == MAT_DIELECTRIC) act = BEND;                // either side (:149-160, :246-257)
    else if (outside && m.kind == MAT_REFLECTIVE) act = BEND;     // :238-245
    else if (outside && m.kind == MAT_DIFFUSE && !m.black) act = DIFFUSE; // :211-224
    else if (outside && m.kind == MAT_DIRECTIONAL) act = DIRECTIONAL;     // :226-237
    if (act == TERM) return SHADE_DONE; // emitted, or one of the reference's `_ => 0` arms
    RT_STAMP(prof, 7);

    // (2) Circle::hit's normal: (p - center).normalize(), always outward (rt_common.rs:103)
    if (is_circle) {
        const double vx = hx - cx, vy = hy - cy;
        const double rl = rlen2(vx, vy);
        nx = vx * rl;
        ny = vy * rl;
    }

    RT_STAMP(prof, 8);
    if (act == DIRECTIONAL) {
        if (p.dx * -nx + p.dy * -ny >= m.angle) {
            const double coeff = 1.0 / (h.t + m.d);
            value = sv * (float)coeff;
        }
        return SHADE_DONE;
    }

    double rx, ry, qx, qy;
    if (act == BEND) {
        const bool diel = m.kind == MAT_DIELECTRIC;
        const bool entering = diel && outside;
        // new origin: p - n*1e5*eps when entering glass (:248), p + n*1e4*eps otherwise (:151, :240)
        const double kofs = entering ? 100000.0 : 10000.0;
        const double tx = nx * kofs * EPS64, ty = ny * kofs * EPS64;
        qx = entering ? hx - tx : hx + tx;
        qy = entering ? hy - ty : hy + ty;
        // the normal the reference passes to refract / reflect: -n from inside, n from outside
        const double Nx = outside ? nx : -nx, Ny = outside ? ny : -ny;
        const double ndi = Nx * p.dx + Ny * p.dy;
        bool bounce = true;
        if (diel) {
            // glam refract(dir, N, eta); eta = ior leaving (:150-152), 1/ior entering (:247-249)
            const double eta = FORWARD ? (outside ? S->inv_ior_fwd[m.ior_row] : S->ior_fwd[m.ior_row])
                                       : (outside ? S->inv_ior[m.ior_row * NBINS + p.li] : S->ior[m.ior_row * NBINS + p.li]);
            const double kk = 1.0 - eta * eta * (1.0 - ndi * ndi);
            if (kk >= 0.0) {
                const double mm = eta * ndi + sqrt_exact(kk);
                rx = eta * p.dx - mm * Nx;
                ry = eta * p.dy - mm * Ny;
                bounce = (rx == 0.0 && ry == 0.0); // `refracted_ray == DVec2::ZERO`
            }
            if (!outside) p.thr = p.thr * expf_fixed((float)(-h.t) * sv); // Beer-Lambert on the way out (:159)
        }
        if (bounce) {
            // glam reflect: dir - 2 * dir.dot(N) * N   (dir.dot(N) has the bits of N.dot(dir))
            const double k2 = 2.0 * ndi;
            rx = p.dx - k2 * Nx;
            ry = p.dy - k2 * Ny;
        }
        RT_STAMP(prof, 9);
    } else {
        // Diffuse (:211-224): new origin p + n*1e4*eps, factor `absorption[lambda]`; direction drawn by the wave.
        // A normal that is the zero vector (a caller's segment normal) or NaN (a hit point that rounds onto a
        // circle's centre: the circle is smaller than the spacing of f64 at its coordinates) has a positive product
        // with no direction: the reference's rejection loop (rt_common.rs:376-391) never returns for it, and neither
        // would the wave's.  Such a path ends here and contributes nothing.
        if (!(__builtin_fabs(nx) + __builtin_fabs(ny) > 0.0)) return SHADE_DONE;
        p.ox = hx + nx * 10000.0 * EPS64;
        p.oy = hy + ny * 10000.0 * EPS64;
        p.thr = p.thr * sv;
        p.depth_left -= 1;
        dnx = nx;
        dny = ny;
        return SHADE_DIFFUSE;
    }
    p.ox = qx; p.oy = qy; p.dx = rx; p.dy = ry;
    p.depth_left -= 1;
    return SHADE_CONTINUE;
}

// Ray2d::rand_hemisphere (rt_common.rs:376-391) for every lane with `need`, entered by the whole wave.
//
// The reference loops per ray: draw v in [-1,1)^2, keep it if |v|^2 <= 1 and v.normalize().dot(n) > 0.
// Attempt k of a path takes its two numbers from Philox(counter = path words, bounce << 20 | k), so ANY
// lane can evaluate it.  With nd lanes needing a direction, each gets H = 64 / nd helper lanes that test
// attempts base .. base+H-1 at once; the lowest accepted attempt wins (the same vector the sequential
// loop would stop at) and is shuffled back.  A handful of diffuse lanes would otherwise hold the wave
// for the maximum of their geometric trial counts (~6 rounds at 39 % acceptance) at < 10 % lane use.
//
// Acceptance without the square root: `v.normalize().dot(n) > 0` has the sign of the unnormalised
// a + b = v.x n.x + v.y n.y whenever |a + b| > 2^-40 (|a| + |b|) (the normalised sum differs from
// (a + b)/|v| by less than 2^-50 (|a| + |b|)/|v|); closer calls evaluate the reference's own expression.
// The accepted vector is normalised once, as the reference does: v * (1 / sqrt(v.v)).
// HELPERS[nd] = H | M << 8 with H = 64 / nd helper lanes per direction and M = ceil(2^16 / H), so that
// lane / H = lane * M >> 16 for lane < 64 (runtime integer division costs ~25 instructions).
__constant__ const uint32_t HELPERS[65] = { 0x0, 0x40040, 0x80020, 0xC3115, 0x100010, 0x15560C, 0x199A0A, 0x1C7209, 0x200008, 0x249307, 0x2AAB06, 0x333405, 0x333405, 0x400004, 0x400004, 0x400004, 0x400004, 0x555603, 0x555603, 0x555603, 0x555603, 0x555603, 0x800002, 0x800002, 0x800002, 0x800002, 0x800002, 0x800002, 0x800002, 0x800002, 0x800002, 0x800002, 0x800002, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001, 0x1000001 };

RT_DEV void rand_hemisphere_wave(bool need, double nx, double ny, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t bounce,
                                 uint32_t k0, uint32_t k1, double &rx, double &ry, Prof &prof)
{
    uint64_t pending = __ballot(need);
    if (pending == 0) return;
#ifdef RT_PROF
    prof.t[15] += (unsigned long long)__popcll(pending);
#endif
    const int lane = lane_id();
    bool still = need;
    uint32_t base = 0;
    double ovx = 0.0, ovy = 0.0, ol2 = 1.0;
    while (pending != 0) {
#ifdef RT_PROF
        prof.t[14] += 1;
#endif
        const int nd = __popcll(pending);
        const int H = HELPERS[nd] & 0xFF; // 64 / nd >= 1, wave-uniform
        int tl = lane, toff = 0;
        bool helper = still;
        if (H >= 2) {
            const int trank = (lane * (int)(HELPERS[nd] >> 8)) >> 16; // lane / H
            toff = lane - trank * H;
            helper = trank < nd;
            // the trank-th set bit of `pending` is the lane this one helps: every pending lane sends its id to
            // the lane whose index is its rank (forward permute), and helpers fetch it from lane trank
            // (the other lanes send to lane 63, which nobody reads: nd <= 32 here)
            const int sent = __builtin_amdgcn_ds_permute((still ? rank_in_mask(pending) : 63) << 2, lane);
            tl = __builtin_amdgcn_ds_bpermute((trank < nd ? trank : 0) << 2, sent);
        }
        const uint32_t tc0 = __shfl(c0, tl), tc1 = __shfl(c1, tl), tc2 = __shfl(c2, tl);
        const uint32_t tb = __shfl(bounce, tl), tbase = __shfl(base, tl);
        const double tnx = __shfl(nx, tl), tny = __shfl(ny, tl);
        const uint32_t attempt = tbase + (uint32_t)toff;
        const U4 w = philox4x32_10(tc0, tc1, tc2, (tb << 20) | (attempt & 0xFFFFFu), k0, k1);
        const double vx = u01(w.x, w.y) * 2.0 - 1.0;
        const double vy = u01(w.z, w.w) * 2.0 - 1.0;
        const double l2 = vx * vx + vy * vy;
        bool ok = false;
        if (helper && l2 <= 1.0) {
            const double a = vx * tnx, b = vy * tny;
            const double du = a + b;
            if (__builtin_fabs(du) > 0x1.0p-40 * (__builtin_fabs(a) + __builtin_fabs(b))) {
                ok = du > 0.0;
            } else {
                const double rl = rsqrt_exact(l2);
                ok = (vx * rl) * tnx + (vy * rl) * tny > 0.0;
            }
        }
        const uint64_t okm = __ballot(ok);
        // a pending lane reads the verdicts of its own helpers: bits [rank * H, rank * H + H)
        const int seg_lo = (H >= 2) ? rank_in_mask(pending) * H : lane;
        const uint64_t seg_mask = (H >= 64) ? ~0ull : ((1ull << H) - 1ull);
        const uint64_t seg = still ? ((okm >> seg_lo) & seg_mask) : 0ull;
        const int winner = seg ? seg_lo + (__ffsll((long long)seg) - 1) : lane;
        const double wvx = __shfl(vx, winner), wvy = __shfl(vy, winner), wl2 = __shfl(l2, winner);
        // selects, not branches: around a branch the compiler keeps a second copy of the three doubles and moves them
        // back every round
        const bool won = seg != 0ull;
        ovx = won ? wvx : ovx;
        ovy = won ? wvy : ovy;
        ol2 = won ? wl2 : ol2;
        still = still & !won;
        base += still ? (uint32_t)H : 0u;
        pending = __ballot(still);
    }
    if (need) {
        const double rl = rsqrt_exact(ol2);
        rx = ovx * rl;
        ry = ovy * rl;
    }
}

// Wavelength bin of lambda-sample l of (pixel, global sample): word l + 2 of the sample's Philox stream
// (spectrum.rs:185-191 draws it with rand::random).
RT_DEV int lambda_bin(uint32_t pix, uint32_t gsample, uint32_t l, uint32_t k0, uint32_t k1)
{
    const uint32_t call = 1u + (l + 2u) / 4u, word = (l + 2u) % 4u;
    const U4 wl = philox4x32_10(pix, gsample, 0xFFFFFFFFu, call, k0, k1);
    const uint32_t r = (word == 0u) ? wl.x : (word == 1u) ? wl.y : (word == 2u) ? wl.z : wl.w;
    return (int)(((uint64_t)r * (uint64_t)NBINS) >> 32);
}

// Wavelength bins of four lambda-samples of (pixel, global sample), one byte each: block q holds lambda-samples
// 4 q - 2 .. 4 q + 1 in bytes 0 .. 3 (lambda-sample l is word (l + 2) % 4 of call 1 + (l + 2) / 4, as lambda_bin).
RT_DEV uint32_t lambda_bins4(uint32_t pix, uint32_t gsample, uint32_t q, uint32_t k0, uint32_t k1)
{
    const U4 wl = philox4x32_10(pix, gsample, 0xFFFFFFFFu, 1u + q, k0, k1);
    auto bin = [](uint32_t r) { return (uint32_t)(((uint64_t)r * (uint64_t)NBINS) >> 32); };
    return bin(wl.x) | (bin(wl.y) << 8) | (bin(wl.z) << 16) | (bin(wl.w) << 24);
}

// Primary ray of (pixel, global sample) (librt2d_reverse.rs:88-94) - the reference draws ONE ray per sample and
// traces it once per lambda-sample (:90-101) - plus the wavelength bins of its first two lambda-samples in bytes
// 2 and 3 of `bins` (block 0 of lambda_bins4: bytes 0 and 1 belong to the direction angle).  Philox counter
// (pixel, sample, 0xFFFFFFFF, call): call 0 = pixel jitter, call 1 = direction angle (words 0, 1) and the first two
// wavelength draws, calls 2.. = four wavelength draws each.
RT_DEV void generate_primary(double &ox, double &oy, double &dx, double &dy, uint32_t &bins, int px, int py, int width,
                             uint32_t gsample, double px_to_world, uint32_t k0, uint32_t k1)
{
    const uint32_t pix = (uint32_t)(width * py + px);
    const U4 w = philox4x32_10(pix, gsample, 0xFFFFFFFFu, 0u, k0, k1);
    const U4 w1 = philox4x32_10(pix, gsample, 0xFFFFFFFFu, 1u, k0, k1);
    const double x = ((double)px + u01(w.x, w.y)) - 0.5;
    const double y = ((double)py + u01(w.z, w.w)) - 0.5;
    ox = x * px_to_world;
    oy = y * px_to_world;
    sincos2pi(u01(w1.x, w1.y), dy, dx);
    auto bin = [](uint32_t r) { return (uint32_t)(((uint64_t)r * (uint64_t)NBINS) >> 32); };
    bins = (bin(w1.z) << 16) | (bin(w1.w) << 24);
}

} // namespace rt2d
