// Device-side arithmetic shared by the 2d and 3d trace kernels (gfx950).
//
// Everything here is built from IEEE + - * / sqrt only and must be compiled
// with -ffp-contract=off: the reference (Rust) never fuses a*b+c, and hit
// primitive ids / bounce counts are only reproducible bit for bit when every
// f64 operation rounds the same way on the host restatement and on the device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

namespace rtd {

constexpr int NBINS = 95;              // SPECTRUM_SAMPLES, 2d/src/spectrum.rs:106
constexpr double EPS64 = 0x1.0p-52;    // f64::EPSILON

// ---------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. 2011).  Stands in for
// rand::random (2d/src/librt2d_reverse.rs:89, rt_common.rs:370,379-380,
// spectrum.rs:186, 3d/src/rt.rs:41,181): the key is the caller's seed, the
// counter names (pixel | ray, sample, lambda-sample, bounce/attempt), so a
// number never depends on which lane or launch asks for it.
struct U4 { uint32_t x, y, z, w; };

RT_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0; // one v_mad_u64_u32 each
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // three-way xor in one v_bitop3_b32 (truth table 0x96)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{ c0, c1, c2, c3 };
}

// 53-bit uniform in [0,1): ((hi:lo) >> 11) * 2^-53, as rand's f64 sampling does.
RT_DEV double u01(uint32_t lo, uint32_t hi)
{
    // hi * 2^-32 + (lo >> 11) * 2^-53: both conversions, the product and the sum are exact
    return __builtin_fma((double)hi, 0x1.0p-32, (double)(lo >> 11) * 0x1.0p-53);
}

// ---------------------------------------------------------------------------
// (sin, cos)(2*pi*u), u in [0,1): quadrant reduction + degree 15/16 Taylor
// polynomials on [-pi/4, pi/4], Horner form, no fused operations.  Fixed
// evaluation order so host and device agree to the last bit (libm / ocml
// sin/cos do not).  Used where the reference calls DVec2::from_angle
// (2d/src/rt_common.rs:371).
RT_DEV void sincos2pi(double u, double &s, double &c)
{
    const double x = u * 4.0;
    const int q = (int)(x + 0.5);
    const double r = x - (double)q;
    const double th = r * 0x1.921fb54442d18p+0;
    const double z = th * th;
    double ps = -0x1.ae7f3e733b81fp-41;
    ps = ps * z + 0x1.6124613a86d09p-33;
    ps = ps * z + -0x1.ae64567f544e4p-26;
    ps = ps * z + 0x1.71de3a556c734p-19;
    ps = ps * z + -0x1.a01a01a01a01ap-13;
    ps = ps * z + 0x1.1111111111111p-7;
    ps = ps * z + -0x1.5555555555555p-3;
    const double sn = th + (th * z) * ps;
    double pc = 0x1.ae7f3e733b81fp-45;
    pc = pc * z + -0x1.93974a8c07c9dp-37;
    pc = pc * z + 0x1.1eed8eff8d898p-29;
    pc = pc * z + -0x1.27e4fb7789f5cp-22;
    pc = pc * z + 0x1.a01a01a01a01ap-16;
    pc = pc * z + -0x1.6c16c16c16c17p-10;
    pc = pc * z + 0x1.5555555555555p-5;
    const double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    const int k = q & 3;
    s = (k == 0) ? sn : (k == 1) ? cs : (k == 2) ? -sn : -cs;
    c = (k == 0) ? cs : (k == 1) ? -sn : (k == 2) ? -cs : sn;
}

// e^x in f32 with a fixed evaluation order (f32::exp at
// 2d/src/librt2d_reverse.rs:159 is platform libm): n = round(x log2 e),
// two-constant ln2 reduction, degree-7 polynomial, scaling by 2^n.
RT_DEV float expf_fixed(float x)
{
    if (x != x) return x;
    if (x > 88.72283f) return __builtin_inff();
    if (x < -103.97f) return 0.0f;
    const float fn = x * 0x1.715476p+0f;
    int n = (int)(fn + (fn < 0.0f ? -0.5f : 0.5f));
    const float nf = (float)n;
    float r = x - nf * 0x1.63p-1f;
    r = r - nf * -0x1.bd0106p-13f;
    float p = 0x1.a01a02p-13f;
    p = p * r + 0x1.6c16c2p-10f;
    p = p * r + 0x1.111112p-7f;
    p = p * r + 0x1.555556p-5f;
    p = p * r + 0x1.555556p-3f;
    p = p * r + 0.5f;
    float e = 1.0f + (r + (r * r) * p);
    if (n < -125) { e = e * 0x1.0p-64f; n += 64; }
    if (n > 127) { e = e * 2.0f; n -= 1; }
    return e * __uint_as_float((uint32_t)(n + 127) << 23);
}

// The key the hit searches compare distances by: its signed integer order must be f64::total_cmp's (the comparison
// min_by / sort_by use at librt2d_reverse.rs:139, rt_common.rs:302, 3d/src/rt.rs:224), for which -0.0 sorts before
// +0.0.  Every distance a hit test accepts is positive, +0 or -0 (`t0 > 0.`, `t1 > 0.` for circles, rt_common.rs:90-98;
// `t >= 0.`, which holds for -0.0, for segments and spheres, rt_common.rs:258, 3d/src/rt.rs:96-110) - never NaN, never
// below zero - and on those the plain bit pattern, read as a signed integer, already has that order: -0.0 is the most
// negative integer, +0.0 is zero, positive values ascend.  (The general key flips the low 63 bits of negative values;
// five more instructions per test that would change nothing here.)
RT_DEV int64_t hit_key(double t) { return (int64_t)__double_as_longlong(t); }
RT_DEV double key_hit(int64_t k) { return __longlong_as_double(k); }

RT_DEV int hi32(double x) { return __double2hiint(x); }

// fl(1 / fl(sqrt(x))): the two IEEE operations behind glam's `normalize` (v * length().recip()), bit for bit, in 16
// instructions where the compiler's generic expansions of sqrt and of the division take 30.
//   s:  the compiler's own square-root sequence (v_rsq_f64, one coupled refinement of g ~ sqrt(x) and h ~ 1 / (2 sqrt(x)),
//       two residual corrections) without the scaling it wraps around it for arguments near the ends of the exponent
//       range - for x in [2^-500, 2^500] nothing here comes near them, and the scaling is by powers of two, so s has
//       the bits the generic form produces.
//   1 / s: y = 2 h is already 1 / s to about 2^-49; y <- y + y (1 - s y) once brings it within half a unit in the last
//       place (+ 2^-98 relative), and once more gives the correctly rounded reciprocal.  Why: scale s into [1, 2) and
//       y into (1/2, 1]; with |y - 1/s| <= 2^-54 (1 + 2^-43) the residual e = 1 - s y is a multiple of 2^-105 below
//       2^-53 in size, so the fused step forms it exactly, and the value rounded by the last step is (1/s)(1 - e^2).
//       1/s = 2^52 / S (S the integer significand) lies at least 1 / (S 2^54) from every midpoint (2K + 1) 2^-54 of two
//       neighbouring doubles, because 2^106 - (2K + 1) S is a non-zero integer; (1/s) e^2 <= (2^52 / S) s^2 2^-108
//       (1 + 2^-43) stays below that distance whenever s^2 (1 + 2^-43) < 4: for every s but the few next to 2.  Those
//       (the top 20 fraction bits all ones: one value in a million) and arguments outside [2^-500, 2^500] - zero,
//       infinities and NaN among them - take the generic operations.
// tools/ubench/rsqrt_check.hip compares this and sqrt_exact below with the generic operations on the GPU, bit for bit,
// on five classes of arguments (random in the tracer's range, random bits incl. all specials, significands of mostly
// ones or zeros, exact squares and their neighbours, square roots next to 2); tests/test_gpu_math.py runs it.
RT_DEV double rsqrt_exact(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0, h = 0.5 * y0;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    const double s = __builtin_fma(d, h, g);
    double y = h + h;
    double e = __builtin_fma(-s, y, 1.0);
    y = __builtin_fma(e, y, y);
    e = __builtin_fma(-s, y, 1.0);
    y = __builtin_fma(e, y, y);
    // x in [2^-500, 2^500): one unsigned comparison on its high word (sign and exponent; NaN, infinities, zero,
    // denormals and negatives all fall outside); the top 20 fraction bits of s not all ones: another
    const bool plain = (uint32_t)__double2hiint(x) - 0x20B00000u < 0x5F400000u - 0x20B00000u &&
                       ((uint32_t)__double2hiint(s) & 0xFFFFFu) != 0xFFFFFu;
    if (__builtin_expect(!plain, 0)) y = 1.0 / __builtin_sqrt(x);
    return y;
}

// fl(sqrt(x)) alone, for the discriminants of the exact hit tests: the same sequence without the compiler's scaling
// (10 instructions against 17); arguments outside [2^-500, 2^500) - zero among them - take the generic operation.
RT_DEV double sqrt_exact(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0, h = 0.5 * y0;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    double s = __builtin_fma(d, h, g);
    const bool plain = (uint32_t)__double2hiint(x) - 0x20B00000u < 0x5F400000u - 0x20B00000u;
    if (__builtin_expect(!plain, 0)) s = __builtin_sqrt(x);
    return s;
}

// fl(x / 100.0f), the division in `get_background_color(..) / 100.` (3d/src/rt.rs:234), in three instructions where the
// generic f32 division takes ten: q0 = x * fl(1/100), the residual x - 100 q0 (exact in a fused step), q0 + residual *
// fl(1/100).  Same bits for EVERY float whose exponent field is 3 .. 254 (2^-124 <= |x| < inf) and for +0: checked by
// exhaustion over all 2^32 bit patterns on the GPU (rt_selftest_math, class 5; the only patterns that differ are
// -0, the denormals and the two binades above them - whose quotients are denormal - and the infinities).  The caller's
// arguments are sky colours in {+0} and [2^-24, 1].
RT_DEV float div_by_100(float x)
{
    const float r = 0.01f; // fl(1 / 100) = 0x3C23D70A
    const float q0 = x * r;
    const float rem = __builtin_fmaf(-q0, 100.0f, x);
    return __builtin_fmaf(rem, r, q0);
}

// 1/len with len = sqrt(x.x + y.y): glam's normalize multiplies by the reciprocal.
RT_DEV double rlen2(double x, double y) { return rsqrt_exact(x * x + y * y); }
RT_DEV double rlen3(double x, double y, double z) { return rsqrt_exact((x * x + y * y) + z * z); }

// Diagnostic build only (-DRT_PROF): per-wave cycle shares of the trace loop's
// sections, stamped with s_memtime.  Never compiled into the shipped library.
#ifdef RT_PROF
struct Prof {
    unsigned long long t[20];
    unsigned long long last;
    unsigned long long *octants; // 3d: [8][4] rays, rays that go beyond the root, their node visits and the squares of those
};
#define RT_STAMP(prof, i)                                                              \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                            \
        (prof).t[i] += now_ - (prof).last;                                             \
        (prof).last = now_;                                                            \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
#else
struct Prof {};
#define RT_STAMP(prof, i) do { } while (0)
#endif

// wave64 helpers
RT_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
RT_DEV int rank_in_mask(uint64_t mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

} // namespace rtd
