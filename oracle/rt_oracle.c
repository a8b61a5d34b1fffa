/*
 * rt_oracle.c — CPU restatement of tatref/raytracer's per-ray hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see rt_oracle.h for both notes).
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference root).  The code keeps the reference's operation order; it must be
 * built with -ffp-contract=off and without -ffast-math.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* CIE 1931 2-degree colour matching functions, 360..830 nm step 5 nm.
 * Data table of 2d/src/spectrum.rs:8-104 (values as f32).                     */
static const float CMF[RTO_NBINS * 3] = {
    0.0001299f, 0.000003917f, 0.0006061f,   /* 360 nm */
    0.0002321f, 0.000006965f, 0.001086f,   /* 365 nm */
    0.0004149f, 0.00001239f, 0.001946f,   /* 370 nm */
    0.0007416f, 0.00002202f, 0.003486f,   /* 375 nm */
    0.001368f, 0.000039f, 0.006450001f,   /* 380 nm */
    0.002236f, 0.000064f, 0.01054999f,   /* 385 nm */
    0.004243f, 0.00012f, 0.02005001f,   /* 390 nm */
    0.00765f, 0.000217f, 0.03621f,   /* 395 nm */
    0.01431f, 0.000396f, 0.06785001f,   /* 400 nm */
    0.02319f, 0.00064f, 0.1102f,   /* 405 nm */
    0.04351f, 0.00121f, 0.2074f,   /* 410 nm */
    0.07763f, 0.00218f, 0.3713f,   /* 415 nm */
    0.13438f, 0.004f, 0.6456f,   /* 420 nm */
    0.21477f, 0.0073f, 1.0390501f,   /* 425 nm */
    0.2839f, 0.0116f, 1.3856f,   /* 430 nm */
    0.3285f, 0.01684f, 1.62296f,   /* 435 nm */
    0.34828f, 0.023f, 1.74706f,   /* 440 nm */
    0.34806f, 0.0298f, 1.7826f,   /* 445 nm */
    0.3362f, 0.038f, 1.77211f,   /* 450 nm */
    0.3187f, 0.048f, 1.7441f,   /* 455 nm */
    0.2908f, 0.06f, 1.6692f,   /* 460 nm */
    0.2511f, 0.0739f, 1.5281f,   /* 465 nm */
    0.19536f, 0.09098f, 1.28764f,   /* 470 nm */
    0.1421f, 0.1126f, 1.0419f,   /* 475 nm */
    0.09564f, 0.13902f, 0.8129501f,   /* 480 nm */
    0.05795001f, 0.1693f, 0.6162f,   /* 485 nm */
    0.03201f, 0.20802f, 0.46518f,   /* 490 nm */
    0.0147f, 0.2586f, 0.3533f,   /* 495 nm */
    0.0049f, 0.323f, 0.272f,   /* 500 nm */
    0.0024f, 0.4073f, 0.2123f,   /* 505 nm */
    0.0093f, 0.503f, 0.1582f,   /* 510 nm */
    0.0291f, 0.6082f, 0.1117f,   /* 515 nm */
    0.06327f, 0.71f, 0.07824999f,   /* 520 nm */
    0.1096f, 0.7932f, 0.05725001f,   /* 525 nm */
    0.1655f, 0.862f, 0.04216f,   /* 530 nm */
    0.2257499f, 0.9148501f, 0.02984f,   /* 535 nm */
    0.2904f, 0.954f, 0.0203f,   /* 540 nm */
    0.3597f, 0.9803f, 0.0134f,   /* 545 nm */
    0.4334499f, 0.9949501f, 0.008749999f,   /* 550 nm */
    0.5120501f, 1.0f, 0.005749999f,   /* 555 nm */
    0.5945f, 0.995f, 0.0039f,   /* 560 nm */
    0.6784f, 0.9786f, 0.002749999f,   /* 565 nm */
    0.7621f, 0.952f, 0.0021f,   /* 570 nm */
    0.8425f, 0.9154f, 0.0018f,   /* 575 nm */
    0.9163f, 0.87f, 0.001650001f,   /* 580 nm */
    0.9786f, 0.8163f, 0.0014f,   /* 585 nm */
    1.0263f, 0.757f, 0.0011f,   /* 590 nm */
    1.0567f, 0.6949f, 0.001f,   /* 595 nm */
    1.0622f, 0.631f, 0.0008f,   /* 600 nm */
    1.0456f, 0.5668f, 0.0006f,   /* 605 nm */
    1.0026f, 0.503f, 0.00034f,   /* 610 nm */
    0.9384f, 0.4412f, 0.00024f,   /* 615 nm */
    0.8544499f, 0.381f, 0.00019f,   /* 620 nm */
    0.7514f, 0.321f, 0.0001f,   /* 625 nm */
    0.6424f, 0.265f, 0.00004999999f,   /* 630 nm */
    0.5419f, 0.217f, 0.00003f,   /* 635 nm */
    0.4479f, 0.175f, 0.00002f,   /* 640 nm */
    0.3608f, 0.1382f, 0.00001f,   /* 645 nm */
    0.2835f, 0.107f, 0.0f,   /* 650 nm */
    0.2187f, 0.0816f, 0.0f,   /* 655 nm */
    0.1649f, 0.061f, 0.0f,   /* 660 nm */
    0.1212f, 0.04458f, 0.0f,   /* 665 nm */
    0.0874f, 0.032f, 0.0f,   /* 670 nm */
    0.0636f, 0.0232f, 0.0f,   /* 675 nm */
    0.04677f, 0.017f, 0.0f,   /* 680 nm */
    0.0329f, 0.01192f, 0.0f,   /* 685 nm */
    0.0227f, 0.00821f, 0.0f,   /* 690 nm */
    0.01584f, 0.005723f, 0.0f,   /* 695 nm */
    0.01135916f, 0.004102f, 0.0f,   /* 700 nm */
    0.008110916f, 0.002929f, 0.0f,   /* 705 nm */
    0.005790346f, 0.002091f, 0.0f,   /* 710 nm */
    0.004109457f, 0.001484f, 0.0f,   /* 715 nm */
    0.002899327f, 0.001047f, 0.0f,   /* 720 nm */
    0.00204919f, 0.00074f, 0.0f,   /* 725 nm */
    0.001439971f, 0.00052f, 0.0f,   /* 730 nm */
    0.0009999493f, 0.0003611f, 0.0f,   /* 735 nm */
    0.0006900786f, 0.0002492f, 0.0f,   /* 740 nm */
    0.0004760213f, 0.0001719f, 0.0f,   /* 745 nm */
    0.0003323011f, 0.00012f, 0.0f,   /* 750 nm */
    0.0002348261f, 0.0000848f, 0.0f,   /* 755 nm */
    0.0001661505f, 0.00006f, 0.0f,   /* 760 nm */
    0.000117413f, 0.0000424f, 0.0f,   /* 765 nm */
    0.00008307527f, 0.00003f, 0.0f,   /* 770 nm */
    0.00005870652f, 0.0000212f, 0.0f,   /* 775 nm */
    0.00004150994f, 0.00001499f, 0.0f,   /* 780 nm */
    0.00002935326f, 0.0000106f, 0.0f,   /* 785 nm */
    0.00002067383f, 0.0000074657f, 0.0f,   /* 790 nm */
    0.00001455977f, 0.0000052578f, 0.0f,   /* 795 nm */
    0.00001025398f, 0.0000037029f, 0.0f,   /* 800 nm */
    0.000007221456f, 0.0000026078f, 0.0f,   /* 805 nm */
    0.000005085868f, 0.0000018366f, 0.0f,   /* 810 nm */
    0.000003581652f, 0.0000012934f, 0.0f,   /* 815 nm */
    0.000002522525f, 0.00000091093f, 0.0f,   /* 820 nm */
    0.000001776509f, 0.00000064153f, 0.0f,   /* 825 nm */
    0.000001251141f, 0.00000045181f, 0.0f,   /* 830 nm */
};

const float *rto_cmf_table(void) { return CMF; }

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1,2,3").
 * Replaces rand::random at 2d/src/librt2d_reverse.rs:89, rt_common.rs:370,
 * 379-380, spectrum.rs:186 and 3d/src/rt.rs:41,181.                           */
void rto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u01(uint32_t lo, uint32_t hi)
{
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (double)(v >> 11) * 0x1.0p-53;
}

typedef struct {
    uint32_t key[2];
    uint32_t c0, c1, c2;
} rng_stream;

static void rng_draw(const rng_stream *s, uint32_t c3, uint32_t out[4])
{
    uint32_t ctr[4] = { s->c0, s->c1, s->c2, c3 };
    rto_philox4x32_10(ctr, s->key, out);
}

/* ------------------------------------------------------------------------- */
/* Deterministic elementary functions.  The reference calls the platform libm
 * (f64::sin_cos through glam's from_angle at rt_common.rs:371, f32::exp at
 * librt2d_reverse.rs:159); their last bits are platform dependent, so the
 * restatement fixes one evaluation scheme built from + - * only, which a
 * device can reproduce bit for bit.                                           */

/* (sin, cos) of 2*pi*u for u in [0,1): quadrant reduction then degree-15/16
 * Taylor polynomials on [-pi/4, pi/4] (truncation error < 5e-17).            */
void rto_sincos2pi(double u, double *s, double *c)
{
    double x = u * 4.0;
    int q = (int)(x + 0.5);
    double r = x - (double)q;
    double th = r * 0x1.921fb54442d18p+0;
    double z = th * th;
    double ps = -0x1.ae7f3e733b81fp-41;
    ps = ps * z + 0x1.6124613a86d09p-33;
    ps = ps * z + -0x1.ae64567f544e4p-26;
    ps = ps * z + 0x1.71de3a556c734p-19;
    ps = ps * z + -0x1.a01a01a01a01ap-13;
    ps = ps * z + 0x1.1111111111111p-7;
    ps = ps * z + -0x1.5555555555555p-3;
    double sn = th + (th * z) * ps;
    double pc = 0x1.ae7f3e733b81fp-45;
    pc = pc * z + -0x1.93974a8c07c9dp-37;
    pc = pc * z + 0x1.1eed8eff8d898p-29;
    pc = pc * z + -0x1.27e4fb7789f5cp-22;
    pc = pc * z + 0x1.a01a01a01a01ap-16;
    pc = pc * z + -0x1.6c16c16c16c17p-10;
    pc = pc * z + 0x1.5555555555555p-5;
    double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    switch (q & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
    }
}

/* e^x in f32: n = round(x*log2 e), Cody-Waite reduction, degree-7 polynomial. */
float rto_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -103.97f) return 0.0f;
    float fn = x * 0x1.715476p+0f;
    int n = (int)(fn + (fn < 0.0f ? -0.5f : 0.5f));
    float nf = (float)n;
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
    union { uint32_t u; float f; } sc;
    sc.u = (uint32_t)(n + 127) << 23;
    return e * sc.f;
}

/* ------------------------------------------------------------------------- */
/* glam f64 vector helpers (glam 0.33 / 0.30 published definitions).           */
typedef struct { double x, y; } dv2;
typedef struct { double x, y, z; } dv3;

static double dot2(dv2 a, dv2 b) { return a.x * b.x + a.y * b.y; }
static dv2 normalize2(dv2 v)
{
    double rcp = 1.0 / sqrt(dot2(v, v));
    dv2 o = { v.x * rcp, v.y * rcp };
    return o;
}
static dv2 reflect2(dv2 v, dv2 n)
{
    double k = 2.0 * dot2(v, n);
    dv2 o = { v.x - k * n.x, v.y - k * n.y };
    return o;
}
static dv2 refract2(dv2 v, dv2 n, double eta)
{
    double ndi = dot2(n, v);
    double k = 1.0 - eta * eta * (1.0 - ndi * ndi);
    if (k >= 0.0) {
        double m = eta * ndi + sqrt(k);
        dv2 o = { eta * v.x - m * n.x, eta * v.y - m * n.y };
        return o;
    }
    dv2 z = { 0.0, 0.0 };
    return z;
}
static double dot3(dv3 a, dv3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static dv3 normalize3(dv3 v)
{
    double rcp = 1.0 / sqrt(dot3(v, v));
    dv3 o = { v.x * rcp, v.y * rcp, v.z * rcp };
    return o;
}
static dv3 cross3(dv3 a, dv3 b)
{
    dv3 o = { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y };
    return o;
}

/* f64::total_cmp(a, b) < 0 (the order min_by / sort_by use at
 * librt2d_reverse.rs:139, rt_common.rs:302, 3d/src/rt.rs:224).                */
static int total_lt(double a, double b)
{
    int64_t ia, ib;
    memcpy(&ia, &a, 8);
    memcpy(&ib, &b, 8);
    ia ^= (int64_t)((uint64_t)(ia >> 63) >> 1);
    ib ^= (int64_t)((uint64_t)(ib >> 63) >> 1);
    return ia < ib;
}

/* ------------------------------------------------------------------------- */
/* 2d primitives                                                               */
typedef struct {
    double t;
    dv2 p, n;
    int side; /* 0 Inside, 1 Outside */
} hit2d;

/* Circle::hit + Circle::is_inside, 2d/src/rt_common.rs:75-111 */
static int circle_hit(dv2 o, dv2 d, dv2 ctr, double r, hit2d *h)
{
    dv2 oc = { o.x - ctr.x, o.y - ctr.y };
    double a = 1.0;
    double b = 2.0 * dot2(d, oc);
    double c = dot2(oc, oc) - r * r;
    double delta = b * b - 4.0 * a * c;
    if (delta < 0.0) return 0;
    double sqrt_delta = sqrt(delta);
    double t0 = (-b - sqrt_delta) / (2.0 * a);
    double t1 = (-b + sqrt_delta) / (2.0 * a);
    double t;
    if (t0 > 0.0) t = t0;
    else if (t1 > 0.0) t = t1;
    else return 0;
    dv2 p = { o.x + t * d.x, o.y + t * d.y };
    dv2 pc = { p.x - ctr.x, p.y - ctr.y };
    dv2 n = normalize2(pc);
    dv2 co = { ctr.x - o.x, ctr.y - o.y };
    h->t = t; h->p = p; h->n = n;
    h->side = (dot2(co, co) < r * r) ? 0 : 1;
    return 1;
}

/* Segment::hit, 2d/src/rt_common.rs:242-276 (n is the stored Segment::new normal, :209-212) */
static int segment_hit(dv2 o, dv2 r, dv2 a, dv2 b, dv2 n, hit2d *h)
{
    dv2 s = { b.x - a.x, b.y - a.y };
    double denom = r.x * s.y - r.y * s.x;
    if (fabs(denom) < 0x1.0p-52) return 0;
    dv2 diff = { a.x - o.x, a.y - o.y };
    double t = (diff.x * s.y - diff.y * s.x) / denom;
    double u = (diff.x * r.y - diff.y * r.x) / denom;
    if (t >= 0.0 && u >= 0.0 && u <= 1.0) {
        h->t = t;
        h->p.x = o.x + t * r.x;
        h->p.y = o.y + t * r.y;
        h->n = n;
        h->side = (dot2(n, r) < 0.0) ? 1 : 0;
        return 1;
    }
    return 0;
}

int rto2d_circle_hit(double ox, double oy, double dx, double dy, double cx, double cy, double r,
                     double *t, double *px, double *py, double *nx, double *ny, int *side)
{
    dv2 o = { ox, oy }, d = { dx, dy }, c = { cx, cy };
    hit2d h;
    if (!circle_hit(o, d, c, r, &h)) return 0;
    *t = h.t; *px = h.p.x; *py = h.p.y; *nx = h.n.x; *ny = h.n.y; *side = h.side;
    return 1;
}

int rto2d_segment_hit(double ox, double oy, double dx, double dy, double ax, double ay, double bx,
                      double by, double nx, double ny, double *t, double *px, double *py, int *side)
{
    dv2 o = { ox, oy }, d = { dx, dy }, a = { ax, ay }, b = { bx, by }, n = { nx, ny };
    hit2d h;
    if (!segment_hit(o, d, a, b, n, &h)) return 0;
    *t = h.t; *px = h.p.x; *py = h.p.y; *side = h.side;
    return 1;
}

/* ------------------------------------------------------------------------- */
/* Object list rebuilt from the flat arrays: object -> circle | run of segments */
typedef struct {
    int is_circle;
    int first;  /* index into the circle or segment arrays */
    int count;  /* segments in the strip (1 for a circle) */
} obj_shape;

typedef struct {
    const rto2d_scene *sc;
    obj_shape *shapes;   /* n_objects, in object (scan) order */
    int *diffuse_black;  /* per object: Diffuse with an all-zero spectrum */
} prep2d;

static int prep2d_init(prep2d *pp, const rto2d_scene *sc)
{
    pp->sc = sc;
    pp->shapes = (obj_shape *)calloc((size_t)(sc->n_objects > 0 ? sc->n_objects : 1), sizeof(obj_shape));
    pp->diffuse_black = (int *)calloc((size_t)(sc->n_objects > 0 ? sc->n_objects : 1), sizeof(int));
    if (!pp->shapes || !pp->diffuse_black) return -1;
    for (int i = 0; i < sc->n_objects; ++i) pp->shapes[i].count = 0;
    for (int i = 0; i < sc->n_circles; ++i) {
        int o = sc->cobj[i];
        if (o < 0 || o >= sc->n_objects || pp->shapes[o].count != 0) return -2;
        pp->shapes[o].is_circle = 1;
        pp->shapes[o].first = i;
        pp->shapes[o].count = 1;
    }
    for (int i = 0; i < sc->n_segments; ++i) {
        int o = sc->sobj[i];
        if (o < 0 || o >= sc->n_objects || pp->shapes[o].is_circle) return -2;
        if (pp->shapes[o].count == 0) pp->shapes[o].first = i;
        else if (pp->shapes[o].first + pp->shapes[o].count != i) return -2; /* strip must be contiguous */
        pp->shapes[o].count++;
    }
    for (int o = 0; o < sc->n_objects; ++o) {
        const rto2d_material *m = &sc->materials[o];
        if (m->kind == RTO_DIFFUSE) {
            /* `absorption == Spectrum::default()`, librt2d_reverse.rs:212 */
            const float *sp = sc->spectra + (size_t)m->spectrum * RTO_NBINS;
            int all_zero = 1;
            for (int k = 0; k < RTO_NBINS; ++k) if (sp[k] != 0.0f) all_zero = 0;
            pp->diffuse_black[o] = all_zero;
        }
    }
    return 0;
}

static void prep2d_free(prep2d *pp)
{
    free(pp->shapes);
    free(pp->diffuse_black);
}

typedef struct {
    int64_t bounces;     /* trace_ray invocations with depth > 0 */
    int32_t first_prim;  /* pid of the primary ray's nearest hit, -1 = miss */
    double first_t;
} path_info;

/* A normal no direction can have a positive product with: the zero vector (a caller's segment normal), or NaN
 * (Circle::hit's `(p - center).normalize()`, rt_common.rs:103, when the hit point rounds onto the centre - a circle
 * smaller than the spacing of f64 at its coordinates).  The reference's rejection loop below never returns for such a
 * normal; here, and in the kernels, the path ends at that hit and contributes nothing. */
static int no_direction(dv2 n)
{
    return !(fabs(n.x) + fabs(n.y) > 0.0);
}

/* Ray2d::rand_hemisphere, 2d/src/rt_common.rs:376-391 */
static dv2 rand_hemisphere(const rng_stream *rs, uint32_t bounce, dv2 n)
{
    for (uint32_t attempt = 0;; ++attempt) {
        uint32_t w[4];
        rng_draw(rs, (bounce << 20) | (attempt & 0xFFFFFu), w);
        dv2 v = { u01(w[0], w[1]) * 2.0 - 1.0, u01(w[2], w[3]) * 2.0 - 1.0 };
        if (dot2(v, v) <= 1.0) {
            v = normalize2(v);
            if (dot2(v, n) > 0.0) return v;
        }
    }
}

/* Ior::ior, 2d/src/rt_common.rs:442-450 */
static double ior_at(const rto2d_material *m, double lambda)
{
    double l = lambda / 1000.0;
    if (m->ior_kind == 0) return m->ior_a;
    return m->ior_a + m->ior_b / (l * l);
}

/* objects.iter().filter_map(hit).min_by(total_cmp): first minimum wins (librt2d_reverse.rs:135-139,
 * librt2d_forward.rs:80-84); Strip::hit = stable sort by t, first element (rt_common.rs:295-305). */
static int scan_objects(const prep2d *pp, dv2 o, dv2 d, hit2d *out_best, int *out_obj, int *out_pid)
{
    const rto2d_scene *sc = pp->sc;
    int have = 0, best_obj = -1, best_pid = -1;
    hit2d best;
    memset(&best, 0, sizeof best);
    for (int ob = 0; ob < sc->n_objects; ++ob) {
        const obj_shape *sh = &pp->shapes[ob];
        hit2d h;
        int got = 0, pid = -1;
        if (sh->count == 0) continue;
        if (sh->is_circle) {
            int i = sh->first;
            dv2 c = { sc->ccx[i], sc->ccy[i] };
            got = circle_hit(o, d, c, sc->cr[i], &h);
            pid = sc->cpid[i];
        } else {
            for (int k = 0; k < sh->count; ++k) {
                int i = sh->first + k;
                dv2 a = { sc->sax[i], sc->say[i] }, b = { sc->sbx[i], sc->sby[i] };
                dv2 n = { sc->snx[i], sc->sny[i] };
                hit2d hs;
                if (segment_hit(o, d, a, b, n, &hs)) {
                    if (!got || total_lt(hs.t, h.t)) { h = hs; pid = sc->spid[i]; got = 1; }
                }
            }
        }
        if (got && (!have || total_lt(h.t, best.t))) {
            best = h; best_obj = ob; best_pid = pid; have = 1;
        }
    }
    *out_best = best; *out_obj = best_obj; *out_pid = best_pid;
    return have;
}

/* ReverseRenderer::trace_ray, 2d/src/librt2d_reverse.rs:112-262 */
static float trace_ray2d(const prep2d *pp, dv2 o, dv2 d, int lambda_idx, double lambda, int depth,
                         int recursion_limit, const rng_stream *rs, path_info *info)
{
    if (depth == 0) return 0.0f;
    const rto2d_scene *sc = pp->sc;
    int bounce = recursion_limit - depth;
    info->bounces++;

    int best_obj = -1, best_pid = -1;
    hit2d best;
    int have = scan_objects(pp, o, d, &best, &best_obj, &best_pid);
    if (bounce == 0) {
        info->first_prim = have ? best_pid : -1;
        info->first_t = have ? best.t : -1.0;
    }
    if (!have) return 0.0f;

    const rto2d_material *m = &sc->materials[best_obj];
    const float *spec = sc->spectra + (size_t)m->spectrum * RTO_NBINS;
    const double EPS = 0x1.0p-52;
    if (best.side == 0) {
        /* Inside, :146-207 */
        if (m->kind == RTO_EMISSIVE) return spec[lambda_idx];
        if (m->kind == RTO_DIELECTRIC) {
            double ior = ior_at(m, lambda);
            dv2 p = { best.p.x + best.n.x * 10000.0 * EPS, best.p.y + best.n.y * 10000.0 * EPS };
            dv2 nn = { -best.n.x, -best.n.y };
            dv2 rr = refract2(d, nn, ior);
            dv2 nd = (rr.x == 0.0 && rr.y == 0.0) ? reflect2(d, nn) : rr;
            float power = trace_ray2d(pp, p, nd, lambda_idx, lambda, depth - 1, recursion_limit, rs, info);
            return power * rto_expf((float)(-best.t) * spec[lambda_idx]);
        }
        return 0.0f;
    }
    /* Outside, :208-261 */
    switch (m->kind) {
    case RTO_DIFFUSE: {
        if (pp->diffuse_black[best_obj]) return 0.0f;
        if (no_direction(best.n)) return 0.0f;
        dv2 p = { best.p.x + best.n.x * 10000.0 * EPS, best.p.y + best.n.y * 10000.0 * EPS };
        dv2 nd = rand_hemisphere(rs, (uint32_t)bounce, best.n);
        float light = trace_ray2d(pp, p, nd, lambda_idx, lambda, depth - 1, recursion_limit, rs, info);
        return spec[lambda_idx] * light;
    }
    case RTO_EMISSIVE:
        return spec[lambda_idx];
    case RTO_DIRECTIONAL: {
        dv2 nn = { -best.n.x, -best.n.y };
        if (dot2(d, nn) >= m->angle) {
            double coeff = 1.0 / (best.t + m->d);
            return spec[lambda_idx] * (float)coeff;
        }
        return 0.0f;
    }
    case RTO_REFLECTIVE: {
        dv2 p = { best.p.x + best.n.x * 10000.0 * EPS, best.p.y + best.n.y * 10000.0 * EPS };
        dv2 nd = reflect2(d, best.n);
        return trace_ray2d(pp, p, nd, lambda_idx, lambda, depth - 1, recursion_limit, rs, info);
    }
    case RTO_DIELECTRIC: {
        double ior = ior_at(m, lambda);
        dv2 p = { best.p.x - best.n.x * 100000.0 * EPS, best.p.y - best.n.y * 100000.0 * EPS };
        dv2 rr = refract2(d, best.n, 1.0 / ior);
        dv2 nd = (rr.x == 0.0 && rr.y == 0.0) ? reflect2(d, best.n) : rr;
        return trace_ray2d(pp, p, nd, lambda_idx, lambda, depth - 1, recursion_limit, rs, info);
    }
    default:
        return 0.0f;
    }
}

static double lambda_of(int idx) { return (double)idx * 5.0 + 360.0; } /* spectrum.rs:185-191 */

int rto2d_trace_batch(const rto2d_scene *sc, int64_t n, const double *ox, const double *oy,
                      const double *dx, const double *dy, const int32_t *lambda_idx, int32_t depth,
                      uint64_t seed, float *radiance, int32_t *first_prim, double *first_t,
                      int32_t *bounces)
{
    prep2d pp;
    int rc = prep2d_init(&pp, sc);
    if (rc) { prep2d_free(&pp); return rc; }
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
        rng_stream rs = { { (uint32_t)seed, (uint32_t)(seed >> 32) }, (uint32_t)i,
                          0x80000000u | (uint32_t)((uint64_t)i >> 32), 0u };
        path_info info = { 0, -1, -1.0 };
        dv2 o = { ox[i], oy[i] }, d = { dx[i], dy[i] };
        int li = lambda_idx[i];
        float v = trace_ray2d(&pp, o, d, li, lambda_of(li), depth, depth, &rs, &info);
        radiance[i] = v;
        if (first_prim) first_prim[i] = info.first_prim;
        if (first_t) first_t[i] = info.first_t;
        if (bounces) bounces[i] = (int32_t)info.bounces;
    }
    prep2d_free(&pp);
    return 0;
}

/* ReverseRenderer::compute_pixel, 2d/src/librt2d_reverse.rs:80-110, up to the final division: total[] is the
 * sum over the samples (:107) */
static void compute_pixel2d_sum(const prep2d *pp, const rto2d_params *p, int px, int py, float *total,
                                uint64_t *rays)
{
    uint32_t key[2] = { (uint32_t)p->seed, (uint32_t)(p->seed >> 32) };
    uint32_t pix = (uint32_t)(p->width * py + px);
    for (int k = 0; k < RTO_NBINS; ++k) total[k] = 0.0f;
    for (int s = 0; s < p->spp; ++s) {
        uint32_t gs = (uint32_t)(p->sample_offset + s);
        rng_stream gen = { { key[0], key[1] }, pix, gs, 0xFFFFFFFFu };
        uint32_t w[4], w1[4];
        rng_draw(&gen, 0u, w);
        rng_draw(&gen, 1u, w1);
        /* pixel.as_dvec2() + rand::<DVec2>() - DVec2::ONE / 2.  (:89) */
        dv2 o = { ((double)px + u01(w[0], w[1])) - 0.5, ((double)py + u01(w[2], w[3])) - 0.5 };
        o.x = o.x * p->px_to_world;
        o.y = o.y * p->px_to_world;
        /* Ray2d::rand, rt_common.rs:369-374 */
        dv2 d;
        rto_sincos2pi(u01(w1[0], w1[1]), &d.y, &d.x);

        float spp_spec[RTO_NBINS];
        for (int k = 0; k < RTO_NBINS; ++k) spp_spec[k] = 0.0f;
        for (int l = 0; l < p->lambda_samples; ++l) {
            uint32_t call = 1u + (uint32_t)(l + 2) / 4u, word = (uint32_t)(l + 2) % 4u, wl[4];
            if (call == 1u) memcpy(wl, w1, sizeof wl);
            else rng_draw(&gen, call, wl);
            int li = (int)(((uint64_t)wl[word] * RTO_NBINS) >> 32);
            rng_stream rs = { { key[0], key[1] }, pix, gs, (uint32_t)l };
            path_info info = { 0, -1, -1.0 };
            float v = trace_ray2d(pp, o, d, li, lambda_of(li), p->recursion_limit, p->recursion_limit,
                                  &rs, &info);
            spp_spec[li] += v;
            *rays += (uint64_t)info.bounces;
        }
        for (int k = 0; k < RTO_NBINS; ++k) {
            float v = spp_spec[k] / (float)p->lambda_samples * (float)RTO_NBINS; /* :104-105 */
            total[k] += v;                                                     /* :107 */
        }
    }
}

static void compute_pixel2d(const prep2d *pp, const rto2d_params *p, int px, int py, float *total,
                            uint64_t *rays)
{
    compute_pixel2d_sum(pp, p, px, py, total, rays);
    for (int k = 0; k < RTO_NBINS; ++k) total[k] = total[k] / (float)p->spp; /* :109 */
}

int rto2d_compute_pixel(const rto2d_scene *sc, const rto2d_params *p, int32_t px, int32_t py,
                        float *spectrum95, uint64_t *rays)
{
    prep2d pp;
    int rc = prep2d_init(&pp, sc);
    if (rc) { prep2d_free(&pp); return rc; }
    uint64_t r = 0;
    compute_pixel2d(&pp, p, px, py, spectrum95, &r);
    if (rays) *rays = r;
    prep2d_free(&pp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* ReverseRenderer::denoise, 2d/src/librt2d_reverse.rs:337-422: adaptive re-sampling of the noisiest pixels.
 * Images are RawImage planes: rgb[W*H][3] (PixelData.value) and weight[W*H], idx = W*y + x (img.rs:93-99). */

/* |merged - loop| (img.rs Sub :279-300, abs :144-155), then RawImage::convolution with the 3x3 Laplace kernel
 * (img.rs:167-185: output (x, y) = sum over i, j in 0..3, i outer, of in(x + i, y + j) * k[i][j], only for
 * x < W - 3 and y < H - 3, the rest stays 0), then value = splat(length) (librt2d_reverse.rs:353-358) and the
 * sort key value.length() (:368).  key[idx] is what the reference sorts by. */
static void denoise_keys(int W, int H, const float *merged_rgb, const float *loop_rgb, float *key)
{
    static const float K[3][3] = { { -1.f, -1.f, -1.f }, { -1.f, 8.f, -1.f }, { -1.f, -1.f, -1.f } };
    size_t n = (size_t)W * (size_t)H;
    float *diff = (float *)malloc(n * 3 * sizeof(float));
    for (size_t i = 0; i < n * 3; ++i) diff[i] = fabsf(merged_rgb[i] - loop_rgb[i]);
    for (size_t i = 0; i < n; ++i) key[i] = 0.0f; /* length of splat(0) */
    for (int x = 0; x < W - 3; ++x)
        for (int y = 0; y < H - 3; ++y) {
            float v[3] = { 0.f, 0.f, 0.f };
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    const float *px = diff + 3 * ((size_t)W * (size_t)(y + j) + (size_t)(x + i));
                    for (int c = 0; c < 3; ++c) v[c] = v[c] + px[c] * K[i][j];
                }
            float len = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); /* glam Vec3::length */
            key[(size_t)W * (size_t)y + (size_t)x] = sqrtf((len * len + len * len) + len * len);
        }
    free(diff);
}

typedef struct { float key; int64_t idx; } keyed;

/* sort_by(total_cmp) is stable; the reference then reverses (:368-369): descending key, and among equal
 * keys descending index.  Keys are non-negative or NaN-free here (sqrt of sums of squares), so the float
 * order is total_cmp's order. */
static int keyed_cmp_desc(const void *a, const void *b)
{
    const keyed *x = (const keyed *)a, *y = (const keyed *)b;
    int32_t kx, ky;
    memcpy(&kx, &x->key, 4);
    memcpy(&ky, &y->key, 4);
    kx ^= (int32_t)((uint32_t)(kx >> 31) >> 1);
    ky ^= (int32_t)((uint32_t)(ky >> 31) >> 1);
    if (kx != ky) return kx > ky ? -1 : 1;
    if (x->idx != y->idx) return x->idx > y->idx ? -1 : 1;
    return 0;
}

/* The selection of denoise (:346-392): keys, stable sort + reverse, top fraction, +-mask_size neighbourhoods.
 * Returns the recompute set as a malloc'ed list in row-major order of the image grown by mask_size on every
 * side (the reference's HashSet has no order; results do not depend on it). */
static size_t select_noisy(int W, int H, float top, int32_t mask_size, const float *merged_rgb, const float *loop_rgb,
                           float *out_key, int32_t **out_lx, int32_t **out_ly)
{
    const size_t n = (size_t)W * (size_t)H;
    float *key = (float *)malloc(n * sizeof(float));
    keyed *order = (keyed *)malloc(n * sizeof(keyed));
    denoise_keys(W, H, merged_rgb, loop_rgb, key);
    if (out_key) memcpy(out_key, key, n * sizeof(float));
    for (size_t i = 0; i < n; ++i) { order[i].key = key[i]; order[i].idx = (int64_t)i; }
    qsort(order, n, sizeof(keyed), keyed_cmp_desc);
    /* `(total_pixels as f32 * denoiser.top) as usize` (:372-373): f32 product, saturating cast */
    float topf = (float)(int32_t)(W * H) * top;
    size_t n_top = topf > 0.0f ? (topf >= (float)n ? n : (size_t)topf) : 0;
    /* the HashSet of mask pixels (:380-392), here a bitmap over the image grown by mask_size on every side:
     * pixels outside the image are recomputed too (and dropped by draw_pixel's Err) */
    const int m = mask_size, EW = W + 2 * m, EH = H + 2 * m;
    uint8_t *mask = (uint8_t *)calloc((size_t)EW * (size_t)EH, 1);
    for (size_t k = 0; k < n_top; ++k) {
        int x = (int)(order[k].idx % W), y = (int)(order[k].idx / W);
        for (int i = -m; i <= m; ++i)
            for (int j = -m; j <= m; ++j) mask[(size_t)EW * (size_t)(y + j + m) + (size_t)(x + i + m)] = 1;
    }
    size_t cnt = 0;
    for (size_t i = 0; i < (size_t)EW * (size_t)EH; ++i) cnt += mask[i];
    int32_t *lx = (int32_t *)malloc((cnt + 1) * sizeof(int32_t)), *ly = (int32_t *)malloc((cnt + 1) * sizeof(int32_t));
    size_t c = 0;
    for (int ey = 0; ey < EH; ++ey)
        for (int ex = 0; ex < EW; ++ex)
            if (mask[(size_t)EW * (size_t)ey + (size_t)ex]) { lx[c] = ex - m; ly[c] = ey - m; ++c; }
    free(mask); free(order); free(key);
    *out_lx = lx; *out_ly = ly;
    return cnt;
}

int rto2d_denoise(const rto2d_scene *sc, const rto2d_params *p, float top, int32_t mask_size,
                  float oversampling_factor, int32_t n_threads, float *merged_rgb, float *merged_weight,
                  const float *loop_rgb, float *out_key, int64_t *out_recomputed, uint64_t *rays)
{
    if (!sc || !p || !merged_rgb || !merged_weight || !loop_rgb || mask_size < 0) return -1;
    const int W = p->width, H = p->height;
    prep2d pp;
    int rc = prep2d_init(&pp, sc);
    if (rc) { prep2d_free(&pp); return rc; }
    int32_t *lx, *ly;
    size_t cnt = select_noisy(W, H, top, mask_size, merged_rgb, loop_rgb, out_key, &lx, &ly);
    /* recompute (:402-409) and add (:410-418): value += spectrum.to_dvec3(), weight += spp * oversampling_factor */
    uint64_t total_rays = 0;
    const float wadd = (float)p->spp * oversampling_factor;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads > 0 ? n_threads : 1) reduction(+ : total_rays)
    for (int64_t k = 0; k < (int64_t)cnt; ++k) {
        float spec[RTO_NBINS], rgb[3];
        uint64_t r = 0;
        compute_pixel2d(&pp, p, lx[k], ly[k], spec, &r);
        total_rays += r;
        if (lx[k] < 0 || ly[k] < 0 || lx[k] >= W || ly[k] >= H) continue; /* pixel_to_idx -> Err */
        rto_resolve_pixel(spec, rgb);
        size_t idx = (size_t)W * (size_t)ly[k] + (size_t)lx[k];
        for (int ch = 0; ch < 3; ++ch) merged_rgb[3 * idx + ch] = merged_rgb[3 * idx + ch] + rgb[ch];
        merged_weight[idx] = merged_weight[idx] + wadd;
    }
    if (out_recomputed) *out_recomputed = (int64_t)cnt;
    if (rays) *rays = total_rays;
    free(lx); free(ly);
    prep2d_free(&pp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Progressive accumulation in linear light: the loop of endless_render (librt2d_reverse.rs:424-510) with the
 * pass merge done on spectral sums and per-pixel sample counts - sum f32[W*H][95], count f32[W*H] - where the
 * reference adds encoded colours (`merged_image + loop_image`, img.rs:257-274).  Not reference arithmetic: this
 * states the merge the GPU library's progressive entry points are specified to perform.                       */
static void resolve_div(const float *sum95, float div, float rgb[3])
{
    float v[RTO_NBINS];
    for (int k = 0; k < RTO_NBINS; ++k) v[k] = sum95[k] / div;
    rto_resolve_pixel(v, rgb);
}

/* One whole-image pass (global_render, :282-302) of p->spp samples from p->sample_offset: loop_rgb = sRGB(pass / spp),
 * sum += pass, count += spp, merged_rgb = sRGB(sum / count). */
int rto2d_progressive_pass(const rto2d_scene *sc, const rto2d_params *p, int32_t n_threads, float *sum,
                           float *count, float *merged_rgb, float *loop_rgb, uint64_t *rays)
{
    if (!sc || !p || !sum || !count || !merged_rgb || !loop_rgb) return -1;
    prep2d pp;
    int rc = prep2d_init(&pp, sc);
    if (rc) { prep2d_free(&pp); return rc; }
    uint64_t total_rays = 0;
    const int64_t n = (int64_t)p->width * p->height;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads > 0 ? n_threads : 1) reduction(+ : total_rays)
    for (int64_t idx = 0; idx < n; ++idx) {
        float pass[RTO_NBINS];
        uint64_t r = 0;
        compute_pixel2d_sum(&pp, p, (int)(idx % p->width), (int)(idx / p->width), pass, &r);
        total_rays += r;
        float *s = sum + (size_t)idx * RTO_NBINS;
        for (int k = 0; k < RTO_NBINS; ++k) s[k] = s[k] + pass[k];
        count[idx] = count[idx] + (float)p->spp;
        resolve_div(pass, (float)p->spp, loop_rgb + 3 * (size_t)idx);
        resolve_div(s, count[idx], merged_rgb + 3 * (size_t)idx);
    }
    if (rays) *rays = total_rays;
    prep2d_free(&pp);
    return 0;
}

/* One adaptive pass: the selection of denoise on (merged_rgb, loop_rgb); every selected pixel takes p->spp new
 * samples from p->sample_offset; inside the image they join sum and count and merged_rgb is refreshed. */
int rto2d_progressive_resample(const rto2d_scene *sc, const rto2d_params *p, float top, int32_t mask_size,
                               int32_t n_threads, float *sum, float *count, float *merged_rgb,
                               const float *loop_rgb, int64_t *out_recomputed, uint64_t *rays)
{
    if (!sc || !p || !sum || !count || !merged_rgb || !loop_rgb || mask_size < 0) return -1;
    const int W = p->width, H = p->height;
    prep2d pp;
    int rc = prep2d_init(&pp, sc);
    if (rc) { prep2d_free(&pp); return rc; }
    int32_t *lx, *ly;
    size_t cnt = select_noisy(W, H, top, mask_size, merged_rgb, loop_rgb, NULL, &lx, &ly);
    uint64_t total_rays = 0;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads > 0 ? n_threads : 1) reduction(+ : total_rays)
    for (int64_t k = 0; k < (int64_t)cnt; ++k) {
        float add[RTO_NBINS];
        uint64_t r = 0;
        compute_pixel2d_sum(&pp, p, lx[k], ly[k], add, &r);
        total_rays += r;
        if (lx[k] < 0 || ly[k] < 0 || lx[k] >= W || ly[k] >= H) continue;
        size_t idx = (size_t)W * (size_t)ly[k] + (size_t)lx[k];
        float *s = sum + idx * RTO_NBINS;
        for (int b = 0; b < RTO_NBINS; ++b) s[b] = s[b] + add[b];
        count[idx] = count[idx] + (float)p->spp;
        resolve_div(s, count[idx], merged_rgb + 3 * idx);
    }
    if (out_recomputed) *out_recomputed = (int64_t)cnt;
    if (rays) *rays = total_rays;
    free(lx); free(ly);
    prep2d_free(&pp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Forward light tracer, 2d/src/librt2d_forward.rs.  Upstream it is a stub: trace_ray hands back only what its
 * innermost call returns, so a path yields one (ray, power, None) entry - its LAST ray, when that ray escapes
 * the scene - or nothing; every emitter shoots 1000 rays of wavelength index 0 with `lambda = 1.`, depth 5;
 * sample_out_ray exists for circles only (todo!() for segments and strips, rt_common.rs:278-280, 322-324).   */

/* ForwardRenderer::trace_ray, :66-195.  Returns 1 when the path escapes, with its last ray. */
static int forward_trace(const prep2d *pp, dv2 o, dv2 d, int lambda_idx, double lambda, int depth, int depth0,
                         float power, const rng_stream *rs, dv2 *last_o, dv2 *last_d, float *last_power,
                         uint64_t *rays)
{
    if (depth == 0) return 0; /* :76-78 */
    const rto2d_scene *sc = pp->sc;
    int bounce = depth0 - depth;
    *rays += 1;
    int best_obj = -1, best_pid = -1;
    hit2d best;
    if (!scan_objects(pp, o, d, &best, &best_obj, &best_pid)) { /* :85-87 */
        *last_o = o; *last_d = d; *last_power = power;
        return 1;
    }
    const rto2d_material *m = &sc->materials[best_obj];
    const float *spec = sc->spectra + (size_t)m->spectrum * RTO_NBINS;
    const double EPS = 0x1.0p-52;
    if (best.side == 0) { /* Inside, :94-119 */
        if (m->kind != RTO_DIELECTRIC) return 0;
        double ior = ior_at(m, lambda);
        dv2 p = { best.p.x + best.n.x * 10000.0 * EPS, best.p.y + best.n.y * 10000.0 * EPS };
        dv2 nn = { -best.n.x, -best.n.y };
        dv2 rr = refract2(d, nn, ior);
        dv2 nd = (rr.x == 0.0 && rr.y == 0.0) ? reflect2(d, nn) : rr;
        float new_power = power * rto_expf((float)(-best.t) * spec[lambda_idx]);
        return forward_trace(pp, p, nd, lambda_idx, lambda, depth - 1, depth0, new_power, rs, last_o, last_d, last_power, rays);
    }
    switch (m->kind) { /* Outside, :120-194 */
    case RTO_DIFFUSE: {
        if (pp->diffuse_black[best_obj]) return 0;
        if (no_direction(best.n)) return 0;
        dv2 p = { best.p.x + best.n.x * 10000.0 * EPS, best.p.y + best.n.y * 10000.0 * EPS };
        dv2 nd = rand_hemisphere(rs, (uint32_t)bounce, best.n);
        float new_power = spec[lambda_idx] * power;
        if (!(new_power > 0.0f)) return 0;
        return forward_trace(pp, p, nd, lambda_idx, lambda, depth - 1, depth0, new_power, rs, last_o, last_d, last_power, rays);
    }
    case RTO_REFLECTIVE: {
        dv2 p = { best.p.x + best.n.x * 10000.0 * EPS, best.p.y + best.n.y * 10000.0 * EPS };
        dv2 nd = reflect2(d, best.n);
        return forward_trace(pp, p, nd, lambda_idx, lambda, depth - 1, depth0, power, rs, last_o, last_d, last_power, rays);
    }
    case RTO_DIELECTRIC: {
        double ior = ior_at(m, lambda);
        dv2 p = { best.p.x - best.n.x * 100000.0 * EPS, best.p.y - best.n.y * 100000.0 * EPS };
        dv2 rr = refract2(d, best.n, 1.0 / ior);
        dv2 nd = (rr.x == 0.0 && rr.y == 0.0) ? reflect2(d, best.n) : rr;
        return forward_trace(pp, p, nd, lambda_idx, lambda, depth - 1, depth0, power, rs, last_o, last_d, last_power, rays);
    }
    default: /* Emissive, DirectionalEmissive: `return vec![]` */
        return 0;
    }
}

/* line_drawing::XiaolinWu<f64, i32> (crate line_drawing 1.0.1, a dependency that is not in the reference tree;
 * restated from its published source): the iterator the reference draws a ray with (:222-236), each item added
 * to the image as draw_pixel((x, y), {value: ONE * v as f32, weight: 1}, Blending::Add). */
static void xiaolin_wu_add(double sx, double sy, double ex, double ey, int W, int H, float *rgb, float *weight)
{
    int steep = fabs(ey - sy) > fabs(ex - sx);
    if (steep) { double t = sx; sx = sy; sy = t; t = ex; ex = ey; ey = t; }
    if (sx > ex) { double t = sx; sx = ex; ex = t; t = sy; sy = ey; ey = t; }
    double dx = ex - sx;
    double gradient = (dx == 0.0) ? 1.0 : (ey - sy) / dx;
    int32_t x = (int32_t)round(sx), end_x = (int32_t)round(ex);
    double y = sy;
    int lower = 0;
    while (x <= end_x) {
        double fpart = y - floor(y);
        int32_t yi = (int32_t)y; /* NumCast: toward zero */
        if (lower) yi += 1;
        int px = steep ? yi : x, py = steep ? x : yi;
        double v;
        if (lower) {
            lower = 0;
            x += 1;
            y += gradient;
            v = fpart;
        } else {
            if (fpart > 0.0) lower = 1;
            else { x += 1; y += gradient; }
            v = 1.0 - fpart;
        }
        if (px >= 0 && px < W && py >= 0 && py < H) { /* pixel_to_idx, img.rs:93-99 */
            size_t idx = (size_t)W * (size_t)py + (size_t)px;
            float f = (float)v;
            rgb[3 * idx] += f; rgb[3 * idx + 1] += f; rgb[3 * idx + 2] += f;
            weight[idx] += 1.0f;
        }
    }
}

void rto_xiaolin_wu_add(double sx, double sy, double ex, double ey, int32_t W, int32_t H, float *rgb, float *weight)
{
    xiaolin_wu_add(sx, sy, ex, ey, W, H, rgb, weight);
}

/* ForwardRenderer::global_render, :198-249, ADDED onto the given RawImage planes (the reference starts from
 * RawImage::new: value 0, weight 1).  Emitters in object order; out_lines (nullable, [n_emitters * rays][5]) =
 * escaped flag, origin, direction of every path's last ray.  Returns -3 for an emitter that is not a circle. */
int rto2d_forward_render(const rto2d_scene *sc, int32_t width, int32_t height, int32_t rays_per_emitter, int32_t depth,
                         int32_t sample_offset, uint64_t seed, float *rgb, float *weight, double *out_lines,
                         int64_t *out_n_lines, uint64_t *rays)
{
    if (!sc || !rgb || !weight || rays_per_emitter < 0 || depth < 0) return -1;
    prep2d pp;
    int rc = prep2d_init(&pp, sc);
    if (rc) { prep2d_free(&pp); return rc; }
    uint64_t total_rays = 0;
    int64_t n_lines = 0, slot = 0;
    const int lambda_idx = 0;   /* sample_emission, :211-217 */
    const double lambda = 1.0;
    for (int ob = 0; ob < sc->n_objects; ++ob) {
        const rto2d_material *m = &sc->materials[ob];
        if (m->kind != RTO_EMISSIVE && m->kind != RTO_DIRECTIONAL) continue;
        if (!pp.shapes[ob].is_circle) { prep2d_free(&pp); return -3; }
        int ci = pp.shapes[ob].first;
        for (int k = 0; k < rays_per_emitter; ++k, ++slot) {
            rng_stream rs = { { (uint32_t)seed, (uint32_t)(seed >> 32) }, 0x40000000u | (uint32_t)ob,
                              (uint32_t)(sample_offset + k), 0u };
            /* Circle::sample_out_ray, rt_common.rs:113-120 */
            uint32_t w[4];
            rng_stream gen = rs;
            gen.c2 = 0xFFFFFFFFu;
            rng_draw(&gen, 0u, w);
            dv2 n;
            rto_sincos2pi(u01(w[0], w[1]), &n.y, &n.x);
            const double delta = 0.0001;
            dv2 o = { sc->ccx[ci] + (sc->cr[ci] + delta) * n.x, sc->ccy[ci] + (sc->cr[ci] + delta) * n.y };
            dv2 lo, ld;
            float lp;
            int escaped = forward_trace(&pp, o, n, lambda_idx, lambda, depth, depth, 1.0f, &rs, &lo, &ld, &lp, &total_rays);
            if (out_lines) {
                double *q = out_lines + 5 * slot;
                q[0] = (double)escaped;
                q[1] = escaped ? lo.x : 0.0; q[2] = escaped ? lo.y : 0.0; q[3] = escaped ? ld.x : 0.0; q[4] = escaped ? ld.y : 0.0;
            }
            if (escaped) {
                ++n_lines;
                const double t_max = 1000.0; /* hit is None, :220 */
                xiaolin_wu_add(lo.x, lo.y, lo.x + t_max * ld.x, lo.y + t_max * ld.y, width, height, rgb, weight);
            }
        }
    }
    if (out_n_lines) *out_n_lines = n_lines;
    if (rays) *rays = total_rays;
    prep2d_free(&pp);
    return 0;
}

/* One channel of palette 0.7.7's `Xyz<D65, f32>::into_color() -> Srgb<f32>` after the matrix (spectrum.rs:258-259):
 * the sRGB transfer function, then - `into_color` is the CLAMPED conversion (`from_color_unclamped(..).clamp()`),
 * unlike `into_color_unclamped` - the channel limited to [0, 1].  Every image in the reference's gallery has 127 =
 * (1 / (1 + 1) * 255) as u8 as its largest byte, which is what a channel capped at 1 gives under Reinhard(1). */
static inline float srgb_channel(float c)
{
    const float e = (c <= 0.0031308f) ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    return (e < 0.0f) ? 0.0f : (e > 1.0f) ? 1.0f : e;
}

/* Spectrum::to_xyz + to_dvec3, 2d/src/spectrum.rs:242-278; palette Xyz<D65> -> Srgb:
 * linear = M * XYZ (sRGB/D65 matrix), then the sRGB transfer function and the clamp to [0, 1]. */
void rto_resolve_pixel(const float spectrum[RTO_NBINS], float rgb[3])
{
    float x = 0.0f, y = 0.0f, z = 0.0f;
    for (int i = 0; i < RTO_NBINS; ++i) {
        x = x + spectrum[i] * CMF[3 * i + 0];
        y = y + spectrum[i] * CMF[3 * i + 1];
        z = z + spectrum[i] * CMF[3 * i + 2];
    }
    float lin[3];
    lin[0] = 3.2404542f * x + -1.5371385f * y + -0.4985314f * z;
    lin[1] = -0.9692660f * x + 1.8760108f * y + 0.0415560f * z;
    lin[2] = 0.0556434f * x + -0.2040259f * y + 1.0572252f * z;
    for (int k = 0; k < 3; ++k) rgb[k] = srgb_channel(lin[k]);
}

/* The resolve in two steps around a sum over processes (to_xyz is linear): xyz = sum_b spectrum_sum[b] * cmf[b],
 * then (xyz / total_spp) -> sRGB as in rto_resolve_pixel. */
void rto_project_xyz(const float spectrum_sum[RTO_NBINS], float xyz[3])
{
    float x = 0.0f, y = 0.0f, z = 0.0f;
    for (int i = 0; i < RTO_NBINS; ++i) {
        x = x + spectrum_sum[i] * CMF[3 * i + 0];
        y = y + spectrum_sum[i] * CMF[3 * i + 1];
        z = z + spectrum_sum[i] * CMF[3 * i + 2];
    }
    xyz[0] = x; xyz[1] = y; xyz[2] = z;
}

void rto_encode_xyz(const float xyz[3], float total_spp, float rgb[3])
{
    const float x = xyz[0] / total_spp, y = xyz[1] / total_spp, z = xyz[2] / total_spp;
    float lin[3];
    lin[0] = 3.2404542f * x + -1.5371385f * y + -0.4985314f * z;
    lin[1] = -0.9692660f * x + 1.8760108f * y + 0.0415560f * z;
    lin[2] = 0.0556434f * x + -0.2040259f * y + 1.0572252f * z;
    for (int k = 0; k < 3; ++k) rgb[k] = srgb_channel(lin[k]);
}

/* RawImage::convert_to_image(Reinhard), 2d/src/img.rs:187-218 (alpha 255) and
 * 3d/src/img.rs:126-157 (param 1, alpha 254).  `as u8` saturates, NaN -> 0.   */
void rto_reinhard_rgba8(const float *rgb, int64_t n_pixels, float param, uint8_t alpha, uint8_t *out)
{
    for (int64_t i = 0; i < n_pixels; ++i) {
        for (int k = 0; k < 3; ++k) {
            float v = rgb[3 * i + k];
            float m = v / (param + v) * 255.0f;
            uint8_t b;
            if (!(m == m)) b = 0;
            else if (m <= 0.0f) b = 0;
            else if (m >= 255.0f) b = 255;
            else b = (uint8_t)m;
            out[4 * i + k] = b;
        }
        out[4 * i + 3] = alpha;
    }
}

/* ReverseRenderer::global_render + render_column, 2d/src/librt2d_reverse.rs:264-302:
 * columns one after another, the rows of a column on the thread pool.         */
int rto2d_render(const rto2d_scene *sc, const rto2d_params *p, int32_t n_threads, float *out_rgb,
                 float *out_weight, float *out_spectrum, uint64_t *rays)
{
    prep2d pp;
    int rc = prep2d_init(&pp, sc);
    if (rc) { prep2d_free(&pp); return rc; }
    uint64_t total_rays = 0;
    if (n_threads < 1) n_threads = 1;
    for (int i = 0; i < p->width; ++i) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : total_rays)
        for (int j = 0; j < p->height; ++j) {
            float spec[RTO_NBINS];
            uint64_t r = 0;
            compute_pixel2d(&pp, p, i, j, spec, &r);
            total_rays += r;
            size_t idx = (size_t)p->width * (size_t)j + (size_t)i; /* img.rs:93-99 */
            if (out_spectrum) memcpy(out_spectrum + idx * RTO_NBINS, spec, sizeof spec);
            if (out_rgb) rto_resolve_pixel(spec, out_rgb + idx * 3);
            if (out_weight) out_weight[idx] = (float)p->spp; /* :293 */
        }
    }
    if (rays) *rays = total_rays;
    prep2d_free(&pp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* 3d crate                                                                    */
typedef struct {
    double t;
    dv3 p, n;
    int side;
} hit3d;

/* Sphere::hit, 3d/src/rt.rs:71-118 */
static int sphere_hit(dv3 o, dv3 d, dv3 ctr, double radius, hit3d *h)
{
    dv3 oc = { ctr.x - o.x, ctr.y - o.y, ctr.z - o.z };
    double a = dot3(d, d);
    double hh = dot3(d, oc);
    double c = dot3(oc, oc) - radius * radius;
    double delta = hh * hh - a * c;
    if (delta < 0.0) return 0;
    double t = (hh - sqrt(delta)) / a;
    int side = 1;
    if (!(t >= 0.0)) {
        t = (hh + sqrt(delta)) / a;
        side = 0;
        if (!(t >= 0.0)) return 0;
    }
    dv3 p = { o.x + d.x * t, o.y + d.y * t, o.z + d.z * t };
    dv3 pc = { p.x - ctr.x, p.y - ctr.y, p.z - ctr.z };
    h->t = t; h->p = p; h->n = normalize3(pc); h->side = side;
    return 1;
}

int rto3d_sphere_hit(const double o[3], const double d[3], const double c[3], double r, double *t,
                     double p[3], double n[3], int *side)
{
    dv3 oo = { o[0], o[1], o[2] }, dd = { d[0], d[1], d[2] }, cc = { c[0], c[1], c[2] };
    hit3d h;
    if (!sphere_hit(oo, dd, cc, r, &h)) return 0;
    *t = h.t; p[0] = h.p.x; p[1] = h.p.y; p[2] = h.p.z;
    n[0] = h.n.x; n[1] = h.n.y; n[2] = h.n.z; *side = h.side;
    return 1;
}

/* World::get_background_color, 3d/src/rt.rs:275-287 */
static void background3d(dv3 d, float rgb[3])
{
    float a = 0.5f * ((float)d.y + 1.0f);
    rgb[0] = (1.0f - a) * 1.0f + a * 0.0f;
    rgb[1] = (1.0f - a) * 1.0f + a * 0.7f;
    rgb[2] = (1.0f - a) * 1.0f + a * 1.0f;
    dv3 sun = { 0.3, 1.0, 0.0 };
    sun = normalize3(sun);
    if (dot3(d, sun) > 0.99) { rgb[0] = 240.0f; rgb[1] = 255.0f; rgb[2] = 170.0f; }
}

/* random_on_hemisphere, 3d/src/rt.rs:40-45 */
static dv3 random_on_hemisphere(const rng_stream *rs, uint32_t bounce, dv3 n)
{
    uint32_t w0[4], w1[4];
    rng_draw(rs, (bounce << 20) | 0u, w0);
    rng_draw(rs, (bounce << 20) | 1u, w1);
    dv3 v = { u01(w0[0], w0[1]) - 0.5 * 1.0, u01(w0[2], w0[3]) - 0.5 * 1.0, u01(w1[0], w1[1]) - 0.5 * 1.0 };
    v = normalize3(v);
    if (dot3(n, v) > 0.0) return v;
    dv3 m = { -v.x, -v.y, -v.z };
    return m;
}

/* World::trace_ray, 3d/src/rt.rs:216-236 */
static void trace_ray3d(const rto3d_scene *sc, dv3 o, dv3 d, int bounce_limit, int limit0,
                        const rng_stream *rs, path_info *info, float rgb[3])
{
    if (bounce_limit <= 0) { rgb[0] = rgb[1] = rgb[2] = 0.0f; return; }
    int bounce = limit0 - bounce_limit;
    info->bounces++;
    int have = 0, best_i = -1;
    hit3d best;
    memset(&best, 0, sizeof best);
    for (int i = 0; i < sc->n_spheres; ++i) {
        dv3 c = { sc->cx[i], sc->cy[i], sc->cz[i] };
        hit3d h;
        if (sphere_hit(o, d, c, sc->r[i], &h)) {
            if (!have || total_lt(h.t, best.t)) { best = h; best_i = i; have = 1; }
        }
    }
    if (bounce == 0) {
        info->first_prim = have ? best_i : -1;
        info->first_t = have ? best.t : -1.0;
    }
    if (have) {
        dv3 r = random_on_hemisphere(rs, (uint32_t)bounce, best.n);
        dv3 bd = { r.x + best.n.x, r.y + best.n.y, r.z + best.n.z };
        bd = normalize3(bd);
        double off = 100.0 * 0x1.0p-52;
        dv3 p = { best.p.x + best.n.x * off, best.p.y + best.n.y * off, best.p.z + best.n.z * off };
        float sub[3];
        trace_ray3d(sc, p, bd, bounce_limit - 1, limit0, rs, info, sub);
        rgb[0] = 0.5f * sub[0]; rgb[1] = 0.5f * sub[1]; rgb[2] = 0.5f * sub[2];
    } else {
        float bg[3];
        background3d(d, bg);
        rgb[0] = bg[0] / 100.0f; rgb[1] = bg[1] / 100.0f; rgb[2] = bg[2] / 100.0f;
    }
}

int rto3d_trace_batch(const rto3d_scene *sc, int64_t n, const double *o, const double *d,
                      int32_t bounce_limit, uint64_t seed, float *rgb, int32_t *first_prim,
                      double *first_t, int32_t *bounces)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
        rng_stream rs = { { (uint32_t)seed, (uint32_t)(seed >> 32) }, (uint32_t)i,
                          0x80000000u | (uint32_t)((uint64_t)i >> 32), 0u };
        path_info info = { 0, -1, -1.0 };
        dv3 oo = { o[3 * i], o[3 * i + 1], o[3 * i + 2] }, dd = { d[3 * i], d[3 * i + 1], d[3 * i + 2] };
        trace_ray3d(sc, oo, dd, bounce_limit, bounce_limit, &rs, &info, rgb + 3 * i);
        if (first_prim) first_prim[i] = info.first_prim;
        if (first_t) first_t[i] = info.first_t;
        if (bounces) bounces[i] = (int32_t)info.bounces;
    }
    return 0;
}

/* Camera::new, 3d/src/rt.rs:139-178 */
typedef struct { dv3 position, pixel00, du, dv; } camera3d;

static int camera3d_init(const rto3d_scene *sc, camera3d *cam)
{
    dv3 pos = { sc->cam_pos[0], sc->cam_pos[1], sc->cam_pos[2] };
    dv3 dir = { sc->cam_dir[0], sc->cam_dir[1], sc->cam_dir[2] };
    dv3 up = { sc->cam_up[0], sc->cam_up[1], sc->cam_up[2] };
    /* assert!(dir.is_normalized()); assert!(up.is_normalized());  glam: |len^2 - 1| <= 2e-4 */
    if (!(fabs(dot3(dir, dir) - 1.0) <= 2e-4) || !(fabs(dot3(up, up) - 1.0) <= 2e-4)) return -3;
    double vh = 2.0;
    double vw = vh * ((double)sc->width / (double)sc->height);
    dv3 right = cross3(dir, up);
    dv3 vu = { right.x * vw, right.y * vw, right.z * vw };
    dv3 vv = { up.x * vh, up.y * vh, up.z * vh };
    dv3 du = { vu.x / (double)sc->width, vu.y / (double)sc->width, vu.z / (double)sc->width };
    dv3 dv = { vv.x / (double)sc->height, vv.y / (double)sc->height, vv.z / (double)sc->height };
    double f = sc->focal_length;
    dv3 ul = { ((pos.x + f * dir.x) - vu.x / 2.0) - vv.x / 2.0, ((pos.y + f * dir.y) - vu.y / 2.0) - vv.y / 2.0,
               ((pos.z + f * dir.z) - vu.z / 2.0) - vv.z / 2.0 };
    dv3 p00 = { ul.x + (du.x + dv.x) / 2.0, ul.y + (du.y + dv.y) / 2.0, ul.z + (du.z + dv.z) / 2.0 };
    cam->position = pos; cam->pixel00 = p00; cam->du = du; cam->dv = dv;
    return 0;
}

/* Camera::get_ray, 3d/src/rt.rs:180-191 */
static void camera3d_ray(const camera3d *cam, const rto3d_scene *sc, uint64_t seed, int i, int j,
                         uint32_t sample, dv3 *o, dv3 *d)
{
    rng_stream gen = { { (uint32_t)seed, (uint32_t)(seed >> 32) }, (uint32_t)(sc->width * j + i), sample,
                       0xFFFFFFFFu };
    uint32_t w[4];
    rng_draw(&gen, 0u, w);
    double rx = u01(w[0], w[1]) - 1.0 / 2.0, ry = u01(w[2], w[3]) - 1.0 / 2.0;
    double a = (double)i + rx, b = (double)j + ry;
    dv3 pc = { (cam->pixel00.x + a * cam->du.x) + b * cam->dv.x, (cam->pixel00.y + a * cam->du.y) + b * cam->dv.y,
               (cam->pixel00.z + a * cam->du.z) + b * cam->dv.z };
    dv3 v = { pc.x - cam->position.x, pc.y - cam->position.y, pc.z - cam->position.z };
    *d = normalize3(v);
    *o = cam->position;
}

int rto3d_camera_rays(const rto3d_scene *sc, const rto3d_params *p, int64_t n, const int32_t *pi,
                      const int32_t *pj, const int32_t *sample, double *o, double *d)
{
    camera3d cam;
    int rc = camera3d_init(sc, &cam);
    if (rc) return rc;
    for (int64_t k = 0; k < n; ++k) {
        dv3 oo, dd;
        camera3d_ray(&cam, sc, p->seed, pi[k], pj[k], (uint32_t)sample[k], &oo, &dd);
        o[3 * k] = oo.x; o[3 * k + 1] = oo.y; o[3 * k + 2] = oo.z;
        d[3 * k] = dd.x; d[3 * k + 1] = dd.y; d[3 * k + 2] = dd.z;
    }
    return 0;
}

static void pixel3d(const rto3d_scene *sc, const camera3d *cam, const rto3d_params *p, int i, int j, int s,
                    float rgb[3], uint64_t *rays)
{
    uint32_t gs = (uint32_t)(p->sample_offset + s);
    dv3 o, d;
    camera3d_ray(cam, sc, p->seed, i, j, gs, &o, &d);
    rng_stream rs = { { (uint32_t)p->seed, (uint32_t)(p->seed >> 32) }, (uint32_t)(sc->width * j + i), gs, 0u };
    path_info info = { 0, -1, -1.0 };
    trace_ray3d(sc, o, d, p->bounce_limit, p->bounce_limit, &rs, &info, rgb);
    *rays += (uint64_t)info.bounces;
}

/* World::render + compute_pixel, 3d/src/rt.rs:238-273: sum over spp, skip the
 * 10x10 corner, store at (i, H-1-j).                                          */
int rto3d_render(const rto3d_scene *sc, const rto3d_params *p, int32_t n_threads,
                 int32_t threads_over_spp, float *out_rgb, uint64_t *rays)
{
    camera3d cam;
    int rc = camera3d_init(sc, &cam);
    if (rc) return rc;
    if (n_threads < 1) n_threads = 1;
    uint64_t total_rays = 0;
    memset(out_rgb, 0, sizeof(float) * 3 * (size_t)sc->width * (size_t)sc->height);
    if (threads_over_spp) {
        for (int i = 0; i < sc->width; ++i) {
            for (int j = 0; j < sc->height; ++j) {
                if (i < 10 && j < 10) continue;
                float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma omp parallel for schedule(static) num_threads(n_threads) reduction(+ : r, g, b, total_rays)
                for (int s = 0; s < p->spp; ++s) {
                    float c[3];
                    uint64_t rr = 0;
                    pixel3d(sc, &cam, p, i, j, s, c, &rr);
                    r += c[0]; g += c[1]; b += c[2];
                    total_rays += rr;
                }
                size_t idx = ((size_t)sc->width * (size_t)(sc->height - j - 1) + (size_t)i) * 3;
                out_rgb[idx] = r; out_rgb[idx + 1] = g; out_rgb[idx + 2] = b;
            }
        }
    } else {
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : total_rays)
        for (int j = 0; j < sc->height; ++j) {
            for (int i = 0; i < sc->width; ++i) {
                if (i < 10 && j < 10) continue;
                float r = 0.0f, g = 0.0f, b = 0.0f;
                for (int s = 0; s < p->spp; ++s) {
                    float c[3];
                    uint64_t rr = 0;
                    pixel3d(sc, &cam, p, i, j, s, c, &rr);
                    r += c[0]; g += c[1]; b += c[2];
                    total_rays += rr;
                }
                size_t idx = ((size_t)sc->width * (size_t)(sc->height - j - 1) + (size_t)i) * 3;
                out_rgb[idx] = r; out_rgb[idx + 1] = g; out_rgb[idx + 2] = b;
            }
        }
    }
    if (rays) *rays = total_rays;
    return 0;
}
