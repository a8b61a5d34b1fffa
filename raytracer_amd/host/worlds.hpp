// Scene builders with the geometry and materials of 2d/src/worlds.rs, producing
// rt2d_host::World.  They are inputs of the hot path, not part of it: the same
// flattened arrays feed the GPU library and, in the tests, the CPU checker.
#pragma once

#include "rt2d.hpp"

namespace rt2d_host {
namespace worlds {

using SC = SpectrumColor;

inline Spectrum em(SC c) { return Spectrum::emission_from_color(c); }
inline Spectrum ab(SC c) { return Spectrum::absorption_from_color(c); }
inline Object circle(double x, double y, double r, const Material &m) { return Object{ Circle{ DVec2(x, y), r }, m }; }

// Philox4x32-10 on the host, only to make complex_world reproducible (the
// reference draws it from an unseeded generator, worlds.rs:332-335).
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = { ctr[0], ctr[1], ctr[2], ctr[3] }, k[2] = { key[0], key[1] };
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n[4] = { (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], (uint32_t)p0 };
        for (int i = 0; i < 4; ++i) c[i] = n[i];
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}

inline double unit_f64(uint32_t lo, uint32_t hi) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53; }

// worlds.rs:245-265
inline World simple_world(const Camera &cam, double, uint64_t)
{
    World w{ cam, {} };
    w.objects.push_back(circle(cam.center.x, cam.center.y, 1.0, Material::emissive(em(SC::White) * 10.0f)));
    w.objects.push_back(Object::segment(DVec2(300, 200), DVec2(500, 200), Material::diffuse(em(SC::White))));
    return w;
}

inline void cornell_walls(World &w, const Camera &cam, SC right, SC left)
{
    w.objects.push_back(circle(cam.center.x, cam.center.y, 5.0, Material::emissive(em(SC::White) * 0.5f)));
    w.objects.push_back(Object::segment(DVec2(200, 100), DVec2(600, 100), Material::diffuse(em(SC::White))));
    w.objects.push_back(Object::segment(DVec2(600, 100), DVec2(600, 500), Material::diffuse(em(right))));
    w.objects.push_back(Object::segment(DVec2(200, 500), DVec2(200, 100), Material::diffuse(em(left))));
}

// worlds.rs:13-83 (its subsurface material is commented out upstream: light and three walls remain)
inline World cornell_box_sss(const Camera &cam, double, uint64_t)
{
    World w{ cam, {} };
    cornell_walls(w, cam, SC::White, SC::White);
    return w;
}

// worlds.rs:85-155: four absorbing Cauchy lenses
inline World cornell_box_absorption(const Camera &cam, double, uint64_t)
{
    World w{ cam, {} };
    cornell_walls(w, cam, SC::White, SC::White);
    const Ior glass = Ior::cauchy(1.45, 0.05);
    w.objects.push_back(circle(300, 200, 50, Material::dielectric(glass, ab(SC::White) * 0.01f)));
    w.objects.push_back(circle(500, 400, 50, Material::dielectric(glass, ab(SC::Cyan) * 0.01f)));
    w.objects.push_back(circle(300, 400, 50, Material::dielectric(glass, ab(SC::Cyan) * 0.05f)));
    w.objects.push_back(circle(500, 200, 50, Material::dielectric(glass, ab(SC::Orange) * 1.0f)));
    return w;
}

// worlds.rs:157-243: constant-ior vs dispersive lens
inline World cornell_box(const Camera &cam, double, uint64_t)
{
    World w{ cam, {} };
    cornell_walls(w, cam, SC::Red, SC::Blue);
    w.objects.push_back(circle(300, 200, 50, Material::diffuse(em(SC::White))));
    w.objects.push_back(circle(500, 400, 50, Material::dieletric(1.5, ab(SC::White))));
    w.objects.push_back(circle(300, 400, 50, Material::dielectric(Ior::cauchy(1.45, 0.05), ab(SC::White))));
    const int sides = 4;
    std::vector<DVec2> pts;
    for (int i = 0; i < sides; ++i)
        pts.push_back(DVec2::from_angle((double)i / (double)sides * 2.0 * PI) * 50.0 + DVec2(500, 200));
    std::vector<DVec2> rev(pts.rbegin(), pts.rend());
    w.objects.push_back(Object::from_points(rev, Material::diffuse(em(SC::White))));
    return w;
}

// worlds.rs:267-317
inline World spectrum_world(const Camera &cam, double, uint64_t)
{
    World w{ cam, {} };
    const double spacing = 80.0;
    w.objects.push_back(Object::segment(DVec2(0, 300), DVec2(800, 300), Material::diffuse(Spectrum())));
    const std::vector<SC> colors = iter_colors();
    for (size_t idx = 0; idx < colors.size(); ++idx) {
        const double fi = (double)idx;
        w.objects.push_back(circle(100.0 + fi * spacing, 150, 10, Material::emissive(em(colors[idx]) * 0.1f)));
        w.objects.push_back(circle(100.0 + fi * spacing, 450, 10, Material::emissive(em(SC::White) * 0.1f)));
        const Material wall = Material::diffuse(em(colors[idx]));
        const double x0 = 100.0 - spacing / 2.0 + fi * spacing, x1 = 100.0 - spacing / 2.0 + (double)(idx + 1) * spacing;
        w.objects.push_back(Object::segment(DVec2(x0, 400), DVec2(x1, 400), wall));
        w.objects.push_back(Object::segment(DVec2(x0, 600), DVec2(x0, 400), wall));
    }
    return w;
}

// worlds.rs:319-343, the README's stress test: 200 circles r = 5, uniform centre
// in [0,800) x [0,600), material uniform among {mirror, dim white light, white
// diffuse, dispersive glass}.  `seed` replaces the reference's unseeded generator:
// circle i takes its material from word 0 of Philox(counter {i,0,0,0}) and its
// centre from the two f64 of Philox(counter {i,1,0,0}).
inline World complex_world(const Camera &cam, double, uint64_t, uint64_t seed = 1)
{
    World w{ cam, {} };
    const Material mats[4] = { Material::reflective(), Material::emissive(em(SC::White) * 0.06f), Material::diffuse(em(SC::White)),
                               Material::dielectric(Ior::cauchy(1.45, 0.0354), em(SC::White)) };
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (uint32_t i = 0; i < 200; ++i) {
        uint32_t c0[4] = { i, 0, 0, 0 }, c1[4] = { i, 1, 0, 0 }, a[4], b[4];
        philox4x32_10(c0, key, a);
        philox4x32_10(c1, key, b);
        const int m = (int)(((uint64_t)a[0] * 4u) >> 32);
        w.objects.push_back(circle(unit_f64(b[0], b[1]) * 800.0, unit_f64(b[2], b[3]) * 600.0, 5.0, mats[m]));
    }
    return w;
}

// worlds.rs:345-541: the gallery's sample scene, animated by t
inline World sample_world(const Camera &cam, double t, uint64_t)
{
    World w{ cam, {} };
    const Material glass = Material::dielectric(Ior::cauchy(1.45, 0.0354), ab(SC::White));
    w.objects.push_back(circle(600, 400, 20, Material::emissive(em(SC::White) * 0.1f)));
    double angle = 2.0 * PI * t * 2.0;
    const double phi = PI * 2.0 / 3.0;
    w.objects.push_back(circle(600.0 + 50.0 * std::sin(angle + 0.0 * phi), 400.0 + 50.0 * std::cos(angle + 0.0 * phi), 10, glass));
    w.objects.push_back(circle(600.0 + 50.0 * std::sin(angle + 1.0 * phi), 400.0 + 50.0 * std::cos(angle + 1.0 * phi), 10,
                               Material::diffuse(em(SC::Red))));
    const double phi_2 = phi / 10.0;
    const DVec2 ma(600.0 + 50.0 * std::sin(angle + 2.0 * phi - phi_2), 400.0 + 50.0 * std::cos(angle + 2.0 * phi - phi_2));
    const DVec2 mb(600.0 + 50.0 * std::sin(angle + 2.0 * phi + phi_2), 400.0 + 50.0 * std::cos(angle + 2.0 * phi + phi_2));
    w.objects.push_back(Object{ Segment(ma, mb), Material::reflective() });
    w.objects.push_back(circle(50.0 + std::sin(angle) * 100.0, 100, 20, Material::emissive(em(SC::White) * 0.1f)));

    // wavy strip: degree-29 Bezier through 30 points, 99 segments
    const double start_x = 400.0, end_x = 800.0, y = 100.0;
    const int n_points = 30;
    std::vector<DVec2> pts;
    for (int i = 0; i < n_points; ++i) {
        const double f = (double)i / (double)n_points;
        const double x = start_x + (end_x - start_x) * f;
        const double dy = std::sin((f + t) * 2.0 * PI * 2.0) * 10.0;
        pts.push_back(DVec2(x, y + dy));
    }
    w.objects.push_back(Object{ Strip{ Bezier{ pts }.as_segments(100) },
                                Material::dielectric(Ior::cauchy(1.45, 0.05), ab(SC::White)) });

    const Material white = Material::diffuse(em(SC::White));
    w.objects.push_back(Object{ Segment(DVec2(400, 0), DVec2(800, 0)), white });
    w.objects.push_back(Object{ Segment(DVec2(0, 250), DVec2(100, 250)), white });
    w.objects.push_back(Object{ Segment(DVec2(110, 250), DVec2(200, 250)), white });
    w.objects.push_back(Object{ Segment(DVec2(200, 250), DVec2(200, 500)), white });

    // rotating box, one colour per side
    const DVec2 center(350, 250);
    const double r = 50.0;
    angle = 2.0 * PI * t;
    const double d_phi = 2.0 * PI / 4.0;
    const SC side_colors[4] = { SC::Red, SC::Green, SC::Blue, SC::Yellow };
    for (int k = 0; k < 4; ++k)
        w.objects.push_back(Object::segment(center + DVec2::from_angle(angle + (double)(k + 1) * d_phi) * r,
                                            center + DVec2::from_angle(angle + (double)k * d_phi) * r,
                                            Material::diffuse(em(side_colors[k]))));

    w.objects.push_back(Object::from_points({ DVec2(50, 300), DVec2(200, 300), DVec2(125, 350) }, glass));
    w.objects.push_back(Object{ Segment(DVec2(720, 480), DVec2(700, 500)), Material::reflective() });
    w.objects.push_back(Object::bezier({ DVec2(200, 50), DVec2(300, 50), DVec2(300, 150) }, 4, Material::reflective()));
    return w;
}

inline World by_name(const std::string &name, const Camera &cam, double t, uint64_t idx, uint64_t seed)
{
    if (name == "sample_world") return sample_world(cam, t, idx);
    if (name == "complex_world") return complex_world(cam, t, idx, seed);
    if (name == "cornell_box") return cornell_box(cam, t, idx);
    if (name == "cornell_box_absorption") return cornell_box_absorption(cam, t, idx);
    if (name == "cornell_box_sss") return cornell_box_sss(cam, t, idx);
    if (name == "spectrum_world") return spectrum_world(cam, t, idx);
    if (name == "simple_world") return simple_world(cam, t, idx);
    throw std::invalid_argument("unknown world: " + name);
}

} // namespace worlds
} // namespace rt2d_host
