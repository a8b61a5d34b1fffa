// C entry points over the host-side scene builders, so that Python harnesses
// (tests, bench.py) feed the GPU library and the CPU checker with exactly the
// scenes the C++ front-ends use.  CPU only; no HIP here.
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>

#include "png.hpp"
#include "rt3d.hpp"
#include "worlds.hpp"

namespace {
thread_local std::string g_err;

struct World2dHandle {
    rt2d_host::World world;
    rt2d_host::FlatScene flat;
};

struct Scene3dHandle {
    rt3d_host::Camera cam;
    std::vector<rt3d_host::Sphere> spheres;
    std::vector<double> cx, cy, cz, r;
};
} // namespace

extern "C" {

const char *rth_last_error(void) { return g_err.c_str(); }

// Build one of the scenes of 2d/src/worlds.rs by name; camera as in
// 2d/src/bin/2d_reverse.rs:37-40 for a width x height world.
void *rth_world2d_build(const char *name, double t, uint64_t idx, uint64_t seed, double width, double height)
{
    try {
        rt2d_host::Camera cam{ rt2d_host::DVec2(width, height) / 2.0, rt2d_host::DVec2(width, height) / 2.0 };
        auto h = std::make_unique<World2dHandle>();
        h->world = rt2d_host::worlds::by_name(name ? name : "", cam, t, idx, seed);
        h->flat = rt2d_host::flatten(h->world);
        return h.release();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

// Pointers stay valid until rth_world2d_free.
int32_t rth_world2d_desc(void *handle, rt2d_scene_desc *out)
{
    if (!handle || !out) { g_err = "null argument"; return -1; }
    *out = static_cast<World2dHandle *>(handle)->flat.desc();
    return 0;
}

void rth_world2d_free(void *handle) { delete static_cast<World2dHandle *>(handle); }

// Scene and camera of 3d/src/bin/standalone.rs:21-49 at the given image size.
void *rth_scene3d_standalone(int32_t width, int32_t height, int32_t use_bvh, rt3d_scene_desc *out)
{
    if (!out) { g_err = "null argument"; return nullptr; }
    auto h = std::make_unique<Scene3dHandle>();
    rt3d_host::standalone_scene(width, height, h->cam, h->spheres);
    for (const auto &s : h->spheres) { h->cx.push_back(s.center.x); h->cy.push_back(s.center.y); h->cz.push_back(s.center.z); h->r.push_back(s.radius); }
    rt3d_scene_desc d{};
    d.n_spheres = (int32_t)h->spheres.size();
    d.use_bvh = use_bvh;
    d.cx = h->cx.data(); d.cy = h->cy.data(); d.cz = h->cz.data(); d.r = h->r.data();
    const double p[3] = { h->cam.position.x, h->cam.position.y, h->cam.position.z };
    const double v[3] = { h->cam.dir.x, h->cam.dir.y, h->cam.dir.z }, u[3] = { h->cam.up.x, h->cam.up.y, h->cam.up.z };
    for (int a = 0; a < 3; ++a) { d.cam_pos[a] = p[a]; d.cam_dir[a] = v[a]; d.cam_up[a] = u[a]; }
    d.focal_length = h->cam.focal_length;
    d.width = width; d.height = height;
    *out = d;
    return h.release();
}

void rth_scene3d_free(void *handle) { delete static_cast<Scene3dHandle *>(handle); }

int32_t rth_png_write(const char *path, const uint8_t *rgba8, int32_t width, int32_t height)
{
    if (!path || !rgba8 || width <= 0 || height <= 0) { g_err = "bad argument"; return -1; }
    if (!rt_png::write_rgba8(path, rgba8, width, height)) { g_err = std::string("cannot write ") + path; return -2; }
    return 0;
}

} // extern "C"
