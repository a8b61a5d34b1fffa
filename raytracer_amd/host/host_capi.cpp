// C entry points over the host-side scene builders, so that Python harnesses
// (tests, bench.py) feed the GPU library and the CPU checker with exactly the
// scenes the C++ front-ends use.  CPU only; no HIP here.
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>

#include "checkpoint.hpp"
#include "png.hpp"
#include "rt3d.hpp"
#include "worlds.hpp"

namespace {
thread_local std::string g_err;

struct World2dHandle {
    rt2d_host::World world;
    rt2d_host::FlatScene flat;
    rt2d_host::RawImage image{ 0, 0 }; // the RawImage of a loaded checkpoint
    std::vector<uint8_t> encoded;      // result of the last size query of rth_world2d_encode
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

// ---- the reference's checkpoint format (2d/src/bin/gui.rs:24-45), see checkpoint.hpp ----
// A World (is_checkpoint = 0) or a CheckPoint { world, raw_image } (1) from MessagePack bytes.
void *rth_world2d_decode(const uint8_t *data, uint64_t size, int32_t is_checkpoint)
{
    try {
        if (!data) throw std::invalid_argument("null data");
        auto h = std::make_unique<World2dHandle>();
        if (is_checkpoint) {
            rt2d_host::checkpoint::CheckPoint c = rt2d_host::checkpoint::decode_checkpoint(data, (size_t)size);
            h->world = std::move(c.world);
            h->image = std::move(c.raw_image);
        } else {
            h->world = rt2d_host::checkpoint::decode_world(data, (size_t)size);
        }
        h->flat = rt2d_host::flatten(h->world);
        return h.release();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

void *rth_checkpoint_load(const char *path)
{
    try {
        auto h = std::make_unique<World2dHandle>();
        rt2d_host::checkpoint::CheckPoint c = rt2d_host::checkpoint::load_checkpoint(path ? path : "");
        h->world = std::move(c.world);
        h->image = std::move(c.raw_image);
        h->flat = rt2d_host::flatten(h->world);
        return h.release();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

// The world's MessagePack bytes (with_image = 0) or a whole CheckPoint with the given RawImage planes
// (rgb [w*h][3], weight [w*h]).  Returns the encoded size; copies at most `cap` bytes into out.  With out = NULL
// the bytes are kept and handed to the next call with a buffer (size query, then fetch, with the same arguments).
int64_t rth_world2d_encode(void *handle, int32_t with_image, const float *rgb, const float *weight, int32_t w, int32_t h,
                           uint8_t *out, uint64_t cap)
{
    try {
        if (!handle) throw std::invalid_argument("null handle");
        auto *wh = static_cast<World2dHandle *>(handle);
        if (out && !wh->encoded.empty()) { // the size query just before encoded the same arguments
            const std::vector<uint8_t> bytes = std::move(wh->encoded);
            wh->encoded.clear();
            memcpy(out, bytes.data(), (size_t)std::min<uint64_t>(cap, bytes.size()));
            return (int64_t)bytes.size();
        }
        std::vector<uint8_t> bytes;
        if (with_image) {
            if (!rgb || !weight || w < 0 || h < 0) throw std::invalid_argument("bad image");
            rt2d_host::checkpoint::CheckPoint c;
            c.world = wh->world;
            c.raw_image = rt2d_host::RawImage(w, h);
            for (size_t i = 0; i < c.raw_image.data.size(); ++i) {
                c.raw_image.data[i].weight = weight[i];
                c.raw_image.data[i].value = rt2d_host::Vec3{ rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] };
            }
            bytes = rt2d_host::checkpoint::encode_checkpoint(c);
        } else {
            bytes = rt2d_host::checkpoint::encode_world(wh->world);
        }
        if (out) memcpy(out, bytes.data(), (size_t)std::min<uint64_t>(cap, bytes.size()));
        const int64_t n = (int64_t)bytes.size();
        if (!out) wh->encoded = std::move(bytes); // kept for the call that fetches them
        return n;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Size and planes of a loaded checkpoint's RawImage; rgb / weight nullable.
int32_t rth_world2d_image(void *handle, int32_t *w, int32_t *h, float *rgb, float *weight)
{
    if (!handle || !w || !h) { g_err = "null argument"; return -1; }
    const auto *wh = static_cast<World2dHandle *>(handle);
    *w = wh->image.width; *h = wh->image.height;
    for (size_t i = 0; i < wh->image.data.size(); ++i) {
        if (weight) weight[i] = wh->image.data[i].weight;
        if (rgb) { rgb[3 * i] = wh->image.data[i].value.x; rgb[3 * i + 1] = wh->image.data[i].value.y; rgb[3 * i + 2] = wh->image.data[i].value.z; }
    }
    return 0;
}

// camera.center, camera.size
int32_t rth_world2d_camera(void *handle, double out[4])
{
    if (!handle || !out) { g_err = "null argument"; return -1; }
    const auto &c = static_cast<World2dHandle *>(handle)->world.camera;
    out[0] = c.center.x; out[1] = c.center.y; out[2] = c.size.x; out[3] = c.size.y;
    return 0;
}

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
