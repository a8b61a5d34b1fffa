// Host-side mirror of the 3d crate's interface (3d/src/rt.rs, 3d/src/img.rs,
// 3d/src/bin/standalone.rs) in C++; the tracing runs in the HIP library behind
// include/rt_abi.h.
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

namespace rt3d_host {

struct DVec3 { double x = 0.0, y = 0.0, z = 0.0; };
struct Color { float x = 0.0f, y = 0.0f, z = 0.0f; };

struct Ray { DVec3 origin, dir; };          // `Ray2d` in 3d/src/rt.rs:12-25
struct Sphere { DVec3 center; double radius = 0.0; }; // rt.rs:60-70; the only Object variant (rt.rs:47-50)

// Arguments of Camera::new (rt.rs:139-147); the derived pixel grid is computed by the library.
struct Camera {
    DVec3 position, dir, up;
    double focal_length = 1.0;
    int32_t image_width = 640, image_height = 480;
};

struct RenderParams { size_t spp = 1, bounce_limit = 10; uint64_t seed = 0; bool use_bvh = true; }; // rt.rs:194-197

// 3d/src/img.rs:19-33: flat RGB f32
struct RawImage {
    int32_t width = 0, height = 0;
    std::vector<float> data;
    RawImage(int32_t w, int32_t h) : width(w), height(h), data((size_t)w * (size_t)h * 3, 0.0f) {}
};

class RtError : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

// rt.rs:199-214
class World {
public:
    World(const Camera &camera, const std::vector<Sphere> &objects, const RenderParams &params, int device = 0)
        : camera_(camera), params_(params)
    {
        // the structs this file passes were laid out by the include/rt_abi.h it was compiled with
        if (rt_abi_version() != RT_ABI_VERSION) throw RtError("librt_amd.so was built from another generation of include/rt_abi.h");
        check(rt_ctx_create(device, &ctx_));
        std::vector<double> cx, cy, cz, r;
        for (const Sphere &s : objects) { cx.push_back(s.center.x); cy.push_back(s.center.y); cz.push_back(s.center.z); r.push_back(s.radius); }
        rt3d_scene_desc d{};
        d.n_spheres = (int32_t)objects.size();
        d.use_bvh = params.use_bvh ? 1 : 0;
        d.cx = cx.data(); d.cy = cy.data(); d.cz = cz.data(); d.r = r.data();
        const double p[3] = { camera.position.x, camera.position.y, camera.position.z };
        const double v[3] = { camera.dir.x, camera.dir.y, camera.dir.z }, u[3] = { camera.up.x, camera.up.y, camera.up.z };
        for (int a = 0; a < 3; ++a) { d.cam_pos[a] = p[a]; d.cam_dir[a] = v[a]; d.cam_up[a] = u[a]; }
        d.focal_length = camera.focal_length;
        d.width = camera.image_width; d.height = camera.image_height;
        check(rt3d_scene_create(ctx_, &d, &scene_)); // fails where Camera::new's asserts would panic
    }
    ~World()
    {
        if (scene_) rt3d_scene_destroy(scene_);
        if (ctx_) rt_ctx_destroy(ctx_);
    }
    World(const World &) = delete;
    World &operator=(const World &) = delete;

    // rt.rs:252-273
    RawImage render()
    {
        RawImage img(camera_.image_width, camera_.image_height);
        rt3d_render_params p{};
        p.spp = (int32_t)params_.spp; p.bounce_limit = (int32_t)params_.bounce_limit; p.seed = params_.seed;
        check(rt3d_render(ctx_, scene_, &p, img.data.data()));
        return img;
    }

    // rt.rs:216-236 (the object slice argument is the world's own list, as at every call site).  The reference draws
    // fresh random numbers for every call's bounces (rt.rs:41): each call here takes the next random stream of this
    // world; trace_ray_at names the stream, for a call that has to be reproduced.
    Color trace_ray(const Ray &ray, size_t bounce_limit) { return trace_ray_at(ray, bounce_limit, next_stream_++); }
    Color trace_ray_at(const Ray &ray, size_t bounce_limit, uint64_t stream)
    {
        const double o[3] = { ray.origin.x, ray.origin.y, ray.origin.z }, d[3] = { ray.dir.x, ray.dir.y, ray.dir.z };
        float rgb[3];
        check(rt3d_trace_batch(ctx_, scene_, 1, o, d, (int32_t)bounce_limit, params_.seed + stream * 0x9E3779B97F4A7C15ull, rgb,
                               nullptr, nullptr, nullptr));
        return Color{ rgb[0], rgb[1], rgb[2] };
    }

    rt_ctx *context() const { return ctx_; }

private:
    static void check(int32_t rc)
    {
        if (rc != RT_OK) throw RtError(std::string("rt_abi call failed (") + std::to_string(rc) + "): " + rt_last_error());
    }
    Camera camera_;
    RenderParams params_;
    rt_ctx *ctx_ = nullptr;
    rt3d_scene *scene_ = nullptr;
    uint64_t next_stream_ = 0; // random stream of the next trace_ray call
};

// The scene of 3d/src/bin/standalone.rs:21-49: a 6x6x6 grid of spheres r = 0.8
// with pitch 3, one sphere r = 5 and a ground sphere r = 10000; camera at
// (0,2,0) looking down -Z rotated pi/16 towards -X, focal length 2.
inline void standalone_scene(int32_t width, int32_t height, Camera &cam, std::vector<Sphere> &objects)
{
    const double a = 3.14159265358979323846 / 16.0;
    cam.position = DVec3{ 0.0, 2.0, 0.0 };
    cam.dir = DVec3{ -std::sin(a), -0.0, -std::cos(a) }; // -Z.rotate_towards(X, a)
    cam.up = DVec3{ 0.0, 1.0, 0.0 };
    cam.focal_length = 2.0;
    cam.image_width = width;
    cam.image_height = height;
    objects.clear();
    for (int i = -3; i < 3; ++i)
        for (int j = -3; j < 3; ++j)
            for (int k = -3; k < 3; ++k)
                objects.push_back(Sphere{ DVec3{ (double)i * 3.0, (double)j * 3.0, -20.0 + (double)k * 3.0 }, 0.8 });
    objects.push_back(Sphere{ DVec3{ 0.0, 5.0, -20.0 }, 5.0 });
    objects.push_back(Sphere{ DVec3{ 0.0, -10000.0, -20.0 }, 10000.0 });
}

} // namespace rt3d_host
