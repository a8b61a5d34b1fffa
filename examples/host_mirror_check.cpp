// Call semantics of the C++ host mirror that a rendered image does not show, checked on the GPU and printed as
// "name value" lines (tests/test_gpu_examples.py reads them):
//  - ReverseRenderer::trace_ray / rt3d World::trace_ray draw fresh random numbers per call, as the reference's
//    do (2d/src/rt_common.rs:376-391, 3d/src/rt.rs:40-45), and trace_ray_at reproduces a named stream;
//  - a renderer handed a different world in the same variable renders the new world (the reference borrows
//    `&World` per call, 2d/src/librt2d_reverse.rs:282-285, and its batch binary builds one per frame,
//    2d/src/bin/2d_reverse.rs:45-51).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../raytracer_amd/host/rt3d.hpp"
#include "../raytracer_amd/host/worlds.hpp"

int main()
{
    try {
        {
            using namespace rt2d_host;
            ReverseRenderParams params;
            params.width = 64; params.height = 48; params.spp = 4; params.lambda_samples = 2; params.recursion_limit = 10;
            params.px_to_world = 800.0 / 64.0;
            const Camera camera{ DVec2(800.0, 600.0) / 2.0, DVec2(800.0, 600.0) / 2.0 };
            ReverseRenderer renderer(params);
            const World stress = worlds::by_name("complex_world", camera, 0.0, 0, 1);
            int distinct = 0, repeat_equal = 0, first_is_stream0 = 0, n = 0;
            for (int k = 0; k < 64; ++k) {
                const Ray2d ray{ DVec2(400.0, 300.0), DVec2::from_angle(2.0 * PI * (k + 0.5) / 64.0) };
                ReverseRenderer fresh(params);
                const float first = fresh.trace_ray(stress, ray, 40, 560.0, 10);
                float sum_abs = 0.0f;
                bool differs = false;
                for (int c = 0; c < 8; ++c) {
                    const float v = fresh.trace_ray(stress, ray, 40, 560.0, 10);
                    differs = differs || v != first;
                    sum_abs += v;
                }
                (void)sum_abs;
                const float a = renderer.trace_ray_at(stress, ray, 40, 560.0, 10, 5), b = renderer.trace_ray_at(stress, ray, 40, 560.0, 10, 5);
                const float s0 = renderer.trace_ray_at(stress, ray, 40, 560.0, 10, 0);
                ++n;
                distinct += differs ? 1 : 0;
                repeat_equal += (std::memcmp(&a, &b, sizeof a) == 0) ? 1 : 0;
                first_is_stream0 += (std::memcmp(&first, &s0, sizeof s0) == 0) ? 1 : 0;
            }
            printf("trace_ray2d_rays %d\n", n);
            printf("trace_ray2d_rays_with_distinct_calls %d\n", distinct);
            printf("trace_ray2d_at_repeats %d\n", repeat_equal);
            printf("trace_ray2d_first_call_is_stream0 %d\n", first_is_stream0);

            // one variable, two worlds: the second render must show the second world
            std::vector<RawImage> got;
            for (int frame = 0; frame < 2; ++frame) {
                const World world = worlds::by_name("sample_world", camera, frame * 0.5, (uint64_t)frame, 1);
                printf("world_slot_%d %p objects %zu\n", frame, (const void *)&world, world.objects.size());
                got.push_back(renderer.global_render(world, params));
            }
            ReverseRenderer other(params);
            const World w1 = worlds::by_name("sample_world", camera, 0.5, 1, 1);
            const RawImage want1 = other.global_render(w1, params);
            // a pixel's samples meet in f32 atomics, so two renders of one world may differ in a last bit; two worlds
            // differ by whole objects
            auto max_diff = [](const RawImage &a, const RawImage &b) {
                float m = 0.0f;
                for (size_t i = 0; i < a.data.size(); ++i) {
                    m = std::fmax(m, std::fabs(a.data[i].value.x - b.data[i].value.x));
                    m = std::fmax(m, std::fabs(a.data[i].value.y - b.data[i].value.y));
                    m = std::fmax(m, std::fabs(a.data[i].value.z - b.data[i].value.z));
                }
                return m;
            };
            printf("rebind_frame1_matches_fresh_renderer %d\n", max_diff(got[1], want1) < 1e-4f ? 1 : 0);
            printf("rebind_frames_differ %d\n", max_diff(got[0], got[1]) > 1e-2f ? 1 : 0);
        }
        {
            using namespace rt3d_host;
            RenderParams params;
            Camera camera;
            std::vector<Sphere> objects;
            standalone_scene(64, 48, camera, objects);
            World world(camera, objects, params);
            const Ray ray{ camera.position, DVec3{ 0.0, -0.6, -0.8 } }; // down onto the ground sphere: a diffuse bounce follows
            const Color first = world.trace_ray_at(ray, 10, 0);
            int distinct = 0;
            for (int c = 0; c < 8; ++c) {
                const Color v = world.trace_ray(ray, 10);
                distinct += (v.x != first.x || v.y != first.y || v.z != first.z) ? 1 : 0;
            }
            const Color a = world.trace_ray_at(ray, 10, 3), b = world.trace_ray_at(ray, 10, 3);
            printf("trace_ray3d_distinct_of_8 %d\n", distinct);
            printf("trace_ray3d_at_repeats %d\n", (a.x == b.x && a.y == b.y && a.z == b.z) ? 1 : 0);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
