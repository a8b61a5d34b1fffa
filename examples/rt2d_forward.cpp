// Front-end with the parameters of the reference's forward-tracer binary (2d/src/bin/2d_forward.rs:15-47):
// 800x600, simple_world, one global_render, Reinhard{1.} PNG.
// Usage: rt2d_forward [world=simple_world] [out=out_forward.png] [loops=0]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../raytracer_amd/host/png.hpp"
#include "../raytracer_amd/host/worlds.hpp"

using namespace rt2d_host;

int main(int argc, char **argv)
{
    const std::string world_name = argc > 1 ? argv[1] : "simple_world";
    const std::string out = argc > 2 ? argv[2] : "out_forward.png";
    const size_t loops = argc > 3 ? (size_t)atoi(argv[3]) : 0;
    const int width = 800, height = 600;
    ForwardRenderParams params;
    params.width = width;
    params.height = height;
    params.stop_condition.max_loops = loops ? loops : 1;
    params.recursion_limit = 2;
    params.lambda_samples = 1;
    const Camera camera{ DVec2((double)width, (double)height) / 2.0, DVec2((double)width, (double)height) / 2.0 };
    try {
        const World world = worlds::by_name(world_name, camera, 0.0, 0, 1);
        ForwardRenderer renderer(params);
        const auto t0 = std::chrono::steady_clock::now();
        const RawImage raw = loops ? renderer.endless_render(world, [](size_t n, const RawImage &) { printf("loop %zu\n", n); return true; })
                                   : renderer.global_render(world);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::vector<float> rgb((size_t)width * height * 3);
        for (size_t i = 0; i < raw.data.size(); ++i) {
            rgb[3 * i] = raw.data[i].value.x; rgb[3 * i + 1] = raw.data[i].value.y; rgb[3 * i + 2] = raw.data[i].value.z;
        }
        std::vector<uint8_t> rgba((size_t)width * height * 4);
        if (rt_tonemap_reinhard(renderer.context(), rgb.data(), (int64_t)width * height, 1.0f, 255, rgba.data()) != RT_OK)
            throw RtError(rt_last_error());
        if (!rt_png::write_rgba8(out.c_str(), rgba.data(), width, height)) throw RtError("cannot write " + out);
        printf("elapsed = %.3f s\n", s);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
