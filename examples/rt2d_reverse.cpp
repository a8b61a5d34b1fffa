// Front-end with the parameters of the reference's batch renderer (2d/src/bin/2d_reverse.rs:15-58):
// 800x600, spp 500, recursion limit 10, 2 lambda samples, frames of sample_world(t), Reinhard{1.} PNGs.
// Usage: rt2d_reverse [frames=1] [spp=500] [world=sample_world] [width=800] [out_prefix=out_] [loops=0] [denoise=0] [device_loop=0]
// loops > 0 renders every frame with endless_render under StopCondition::MaxLoops(loops) (librt2d_reverse.rs:424-510)
// instead of one global_render; denoise = 1 adds the Denoiser the reference binary constructs (2d_reverse.rs:21-26);
// device_loop = 1 runs that loop with its state on the GPU (linear merging, see ReverseRenderer::endless_render_device).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../raytracer_amd/host/png.hpp"
#include "../raytracer_amd/host/worlds.hpp"

using namespace rt2d_host;

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 1;
    const size_t spp = argc > 2 ? (size_t)atoi(argv[2]) : 500;
    const std::string world_name = argc > 3 ? argv[3] : "sample_world";
    const int width = argc > 4 ? atoi(argv[4]) : 800;
    const std::string prefix = argc > 5 ? argv[5] : "out_";
    const size_t loops = argc > 6 ? (size_t)atoi(argv[6]) : 0;
    const bool with_denoiser = argc > 7 && atoi(argv[7]) != 0;
    const bool device_loop = argc > 8 && atoi(argv[8]) != 0;
    const int height = width * 3 / 4;

    ReverseRenderParams params;
    params.width = width;
    params.height = height;
    params.spp = spp;
    params.recursion_limit = 10;
    params.lambda_samples = 2;
    params.px_to_world = 800.0 / width;
    params.stop_condition.max_loops = loops;
    if (with_denoiser) {
        params.has_denoiser = true;
        params.denoiser.top = 0.01f;
        params.denoiser.oversampling_factor = 10.0f;
        params.denoiser.mask_size = 2;
        params.denoiser.passes = 1;
    }
    const Camera camera{ DVec2(800.0, 600.0) / 2.0, DVec2(800.0, 600.0) / 2.0 };
    try {
        ReverseRenderer renderer(params);
        const auto t0 = std::chrono::steady_clock::now();
        for (int idx = 0; idx < frames; ++idx) {
            printf("%d/%d\n", idx, frames - 1);
            const double t = (double)idx / (double)(frames > 1 ? frames : 60);
            const World world = worlds::by_name(world_name, camera, t, (uint64_t)idx, 1);
            auto progress = [](size_t n, const RawImage &) { printf("loop %zu\n", n); return true; };
            RawImage raw = loops == 0 ? renderer.global_render(world, params)
                           : device_loop ? renderer.endless_render_device(world, progress)
                                         : renderer.endless_render(world, progress);
            annotate(raw, camera, DVec2(0.0, 0.0)); // 2d_reverse.rs:52
            std::vector<float> rgb((size_t)width * height * 3);
            for (size_t i = 0; i < raw.data.size(); ++i) {
                rgb[3 * i] = raw.data[i].value.x; rgb[3 * i + 1] = raw.data[i].value.y; rgb[3 * i + 2] = raw.data[i].value.z;
            }
            std::vector<uint8_t> rgba((size_t)width * height * 4);
            if (rt_tonemap_reinhard(renderer.context(), rgb.data(), (int64_t)width * height, 1.0f, 255, rgba.data()) != RT_OK)
                throw RtError(rt_last_error());
            char name[256];
            snprintf(name, sizeof name, "%s%04d.png", prefix.c_str(), idx);
            if (!rt_png::write_rgba8(name, rgba.data(), width, height)) throw RtError(std::string("cannot write ") + name);
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rt_stats st{};
        rt_ctx_get_stats(renderer.context(), &st);
        printf("elapsed = %.3f s, %.1f Mrays/s in the trace kernel\n", s, (double)st.rays / st.kernel_ms / 1e3);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
