// Front-end with the scene and parameters of 3d/src/bin/standalone.rs:20-72: 640x480, 218 spheres, spp 512,
// bounce limit 10, Reinhard tone mapping (alpha 254), one PNG.
// Usage: rt3d_standalone [spp=512] [width=640] [height=480] [bvh=1] [out=out_foc=2_idx=0.png]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../raytracer_amd/host/png.hpp"
#include "../raytracer_amd/host/rt3d.hpp"

using namespace rt3d_host;

int main(int argc, char **argv)
{
    RenderParams params;
    params.spp = argc > 1 ? (size_t)atoi(argv[1]) : 512;
    params.bounce_limit = 10;
    const int width = argc > 2 ? atoi(argv[2]) : 640, height = argc > 3 ? atoi(argv[3]) : 480;
    params.use_bvh = argc > 4 ? atoi(argv[4]) != 0 : true;
    const std::string out = argc > 5 ? argv[5] : "out_foc=2_idx=0.png";
    Camera camera;
    std::vector<Sphere> objects;
    standalone_scene(width, height, camera, objects);
    try {
        World world(camera, objects, params);
        const auto t0 = std::chrono::steady_clock::now();
        printf("Rendering...\n");
        const RawImage raw = world.render();
        printf("%.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        std::vector<uint8_t> rgba((size_t)width * height * 4);
        if (rt_tonemap_reinhard(world.context(), raw.data.data(), (int64_t)width * height, 1.0f, 254, rgba.data()) != RT_OK)
            throw RtError(rt_last_error());
        if (!rt_png::write_rgba8(out, rgba.data(), width, height)) throw RtError("cannot write " + out);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
