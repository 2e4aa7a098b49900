// Headless command-line renderer: the knobs of the reference's main.cpp (CR/main.cpp:3-18) and of
// its settings panel (scene, viewport, texture size, samples, depth of field, output name;
// CR/Renderer.cpp:296-480) as flags.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "renderer.hpp"

static void usage() {
    std::fprintf(stderr,
                 "usage: rt_render --scene FILE [--assets DIR] [--out FILE.bmp] [--size WxH] [--viewport WxH]\n"
                 "                 [--spp N] [--dof APERTURE FOCAL] [--seed S] [--mode gpu|cpu] [--strict] [--device D]\n");
}

int main(int argc, char** argv) {
    std::string scene, assets, out = "output.bmp", mode = "gpu";
    int tw = 1920, th = 1080, vw = 0, vh = 0, spp = 1, device = 0;
    bool dof = false, strict = false;
    double aperture = 10, focal = 10;
    unsigned long long seed = 1;
    for (int i = 1; i < argc; i++) {
        auto need = [&](int n) {
            if (i + n >= argc) {
                usage();
                std::exit(2);
            }
        };
        if (!std::strcmp(argv[i], "--scene")) need(1), scene = argv[++i];
        else if (!std::strcmp(argv[i], "--assets")) need(1), assets = argv[++i];
        else if (!std::strcmp(argv[i], "--out")) need(1), out = argv[++i];
        else if (!std::strcmp(argv[i], "--size")) need(1), std::sscanf(argv[++i], "%dx%d", &tw, &th);
        else if (!std::strcmp(argv[i], "--viewport")) need(1), std::sscanf(argv[++i], "%dx%d", &vw, &vh);
        else if (!std::strcmp(argv[i], "--spp")) need(1), spp = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--dof")) need(2), dof = true, aperture = std::atof(argv[i + 1]), focal = std::atof(argv[i + 2]), i += 2;
        else if (!std::strcmp(argv[i], "--seed")) need(1), seed = std::strtoull(argv[++i], nullptr, 10);
        else if (!std::strcmp(argv[i], "--mode")) need(1), mode = argv[++i];
        else if (!std::strcmp(argv[i], "--strict")) strict = true;
        else if (!std::strcmp(argv[i], "--device")) need(1), device = std::atoi(argv[++i]);
        else {
            usage();
            return 2;
        }
    }
    if (scene.empty()) {
        usage();
        return 2;
    }
    if (vw <= 0 || vh <= 0) vw = tw, vh = th;
    try {
        Renderer r(vw, vh, tw, th, scene, assets, device);
        r.seed = seed;
        r.mode = (mode == "cpu" ? RT_MODE_CPU_REF : RT_MODE_GPU_REF) | (strict ? RT_F_STRICT_TRAVERSAL : 0u);
        r.settings.resampling_size = spp;
        r.settings.depthOfField.enabled = dof;
        r.settings.depthOfField.aperture = aperture;
        r.settings.depthOfField.focalDistance = focal;
        r.outputName = out;
        r.Render();
        r.SaveTextureToBMP();
        const rt_stats& s = r.lastStats;
        const double rays = (double)(s.rays_primary + s.rays_secondary + s.rays_shadow);
        std::printf("%s: %dx%d, %d spp, %.0f rays, kernel %.2f ms (%.1f Mrays/s), total %.2f ms\n", out.c_str(), tw, th, spp, rays, s.kernel_ms,
                    rays / (s.kernel_ms * 1e3), s.total_ms);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rt_render: %s\n", e.what());
        return 1;
    }
    return 0;
}
