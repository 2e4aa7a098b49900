// Headless command-line renderer: the knobs of the reference's main.cpp (CR/main.cpp:3-18) and of
// its settings panel (scene, viewport, texture size, samples, depth of field, output name;
// CR/Renderer.cpp:296-480) as flags.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "renderer.hpp"

static void usage() {
    std::fprintf(stderr,
                 "usage: rt_render --scene FILE [--assets DIR] [--out FILE.bmp] [--size WxH] [--viewport WxH]\n"
                 "                 [--spp N] [--dof APERTURE FOCAL] [--seed S] [--mode gpu|cpu] [--strict] [--device D] [--gpus N]\n"
                 "                 [--progressive N [--sequence PREFIX]]\n"
                 "  --gpus N          render whole frames on GPUs 0 .. N-1 (tiles interleaved over them, gathered on GPU 0)\n"
                 "  --progressive N   render the frame as N x N segments, one call each, with a progress line per segment\n"
                 "  --sequence PREFIX also write PREFIX_0001.bmp, ... after every row of segments\n");
}

int main(int argc, char** argv) {
    std::string scene, assets, out = "output.bmp", mode = "gpu", sequence;
    int tw = 1920, th = 1080, vw = 0, vh = 0, spp = 1, device = 0, progressive = 0, gpus = 1;
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
        else if (!std::strcmp(argv[i], "--gpus")) need(1), gpus = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--progressive")) need(1), progressive = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--sequence")) need(1), sequence = argv[++i];
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
        Renderer r(vw, vh, tw, th, scene, assets, device, gpus);
        r.seed = seed;
        r.mode = (mode == "cpu" ? RT_MODE_CPU_REF : RT_MODE_GPU_REF) | (strict ? RT_F_STRICT_TRAVERSAL : 0u);
        r.settings.resampling_size = spp;
        r.settings.depthOfField.enabled = dof;
        r.settings.depthOfField.aperture = aperture;
        r.settings.depthOfField.focalDistance = focal;
        r.outputName = out;
        if (progressive > 0) {
            // The reference's per-frame tile mode (CR/Renderer_CUDA.cpp:421-436) with its progress / time-left line
            // (CR/RayTracer.cu:120-131), to a file sequence instead of a window.
            r.segmentation = progressive;
            const auto t0 = std::chrono::steady_clock::now();
            const int total = progressive * progressive;
            double rays = 0, kernel_ms = 0;
            int done = 0, rows = 0;
            bool more = true;
            while (more) {
                more = r.RenderNextSegment();
                done++;
                rays += (double)(r.lastStats.rays_primary + r.lastStats.rays_secondary + r.lastStats.rays_shadow);
                kernel_ms += r.lastStats.kernel_ms;
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                const double frac = done < total ? (double)done / total : 1.0;
                std::fprintf(stderr, "\rprogress %5.1f%%  elapsed %.2f s  left %.2f s   ", 100.0 * frac, el, el / frac * (1.0 - frac));
                if (!sequence.empty() && (done % progressive == 0 || !more)) {
                    char name[32];
                    std::snprintf(name, sizeof name, "_%04d.bmp", ++rows);
                    r.outputName = sequence + name;
                    r.SaveTextureToBMP();
                }
            }
            std::fprintf(stderr, "\n");
            r.outputName = out;
            r.SaveTextureToBMP();
            std::printf("%s: %dx%d, %d spp, %d segments, %.0f rays, kernels %.2f ms (%.1f Mrays/s)\n", out.c_str(), tw, th, spp, done, rays,
                        kernel_ms, rays / (kernel_ms * 1e3));
            return 0;
        }
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
