// Host code of the device library under AddressSanitizer (the kernels themselves cannot be instrumented on this pool): every
// host-side path of the C ABI once, on the GPU box - scene creation with the three tree builders (with the tree check), tile /
// frame / no-allocation / several-GPUs-in-one-process renders, pruned and in reference order, with and without the
// three-kernel render, the pack / unpack kernels, the primary-ray probe, error returns.  Images of the variants must agree.
// Built and run by tools/sanitize_device.sh.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

#include "rt_amd.h"

#define CHECK(x)                                                                     \
    do {                                                                             \
        if ((x) != 0) {                                                              \
            std::printf("FAILED %s:%d %s: %s\n", __FILE__, __LINE__, #x, rt_last_error()); \
            std::fflush(stdout);                                                     \
            return 1;                                                                \
        }                                                                            \
    } while (0)
#define EXPECT(x)                                                  \
    do {                                                           \
        if (!(x)) {                                                \
            std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #x); \
            std::fflush(stdout);                                   \
            return 1;                                              \
        }                                                          \
    } while (0)

static int run_scene(const char* path, const char* assets) {
    rt_host_scene* hs = nullptr;
    CHECK(rt_host_scene_load(path, assets, &hs));
    const rt_scene_desc* d = rt_host_scene_desc(hs);
    const int W = 96, H = 72;
    rt_settings st{};
    st.dof_enabled = 1, st.spp = 3, st.focal_distance = 9, st.aperture = 1.5, st.seed = 5, st.flags = RT_MODE_GPU_REF;
    std::vector<uint8_t> first;
    for (int builder = 0; builder < 3; builder++) {
        for (int defer = 0; defer < 2; defer++) {
            setenv("RT_AMD_DEFER", defer ? "1" : "0", 1);
            CHECK(rt_set_tree_builder(builder));
            rt_scene* s = nullptr;
            CHECK(rt_scene_create(d, &s));
            int trees = 0;
            double cost = 0;
            CHECK(rt_scene_check_trees(s, &trees, &cost));
            EXPECT(trees == d->n_meshes);
            std::vector<uint8_t> img((size_t)W * H * 4), img2(img.size()), img3(img.size());
            rt_stats a{}, b{}, c{};
            CHECK(rt_render_frame(s, &st, W, H, W, H, 32, 0, 1, img.data(), &a));
            st.flags |= RT_F_STRICT_TRAVERSAL;
            CHECK(rt_render_tile(s, &st, W, H, W, H, 0, 0, W, H, img2.data(), &b));
            st.flags &= ~RT_F_STRICT_TRAVERSAL;
            EXPECT(img == img2);
            EXPECT(a.rays_primary + a.rays_secondary + a.rays_shadow == b.rays_primary + b.rays_secondary + b.rays_shadow);
            if (first.empty()) first = img;
            EXPECT(img == first);
            // two parts, second call reuses the block list of the first
            CHECK(rt_render_frame(s, &st, W, H, W, H, 16, 0, 2, img3.data(), nullptr));
            CHECK(rt_render_frame(s, &st, W, H, W, H, 16, 1, 2, img3.data(), nullptr));
            EXPECT(img3 == first);
            // device destination + the entry point that allocates nothing, a larger and then a smaller rectangle
            uint8_t* dev = nullptr;
            EXPECT(hipMalloc(&dev, img.size()) == hipSuccess);
            EXPECT(hipMemset(dev, 0, img.size()) == hipSuccess);
            CHECK(rt_render_tile_noalloc(s, &st, W, H, W, H, 0, 0, W, H, dev, &c));
            CHECK(rt_render_tile_noalloc(s, &st, W, H, W, H, 8, 8, 40, 24, dev, &c));
            EXPECT(hipMemcpy(img3.data(), dev, img.size(), hipMemcpyDeviceToHost) == hipSuccess);
            EXPECT(img3 == first);
            // pack / unpack round trip of part 1 of 3
            const long long n = rt_member_pixels(W, H, 16, 1, 3);
            EXPECT(n > 0);
            uint8_t *pack = nullptr, *back = nullptr;
            EXPECT(hipMalloc(&pack, (size_t)n * 4) == hipSuccess && hipMalloc(&back, img.size()) == hipSuccess);
            EXPECT(hipMemset(back, 0, img.size()) == hipSuccess);
            CHECK(rt_pack_tiles(dev, W, H, 16, 1, 3, pack));
            CHECK(rt_unpack_tiles(pack, W, H, 16, 1, 3, back));
            EXPECT(hipDeviceSynchronize() == hipSuccess);
            (void)hipFree(pack), (void)hipFree(back), (void)hipFree(dev);
            // probe
            std::vector<int32_t> prim((size_t)W * H), leaf((size_t)W * H);
            std::vector<double> t((size_t)W * H);
            std::vector<uint32_t> log((size_t)W * H * 8), cnt((size_t)W * H);
            CHECK(rt_trace_primary(s, &st, W, H, W, H, 0, 0, W, H, prim.data(), t.data(), leaf.data(), log.data(), cnt.data(), 8));
            // error returns must leave everything usable
            EXPECT(rt_render_tile(s, &st, W, H, W, H, -1, 0, W, H, img.data(), nullptr) != 0);
            EXPECT(rt_render_tile(s, nullptr, W, H, W, H, 0, 0, W, H, img.data(), nullptr) != 0);
            EXPECT(rt_render_frame(s, &st, W, H, W, H, 32, 3, 2, img.data(), nullptr) != 0);
            rt_scene_destroy(s);
        }
    }
    setenv("RT_AMD_DEFER", "0", 1);
    CHECK(rt_set_tree_builder(0));
    // several members on the one GPU: partition, pack, peer copy, unpack
    for (int members : {1, 2, 5}) {
        std::vector<int> devs((size_t)members, 0);
        rt_scene_group* g = nullptr;
        CHECK(rt_scene_group_create(d, members, devs.data(), &g));
        EXPECT(rt_scene_group_size(g) == members);
        std::vector<uint8_t> img((size_t)W * H * 4);
        rt_stats s{};
        CHECK(rt_render_frame_multi(g, &st, W, H, W, H, 16, img.data(), &s));
        EXPECT(img == first);
        CHECK(rt_render_frame_multi(g, &st, W, H, W, H, 32, img.data(), &s));
        EXPECT(img == first);
        rt_scene_group_destroy(g);
    }
    {
        rt_scene_group* g = nullptr;
        const int bad[2] = {0, 63};
        EXPECT(rt_scene_group_create(d, 2, bad, &g) != 0 && g == nullptr);
    }
    rt_host_scene_free(hs);
    std::printf("%s: ok\n", path);
    std::fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    CHECK(rt_init(0));
    for (int i = 2; i < argc; i++)
        if (run_scene(argv[i], argv[1])) return 1;
    std::printf("device paths under the host sanitizer: clean\n");
    std::fflush(stdout);
    _exit(0);  // not through the exit handlers: the sanitizer's runtime and the HSA runtime's teardown do not get along
}
