#include "rt_amd.h"
#include <cstdio>
#include <vector>
extern "C" int orc_render(const rt_scene_desc*, const rt_settings*, int, int, int, int, int, int, int, int, uint8_t*, double*, rt_stats*, int);
int main(int argc, char** argv) {
    for (int i = 2; i < argc; i++) {
        rt_host_scene* hs = nullptr;
        if (rt_host_scene_load(argv[i], argv[1], &hs)) { std::printf("%s: %s\n", argv[i], rt_last_error()); continue; }
        for (unsigned flags : {RT_MODE_GPU_REF | RT_F_STRICT_TRAVERSAL, RT_MODE_CPU_REF, (RT_MODE_GPU_REF & ~RT_F_BEZIER_STACKED)}) {
            rt_settings st{}; st.dof_enabled = 1; st.spp = 2; st.focal_distance = 9; st.aperture = 1.5; st.seed = 5; st.flags = flags;
            std::vector<uint8_t> img(48 * 36 * 4);
            rt_stats s{};
            int rc = orc_render(rt_host_scene_desc(hs), &st, 48, 36, 48, 36, 0, 0, 48, 36, img.data(), nullptr, &s, 2);
            std::printf("%s flags %x: rc %d rays %llu\n", argv[i], flags, rc, (unsigned long long)(s.rays_primary + s.rays_secondary + s.rays_shadow));
        }
        rt_host_scene_free(hs);
    }
}
