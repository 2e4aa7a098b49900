#include "rt_amd.h"
#include <cstdio>
int main(int argc, char** argv) {
    int bad = 0;
    for (int i = 2; i < argc; i++) {
        rt_host_scene* hs = nullptr;
        int rc = rt_host_scene_load(argv[i], argv[1], &hs);
        if (rc) { std::printf("%s: rc %d %s\n", argv[i], rc, rt_last_error()); bad++; continue; }
        const rt_scene_desc* d = rt_host_scene_desc(hs);
        std::printf("%s: %d prims %d lights %u nodes %u leaves %d warnings\n", argv[i], d->n_primitives, d->n_lights, d->n_nodes, d->n_leaves, rt_host_scene_warning_count(hs));
        rt_host_scene_free(hs);
    }
    return 0;
}
