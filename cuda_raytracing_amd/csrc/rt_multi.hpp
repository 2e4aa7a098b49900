// Multi-GPU frames inside one process: the image is cut into tile x tile squares, tile t (row-major) is rendered by
// member t % N of a group of scene replicas, one per GPU, and the members' tiles travel to the first GPU as one packed
// buffer each (hipMemcpyPeerAsync over xGMI: every peer has its own link to the first GPU, so the copies run side by
// side), where a kernel writes them into the frame.  The reference renders the whole frame with one call on one GPU
// (launchRayTraceKernelParallel, CR/RayTracer.cu:134-169); this is that call for N GPUs.  Also here: the pack / unpack
// kernels on their own (rt_pack_tiles / rt_unpack_tiles), which the one-process-per-GPU path uses around its gather.
// Included by rt_device.hip only.
#ifndef RT_MULTI_HPP
#define RT_MULTI_HPP
#include <thread>

// A member's packed buffer: its tiles in row-major tile order, each tile row by row (edge tiles are narrower / shorter).
// One block per tile of the member; its first thread finds where the tile starts in the packed buffer.
struct PackParams {
    int32_t tex_w, tex_h, tile, part, n_parts, tiles_x, n_tiles_total;
    uint32_t n_pixels;  // of this member
};
// packed index of the first pixel of tile t (which must belong to member `part`)
DEV uint32_t pack_tile_start(const PackParams& Q, int part, int t) {
    // Whole tile rows first (each holds the same number of the member's pixels except where the member owns the narrower
    // last column, and the last row is shorter), then the member's tiles to the left in the tile's own row.  Rows of tiles
    // are few (< 4100 at 32768 pixels and 8-pixel tiles), and the walk runs once per tile.
    const int tile = Q.tile, tx = Q.tiles_x;
    const int ty = (Q.tex_h + tile - 1) / tile;
    const int last_w = Q.tex_w - (tx - 1) * tile, last_h = Q.tex_h - (ty - 1) * tile;
    const int tj = t / tx, ti = t - tj * tx;
    uint32_t start = 0;
    for (int j = 0; j <= tj; j++) {
        const int th = j == ty - 1 ? last_h : tile;
        // member tiles in row j: columns i with (j * tx + i) % n_parts == part
        const int i0 = (part - (int)(((long long)j * tx) % Q.n_parts) + Q.n_parts) % Q.n_parts;
        if (i0 >= tx) continue;
        if (j == tj) {
            start += (uint32_t)th * (uint32_t)tile * (uint32_t)((ti - i0) / Q.n_parts);  // none of those is the last column
            break;
        }
        const int cnt = (tx - 1 - i0) / Q.n_parts + 1;
        const bool has_last = (i0 + (cnt - 1) * Q.n_parts) == tx - 1;
        start += (uint32_t)th * (uint32_t)((cnt - (has_last ? 1 : 0)) * tile + (has_last ? last_w : 0));
    }
    return start;
}
// UNPACK = false: frame -> packed, true: packed -> frame.  blockIdx.x = which of the member's tiles, blockIdx.y = which member
// (first_part + y; its pack starts at packed + y * stride_words).
template <bool UNPACK>
__global__ __launch_bounds__(256) void tiles_copy_kernel(PackParams Q, int first_part, unsigned long long stride_words, const uint32_t* src, uint32_t* dst) {
    const int part = first_part + (int)blockIdx.y;
    const long long t64 = (long long)part + (long long)blockIdx.x * Q.n_parts;
    if (t64 >= Q.n_tiles_total) return;
    const int t = (int)t64;
    __shared__ uint32_t s_start;
    if (threadIdx.x == 0) s_start = pack_tile_start(Q, part, t);
    __syncthreads();
    const int tj = t / Q.tiles_x, ti = t - tj * Q.tiles_x;
    const int x0 = ti * Q.tile, y0 = tj * Q.tile;
    const uint32_t w = (uint32_t)min(Q.tile, Q.tex_w - x0), h = (uint32_t)min(Q.tile, Q.tex_h - y0);
    const size_t pack0 = (size_t)blockIdx.y * stride_words + s_start;
    for (uint32_t p = threadIdx.x; p < w * h; p += 256u) {
        const uint32_t ry = p / w, rx = p - ry * w;
        const size_t at = (size_t)(y0 + (int)ry) * (size_t)Q.tex_w + (size_t)(x0 + (int)rx);
        if (UNPACK) dst[at] = src[pack0 + p];
        else dst[pack0 + p] = src[at];
    }
}
#define RT_UNPACK_PARTS 65535  // members one launch can serve (grid y)

static uint64_t member_pixels(int tex_w, int tex_h, int tile, int part, int n_parts) {
    const int tx = (tex_w + tile - 1) / tile, ty = (tex_h + tile - 1) / tile;
    uint64_t n = 0;
    for (int j = 0; j < ty; j++)
        for (int i = 0; i < tx; i++)
            if ((j * tx + i) % n_parts == part) n += (uint64_t)std::min(tile, tex_w - i * tile) * std::min(tile, tex_h - j * tile);
    return n;
}

static int pack_common(bool unpack, const void* src, void* dst, int tex_w, int tex_h, int tile, int part, int n_parts, uint64_t* n_out) {
    if (tex_w <= 0 || tex_h <= 0 || tex_w > 32768 || tex_h > 32768 || tile <= 0 || (tile & 7) || n_parts <= 0 || part < 0 || part >= n_parts)
        return rt_fail(RT_ERR_INVALID, "tile must be a positive multiple of 8, 0 <= part < n_parts, image size in range");
    PackParams Q;
    Q.tex_w = tex_w, Q.tex_h = tex_h, Q.tile = tile, Q.part = part, Q.n_parts = n_parts;
    Q.tiles_x = (tex_w + tile - 1) / tile;
    Q.n_tiles_total = Q.tiles_x * ((tex_h + tile - 1) / tile);
    const uint64_t n = member_pixels(tex_w, tex_h, tile, part, n_parts);
    Q.n_pixels = (uint32_t)n;
    if (n_out) *n_out = n;
    if (n == 0) return RT_OK;
    if (!src || !dst) return rt_fail(RT_ERR_INVALID, "null buffer");
    const dim3 g((unsigned)((Q.n_tiles_total - part + n_parts - 1) / n_parts)), t(256);  // the member's tiles
    if (unpack) hipLaunchKernelGGL(tiles_copy_kernel<true>, g, t, 0, g_stream, Q, part, 0ull, (const uint32_t*)src, (uint32_t*)dst);
    else hipLaunchKernelGGL(tiles_copy_kernel<false>, g, t, 0, g_stream, Q, part, 0ull, (const uint32_t*)src, (uint32_t*)dst);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

extern "C" long long rt_member_pixels(int tex_w, int tex_h, int tile, int part, int n_parts) {
    if (tex_w <= 0 || tex_h <= 0 || tile <= 0 || n_parts <= 0 || part < 0 || part >= n_parts) return -1;
    return (long long)member_pixels(tex_w, tex_h, tile, part, n_parts);
}
extern "C" int rt_pack_tiles(const uint8_t* image_device, int tex_w, int tex_h, int tile, int part, int n_parts, uint8_t* packed_device) {
    return pack_common(false, image_device, packed_device, tex_w, tex_h, tile, part, n_parts, nullptr);
}
extern "C" int rt_unpack_tiles(const uint8_t* packed_device, int tex_w, int tex_h, int tile, int part, int n_parts, uint8_t* image_device) {
    return pack_common(true, packed_device, image_device, tex_w, tex_h, tile, part, n_parts, nullptr);
}

extern "C" int rt_unpack_tiles_all(const uint8_t* packed_device, size_t stride_bytes, int tex_w, int tex_h, int tile, int first_part, int n_parts,
                                   uint8_t* image_device) {
    if (tex_w <= 0 || tex_h <= 0 || tex_w > 32768 || tex_h > 32768 || tile <= 0 || (tile & 7) || n_parts <= 0 || first_part < 0 || first_part > n_parts ||
        (stride_bytes & 3))
        return rt_fail(RT_ERR_INVALID, "tile must be a positive multiple of 8, 0 <= first_part <= n_parts, stride a multiple of 4, image size in range");
    if (first_part == n_parts) return RT_OK;
    if (!packed_device || !image_device) return rt_fail(RT_ERR_INVALID, "null buffer");
    PackParams Q;
    Q.tex_w = tex_w, Q.tex_h = tex_h, Q.tile = tile, Q.part = first_part, Q.n_parts = n_parts;
    Q.tiles_x = (tex_w + tile - 1) / tile;
    Q.n_tiles_total = Q.tiles_x * ((tex_h + tile - 1) / tile);
    Q.n_pixels = 0;
    for (int p = first_part; p < n_parts; p++)
        if (member_pixels(tex_w, tex_h, tile, p, n_parts) * 4 > stride_bytes) return rt_fail(RT_ERR_INVALID, "rt_unpack_tiles_all: a pack is longer than the stride");
    const unsigned tiles_most = (unsigned)((Q.n_tiles_total - first_part + n_parts - 1) / n_parts);  // the first of the parts owns the most tiles
    if (tiles_most == 0) return RT_OK;
    for (int p0 = first_part; p0 < n_parts; p0 += RT_UNPACK_PARTS) {
        const int here = std::min(RT_UNPACK_PARTS, n_parts - p0);
        hipLaunchKernelGGL(tiles_copy_kernel<true>, dim3(tiles_most, (unsigned)here), dim3(256), 0, g_stream, Q, p0, (unsigned long long)(stride_bytes / 4),
                           (const uint32_t*)packed_device + (size_t)(p0 - first_part) * (stride_bytes / 4), (uint32_t*)image_device);
        HIPCHK(hipGetLastError());
    }
    return RT_OK;
}

// ---- a scene on several GPUs ---------------------------------------------------------------------------------------
struct rt_scene_group {
    std::vector<int> devices;
    std::vector<rt_scene*> scenes;     // one replica per member
    std::vector<uint8_t*> image;       // per member: its frame (device memory on its GPU); member 0's is the assembled frame
    std::vector<uint8_t*> packed;      // per member > 0: its packed tiles on its own GPU
    std::vector<uint8_t*> landing;     // per member > 0: where those arrive on the first GPU (slices of one block, landing[1] is its start)
    size_t image_bytes = 0, packed_bytes = 0;
    std::vector<float> member_ms;      // kernel time of every member in the last frame
};

extern "C" int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" void rt_scene_group_destroy(rt_scene_group* g) {
    if (!g) return;
    for (size_t k = 0; k < g->devices.size(); k++) {
        (void)hipSetDevice(g->devices[k]);
        if (k < g->scenes.size() && g->scenes[k]) rt_scene_destroy(g->scenes[k]);
        if (k < g->image.size() && g->image[k]) (void)hipFree(g->image[k]);
        if (k < g->packed.size() && g->packed[k]) (void)hipFree(g->packed[k]);
    }
    if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
    if (g->landing.size() > 1 && g->landing[1]) (void)hipFree(g->landing[1]);
    delete g;
}

extern "C" int rt_scene_group_create(const rt_scene_desc* d, int n_gpus, const int* devices, rt_scene_group** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_scene_group_create: null out pointer");
    *out = nullptr;
    const int have = rt_device_count();
    if (have <= 0) return rt_fail(RT_ERR_NOGPU, "no HIP device visible");
    if (n_gpus <= 0 || n_gpus > 64) return rt_fail(RT_ERR_INVALID, "rt_scene_group_create: n_gpus must be in [1, 64]");
    rt_scene_group* g = new rt_scene_group;
    struct Guard {
        rt_scene_group* g;
        ~Guard() { if (g) rt_scene_group_destroy(g); }
    } guard{g};
    for (int k = 0; k < n_gpus; k++) {
        const int dev = devices ? devices[k] : k;
        if (dev < 0 || dev >= have) return rt_fail(RT_ERR_INVALID, "rt_scene_group_create: device " + std::to_string(dev) + " does not exist (" + std::to_string(have) + " visible)");
        g->devices.push_back(dev);
    }
    g->scenes.assign(n_gpus, nullptr), g->image.assign(n_gpus, nullptr), g->packed.assign(n_gpus, nullptr), g->landing.assign(n_gpus, nullptr);
    g->member_ms.assign(n_gpus, 0.0f);
    for (int k = 0; k < n_gpus; k++) {
        int rc = rt_init(g->devices[k]);
        if (rc) return rc;
        rc = rt_scene_create(d, &g->scenes[k]);
        if (rc) return rc;
    }
    // the first GPU reads nothing from its peers' memory (the peers push), but peer access lets the copies go GPU to GPU
    HIPCHK(hipSetDevice(g->devices[0]));
    for (int k = 1; k < n_gpus; k++) {
        if (g->devices[k] == g->devices[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g->devices[k], g->devices[0]) == hipSuccess && can) {
            (void)hipSetDevice(g->devices[k]);
            hipError_t e = hipDeviceEnablePeerAccess(g->devices[0], 0);
            if (e != hipSuccess) (void)hipGetLastError();  // already enabled, or not possible: the copy then goes through the host
        }
    }
    (void)hipSetDevice(g->devices[0]);
    guard.g = nullptr;
    *out = g;
    return RT_OK;
}

static int group_buffers(rt_scene_group* g, int tex_w, int tex_h, int tile) {
    const size_t bytes = (size_t)tex_w * tex_h * 4;
    const int n = (int)g->devices.size();
    size_t pk = 0;
    for (int k = 1; k < n; k++) pk = std::max<size_t>(pk, (size_t)member_pixels(tex_w, tex_h, tile, k, n) * 4);
    if (g->image_bytes >= bytes && g->packed_bytes >= pk) return RT_OK;
    for (int k = 0; k < n; k++) {
        HIPCHK(hipSetDevice(g->devices[k]));
        if (g->image[k]) (void)hipFree(g->image[k]);
        if (g->packed[k]) (void)hipFree(g->packed[k]);
        g->image[k] = g->packed[k] = nullptr;
    }
    HIPCHK(hipSetDevice(g->devices[0]));
    if (n > 1 && g->landing[1]) (void)hipFree(g->landing[1]);
    for (int k = 0; k < n; k++) g->landing[k] = nullptr;
    g->image_bytes = g->packed_bytes = 0;
    for (int k = 0; k < n; k++) {
        HIPCHK(hipSetDevice(g->devices[k]));
        HIPCHK(hipMalloc(&g->image[k], bytes));
        if (k > 0 && pk) HIPCHK(hipMalloc(&g->packed[k], pk));
    }
    HIPCHK(hipSetDevice(g->devices[0]));
    if (n > 1 && pk) {
        HIPCHK(hipMalloc(&g->landing[1], pk * (size_t)(n - 1)));
        for (int k = 2; k < n; k++) g->landing[k] = g->landing[1] + pk * (size_t)(k - 1);
    }
    g->image_bytes = bytes, g->packed_bytes = pk;
    return RT_OK;
}

// One frame on all GPUs of the group.  `rgba`: host memory, or device memory of the group's first GPU; tex_w * tex_h * 4
// bytes.  `stats`: counters summed over the members; kernel_ms = the slowest member's kernel time.
extern "C" int rt_render_frame_multi(rt_scene_group* g, const rt_settings* st, int tex_w, int tex_h, int cam_w, int cam_h, int tile,
                                     uint8_t* rgba, rt_stats* stats) {
    auto t0 = std::chrono::steady_clock::now();
    if (!g || !st || !rgba) return rt_fail(RT_ERR_INVALID, "rt_render_frame_multi: null argument");
    if (tile <= 0 || (tile & 7)) return rt_fail(RT_ERR_INVALID, "tile must be a positive multiple of 8");
    const int n = (int)g->devices.size();
    int rc = check_common(g->scenes[0], st, tex_w, tex_h);
    if (rc) return rc;
    rc = group_buffers(g, tex_w, tex_h, tile);
    if (rc) return rc;
    // one host thread per member: each selects its GPU, renders its tiles, packs them and pushes them to the first GPU
    std::vector<int> codes(n, RT_OK);
    std::vector<std::string> messages(n);
    std::vector<rt_stats> part_stats(n);
    auto member = [&](int k) {
        int rc = hipSetDevice(g->devices[k]) == hipSuccess ? RT_OK : rt_fail(RT_ERR_DEVICE, "hipSetDevice failed");
        std::memset(&part_stats[k], 0, sizeof(rt_stats));
        if (!rc) rc = rt_render_frame(g->scenes[k], st, tex_w, tex_h, cam_w, cam_h, tile, k, n, g->image[k], &part_stats[k]);
        if (!rc && k > 0) {
            uint64_t pixels = 0;
            rc = pack_common(false, g->image[k], g->packed[k], tex_w, tex_h, tile, k, n, &pixels);
            if (!rc && pixels) {
                hipError_t e = hipMemcpyPeerAsync(g->landing[k], g->devices[0], g->packed[k], g->devices[k], (size_t)pixels * 4, g_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
                if (e != hipSuccess) rc = rt_fail(RT_ERR_DEVICE, std::string("tile transfer to the first GPU: ") + hipGetErrorString(e));
            }
        }
        codes[k] = rc;
        if (rc) messages[k] = rt_last_error();
    };
    if (n == 1) {
        member(0);
    } else {
        std::vector<std::thread> threads;
        for (int k = 1; k < n; k++) threads.emplace_back(member, k);
        member(0);
        for (std::thread& t : threads) t.join();
    }
    for (int k = 0; k < n; k++)
        if (codes[k]) return rt_fail(codes[k], "GPU " + std::to_string(g->devices[k]) + ": " + messages[k]);
    // the first GPU writes the arrived tiles into its frame
    HIPCHK(hipSetDevice(g->devices[0]));
    if (n > 1 && g->packed_bytes) {
        rc = rt_unpack_tiles_all(g->landing[1], g->packed_bytes, tex_w, tex_h, tile, 1, n, g->image[0]);
        if (rc) return rc;
    }
    const size_t bytes = (size_t)tex_w * tex_h * 4;
    HIPCHK(hipMemcpyAsync(rgba, g->image[0], bytes, is_device_ptr(rgba) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (int k = 0; k < n; k++) {
            const rt_stats& p = part_stats[k];
            stats->rays_primary += p.rays_primary, stats->rays_secondary += p.rays_secondary, stats->rays_shadow += p.rays_shadow;
            stats->rays_shadow_traced += p.rays_shadow_traced, stats->nodes_visited += p.nodes_visited, stats->tris_tested += p.tris_tested;
            stats->shaded_hits += p.shaded_hits, stats->texture_fetches += p.texture_fetches, stats->pixels += p.pixels;
            stats->kernel_ms = std::max(stats->kernel_ms, p.kernel_ms);
            g->member_ms[k] = (float)p.kernel_ms;
        }
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
}

extern "C" int rt_scene_group_size(const rt_scene_group* g) { return g ? (int)g->devices.size() : 0; }
extern "C" double rt_scene_group_member_ms(const rt_scene_group* g, int member) {
    return (g && member >= 0 && member < (int)g->member_ms.size()) ? (double)g->member_ms[member] : 0.0;
}
#endif
