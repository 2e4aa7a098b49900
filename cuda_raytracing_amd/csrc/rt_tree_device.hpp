// Device builder of the four-wide tree: kernels and the host routine that runs them after the scene slab has
// been uploaded.  Included by rt_device.hip only (one translation unit: it uses that file's HIPCHK, g_stream and
// the Tree4 / TreeJob records of rt_scene_build.hpp).
#ifndef RT_TREE_DEVICE_HPP
#define RT_TREE_DEVICE_HPP

// ---- four-wide tree built on the device ---------------------------------------------------------
// The reference builds its tree on the CPU (median split with std::sort, CR/primitive.cpp:552-616) and the
// pruned walk accepts any tree that loses no triangle (see mesh_walk4), so this builder is free to be simple:
// the triangles are sorted by the Morton code of their centres and the sorted run is cut top-down where the code changes
// in its highest varying bit (a linear BVH, collapsed two binary levels at a time), at most max_levels levels deep.
// Boxes are single precision, rounded outward, padded like the host builder's.
DEV uint32_t morton_spread10(uint32_t v) {
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu, v = (v | (v << 8)) & 0x0300F00Fu, v = (v | (v << 4)) & 0x030C30C3u, v = (v | (v << 2)) & 0x09249249u;
    return v;
}
struct TriBounds {
    double lo[3], hi[3];
};
DEV TriBounds tri_bounds(const rt_bvh_leaf& lf) {
    TriBounds b;
    const double v[3][3] = {{lf.O1.x, lf.O1.y, lf.O1.z}, {lf.O2.x, lf.O2.y, lf.O2.z}, {lf.O3.x, lf.O3.y, lf.O3.z}};
#pragma unroll
    for (int k = 0; k < 3; k++) b.lo[k] = fmin(v[0][k], fmin(v[1][k], v[2][k])), b.hi[k] = fmax(v[0][k], fmax(v[1][k], v[2][k]));
    return b;
}
// keys[i] = Morton code of triangle i of the mesh << 32 | i; the tail up to a power of two sorts last
__global__ void tree_keys_kernel(const rt_bvh_leaf* leaves, const uint32_t* mesh_leaves, uint32_t n, uint32_t n_pad, d3 lo, d3 inv_ext,
                                 unsigned long long* keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    if (i >= n) {
        keys[i] = ~0ull;
        return;
    }
    const TriBounds b = tri_bounds(leaves[mesh_leaves[i]]);
    const double c[3] = {0.5 * (b.lo[0] + b.hi[0]), 0.5 * (b.lo[1] + b.hi[1]), 0.5 * (b.lo[2] + b.hi[2])};
    const double l3[3] = {lo.x, lo.y, lo.z}, e3[3] = {inv_ext.x, inv_ext.y, inv_ext.z};
    uint32_t code = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double q = (c[k] - l3[k]) * e3[k] * 1024.0;
        q = q >= 0.0 ? q : 0.0;  // also catches NaN
        const uint32_t cell = q < 1023.0 ? (uint32_t)q : 1023u;
        code |= morton_spread10(cell) << k;
    }
    keys[i] = ((unsigned long long)code << 32) | i;
}
// one compare-exchange pass of a bitonic sorting network over n_pad = 2^m keys
__global__ void tree_sort_pass_kernel(unsigned long long* keys, uint32_t n_pad, uint32_t j, uint32_t k) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    const uint32_t p = i ^ j;
    if (p <= i) return;
    const unsigned long long a = keys[i], b = keys[p];
    const bool up = (i & k) == 0;
    if ((a > b) == up) keys[i] = b, keys[p] = a;
}
// single-precision box of the triangle at every sorted position, and its leaf index
__global__ void tree_boxes_kernel(const rt_bvh_leaf* leaves, const uint32_t* mesh_leaves, const unsigned long long* keys, uint32_t n,
                                  float* boxes, uint32_t* sorted_leaf) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t leaf = mesh_leaves[(uint32_t)keys[p]];
    const TriBounds b = tri_bounds(leaves[leaf]);
    sorted_leaf[p] = leaf;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // a triangle accepted by the double-precision test may lie a rounding error outside its exact bounds
        const double pad = 1e-9 * (1.0 + fabs(b.lo[k]) + fabs(b.hi[k])) + 1e-7 * (b.hi[k] - b.lo[k]);
        boxes[p * 6 + k] = __double2float_rd(b.lo[k] - pad);
        boxes[p * 6 + 3 + k] = __double2float_ru(b.hi[k] + pad);
    }
}
// Work list of the top-down pass: the interior nodes, level after level (level l = items [level_off[l], level_off[l + 1])).
struct TreeItem {
    uint32_t node, b, e;  // DevNode4 index, and its run [b, e) of sorted triangles (e - b >= 2)
};
struct TreeState {
    uint32_t next_node, tail;  // next free DevNode4 index, items appended so far
    uint32_t level_off[32];
};
// first index in [b, e) whose Morton code has the highest bit in which the codes of the run differ; the middle if all agree
DEV uint32_t tree_split_point(const unsigned long long* keys, uint32_t b, uint32_t e) {
    const uint32_t cb = (uint32_t)(keys[b] >> 32), ce = (uint32_t)(keys[e - 1] >> 32);
    if (cb == ce) return b + (e - b) / 2;
    const uint32_t bit = 1u << (31 - __clz(cb ^ ce));
    uint32_t lo = b, hi = e - 1;  // code[lo] has the bit clear, code[hi] has it set
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((uint32_t)(keys[mid] >> 32) & bit) hi = mid;
        else lo = mid;
    }
    return hi;
}
// One level of the top-down pass, one thread per interior node: its run is cut where the Morton code changes in its
// highest varying bit, and both halves once more, which gives up to four children (cells of the Morton grid, like a
// linear BVH collapsed two levels at a time).  A child may not hold more triangles than the levels left below it can
// take (4 per node of the last level); a cut that breaks that rule is replaced by four equal parts.
__global__ void tree_split_kernel(DevNode4* nodes, TreeItem* items, TreeState* st, int level, int max_levels, const unsigned long long* keys,
                                  const float* boxes, const uint32_t* sorted_leaf) {
    const uint32_t i = st->level_off[level] + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= st->level_off[level + 1]) return;
    const TreeItem it = items[i];
    const unsigned long long child_cap = 1ull << (2 * (max_levels - level - 1));
    uint32_t cut[5] = {it.b, it.b, it.b, it.e, it.e};  // child c = [cut[c], cut[c + 1]), empty ones allowed
    {
        const uint32_t m = tree_split_point(keys, it.b, it.e);
        cut[2] = m;
        cut[1] = m - it.b >= 2 ? tree_split_point(keys, it.b, m) : it.b;
        cut[3] = it.e - m >= 2 ? tree_split_point(keys, m, it.e) : it.e;
        bool fits = true;
        for (int c = 0; c < 4; c++) fits = fits && (unsigned long long)(cut[c + 1] - cut[c]) <= child_cap;
        if (!fits) {
            const uint32_t n = it.e - it.b;
            for (int c = 1; c < 4; c++) cut[c] = it.b + (uint32_t)((unsigned long long)n * c / 4);
        }
    }
    DevNode4& nd = nodes[it.node];
    for (int c = 0; c < 4; c++) {
        const uint32_t b = cut[c], e = cut[c + 1];
        uint32_t ref = RT_REF_END;
        float lo[3] = {0.0f, 0.0f, 0.0f}, hi[3] = {0.0f, 0.0f, 0.0f};
        if (e - b == 1u) {
            ref = RT_REF_LEAF | sorted_leaf[b];
            for (int k = 0; k < 3; k++) lo[k] = boxes[b * 6 + k], hi[k] = boxes[b * 6 + 3 + k];
        } else if (e - b > 1u) {
            ref = atomicAdd(&st->next_node, 1u);
            items[atomicAdd(&st->tail, 1u)] = TreeItem{ref, b, e};
        }
        for (int k = 0; k < 3; k++) nd.lo[k][c] = lo[k], nd.hi[k][c] = hi[k];  // interior children: filled in by tree_box_kernel
        nd.child[c] = ref;
    }
}
__global__ void tree_mark_level_kernel(TreeState* st, int level) { st->level_off[level] = st->tail; }
// bottom-up pass, one thread per interior node of a level: the box of an interior child is the union of that child's own
// child boxes, which the levels below have already written
__global__ void tree_box_kernel(DevNode4* nodes, const TreeItem* items, const TreeState* st, int level) {
    const uint32_t i = st->level_off[level] + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= st->level_off[level + 1]) return;
    DevNode4& nd = nodes[items[i].node];
    for (int c = 0; c < 4; c++) {
        const uint32_t ref = nd.child[c];
        if (ref == RT_REF_END || (ref & RT_REF_LEAF)) continue;
        const DevNode4& ch = nodes[ref];
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++) lo[k] = __builtin_inff(), hi[k] = -__builtin_inff();
        for (int cc = 0; cc < 4; cc++) {
            if (ch.child[cc] == RT_REF_END) continue;
            for (int k = 0; k < 3; k++) lo[k] = fminf(lo[k], ch.lo[k][cc]), hi[k] = fmaxf(hi[k], ch.hi[k][cc]);
        }
        for (int k = 0; k < 3; k++) nd.lo[k][c] = lo[k], nd.hi[k][c] = hi[k];
    }
}

// which builder makes the four-wide trees of rt_scene_create: the last rt_set_tree_builder call, else the
// environment variable RT_AMD_TREE_BUILD ("device" or "host"), else the host builder
static int tree_builder_choice() {
    if (g_tree_builder_set) return g_tree_builder;
    const char* e = std::getenv("RT_AMD_TREE_BUILD");
    return (e && std::strcmp(e, "device") == 0) ? 1 : 0;
}
extern "C" int rt_set_tree_builder(int which) {
    if (which != 0 && which != 1) return rt_fail(RT_ERR_INVALID, "rt_set_tree_builder: 0 = host, 1 = device");
    g_tree_builder = which, g_tree_builder_set = true;
    return RT_OK;
}

extern "C" int rt_scene_waves_per_simd(const rt_scene* s) { return s ? s->waves : 0; }
extern "C" int rt_scene_deferred_shadows(const rt_scene* s, int* fallbacks) {
    if (fallbacks) *fallbacks = s ? s->defer_fallbacks : 0;
    return s ? s->defer : 0;
}
extern "C" double rt_scene_tree_build_ms(const rt_scene* s) { return s ? (double)s->tree_build_ms : 0.0; }
extern "C" int rt_scene_shared_walk(const rt_scene* s, double* plain_ms, double* shared_ms) {
    if (!s) return -1;
    if (plain_ms) *plain_ms = 0.0;  // the choice is made from the scene's contents, nothing is timed
    if (shared_ms) *shared_ms = 0.0;
    return s->shared_walk;
}

// Build the trees of T.jobs in place (nodes4 and leaves point into the uploaded slab).
static int run_tree_jobs(const Tree4& T, DevNode4* nodes4, const rt_bvh_leaf* leaves, float* build_ms) {
    *build_ms = 0.0f;
    if (T.jobs.empty()) return RT_OK;
    size_t n_max = 0;
    for (const TreeJob& j : T.jobs) n_max = std::max(n_max, j.leaves.size());
    size_t pad_max = 1;
    while (pad_max < n_max) pad_max <<= 1;
    uint32_t *mesh_leaves = nullptr, *sorted_leaf = nullptr;
    unsigned long long* keys = nullptr;
    float* boxes = nullptr;
    TreeItem* items = nullptr;
    TreeState* state = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Free {
        uint32_t *&a, *&b;
        unsigned long long*& c;
        float*& d;
        TreeItem*& it;
        TreeState*& st;
        hipEvent_t &e0, &e1;
        ~Free() {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            if (c) (void)hipFree(c);
            if (d) (void)hipFree(d);
            if (it) (void)hipFree(it);
            if (st) (void)hipFree(st);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } guard{mesh_leaves, sorted_leaf, keys, boxes, items, state, e0, e1};
    HIPCHK(hipMalloc(&mesh_leaves, n_max * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&sorted_leaf, n_max * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&keys, pad_max * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&boxes, n_max * 6 * sizeof(float)));
    HIPCHK(hipMalloc(&items, n_max * sizeof(TreeItem)));
    HIPCHK(hipMalloc(&state, sizeof(TreeState)));
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, g_stream));
    for (const TreeJob& j : T.jobs) {
        const uint32_t n = (uint32_t)j.leaves.size();
        uint32_t n_pad = 1;
        while (n_pad < n) n_pad <<= 1;
        HIPCHK(hipMemcpyAsync(mesh_leaves, j.leaves.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, g_stream));
        d3 lo = d3{j.lo[0], j.lo[1], j.lo[2]}, inv;
        inv.x = j.hi[0] > j.lo[0] ? 1.0 / (j.hi[0] - j.lo[0]) : 0.0;
        inv.y = j.hi[1] > j.lo[1] ? 1.0 / (j.hi[1] - j.lo[1]) : 0.0;
        inv.z = j.hi[2] > j.lo[2] ? 1.0 / (j.hi[2] - j.lo[2]) : 0.0;
        const unsigned tpb = 256;
        hipLaunchKernelGGL(tree_keys_kernel, dim3((n_pad + tpb - 1) / tpb), dim3(tpb), 0, g_stream, leaves, mesh_leaves, n, n_pad, lo, inv, keys);
        for (uint32_t k = 2; k <= n_pad; k <<= 1)
            for (uint32_t jj = k >> 1; jj > 0; jj >>= 1)
                hipLaunchKernelGGL(tree_sort_pass_kernel, dim3((n_pad + tpb - 1) / tpb), dim3(tpb), 0, g_stream, keys, n_pad, jj, k);
        hipLaunchKernelGGL(tree_boxes_kernel, dim3((n + tpb - 1) / tpb), dim3(tpb), 0, g_stream, leaves, mesh_leaves, keys, n, boxes, sorted_leaf);
        constexpr int MAX_LEVELS = RT_STACK / 3;  // the walk pushes up to three siblings per level
        TreeState init;
        std::memset(&init, 0, sizeof(init));
        init.next_node = j.first + 1, init.tail = 1, init.level_off[1] = 1;
        const TreeItem root_item{j.first, 0u, n};
        HIPCHK(hipMemcpyAsync(state, &init, sizeof(init), hipMemcpyHostToDevice, g_stream));
        HIPCHK(hipMemcpyAsync(items, &root_item, sizeof(root_item), hipMemcpyHostToDevice, g_stream));
        // a level has at most n / 2 interior nodes; its true size is read from the state on the device
        const unsigned level_blocks = (n / 2 + tpb) / tpb;
        for (int l = 0; l < MAX_LEVELS; l++) {
            hipLaunchKernelGGL(tree_split_kernel, dim3(level_blocks), dim3(tpb), 0, g_stream, nodes4, items, state, l, MAX_LEVELS, keys, boxes, sorted_leaf);
            hipLaunchKernelGGL(tree_mark_level_kernel, dim3(1), dim3(1), 0, g_stream, state, l + 2);
        }
        for (int l = MAX_LEVELS - 1; l >= 0; l--)
            hipLaunchKernelGGL(tree_box_kernel, dim3(level_blocks), dim3(tpb), 0, g_stream, nodes4, items, state, l);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(g_stream));  // mesh_leaves is reused by the next mesh
    }
    HIPCHK(hipEventRecord(e1, g_stream));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(build_ms, e0, e1));
    return RT_OK;
}

#endif
