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

// ---- four-wide tree built on the device with the binned surface-area heuristic --------------------------------------------
// The host builder's algorithm (build_tree4 / sah_split) level by level on the device: one block per interior node.  The
// block cuts the node's run of triangles into up to four ranges - split, then keep splitting the range with the largest
// area x count - each cut chosen among 15 bin boundaries on each axis by the surface-area heuristic, and reorders the
// run's indices accordingly.  Nodes whose run is too long for the levels below are cut into equal parts by position.
// Boxes are the unions of the triangles' padded single-precision boxes.
#define RT_DSAH_BINS 16
DEV int sah_ord(float f) {  // float -> int with the same order (its own inverse on the bit pattern)
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
DEV float sah_val(int o) { return __int_as_float(o ^ ((o >> 31) & 0x7fffffff)); }
// padded single-precision box of every triangle of the mesh (by ordinal), and the identity order
__global__ void sah_prepare_kernel(const rt_bvh_leaf* leaves, const uint32_t* mesh_leaves, uint32_t n, float* boxes, uint32_t* idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TriBounds b = tri_bounds(leaves[mesh_leaves[i]]);
    idx[i] = i;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double pad = 1e-9 * (1.0 + fabs(b.lo[k]) + fabs(b.hi[k])) + 1e-7 * (b.hi[k] - b.lo[k]);
        boxes[i * 6 + k] = __double2float_rd(b.lo[k] - pad);
        boxes[i * 6 + 3 + k] = __double2float_ru(b.hi[k] + pad);
    }
}
template <int THREADS>
struct SahShared {
    int bins[3][RT_DSAH_BINS][6];  // box of the triangles whose centre falls into the bin (ordered ints)
    uint32_t cnt[3][RT_DSAH_BINS];
    int cb[6];                     // bounds of the centres
    int box[4][6], half[2][6];     // boxes of the node's ranges; of the two parts of the range being cut
    uint32_t rb[4], re[4];
    int nr, pick, best_axis, best_bin;
    uint32_t left_n, wl[THREADS / 64], wr[THREADS / 64];
    double cost[3 * (RT_DSAH_BINS - 1)];
    float clo[3], scale[3];        // bin of a centre c on axis k: (c - clo[k]) * scale[k], clamped
};
DEV void sah_box_reset(int* b) {
    for (int k = 0; k < 3; k++) b[k] = 0x7fffffff, b[3 + k] = (int)0x80000000;
}
DEV int sah_bin(const float* clo, const float* scale, int k, float c) {
    int id = (int)((c - clo[k]) * scale[k]);
    return id < 0 ? 0 : (id >= RT_DSAH_BINS ? RT_DSAH_BINS - 1 : id);
}
DEV double sah_half_area(const int* b) {
    const double dx = (double)sah_val(b[3]) - (double)sah_val(b[0]), dy = (double)sah_val(b[4]) - (double)sah_val(b[1]),
                 dz = (double)sah_val(b[5]) - (double)sah_val(b[2]);
    return dx * dy + dy * dz + dz * dx;
}
// box of idx[b, e) into dst (block-wide; dst must have been reset and the block synchronised before)
template <int THREADS>
DEV void sah_range_box(const float* boxes, const uint32_t* idx, uint32_t b, uint32_t e, int* dst) {
    for (uint32_t i = b + threadIdx.x; i < e; i += THREADS) {
        const float* bx = boxes + (size_t)idx[i] * 6;
        for (int k = 0; k < 3; k++) atomicMin(&dst[k], sah_ord(bx[k])), atomicMax(&dst[3 + k], sah_ord(bx[3 + k]));
    }
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void tree_sah_kernel(DevNode4* nodes, TreeItem* items, TreeState* st, int level, int max_levels,
                                                                   const float* boxes, uint32_t* idx, uint32_t* tmp, const uint32_t* mesh_leaves) {
    __shared__ SahShared<THREADS> S;
    const uint32_t tid = threadIdx.x;
    for (uint32_t item = st->level_off[level] + blockIdx.x; item < st->level_off[level + 1]; item += gridDim.x) {
        const TreeItem it = items[item];
        const unsigned long long child_cap = 1ull << (2 * (max_levels - level - 1));
        const bool tight = (unsigned long long)(it.e - it.b) > child_cap;
        __syncthreads();  // the previous item's use of S is over
        if (tid == 0) S.rb[0] = it.b, S.re[0] = it.e, S.nr = 1;
        if (tid < 4) sah_box_reset(S.box[tid]);
        __syncthreads();
        sah_range_box<THREADS>(boxes, idx, it.b, it.e, S.box[0]);
        for (;;) {
            __syncthreads();
            if (S.nr >= 4) break;
            if (tid == 0) {  // the range to cut next: those too long for the levels below first, then by area x count
                int pick = -1;
                double pick_key = -1.0;
                for (int r = 0; r < S.nr; r++) {
                    const uint32_t n = S.re[r] - S.rb[r];
                    if (n < 2) continue;
                    const double key = ((unsigned long long)n > child_cap ? 1e300 : 0.0) + sah_half_area(S.box[r]) * (double)n;
                    if (key > pick_key) pick_key = key, pick = r;
                }
                S.pick = pick;
            }
            __syncthreads();
            const int pick = S.pick;
            if (pick < 0) break;
            const uint32_t b = S.rb[pick], e = S.re[pick], n = e - b;
            // ---- choose the cut ----
            if (tid == 0) S.best_axis = -1, S.left_n = n / 2;
            if (tid < 6) S.cb[tid] = tid < 3 ? 0x7fffffff : (int)0x80000000;
            if (tid < 2) sah_box_reset(S.half[tid]);
            for (uint32_t j = tid; j < 3 * RT_DSAH_BINS; j += THREADS) {
                S.cnt[j / RT_DSAH_BINS][j % RT_DSAH_BINS] = 0;
                sah_box_reset(S.bins[j / RT_DSAH_BINS][j % RT_DSAH_BINS]);
            }
            __syncthreads();
            if (!tight && n > 2) {
                for (uint32_t i = b + tid; i < e; i += THREADS) {
                    const float* bx = boxes + (size_t)idx[i] * 6;
                    for (int k = 0; k < 3; k++) {
                        const int c = sah_ord(0.5f * (bx[k] + bx[3 + k]));
                        atomicMin(&S.cb[k], c), atomicMax(&S.cb[3 + k], c);
                    }
                }
                __syncthreads();
                if (tid < 3) {
                    const float lo = sah_val(S.cb[tid]), ext = sah_val(S.cb[3 + tid]) - lo;
                    S.clo[tid] = lo, S.scale[tid] = ext > 0.0f ? (float)RT_DSAH_BINS / ext : 0.0f;  // scale 0: the axis has no extent, not a candidate
                }
                __syncthreads();
                for (uint32_t i = b + tid; i < e; i += THREADS) {
                    const float* bx = boxes + (size_t)idx[i] * 6;
                    int o[6];
                    for (int k = 0; k < 3; k++) o[k] = sah_ord(bx[k]), o[3 + k] = sah_ord(bx[3 + k]);
                    for (int k = 0; k < 3; k++) {
                        if (!(S.scale[k] > 0.0f)) continue;
                        const int id = sah_bin(S.clo, S.scale, k, 0.5f * (bx[k] + bx[3 + k]));
                        atomicAdd(&S.cnt[k][id], 1u);
                        for (int j = 0; j < 3; j++) atomicMin(&S.bins[k][id][j], o[j]), atomicMax(&S.bins[k][id][3 + j], o[3 + j]);
                    }
                }
                __syncthreads();
                if (tid < 3 * (RT_DSAH_BINS - 1)) {  // cost of cutting axis k before bin i
                    const int k = tid / (RT_DSAH_BINS - 1), i = 1 + tid % (RT_DSAH_BINS - 1);
                    double cost = 1e300;
                    if (S.scale[k] > 0.0f) {
                        int L[6], R[6];
                        sah_box_reset(L), sah_box_reset(R);
                        uint32_t nl = 0, nrr = 0;
                        for (int q = 0; q < RT_DSAH_BINS; q++) {
                            if (S.cnt[k][q] == 0) continue;
                            int* dst = q < i ? L : R;
                            (q < i ? nl : nrr) += S.cnt[k][q];
                            for (int j = 0; j < 3; j++) dst[j] = min(dst[j], S.bins[k][q][j]), dst[3 + j] = max(dst[3 + j], S.bins[k][q][3 + j]);
                        }
                        if (nl != 0 && nrr != 0) cost = sah_half_area(L) * (double)nl + sah_half_area(R) * (double)nrr;
                    }
                    S.cost[tid] = cost;
                }
                __syncthreads();
                if (tid == 0) {
                    double best = 1e300;
                    int arg = -1;
                    for (int j = 0; j < 3 * (RT_DSAH_BINS - 1); j++)
                        if (S.cost[j] < best) best = S.cost[j], arg = j;
                    if (arg >= 0) {
                        const int k = arg / (RT_DSAH_BINS - 1), i = 1 + arg % (RT_DSAH_BINS - 1);
                        uint32_t nl = 0;
                        for (int q = 0; q < RT_DSAH_BINS; q++) {  // the boxes of both parts are the unions of their bins
                            if (S.cnt[k][q] == 0) continue;
                            if (q < i) nl += S.cnt[k][q];
                            int* dst = S.half[q < i ? 0 : 1];
                            for (int j = 0; j < 3; j++) dst[j] = min(dst[j], S.bins[k][q][j]), dst[3 + j] = max(dst[3 + j], S.bins[k][q][3 + j]);
                        }
                        S.best_axis = k, S.best_bin = i, S.left_n = nl;
                    }
                }
                __syncthreads();
            }
            const int axis = S.best_axis, cut_bin = S.best_bin;
            const uint32_t left_n = S.left_n, m = b + left_n;
            if (axis >= 0) {
                // ---- reorder: triangles left of the cut first, both parts in their old order ----
                uint32_t doneL = 0, doneR = 0;
                for (uint32_t t0 = b; t0 < e; t0 += THREADS) {
                    const uint32_t i = t0 + tid;
                    const bool valid = i < e;
                    uint32_t t = 0;
                    bool left = false;
                    if (valid) {
                        t = idx[i];
                        const float* bx = boxes + (size_t)t * 6;
                        left = sah_bin(S.clo, S.scale, axis, 0.5f * (bx[axis] + bx[3 + axis])) < cut_bin;
                    }
                    const unsigned long long ml = __ballot(valid && left), mr = __ballot(valid && !left);
                    if ((tid & 63u) == 0) S.wl[tid >> 6] = (uint32_t)__popcll(ml), S.wr[tid >> 6] = (uint32_t)__popcll(mr);
                    __syncthreads();
                    uint32_t offL = doneL, offR = doneR;
                    for (uint32_t w = 0; w < THREADS / 64; w++) {
                        if (w < (tid >> 6)) offL += S.wl[w], offR += S.wr[w];
                        doneL += S.wl[w], doneR += S.wr[w];
                    }
                    const unsigned long long below = (1ull << (tid & 63u)) - 1ull;
                    if (valid && left) tmp[b + offL + (uint32_t)__popcll(ml & below)] = t;
                    if (valid && !left) tmp[m + offR + (uint32_t)__popcll(mr & below)] = t;
                    __syncthreads();
                }
                __threadfence_block();
                for (uint32_t i = b + tid; i < e; i += THREADS) idx[i] = tmp[i];
                __syncthreads();
            }
            if (axis < 0) {  // cut by position: the boxes of the parts are not known yet
                sah_range_box<THREADS>(boxes, idx, b, m, S.half[0]);
                sah_range_box<THREADS>(boxes, idx, m, e, S.half[1]);
            }
            __syncthreads();
            if (tid == 0) {
                for (int r = S.nr; r > pick + 1; r--) {
                    S.rb[r] = S.rb[r - 1], S.re[r] = S.re[r - 1];
                    for (int j = 0; j < 6; j++) S.box[r][j] = S.box[r - 1][j];
                }
                S.re[pick + 1] = e, S.rb[pick + 1] = m, S.re[pick] = m;
                for (int j = 0; j < 6; j++) S.box[pick][j] = S.half[0][j], S.box[pick + 1][j] = S.half[1][j];
                S.nr++;
            }
        }
        // ---- the node ----
        if (tid < 4) {
            const int c = (int)tid;
            DevNode4& nd = nodes[it.node];
            uint32_t ref = RT_REF_END;
            float lo[3] = {0.0f, 0.0f, 0.0f}, hi[3] = {0.0f, 0.0f, 0.0f};
            if (c < S.nr) {
                const uint32_t b = S.rb[c], e = S.re[c];
                for (int k = 0; k < 3; k++) lo[k] = sah_val(S.box[c][k]), hi[k] = sah_val(S.box[c][3 + k]);
                if (e - b == 1u) {
                    ref = RT_REF_LEAF | mesh_leaves[idx[b]];
                } else {
                    ref = atomicAdd(&st->next_node, 1u);
                    items[atomicAdd(&st->tail, 1u)] = TreeItem{ref, b, e};
                }
            }
            for (int k = 0; k < 3; k++) nd.lo[k][c] = lo[k], nd.hi[k][c] = hi[k];
            nd.child[c] = ref;
        }
    }
}

// which builder makes the four-wide trees of rt_scene_create: the last rt_set_tree_builder call, else the
// environment variable RT_AMD_TREE_BUILD ("device" or "host"), else the host builder
static int tree_builder_choice() {
    if (g_tree_builder_set) return g_tree_builder;
    const char* e = std::getenv("RT_AMD_TREE_BUILD");
    if (e && std::strcmp(e, "device") == 0) return 1;
    if (e && std::strcmp(e, "device_sah") == 0) return 2;
    return 0;
}
extern "C" int rt_set_tree_builder(int which) {
    if (which < 0 || which > 2) return rt_fail(RT_ERR_INVALID, "rt_set_tree_builder: 0 = host, 1 = device (Morton order), 2 = device (surface-area heuristic)");
    g_tree_builder = which, g_tree_builder_set = true;
    return RT_OK;
}

extern "C" int rt_scene_waves_per_simd(const rt_scene* s) { return s ? s->waves : 0; }
#ifdef RT_EXPERIMENTS
extern "C" int rt_scene_deferred_shadows(const rt_scene* s, int* fallbacks) {
    if (fallbacks) *fallbacks = s ? s->defer_fallbacks : 0;
    return s ? s->defer : 0;
}
#endif
extern "C" double rt_scene_tree_build_ms(const rt_scene* s) { return s ? (double)s->tree_build_ms : 0.0; }
extern "C" int rt_scene_shared_walk(const rt_scene* s, double* plain_ms, double* shared_ms) {
    if (!s) return -1;
    if (plain_ms) *plain_ms = 0.0;  // the choice is made from the scene's contents, nothing is timed
    if (shared_ms) *shared_ms = 0.0;
    return s->shared_walk;
}

// Build the trees of T.jobs in place (nodes4 and leaves point into the uploaded slab).
static int run_tree_jobs(const Tree4& T, DevNode4* nodes4, const rt_bvh_leaf* leaves, float* build_ms, bool sah) {
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
        constexpr int MAX_LEVELS = RT_STACK / 3;  // the walk pushes up to three siblings per level
        TreeState init;
        std::memset(&init, 0, sizeof(init));
        init.next_node = j.first + 1, init.tail = 1, init.level_off[1] = 1;
        const TreeItem root_item{j.first, 0u, n};
        HIPCHK(hipMemcpyAsync(state, &init, sizeof(init), hipMemcpyHostToDevice, g_stream));
        HIPCHK(hipMemcpyAsync(items, &root_item, sizeof(root_item), hipMemcpyHostToDevice, g_stream));
        if (sah) {
            uint32_t* idx = sorted_leaf;
            uint32_t* tmp = reinterpret_cast<uint32_t*>(keys);
            hipLaunchKernelGGL(sah_prepare_kernel, dim3((n + tpb - 1) / tpb), dim3(tpb), 0, g_stream, leaves, mesh_leaves, n, boxes, idx);
            unsigned long long width = 1;  // level l has at most 4^l interior nodes, and fewer than n / 2 + 1
            for (int l = 0; l < MAX_LEVELS; l++, width *= 4) {
                const unsigned blocks = (unsigned)std::min<unsigned long long>(std::min<unsigned long long>(width, n / 2 + 1), 8192ull);
                if (l < 3)  // the few long runs at the top: large blocks
                    hipLaunchKernelGGL(tree_sah_kernel<1024>, dim3(blocks), dim3(1024), 0, g_stream, nodes4, items, state, l, MAX_LEVELS, boxes, idx, tmp, mesh_leaves);
                else
                    hipLaunchKernelGGL(tree_sah_kernel<128>, dim3(blocks), dim3(128), 0, g_stream, nodes4, items, state, l, MAX_LEVELS, boxes, idx, tmp, mesh_leaves);
                hipLaunchKernelGGL(tree_mark_level_kernel, dim3(1), dim3(1), 0, g_stream, state, l + 2);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(g_stream));  // mesh_leaves is reused by the next mesh
            continue;
        }
        hipLaunchKernelGGL(tree_keys_kernel, dim3((n_pad + tpb - 1) / tpb), dim3(tpb), 0, g_stream, leaves, mesh_leaves, n, n_pad, lo, inv, keys);
        for (uint32_t k = 2; k <= n_pad; k <<= 1)
            for (uint32_t jj = k >> 1; jj > 0; jj >>= 1)
                hipLaunchKernelGGL(tree_sort_pass_kernel, dim3((n_pad + tpb - 1) / tpb), dim3(tpb), 0, g_stream, keys, n_pad, jj, k);
        hipLaunchKernelGGL(tree_boxes_kernel, dim3((n + tpb - 1) / tpb), dim3(tpb), 0, g_stream, leaves, mesh_leaves, keys, n, boxes, sorted_leaf);
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

// Reads the four-wide trees back and checks what the pruned walk relies on: every triangle of a mesh hangs under its tree
// exactly once, no more than RT_STACK / 3 levels deep, and every child box contains the exact bounds of the triangles below
// it.  *cost = the surface-area measure of the trees (expected box tests of a random ray that meets the root box: sum of
// the child-box areas over the root-box area), summed over the trees; *trees = how many there are.
extern "C" int rt_scene_check_trees(const rt_scene* s, int* trees, double* cost) {
    if (!s) return rt_fail(RT_ERR_INVALID, "rt_scene_check_trees: no scene");
    if (trees) *trees = 0;
    if (cost) *cost = 0.0;
    if (s->n_nodes4 == 0) return RT_OK;
    std::vector<DevNode4> nodes(s->n_nodes4);
    std::vector<rt_bvh_leaf> leaves(s->n_leaves);
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemcpy(nodes.data(), s->dev.nodes4, nodes.size() * sizeof(DevNode4), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(leaves.data(), s->dev.leaves, leaves.size() * sizeof(rt_bvh_leaf), hipMemcpyDeviceToHost));
    constexpr int MAX_LEVELS = RT_STACK / 3;
    std::vector<uint8_t> seen(s->n_leaves, 0), node_seen(s->n_nodes4, 0);
    for (size_t mi = 0; mi < s->tree_roots.size(); mi++) {
        if (s->tree_roots[mi] < 0) continue;
        std::vector<uint8_t> mine(s->n_leaves, 0);
        for (uint32_t l : s->tree_tris[mi]) mine[l] = 1;
        struct Ret {
            double lo[3], hi[3];
        };
        size_t found = 0;
        double area_sum = 0.0;
        std::string err;
        auto half_area_f = [](const DevNode4& n, int c) {
            const double dx = (double)n.hi[0][c] - n.lo[0][c], dy = (double)n.hi[1][c] - n.lo[1][c], dz = (double)n.hi[2][c] - n.lo[2][c];
            return dx * dy + dy * dz + dz * dx;
        };
        // bounds of the triangles under node `ni`; checks its child boxes against them
        std::function<Ret(uint32_t, int)> visit = [&](uint32_t ni, int level) -> Ret {
            Ret r;
            for (int k = 0; k < 3; k++) r.lo[k] = 1e300, r.hi[k] = -1e300;
            if (ni >= nodes.size() || level > MAX_LEVELS || node_seen[ni]) {
                if (err.empty()) err = ni >= nodes.size() ? "node index out of range" : level > MAX_LEVELS ? "tree deeper than the walk's stack allows" : "node reached twice";
                return r;
            }
            node_seen[ni] = 1;
            const DevNode4& n = nodes[ni];
            int kids = 0;
            for (int c = 0; c < 4 && err.empty(); c++) {
                const uint32_t ref = n.child[c];
                if (ref == RT_REF_END) continue;
                kids++;
                Ret sub;
                if (ref & RT_REF_LEAF) {
                    const uint32_t l = ref & ~RT_REF_LEAF;
                    if (l >= leaves.size() || !mine[l] || seen[l]) {
                        err = l >= leaves.size() ? "leaf index out of range" : !mine[l] ? "triangle of another mesh" : "triangle under two parents";
                        break;
                    }
                    seen[l] = 1, found++;
                    const rt_bvh_leaf& lf = leaves[l];
                    const double v[3][3] = {{lf.O1.x, lf.O1.y, lf.O1.z}, {lf.O2.x, lf.O2.y, lf.O2.z}, {lf.O3.x, lf.O3.y, lf.O3.z}};
                    for (int k = 0; k < 3; k++) sub.lo[k] = std::fmin(v[0][k], std::fmin(v[1][k], v[2][k])), sub.hi[k] = std::fmax(v[0][k], std::fmax(v[1][k], v[2][k]));
                } else {
                    sub = visit(ref, level + 1);
                }
                for (int k = 0; k < 3; k++) {
                    if (!((double)n.lo[k][c] <= sub.lo[k] && sub.hi[k] <= (double)n.hi[k][c]) && err.empty()) err = "child box does not contain its triangles";
                    r.lo[k] = std::fmin(r.lo[k], sub.lo[k]), r.hi[k] = std::fmax(r.hi[k], sub.hi[k]);
                }
                area_sum += half_area_f(n, c);
            }
            if (kids < 2 && err.empty()) err = "interior node with fewer than two children";
            return r;
        };
        const Ret root = visit((uint32_t)s->tree_roots[mi], 1);
        if (err.empty() && found != s->tree_tris[mi].size()) err = "triangles missing from the tree";
        if (!err.empty()) return rt_fail(RT_ERR_INVALID, "rt_scene_check_trees: mesh " + std::to_string(mi) + ": " + err);
        const double dx = root.hi[0] - root.lo[0], dy = root.hi[1] - root.lo[1], dz = root.hi[2] - root.lo[2];
        const double root_area = dx * dy + dy * dz + dz * dx;
        if (cost && root_area > 0.0) *cost += area_sum / root_area;
        if (trees) ++*trees;
    }
    return RT_OK;
}

#endif
