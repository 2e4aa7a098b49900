// Host-side preparation of a scene for the device: per-primitive geometry and shading records, the four-wide
// tree of the pruned walk (binned surface-area heuristic), the shadow candidate grid.  Included by rt_device.hip
// only (one translation unit: it uses that file's record types and is compiled with -ffp-contract=off like it).
#ifndef RT_SCENE_BUILD_HPP
#define RT_SCENE_BUILD_HPP

// ---- host-side precomputation of DevGeom (same IEEE double operations, in the same order, as the
// reference performs per ray; this file is compiled with -ffp-contract=off on the host too) -------
namespace hostmath {
struct h3 {
    double x, y, z;
};
static h3 H(const rt_vec3& v) { return h3{v.x, v.y, v.z}; }
static h3 operator+(h3 a, h3 b) { return h3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static h3 operator-(h3 a, h3 b) { return h3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static h3 operator*(h3 a, double k) { return h3{a.x * k, a.y * k, a.z * k}; }
static h3 operator/(h3 a, double k) { return h3{a.x / k, a.y / k, a.z / k}; }
static double hdot(h3 a, h3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static h3 hcross(h3 a, h3 b) { return h3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static double hmodule(h3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
static h3 hunit(h3 a) { return a / hmodule(a); }
static void put(DevGeom& g, int i, h3 v) { g.f[i] = v.x, g.f[i + 1] = v.y, g.f[i + 2] = v.z; }
}  // namespace hostmath

static void set_ball(DevGeom& g, hostmath::h3 c, double r) {
    hostmath::put(g, BND_C, c);
    g.f[BND_R] = r * (1.0 + 1e-6) + 1e-9;  // inflated: the pre-test must never reject a true hit
    if (!(g.f[BND_R] == g.f[BND_R]) || !(c.x == c.x) || !(c.y == c.y) || !(c.z == c.z)) g.f[BND_R] = 1e300;  // degenerate: never prune
}

static void cylinder_geom(DevGeom& g, hostmath::h3 O1, hostmath::h3 O2, double R) {
    using namespace hostmath;
    g.type = RT_PRIM_CYLINDER;
    h3 dd = O2 - O1;
    put(g, CYL_O1, O1), put(g, CYL_O2, O2);
    put(g, CYL_DN, hunit(dd));
    g.f[CYL_LEN] = hmodule(dd);
    g.f[CYL_R2] = R * R;
    put(g, CYL_NTOP, hunit(O1 - O2));
    put(g, CYL_NBOT, hunit(O2 - O1));
    double half = 0.5 * hmodule(dd);
    set_ball(g, (O1 + O2) * 0.5, std::sqrt(half * half + R * R));
    g.f[CAP_R] = std::fabs(R) * (1.0 + 1e-6) + 1e-9;  // with CYL_O1, CYL_O2 (= CAP_P0, CAP_P1): the bounding capsule
}

// ---- four-wide tree of the pruned walk (DevNode4) -------------------------------------------------
static float round_down(double x) {
    float f = (float)x;
    return (double)f > x ? std::nextafterf(f, -INFINITY) : f;
}
static float round_up(double x) {
    float f = (float)x;
    return (double)f < x ? std::nextafterf(f, INFINITY) : f;
}

struct TreeJob {  // one mesh whose four-wide tree is built on the device, in place, after the slab has been uploaded
    uint32_t first;                 // index of its root DevNode4; its nodes are first .. first + number of triangles - 1 at most
    double lo[3], hi[3];            // bounds of the mesh (Morton code frame)
    std::vector<uint32_t> leaves;   // leaf indices of its triangles
};
struct Tree4 {
    std::vector<DevNode4> nodes;
    std::vector<uint32_t> node_parent, leaf_parent;   // per binary node / per leaf
    std::vector<int32_t> mesh_root4;                  // per mesh: root DevNode4 index, -1 = use the stackless walk
    std::vector<std::vector<uint32_t>> mesh_tris;     // per mesh with a four-wide tree: the leaf indices of its triangles
    std::vector<TreeJob> jobs;
};
// set by rt_set_tree_builder for the calling thread (like the render stream), read by that thread's next rt_scene_create
static thread_local int g_tree_builder = 0;  // 0 = host (binned surface-area heuristic), 1 = device (Morton order, balanced)
static thread_local bool g_tree_builder_set = false;

// double-precision bounds of what hangs under a child ref: the reference's box of an interior node, the
// triangle's own bounds for a leaf
static void ref_bounds(const rt_scene_desc* d, uint32_t ref, double lo[3], double hi[3]) {
    if (ref & RT_REF_LEAF) {
        const rt_bvh_leaf& lf = d->leaves[ref & ~RT_REF_LEAF];
        const double v[3][3] = {{lf.O1.x, lf.O1.y, lf.O1.z}, {lf.O2.x, lf.O2.y, lf.O2.z}, {lf.O3.x, lf.O3.y, lf.O3.z}};
        for (int k = 0; k < 3; k++) lo[k] = std::fmin(v[0][k], std::fmin(v[1][k], v[2][k])), hi[k] = std::fmax(v[0][k], std::fmax(v[1][k], v[2][k]));
    } else {
        for (int k = 0; k < 3; k++) lo[k] = d->nodes[ref].bmin[k], hi[k] = d->nodes[ref].bmax[k];
    }
}

// ---- the walk's own tree -----------------------------------------------------------------------
// The pruned walk does not have to follow the reference's tree (a median split on cycling axes,
// CR/Renderer_CUDA.cpp:183-236): any structure that never loses a triangle the ray can hit will do,
// because a hit is accepted only after the reference's own box tests have been honoured (mesh_walk4).
// So each mesh gets a four-wide tree built top-down with a binned surface-area heuristic, one triangle
// per leaf slot, at most RT_STACK / 3 levels deep (the walk pushes up to three siblings per level).
struct SahTri {
    uint32_t leaf;  // RT_REF_LEAF | leaf index
    double lo[3], hi[3], c[3];
};

static double half_area(const double lo[3], const double hi[3]) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// Partition tris[b, e) into two non-empty parts; returns the split index.  `balanced` forces a median split.
static size_t sah_split(std::vector<SahTri>& tris, size_t b, size_t e, bool balanced) {
    const size_t n = e - b;
    double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = b; i < e; i++)
        for (int k = 0; k < 3; k++) clo[k] = std::fmin(clo[k], tris[i].c[k]), chi[k] = std::fmax(chi[k], tris[i].c[k]);
    int wide = 0;
    for (int k = 1; k < 3; k++)
        if (chi[k] - clo[k] > chi[wide] - clo[wide]) wide = k;
    auto median = [&]() {
        std::nth_element(tris.begin() + b, tris.begin() + b + n / 2, tris.begin() + e,
                         [wide](const SahTri& x, const SahTri& y) { return x.c[wide] < y.c[wide]; });
        return b + n / 2;
    };
    if (balanced || n <= 2 || !(chi[wide] - clo[wide] > 0)) return median();
#ifndef RT_SAH_BINS
#define RT_SAH_BINS 16
#endif
    constexpr int NB = RT_SAH_BINS;
    double best = 1e300;
    int best_axis = -1, best_bin = 0;
    for (int k = 0; k < 3; k++) {
        const double ext = chi[k] - clo[k];
        if (!(ext > 0)) continue;
        struct Bin {
            double lo[3], hi[3];
            size_t n;
        } bins[NB];
        for (auto& bn : bins) {
            bn.n = 0;
            for (int j = 0; j < 3; j++) bn.lo[j] = 1e300, bn.hi[j] = -1e300;
        }
        for (size_t i = b; i < e; i++) {
            int id = (int)((tris[i].c[k] - clo[k]) / ext * NB);
            id = id < 0 ? 0 : (id >= NB ? NB - 1 : id);
            bins[id].n++;
            for (int j = 0; j < 3; j++) bins[id].lo[j] = std::fmin(bins[id].lo[j], tris[i].lo[j]), bins[id].hi[j] = std::fmax(bins[id].hi[j], tris[i].hi[j]);
        }
        double ra[NB];  // cost of everything right of boundary i (bins i..NB-1)
        {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            size_t cnt = 0;
            for (int i = NB - 1; i > 0; i--) {
                for (int j = 0; j < 3; j++) lo[j] = std::fmin(lo[j], bins[i].lo[j]), hi[j] = std::fmax(hi[j], bins[i].hi[j]);
                cnt += bins[i].n;
                ra[i] = cnt ? half_area(lo, hi) * (double)cnt : 0.0;
            }
        }
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        size_t cnt = 0;
        for (int i = 1; i < NB; i++) {
            for (int j = 0; j < 3; j++) lo[j] = std::fmin(lo[j], bins[i - 1].lo[j]), hi[j] = std::fmax(hi[j], bins[i - 1].hi[j]);
            cnt += bins[i - 1].n;
            if (cnt == 0 || cnt == n) continue;
            const double cost = half_area(lo, hi) * (double)cnt + ra[i];
            if (cost < best) best = cost, best_axis = k, best_bin = i;
        }
    }
    if (best_axis < 0) return median();
    const double ext = chi[best_axis] - clo[best_axis], c0 = clo[best_axis];
    const int ax = best_axis, bb = best_bin;
    auto mid = std::partition(tris.begin() + b, tris.begin() + e, [=](const SahTri& t) {
        int id = (int)((t.c[ax] - c0) / ext * NB);
        id = id < 0 ? 0 : (id >= NB ? NB - 1 : id);
        return id < bb;
    });
    const size_t m = (size_t)(mid - tris.begin());
    return (m == b || m == e) ? median() : m;
}

static void build_tree4(const rt_scene_desc* d, Tree4& T, bool device_build) {
    T.nodes.clear();
    T.jobs.clear();
    T.node_parent.assign(d->n_nodes, RT_REF_END);
    T.leaf_parent.assign(d->n_leaves, RT_REF_END);
    T.mesh_root4.assign(d->n_meshes, -1);
    T.mesh_tris.assign(d->n_meshes, {});
    bool shared = false;  // a node or leaf with two parents has no single ancestor chain: no four-wide walk then
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        for (uint32_t c : {d->nodes[i].left, d->nodes[i].right}) {
            if (c == RT_REF_END) continue;
            uint32_t& slot = (c & RT_REF_LEAF) ? T.leaf_parent[c & ~RT_REF_LEAF] : T.node_parent[c];
            shared = shared || slot != RT_REF_END;
            slot = i;
        }
    }
    for (int mi = 0; mi < d->n_meshes; mi++) {  // roots must not hang under another node either
        const uint32_t r = d->meshes[mi].root;
        if (r != RT_REF_END && !(r & RT_REF_LEAF)) shared = shared || T.node_parent[r] != RT_REF_END;
    }
    if (shared) return;
    constexpr int MAX_LEVELS = RT_STACK / 3;
    for (int mi = 0; mi < d->n_meshes; mi++) {
        const uint32_t root = d->meshes[mi].root;
        if (root == RT_REF_END || (root & RT_REF_LEAF)) continue;
        // the triangles of the mesh, and the nesting of the reference's boxes above them (the walk's argument for
        // keeping a hit without re-running the reference's box tests relies on that nesting)
        bool usable = true;
        std::vector<SahTri> tris;
        std::vector<uint32_t> todo{root};
        while (!todo.empty() && usable) {
            const uint32_t n = todo.back();
            todo.pop_back();
            double plo[3], phi[3];
            ref_bounds(d, n, plo, phi);
            for (uint32_t c : {d->nodes[n].left, d->nodes[n].right}) {
                if (c == RT_REF_END) continue;
                double clo[3], chi[3];
                ref_bounds(d, c, clo, chi);
                for (int k = 0; k < 3; k++)
                    usable = usable && plo[k] <= clo[k] && chi[k] <= phi[k] && std::isfinite(clo[k]) && std::isfinite(chi[k]) &&
                             std::fabs(clo[k]) < 1e6 && std::fabs(chi[k]) < 1e6;
                if (c & RT_REF_LEAF) {
                    SahTri t;
                    t.leaf = c;
                    for (int k = 0; k < 3; k++) t.lo[k] = clo[k], t.hi[k] = chi[k], t.c[k] = 0.5 * (clo[k] + chi[k]);
                    tris.push_back(t);
                } else {
                    todo.push_back(c);
                }
            }
        }
        if (!usable || tris.size() < 2) continue;
        // capacity check: a level holds four times the triangles of the one above
        double cap = 1;
        for (int i = 0; i < MAX_LEVELS; i++) cap *= 4;
        if ((double)tris.size() > cap) continue;
        if (device_build) {
            // every interior node has at least two children, so a mesh of n triangles needs fewer than n nodes
            TreeJob job;
            job.first = (uint32_t)T.nodes.size();
            ref_bounds(d, root, job.lo, job.hi);
            job.leaves.reserve(tris.size());
            for (const SahTri& t : tris) job.leaves.push_back(t.leaf & ~RT_REF_LEAF);
            if (T.nodes.size() + tris.size() >= (1u << 30)) continue;
            T.nodes.resize(T.nodes.size() + tris.size());  // zero-filled here, written by the device
            T.mesh_root4[mi] = (int32_t)job.first;
            T.mesh_tris[mi] = job.leaves;
            T.jobs.push_back(std::move(job));
            continue;
        }

        struct Item {
            size_t b, e;
            uint32_t idx4;
            int level;  // 1 = root
        };
        const size_t first = T.nodes.size();
        T.nodes.push_back(DevNode4{});
        std::vector<Item> work{{0, tris.size(), (uint32_t)first, 1}};
        while (!work.empty()) {
            const Item it = work.back();
            work.pop_back();
            // triangles a child may hold so that its subtree still fits the remaining levels
            double child_cap = 1;
            for (int i = it.level; i < MAX_LEVELS; i++) child_cap *= 4;
            // up to four ranges: split the range, then keep splitting the part with the largest box
            size_t rb[4], re[4];
            int nr = 1;
            rb[0] = it.b, re[0] = it.e;
            while (nr < 4) {
                int pick = -1;
                double pick_key = -1;
                for (int r = 0; r < nr; r++) {
                    const size_t n = re[r] - rb[r];
                    if (n < 2) continue;
                    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                    for (size_t i = rb[r]; i < re[r]; i++)
                        for (int k = 0; k < 3; k++) lo[k] = std::fmin(lo[k], tris[i].lo[k]), hi[k] = std::fmax(hi[k], tris[i].hi[k]);
                    // ranges that are too big for the levels left go first, then by area
                    const double key = ((double)n > child_cap ? 1e300 : 0.0) + half_area(lo, hi) * (double)n;
                    if (key > pick_key) pick_key = key, pick = r;
                }
                if (pick < 0) break;
                // a node with more triangles than one child may hold is cut into equal parts, which then all fit
                const bool tight = (double)(it.e - it.b) > child_cap;
                const size_t m = sah_split(tris, rb[pick], re[pick], tight);
                for (int r = nr; r > pick + 1; r--) rb[r] = rb[r - 1], re[r] = re[r - 1];
                re[pick + 1] = re[pick], rb[pick + 1] = m, re[pick] = m;
                nr++;
            }
            DevNode4 n4;
            std::memset(&n4, 0, sizeof(n4));
            for (int c = 0; c < 4; c++) n4.child[c] = RT_REF_END;
            for (int c = 0; c < nr; c++) {
                double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                for (size_t i = rb[c]; i < re[c]; i++)
                    for (int k = 0; k < 3; k++) lo[k] = std::fmin(lo[k], tris[i].lo[k]), hi[k] = std::fmax(hi[k], tris[i].hi[k]);
                for (int k = 0; k < 3; k++) {
                    // a triangle accepted by the double-precision test may lie a rounding error outside its exact bounds
                    const double pad = 1e-9 * (1.0 + std::fabs(lo[k]) + std::fabs(hi[k])) + 1e-7 * (hi[k] - lo[k]);
                    n4.lo[k][c] = round_down(lo[k] - pad), n4.hi[k][c] = round_up(hi[k] + pad);
                }
                if (re[c] - rb[c] == 1) {
                    n4.child[c] = tris[rb[c]].leaf;
                } else {
                    n4.child[c] = (uint32_t)T.nodes.size();
                    if ((double)(re[c] - rb[c]) > child_cap || it.level + 1 > MAX_LEVELS) usable = false;
                    work.push_back({rb[c], re[c], n4.child[c], it.level + 1});
                    T.nodes.push_back(DevNode4{});
                }
            }
            T.nodes[it.idx4] = n4;
        }
        if (usable && T.nodes.size() < (1u << 30)) {
            T.mesh_root4[mi] = (int32_t)first;
            for (const SahTri& t : tris) T.mesh_tris[mi].push_back(t.leaf & ~RT_REF_LEAF);
        } else {
            T.nodes.resize(first);
        }
    }
}

static void build_shade(const rt_primitive& p, int tex_w, int tex_h, uint32_t tex_off, DevShade& o) {
    using namespace hostmath;
    std::memset(&o, 0, sizeof(o));
    const rt_material& m = p.material;
    o.color[0] = m.color.x, o.color[1] = m.color.y, o.color[2] = m.color.z;
    o.absor[0] = m.absor.x, o.absor[1] = m.absor.y, o.absor[2] = m.absor.z;
    o.diff = m.diff, o.spec = m.spec, o.refl = m.refl, o.refr = m.refr, o.rindex = m.rindex, o.drefl = m.drefl;
    o.type = p.type, o.texture = m.texture, o.is_light = p.is_light;
    o.tex_w = tex_w, o.tex_h = tex_h, o.tex_off = tex_off;
    auto put3 = [&](int i, h3 v) { o.tf[i] = v.x, o.tf[i + 1] = v.y, o.tf[i + 2] = v.z; };
    switch (p.type) {
        case RT_PRIM_SPHERE:
            put3(TF_SPH_O, H(p.u.sphere.O)), put3(TF_SPH_DE, H(p.u.sphere.De)), put3(TF_SPH_DC, H(p.u.sphere.Dc));
            put3(TF_SPH_X, hcross(H(p.u.sphere.Dc), H(p.u.sphere.De)));
            break;
        case RT_PRIM_PLANE:
            put3(TF_PLN_DX, H(p.u.plane.Dx)), put3(TF_PLN_DY, H(p.u.plane.Dy));
            o.tf[TF_PLN_DXX] = hdot(H(p.u.plane.Dx), H(p.u.plane.Dx)), o.tf[TF_PLN_DYY] = hdot(H(p.u.plane.Dy), H(p.u.plane.Dy));
            break;
        case RT_PRIM_SQUARE:
            put3(TF_SQR_O, H(p.u.square.O)), put3(TF_SQR_DX, H(p.u.square.Dx)), put3(TF_SQR_DY, H(p.u.square.Dy));
            o.tf[TF_SQR_DXX] = hdot(H(p.u.square.Dx), H(p.u.square.Dx)), o.tf[TF_SQR_DYY] = hdot(H(p.u.square.Dy), H(p.u.square.Dy));
            break;
        case RT_PRIM_CYLINDER: {
            const h3 O1 = H(p.u.cylinder.O1), O2 = H(p.u.cylinder.O2);
            put3(TF_CYL_O1, O1), put3(TF_CYL_O2, O2), put3(TF_CYL_AX, hunit(O2 - O1));
            o.tf[TF_CYL_LEN] = hmodule(O2 - O1), o.tf[TF_CYL_R] = p.u.cylinder.R;
            break;
        }
        default: break;
    }
}

static void build_geom(const rt_scene_desc* d, const rt_primitive& p, DevGeom& g, std::vector<DevGeom>& segs, const Tree4& T) {
    using namespace hostmath;
    std::memset(&g, 0, sizeof(g));
    g.type = p.type;
    g.f[BND_R] = 1e300;
    switch (p.type) {
        case RT_PRIM_SPHERE:
            put(g, SPH_O, H(p.u.sphere.O));
            g.f[SPH_R2] = p.u.sphere.R * p.u.sphere.R;
            set_ball(g, H(p.u.sphere.O), std::fabs(p.u.sphere.R));
            break;
        case RT_PRIM_PLANE: {
            h3 N = hunit(H(p.u.plane.N));
            put(g, PLN_N, N), put(g, PLN_NR, N * p.u.plane.R);
            break;
        }
        case RT_PRIM_SQUARE: {
            h3 Dx = H(p.u.square.Dx), Dy = H(p.u.square.Dy);
            put(g, SQR_O, H(p.u.square.O)), put(g, SQR_N, hunit(hcross(Dx, Dy))), put(g, SQR_DX, Dx), put(g, SQR_DY, Dy);
            g.f[SQR_LX] = hdot(Dx, Dx), g.f[SQR_LY] = hdot(Dy, Dy);
            set_ball(g, H(p.u.square.O), hmodule(Dx) + hmodule(Dy));
            break;
        }
        case RT_PRIM_CYLINDER:
            cylinder_geom(g, H(p.u.cylinder.O1), H(p.u.cylinder.O2), p.u.cylinder.R);
            break;
        case RT_PRIM_BEZIER: {  // segments of CR/Collision_Bezier.cu:206-222
            h3 O1 = H(p.u.bezier.O1), axis = hunit(H(p.u.bezier.O2) - O1);
            g.aux0 = (uint32_t)segs.size();
            g.aux1 = p.u.bezier.degree > 0 ? p.u.bezier.degree : 0;
            double tmin = 1e300, tmax = -1e300, rmax = 0;
            for (int i = 0; i < g.aux1; i++) {
                h3 p1 = i == 0 ? O1 : O1 + axis * p.u.bezier.Z[i - 1];
                h3 p2 = O1 + axis * p.u.bezier.Z[i];
                double R = i == 0 ? p.u.bezier.R_c : p.u.bezier.R[i];
                DevGeom sg;
                std::memset(&sg, 0, sizeof(sg));
                cylinder_geom(sg, p1, p2, R);
                segs.push_back(sg);
                for (h3 e : {p1, p2}) {
                    double t = hdot(e - O1, axis);
                    tmin = std::fmin(tmin, t), tmax = std::fmax(tmax, t);
                }
                rmax = std::fmax(rmax, std::fabs(R));
            }
            // the true curve (bezier_newton_hit): frame record, then power-basis coefficients
            // c[j] = C(n,j) * sum_i (-1)^(j-i) C(j,i) P[i] of the control points P[0..n]
            const int n = p.u.bezier.degree;
            DevGeom fr, co;
            std::memset(&fr, 0, sizeof(fr)), std::memset(&co, 0, sizeof(co));
            fr.aux1 = n;
            put(fr, BZN_O1, O1), put(fr, BZN_AX, H(p.u.bezier.N)), put(fr, BZN_NX, H(p.u.bezier.Nx)), put(fr, BZN_NY, H(p.u.bezier.Ny));
            if (n >= 1 && n <= RT_BEZIER_MAX_CTRL - 1) {
                double binom[RT_BEZIER_MAX_CTRL][RT_BEZIER_MAX_CTRL];
                for (int a = 0; a <= n; a++) {
                    binom[a][0] = 1.0;
                    for (int b = 1; b <= a; b++) binom[a][b] = (b == a) ? 1.0 : binom[a - 1][b - 1] + binom[a - 1][b];
                }
                for (int which = 0; which < 2; which++) {
                    const double* P = which ? p.u.bezier.R : p.u.bezier.Z;
                    for (int j = 0; j <= n; j++) {
                        double s = 0.0;
                        for (int i = 0; i <= j; i++) s += (((j - i) & 1) ? -1.0 : 1.0) * binom[j][i] * P[i];
                        co.f[(which ? BZN_CR : BZN_CZ) + j] = binom[n][j] * s;
                    }
                }
                for (int i = 0; i <= n; i++) {  // the curve stays inside the hull of its control points
                    tmin = std::fmin(tmin, p.u.bezier.Z[i]), tmax = std::fmax(tmax, p.u.bezier.Z[i]);
                    rmax = std::fmax(rmax, std::fabs(p.u.bezier.R[i]));
                }
            } else {
                fr.aux1 = 0;
            }
            segs.push_back(fr), segs.push_back(co);
            if (tmax >= tmin) {
                double half = 0.5 * (tmax - tmin);
                set_ball(g, O1 + axis * (0.5 * (tmin + tmax)), std::sqrt(half * half + rmax * rmax));
                put(g, CAP_P0, O1 + axis * tmin), put(g, CAP_P1, O1 + axis * tmax);
                g.f[CAP_R] = rmax * (1.0 + 1e-6) + 1e-9;
            } else {
                g.f[CAP_R] = 1e300;  // nothing known: never pruned by the capsule
            }
            break;
        }
        case RT_PRIM_TRIANGLE: {
            h3 V1 = H(p.u.triangle.O1), e1 = H(p.u.triangle.O2) - V1, e2 = H(p.u.triangle.O3) - V1;
            put(g, TRI_V1, V1), put(g, TRI_E1, e1), put(g, TRI_E2, e2), put(g, TRI_N, hunit(hcross(e1, e2)));
            set_ball(g, V1, std::fmax(hmodule(e1), hmodule(e2)));
            break;
        }
        case RT_PRIM_MESH: {
            const rt_mesh& m = d->meshes[p.u.mesh.mesh];
            g.aux0 = m.root;
            g.aux1 = p.u.mesh.mesh;
            double sc = 0;
            h3 lo{0, 0, 0}, hi{0, 0, 0};
            if (m.root != RT_REF_END && !(m.root & RT_REF_LEAF)) {
                const rt_bvh_node& n = d->nodes[m.root];
                for (int k = 0; k < 3; k++) sc = std::fmax(sc, std::fmax(std::fabs(n.bmin[k]), std::fabs(n.bmax[k])));
                lo = h3{n.bmin[0], n.bmin[1], n.bmin[2]}, hi = h3{n.bmax[0], n.bmax[1], n.bmax[2]};
            } else if (m.root != RT_REF_END) {
                const rt_bvh_leaf& lf = d->leaves[m.root & ~RT_REF_LEAF];
                const double c[9] = {lf.O1.x, lf.O1.y, lf.O1.z, lf.O2.x, lf.O2.y, lf.O2.z, lf.O3.x, lf.O3.y, lf.O3.z};
                for (double v : c) sc = std::fmax(sc, std::fabs(v));
                lo = h3{std::fmin(c[0], std::fmin(c[3], c[6])), std::fmin(c[1], std::fmin(c[4], c[7])), std::fmin(c[2], std::fmin(c[5], c[8]))};
                hi = h3{std::fmax(c[0], std::fmax(c[3], c[6])), std::fmax(c[1], std::fmax(c[4], c[7])), std::fmax(c[2], std::fmax(c[5], c[8]))};
            }
            g.f[MSH_SCALE] = 3.0 * sc;
            put(g, MSH_BMIN, lo), put(g, MSH_BMAX, hi);
            g.aux2 = T.mesh_root4[p.u.mesh.mesh] + 1;
            set_ball(g, (lo + hi) * 0.5, 0.5 * hmodule(hi - lo));  // empty mesh: a point ball, never reached
            break;
        }
        default: break;
    }
}

// squared distance between two segments (host twin of seg_seg_dist2)
static double host_seg_seg_dist2(hostmath::h3 p1, hostmath::h3 q1, hostmath::h3 p2, hostmath::h3 q2, double& s_out, double& t_out) {
    using namespace hostmath;
    const h3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const double a = hdot(d1, d1), e = hdot(d2, d2), f = hdot(d2, r), c = hdot(d1, r), b = hdot(d1, d2);
    const double tiny = 1e-200, denom = a * e - b * b;
    double s = (denom > 1e-12 * a * e && denom > tiny) ? std::fmin(std::fmax((b * f - c * e) / denom, 0.0), 1.0) : 0.0;
    if (!(a > tiny)) s = 0.0;
    double t = e > tiny ? (b * s + f) / e : 0.0;
    if (t < 0.0) t = 0.0, s = a > tiny ? std::fmin(std::fmax(-c / a, 0.0), 1.0) : 0.0;
    else if (t > 1.0) t = 1.0, s = a > tiny ? std::fmin(std::fmax((b - c) / a, 0.0), 1.0) : 0.0;
    const h3 g = (p1 + d1 * s) - (p2 + d2 * t);
    s_out = s, t_out = t;
    return hdot(g, g);
}

// ---- candidate grid: which primitives can stand between a cell's points and a lamp -------------------
// Same reasoning as light_candidates on the device, for every point of a cell at once: the cell is a ball
// (centre, rho) and every segment from one of its points to the lamp's ball stays within rho of the capsule
// around centre..lamp.  Meshes and triangles always get the capsule stretched as under RT_F_MESH_RAW_DIR (the
// mode is not known here; the stretched capsule contains the plain one).
static void build_cand_grid(const rt_scene_desc* d, const std::vector<DevGeom>& geom, std::vector<unsigned long long>& grid, double lo[3],
                            double inv[3]) {
    using namespace hostmath;
    grid.clear();
    if (d->n_primitives > 64 || d->n_lights <= 0 || d->n_primitives <= 0) return;
    // bounds of everything bounded, the lamps and the camera
    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
    auto grow = [&](h3 c, double r) {
        const double v[3] = {c.x, c.y, c.z};
        for (int k = 0; k < 3; k++) blo[k] = std::fmin(blo[k], v[k] - r), bhi[k] = std::fmax(bhi[k], v[k] + r);
    };
    // primitives far larger than the rest (walls made of giant spheres, CR/resources/scene_final.txt) are treated like
    // planes: they do not set the extent of the grid
    std::vector<double> radii;
    for (int j = 0; j < d->n_primitives; j++) {
        const DevGeom& g = geom[j];
        if (g.type == RT_PRIM_PLANE) continue;
        if (g.type == RT_PRIM_MESH) {
            if (g.aux0 != RT_REF_END) radii.push_back(0.5 * g.f[MSH_SCALE] / 3.0);
        } else if (g.f[BND_R] < 1e100) {
            radii.push_back(g.f[BND_R]);
        }
    }
    if (radii.empty()) return;
    std::sort(radii.begin(), radii.end());
    const double big = 30.0 * radii[radii.size() / 2] + 1e-9;
    for (int j = 0; j < d->n_primitives; j++) {
        const DevGeom& g = geom[j];
        if (g.type == RT_PRIM_PLANE) continue;
        if (g.type == RT_PRIM_MESH) {
            if (g.aux0 == RT_REF_END) continue;
            grow(h3{g.f[MSH_BMIN], g.f[MSH_BMIN + 1], g.f[MSH_BMIN + 2]}, 0), grow(h3{g.f[MSH_BMAX], g.f[MSH_BMAX + 1], g.f[MSH_BMAX + 2]}, 0);
        } else if (g.f[BND_R] < big) {
            grow(h3{g.f[BND_C], g.f[BND_C + 1], g.f[BND_C + 2]}, g.f[BND_R]);
        }
    }
    std::vector<double> lamp_r(d->n_lights);
    for (int l = 0; l < d->n_lights; l++) {
        const rt_light& L = d->lights[l];
        double Rl = 0;
        if (L.type == RT_LIGHT_SPHERE) Rl = std::fabs(L.R);
        else if (L.type == RT_LIGHT_SQUARE) Rl = hmodule(H(L.Dx)) + hmodule(H(L.Dy));
        lamp_r[l] = Rl * (1.0 + 1e-6) + 1e-9;
        grow(H(L.O), lamp_r[l]);
    }
    grow(H(d->camera.O), 0);
    for (int k = 0; k < 3; k++) {
        if (!(bhi[k] >= blo[k]) || !std::isfinite(blo[k]) || !std::isfinite(bhi[k])) return;
        const double ext = std::fmax(bhi[k] - blo[k], 1e-3), pad = 0.5 * ext;
        blo[k] -= pad, bhi[k] += pad;
        lo[k] = blo[k];
        inv[k] = RT_CGRID / (bhi[k] - blo[k]);
    }
    const double cell[3] = {1.0 / inv[0], 1.0 / inv[1], 1.0 / inv[2]};
    // a cell's points are those whose computed index is this cell: allow for the rounding of the index arithmetic
    const double rho = 0.5 * std::sqrt(cell[0] * cell[0] + cell[1] * cell[1] + cell[2] * cell[2]) * (1.0 + 1e-6) + 1e-9;
    const size_t per_lamp = (size_t)RT_CGRID * RT_CGRID * RT_CGRID;
    grid.assign(per_lamp * d->n_lights * 2, 0ull);  // per cell: [may occlude, occludes every sample]
    // slabs (lamp, iz) are independent: host threads take them in turn
    std::atomic<int> next_slab{0};
    auto worker = [&]() {
    for (;;) {
        const int slab = next_slab.fetch_add(1);
        if (slab >= d->n_lights * RT_CGRID) break;
        const int l = slab / RT_CGRID, iz = slab % RT_CGRID;
        const h3 Lc = H(d->lights[l].O);
        const double Rl = lamp_r[l];
        {
            for (int iy = 0; iy < RT_CGRID; iy++)
                for (int ix = 0; ix < RT_CGRID; ix++) {
                    const h3 A{lo[0] + (ix + 0.5) * cell[0], lo[1] + (iy + 0.5) * cell[1], lo[2] + (iz + 0.5) * cell[2]};
                    const h3 D = Lc - A;
                    const double L2 = hdot(D, D), len = std::sqrt(L2);
                    unsigned long long m = 0, mfull = 0;
                    for (int j = 0; j < d->n_primitives; j++) {
                        const DevGeom& g = geom[j];
                        bool may = true, full = false;
                        if (!(L2 > 1e-200)) {
                        } else if (g.type == RT_PRIM_PLANE) {
                            const h3 N{g.f[PLN_N], g.f[PLN_N + 1], g.f[PLN_N + 2]}, Q{g.f[PLN_NR], g.f[PLN_NR + 1], g.f[PLN_NR + 2]};
                            const double sa = hdot(A - Q, N), sl = hdot(Lc - Q, N);
                            const double tol = 1e-7 * (1.0 + std::fabs(sa) + std::fabs(sl)) + rho;
                            if ((sa > tol && sl > Rl + tol) || (sa < -tol && sl < -(Rl + tol))) may = false;
                        } else if (g.type == RT_PRIM_MESH && g.aux0 != RT_REF_END) {
                            // a point A' + s (T - A') of a shadow segment (|A' - A| <= rho, |T - Lc| <= Rl) lies within
                            // |1 - s| rho + s Rl of A + s D: clip with the largest such radius, then again with the
                            // largest radius inside the overlap found
                            const double smax = (len + rho + Rl) * (1.0 + 1e-6), pad = 1e-9 * (1.0 + g.f[MSH_SCALE] + std::fabs(A.x) + std::fabs(A.y) + std::fabs(A.z));
                            double rc = smax * (Rl + rho) + pad;
                            const double Aa[3] = {A.x, A.y, A.z}, Da[3] = {D.x, D.y, D.z};
                            for (int pass = 0; pass < 3 && may; pass++) {
                                double tlo = 0.0, thi = smax;
                                for (int k = 0; k < 3; k++) {
                                    const double blo_k = g.f[MSH_BMIN + k] - rc, bhi_k = g.f[MSH_BMAX + k] + rc;
                                    if (std::fabs(Da[k]) > 1e-200) {
                                        const double i = 1.0 / Da[k], t1 = (blo_k - Aa[k]) * i, t2 = (bhi_k - Aa[k]) * i;
                                        tlo = std::fmax(tlo, std::fmin(t1, t2)), thi = std::fmin(thi, std::fmax(t1, t2));
                                    } else if (Aa[k] < blo_k || Aa[k] > bhi_k) {
                                        thi = -1.0;
                                    }
                                }
                                if (tlo > thi * (1.0 + 1e-9) + 1e-12) may = false;
                                if (!may) break;
                                const double th = thi * (1.0 + 1e-9) + 1e-12;
                                const double rc2 = std::fmax(rho, std::fabs(1.0 - th) * rho + th * Rl) * (1.0 + 1e-9) + pad;
                                if (!(rc2 < rc)) break;
                                rc = rc2;
                            }
                        } else if (g.type == RT_PRIM_SPHERE && std::fmax(hmodule(h3{g.f[BND_C], g.f[BND_C + 1], g.f[BND_C + 2]} - A),
                                                                         hmodule(h3{g.f[BND_C], g.f[BND_C + 1], g.f[BND_C + 2]} - Lc)) +
                                                                   Rl + rho <
                                                               g.f[BND_R] * (1.0 - 2e-6) - 2e-9 - 1e-6 * (1.0 + len + rho)) {
                            may = false;  // the sphere encloses every shadow segment of the cell: its surface lies beyond the lamp
                        } else if (g.f[BND_R] < 1e100) {
                            const bool stretched = g.type == RT_PRIM_MESH || g.type == RT_PRIM_TRIANGLE;
                            const double smax = stretched ? (len + rho + Rl) * (1.0 + 1e-6) : 1.0, rc = stretched ? smax * (Rl + rho) : Rl + rho;
                            const h3 w = h3{g.f[BND_C], g.f[BND_C + 1], g.f[BND_C + 2]} - A;
                            double sp = hdot(w, D) / L2;
                            sp = std::fmin(std::fmax(sp, 0.0), smax);
                            const h3 e = w - D * sp;
                            const double rr = g.f[BND_R] + rc;
                            // clear of the intersectors' own rounding noise (isect_noise in rt_trace.hpp), for every point of the cell
                            const double noise = 1e-13 * 2.0 * ((hmodule(w) + rho) * (hmodule(w) + rho) + g.f[BND_R] * g.f[BND_R]);
                            if (hdot(e, e) > rr * rr * (1.0 + 1e-9) + noise) may = false;
                            if (may && (g.type == RT_PRIM_CYLINDER || g.type == RT_PRIM_BEZIER) && g.f[CAP_R] < 1e100) {
                                const double reach = g.f[CAP_R] + Rl + rho + 1e-9 * (1.0 + len);
                                const h3 P0{g.f[CAP_P0], g.f[CAP_P0 + 1], g.f[CAP_P0 + 2]}, P1{g.f[CAP_P1], g.f[CAP_P1 + 1], g.f[CAP_P1 + 2]};
                                double ss, tt;
                                const double dd2 = host_seg_seg_dist2(A, Lc, P0, P1, ss, tt);
                                if (dd2 > reach * reach * (1.0 + 1e-6) + noise) may = false;
                                if (may && g.type == RT_PRIM_CYLINDER) {  // umbra of the cylinder for every point of the cell (see light_candidates)
                                    const double Rc = g.f[CAP_R] * (1.0 - 2e-6) - 2e-9, alen = g.f[CYL_LEN];
                                    const double mrg = Rl + rho + 1e-5 * (1.0 + alen + Rc);
                                    const h3 ax{g.f[CYL_DN], g.f[CYL_DN + 1], g.f[CYL_DN + 2]}, a0 = A - P0, a1 = A - P1;
                                    const double ta = std::fmin(std::fmax(hdot(a0, ax), 0.0), alen);
                                    const h3 ea = a0 - ax * ta;
                                    const double far = std::sqrt(std::fmax(hdot(a0, a0), hdot(a1, a1))) + rho;
                                    const double lim = len - rho - Rl - 2e-6 - g.f[CAP_R], deep = Rc - Rl - rho - 1e-6;
                                    if (tt * alen > mrg && (1.0 - tt) * alen > mrg && deep > 0.0 && dd2 + noise < deep * deep * (1.0 - 1e-6) && lim > 0.0 && far < lim * (1.0 - 1e-6) &&
                                        std::sqrt(hdot(ea, ea)) - rho > (g.f[CAP_R] + 2e-6) * (1.0 + 1e-6) + std::sqrt(noise))
                                        full = true;
                                }
                            }
                            if (may && g.type == RT_PRIM_SPHERE) {  // umbra of the sphere for every point of the cell
                                const double rb = g.f[BND_R], r_lo = rb * (1.0 - 2e-6) - 2e-9, dA = hmodule(w);
                                const double dline = std::sqrt(hdot(e, e)) + Rl + rho;
                                if (r_lo * r_lo - dline * dline > 2e-6 * (1.0 + r_lo * r_lo) + noise && dline < r_lo && dA + rho + rb < len - rho - Rl - 2e-6 && dA - rho - rb > 2e-6)
                                    full = true;
                            }
                        }
                        if (may) m |= 1ull << j;
                        if (may && full) mfull |= 1ull << j;
                    }
                    const size_t cell = ((size_t)l * RT_CGRID + iz) * RT_CGRID * RT_CGRID + (size_t)iy * RT_CGRID + ix;
                    grid[2 * cell] = m, grid[2 * cell + 1] = mfull;
                }
        }
    }
    };
    unsigned n_threads = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("OMP_NUM_THREADS")) n_threads = std::min<unsigned>(n_threads, (unsigned)std::max(1, std::atoi(e)));  // the host's share of cores where one is set
    n_threads = std::max(1u, std::min(n_threads, 16u));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; t++) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
}

#endif
