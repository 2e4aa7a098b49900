// Host scene model: text scene grammar, OBJ / packed-mesh loading, BVH build + flattening,
// BMP texture decode, BMP output.  CPU only; produces the flat rt_scene_desc the device
// upload (and the CPU oracle used by the tests) consume.
//
// Behaviour follows the reference's loader so that a scene file means the same thing:
//   grammar + list order        CR/raytracer.cpp:180-275
//   material / primitive keys   CR/primitive.cpp:51-70,101-107,151-157,185-190,237-242,260-281,316-332,344-365
//   light keys + light prims    CR/Light.cpp:16-25,47-52,84-89,101-116
//   camera keys + basis         CR/camera.cpp:19-33,44-55,61-77
//   OBJ reader + transform      CR/primitive.cpp:379-471
//   BVH build                   CR/primitive.cpp:531-616
//   BMP reader                  CR/bmp.cpp:52-96        BMP writer  CR/BmpSave.cpp:5-44
#include "rt_amd.h"
#include "rt_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct V3 {
    double x = 0, y = 0, z = 0;
    V3() {}
    V3(double X, double Y, double Z) : x(X), y(Y), z(Z) {}
    double operator[](int a) const { return a == 0 ? x : (a == 1 ? y : z); }
};
inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(const V3& a, double k) { return V3(a.x * k, a.y * k, a.z * k); }
inline V3 operator/(const V3& a, double k) { return V3(a.x / k, a.y / k, a.z / k); }
inline V3 cross(const V3& a, const V3& b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline double module(const V3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 unit(const V3& a) { return a / module(a); }
// CR/vector3.cpp:142-156
inline V3 vertical(const V3& a) {
    V3 r = cross(a, V3(0, 0, 1));
    if (std::fabs(r.x) < RT_EPS && std::fabs(r.y) < RT_EPS && std::fabs(r.z) < RT_EPS) return V3(1, 0, 0);
    return unit(r);
}
inline rt_vec3 pod(const V3& a) { return rt_vec3{a.x, a.y, a.z}; }
inline void read_v3(std::stringstream& s, V3& v) { s >> v.x >> v.y >> v.z; }

struct Tri {
    V3 O1, O2, O3;
    uint32_t face;
};

struct HostMesh {
    std::vector<Tri> tris;
    V3 vmin{RT_BIG_DIST, RT_BIG_DIST, RT_BIG_DIST}, vmax{-RT_BIG_DIST, -RT_BIG_DIST, -RT_BIG_DIST};
};

struct HostPrim {
    int type = -1;
    bool is_light = false;
    rt_material mat{};
    // union of every primitive's fields; unused ones stay zero
    V3 O, De{0, 0, 1}, Dc{0, 1, 0}, N, Dx, Dy, O1, O2, O3, Nx, Ny, scale, rotation;
    double R = 0, R_c = 0;
    int degree = -1;
    std::vector<double> bzR, bzZ;
    int mesh = -1;
    HostPrim() {
        mat.texture = -1;
    }
};

struct HostLight {
    int type = -1;
    V3 color, O, Dx, Dy;
    double R = 0;
    int light_primitive = -1;
};

}  // namespace

struct rt_host_scene {
    rt_scene_desc desc{};
    std::vector<rt_light> lights;
    std::vector<rt_primitive> prims;
    std::vector<rt_texture> textures;
    std::vector<std::vector<uint8_t>> texture_data;
    std::vector<rt_mesh> meshes;
    std::vector<rt_bvh_node> nodes;
    std::vector<rt_bvh_leaf> leaves;
    std::string asset_dir;
    std::vector<std::string> warnings;
};

namespace {

std::string resolve(const rt_host_scene& hs, const std::string& file) {
    if (file.empty() || file[0] == '/' || hs.asset_dir.empty()) return file;
    return hs.asset_dir + "/" + file;
}

// ---- BMP texture (CR/bmp.cpp:52-96).  The reader takes the pixel rows straight after the
// 54 header bytes whatever bfOffBits says, as the reference does. ---------------------------
int load_texture(rt_host_scene& hs, const std::string& file) {
    std::string path = resolve(hs, file);
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        hs.warnings.push_back("texture not found, primitive left untextured: " + path);
        return -1;
    }
    unsigned char hdr[54];
    if (std::fread(hdr, 1, 54, f) != 54 || hdr[0] != 'B' || hdr[1] != 'M') {
        std::fclose(f);
        hs.warnings.push_back("not a BMP file, primitive left untextured: " + path);
        return -1;
    }
    auto u32 = [&](int o) { return (uint32_t)hdr[o] | ((uint32_t)hdr[o + 1] << 8) | ((uint32_t)hdr[o + 2] << 16) | ((uint32_t)hdr[o + 3] << 24); };
    auto u16 = [&](int o) { return (uint32_t)hdr[o] | ((uint32_t)hdr[o + 1] << 8); };
    uint32_t W = u32(18), H = u32(22), bits = u16(28), comp = u32(30);
    if (bits != 24 || comp != 0 || W == 0 || H == 0 || W > 32768 || H > 32768) {
        std::fclose(f);
        hs.warnings.push_back("unsupported BMP format, primitive left untextured: " + path);
        return -1;
    }
    size_t padded = ((size_t)W * 3 + 3) & ~(size_t)3;
    std::vector<uint8_t> row(padded), rgb((size_t)W * H * 3);
    for (uint32_t i = 0; i < H; i++) {
        size_t got = std::fread(row.data(), 1, padded, f);
        if (got < padded) std::memset(row.data() + got, 0, padded - got);
        for (uint32_t j = 0; j < W; j++) {
            rgb[((size_t)i * W + j) * 3 + 0] = row[j * 3 + 2];
            rgb[((size_t)i * W + j) * 3 + 1] = row[j * 3 + 1];
            rgb[((size_t)i * W + j) * 3 + 2] = row[j * 3 + 0];
        }
    }
    std::fclose(f);
    hs.texture_data.push_back(std::move(rgb));
    rt_texture t{};
    t.width = (int32_t)W;
    t.height = (int32_t)H;
    t.rgb = nullptr;  // patched in finalize()
    hs.textures.push_back(t);
    return (int)hs.textures.size() - 1;
}

// ---- mesh files --------------------------------------------------------------------------
// Vertex transform of a mesh (what CR/primitive.cpp:390-442 does to every vertex): three plane rotations, about X, Y
// and Z in that order, then the scale, then the offset.  The rotation about axis k turns the coordinate pair (k+1, k+2)
// (indices mod 3), so one loop covers all three.  The products and sums per vertex are the reference's, one rounding
// each: the per-mesh tree (and its hash in the tests) depends on the last bit of every coordinate, which is why the
// rotations are not merged into one matrix.
struct Xform {
    double c[3], s[3], k[3], t[3];  // per axis: cosine and sine of the rotation about it, scale, offset
    Xform(const V3& rot, const V3& scale, const V3& o) {
        for (int a = 0; a < 3; a++) c[a] = std::cos(rot[a]), s[a] = std::sin(rot[a]), k[a] = scale[a], t[a] = o[a];
    }
    V3 apply(const V3& v) const {
        double p[3] = {v.x, v.y, v.z};
        for (int a = 0; a < 3; a++) {
            const int i = (a + 1) % 3, j = (a + 2) % 3;
            const double pi = c[a] * p[i] - s[a] * p[j], pj = s[a] * p[i] + c[a] * p[j];
            p[i] = pi, p[j] = pj;
        }
        for (int a = 0; a < 3; a++) p[a] = p[a] * k[a] + t[a];
        return V3(p[0], p[1], p[2]);
    }
};

void add_vertex(HostMesh& m, std::vector<V3>& verts, const V3& v) {
    m.vmin.x = std::min(m.vmin.x, v.x);
    m.vmin.y = std::min(m.vmin.y, v.y);
    m.vmin.z = std::min(m.vmin.z, v.z);
    m.vmax.x = std::max(m.vmax.x, v.x);
    m.vmax.y = std::max(m.vmax.y, v.y);
    m.vmax.z = std::max(m.vmax.z, v.z);
    verts.push_back(v);
}

bool add_face(HostMesh& m, const std::vector<V3>& verts, const int face[3]) {
    for (int k = 0; k < 3; k++)
        if (face[k] < 0 || (size_t)face[k] >= verts.size()) return false;  // the reference indexes out of bounds here
    Tri t;
    t.O1 = verts[face[0]];
    t.O2 = verts[face[1]];
    t.O3 = verts[face[2]];
    t.face = (uint32_t)m.tris.size();
    m.tris.push_back(t);
    return true;
}

// OBJ reader.  What the reference reads (CR/primitive.cpp:395-468: "v x y z" and "f a b c", 1-based, everything
// else ignored) is read the same way, triangle for triangle.  Forms the reference mis-reads are read as OBJ
// means them: "a/b/c", "a/b" and "a//c" corners use the vertex index a (the reference stops at the first '/'
// and builds a degenerate triangle), negative indices count back from the last vertex read so far, polygons
// are cut into a fan around their first corner (the reference keeps the first three corners only), and lines
// may be of any length (the reference cuts them at 255 characters).
bool load_obj(const std::string& path, const Xform& xf, HostMesh& m, std::string& err) {
    std::ifstream file(path);
    if (!file) {
        err = "cannot open mesh file " + path;
        return false;
    }
    std::vector<V3> verts;
    verts.reserve(50000);
    m.tris.reserve(70000);
    size_t bad = 0;
    std::string line;
    std::vector<int> corners;
    while (std::getline(file, line)) {
        std::stringstream ss(line);
        std::string type;
        ss >> type;
        if (type == "v") {
            V3 v;
            ss >> v.x >> v.y >> v.z;
            add_vertex(m, verts, xf.apply(v));
        } else if (type == "f") {
            corners.clear();
            std::string tok;
            while (ss >> tok) {
                char* end = nullptr;
                long idx = std::strtol(tok.c_str(), &end, 10);
                if (end == tok.c_str() || (*end != '\0' && *end != '/')) {  // not a corner: the reference stops reading the line here too
                    break;
                }
                corners.push_back(idx < 0 ? (int)((long)verts.size() + idx) : (int)(idx - 1));
            }
            if (corners.size() < 3) {
                bad++;
                continue;
            }
            for (size_t k = 1; k + 1 < corners.size(); k++) {
                const int face[3] = {corners[0], corners[k], corners[k + 1]};
                if (!add_face(m, verts, face)) bad++;
            }
        }
    }
    if (bad) {
        err = "mesh file " + path + " has faces with fewer than three corners or with corners that index missing vertices";
        return false;
    }
    return true;
}

// Packed mesh: "RTMESH1\0", u32 n_vertices, u32 n_faces, n_vertices*3 f64, n_faces*3 i32 (0-based).
// Same vertices, faces and order as an OBJ would give; exists so large meshes ship as exact doubles.
bool load_rtm(const std::string& path, const Xform& xf, HostMesh& m, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open mesh file " + path;
        return false;
    }
    char magic[8];
    uint32_t nv = 0, nf = 0;
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "RTMESH1", 8) == 0 &&
              std::fread(&nv, 4, 1, f) == 1 && std::fread(&nf, 4, 1, f) == 1 && nv < (1u << 28) && nf < (1u << 28);
    std::vector<double> vb;
    std::vector<int32_t> fb;
    if (ok) {
        vb.resize((size_t)nv * 3);
        fb.resize((size_t)nf * 3);
        ok = std::fread(vb.data(), 8, vb.size(), f) == vb.size() && std::fread(fb.data(), 4, fb.size(), f) == fb.size();
    }
    std::fclose(f);
    if (!ok) {
        err = "malformed packed mesh " + path;
        return false;
    }
    std::vector<V3> verts;
    verts.reserve(nv);
    m.tris.reserve(nf);
    for (uint32_t i = 0; i < nv; i++) add_vertex(m, verts, xf.apply(V3(vb[3 * i], vb[3 * i + 1], vb[3 * i + 2])));
    for (uint32_t i = 0; i < nf; i++) {
        int face[3] = {fb[3 * i], fb[3 * i + 1], fb[3 * i + 2]};
        if (!add_face(m, verts, face)) {
            err = "packed mesh " + path + " has faces that index missing vertices";
            return false;
        }
    }
    return true;
}

// ---- BVH (CR/primitive.cpp:552-616) ------------------------------------------------------
struct Box {
    V3 lo, hi;
};
struct Builder {
    const std::vector<Tri>& tris;
    std::vector<V3> centroid;
    std::vector<rt_bvh_node>& nodes;
    std::vector<rt_bvh_leaf>& leaves;
    uint32_t preorder = 0, max_depth = 0;

    Builder(const std::vector<Tri>& t, std::vector<rt_bvh_node>& n, std::vector<rt_bvh_leaf>& l) : tris(t), nodes(n), leaves(l) {
        centroid.resize(t.size());
        for (size_t i = 0; i < t.size(); i++) centroid[i] = (t[i].O1 + t[i].O2 + t[i].O3) / 3.0;
    }

    // returns the ref of the subtree built from `ids`; box receives its bounds
    uint32_t build(std::vector<uint32_t>& ids, uint32_t depth, Box& box) {
        max_depth = std::max(max_depth, depth);
        if (ids.size() == 1) {
            const Tri& t = tris[ids[0]];
            box.lo = V3(std::min({t.O1.x, t.O2.x, t.O3.x}), std::min({t.O1.y, t.O2.y, t.O3.y}), std::min({t.O1.z, t.O2.z, t.O3.z}));
            box.hi = V3(std::max({t.O1.x, t.O2.x, t.O3.x}), std::max({t.O1.y, t.O2.y, t.O3.y}), std::max({t.O1.z, t.O2.z, t.O3.z}));
            rt_bvh_leaf lf{};
            lf.O1 = pod(t.O1);
            lf.O2 = pod(t.O2);
            lf.O3 = pod(t.O3);
            lf.skip = RT_REF_END;
            lf.preorder = preorder++;
            lf.face = t.face;
            leaves.push_back(lf);
            return RT_REF_LEAF | (uint32_t)(leaves.size() - 1);
        }
        uint32_t self = (uint32_t)nodes.size();
        nodes.push_back(rt_bvh_node{});
        nodes[self].preorder = preorder++;
        int axis = (int)(depth % 3);
        const std::vector<V3>& c = centroid;
        // CR/primitive.cpp:566-569 sorts by the centroid alone, which leaves the order of equal centroids to the library's
        // std::sort; the triangle index breaks such ties here, so the tree does not depend on the standard library
        std::sort(ids.begin(), ids.end(), [axis, &c](uint32_t a, uint32_t b) { return c[a][axis] < c[b][axis] || (c[a][axis] == c[b][axis] && a < b); });
        size_t mid = ids.size() / 2;
        std::vector<uint32_t> l(ids.begin(), ids.begin() + mid), r(ids.begin() + mid, ids.end());
        Box bl, br;
        uint32_t lref = build(l, depth + 1, bl);
        uint32_t rref = build(r, depth + 1, br);
        box.lo = V3(std::min(bl.lo.x, br.lo.x), std::min(bl.lo.y, br.lo.y), std::min(bl.lo.z, br.lo.z));
        box.hi = V3(std::max(bl.hi.x, br.hi.x), std::max(bl.hi.y, br.hi.y), std::max(bl.hi.z, br.hi.z));
        rt_bvh_node& n = nodes[self];
        n.bmin[0] = box.lo.x, n.bmin[1] = box.lo.y, n.bmin[2] = box.lo.z;
        n.bmax[0] = box.hi.x, n.bmax[1] = box.hi.y, n.bmax[2] = box.hi.z;
        n.left = lref;
        n.right = rref;
        n.skip = RT_REF_END;
        return self;
    }

    void set_skip(uint32_t ref, uint32_t skip) {
        if (ref == RT_REF_END) return;
        if (ref & RT_REF_LEAF) {
            leaves[ref & ~RT_REF_LEAF].skip = skip;
            return;
        }
        nodes[ref].skip = skip;
        set_skip(nodes[ref].left, nodes[ref].right);
        set_skip(nodes[ref].right, skip);
    }
};

int add_mesh(rt_host_scene& hs, const HostMesh& hm) {
    rt_mesh m{};
    m.node_base = (uint32_t)hs.nodes.size();
    m.leaf_base = (uint32_t)hs.leaves.size();
    m.root = RT_REF_END;
    if (!hm.tris.empty()) {
        Builder b(hm.tris, hs.nodes, hs.leaves);
        std::vector<uint32_t> ids(hm.tris.size());
        for (size_t i = 0; i < ids.size(); i++) ids[i] = (uint32_t)i;
        Box box;
        m.root = b.build(ids, 0, box);
        b.set_skip(m.root, RT_REF_END);
        m.depth = b.max_depth;
        if (!(m.root & RT_REF_LEAF)) {
            // the root box the renderer tests is the vertex bounding box (CR/Renderer_CUDA.cpp:234-235)
            rt_bvh_node& n = hs.nodes[m.root];
            n.bmin[0] = hm.vmin.x, n.bmin[1] = hm.vmin.y, n.bmin[2] = hm.vmin.z;
            n.bmax[0] = hm.vmax.x, n.bmax[1] = hm.vmax.y, n.bmax[2] = hm.vmax.z;
        }
    }
    m.node_count = (uint32_t)hs.nodes.size() - m.node_base;
    m.leaf_count = (uint32_t)hs.leaves.size() - m.leaf_base;
    hs.meshes.push_back(m);
    return (int)hs.meshes.size() - 1;
}

// ---- keys of a block ---------------------------------------------------------------------
// A block is a list of "key= values" lines.  Which keys a block understands is a table: key -> the field it fills
// (a number, a vector, a whole number); keys with an effect beyond storing a value have a hook.  Keys a block does not know
// are skipped, as in the reference (a chain of string comparisons per class: CR/primitive.cpp:51-70,101-107,151-157,
// 185-190,237-242,260-281,316-332,344-365, CR/Light.cpp:16-25,47-52,101-105, CR/camera.cpp:61-77).
template <class T>
struct Key {
    const char* name;
    double T::*num;
    V3 T::*vec;
    int T::*whole;
};
template <class T, size_t N>
bool read_key(T& obj, const Key<T> (&keys)[N], const std::string& name, std::stringstream& in) {
    for (const Key<T>& k : keys) {
        if (name != k.name) continue;
        if (k.num) in >> obj.*(k.num);
        else if (k.vec) read_v3(in, obj.*(k.vec));
        else if (k.whole) in >> obj.*(k.whole);
        return true;
    }
    return false;
}
#define NUM(T, key, field) {key, &T::field, nullptr, nullptr}
#define VEC(T, key, field) {key, nullptr, &T::field, nullptr}
#define INT(T, key, field) {key, nullptr, nullptr, &T::field}

// material keys are the same for every primitive; they fill the POD record directly
struct MatKey {
    const char* name;
    double rt_material::*num;
    rt_vec3 rt_material::*vec;
};
const MatKey kMaterialKeys[] = {{"color=", nullptr, &rt_material::color}, {"absor=", nullptr, &rt_material::absor}, {"refl=", &rt_material::refl, nullptr},
                                {"refr=", &rt_material::refr, nullptr},   {"diff=", &rt_material::diff, nullptr},   {"spec=", &rt_material::spec, nullptr},
                                {"drefl=", &rt_material::drefl, nullptr}, {"rindex=", &rt_material::rindex, nullptr}};
void read_material_key(rt_host_scene& hs, rt_material& m, const std::string& name, std::stringstream& in) {
    if (name == "texture=") {
        std::string file;
        in >> file;
        m.texture = load_texture(hs, file);
        return;
    }
    for (const MatKey& k : kMaterialKeys) {
        if (name != k.name) continue;
        if (k.num) in >> m.*(k.num);
        else in >> (m.*(k.vec)).x >> (m.*(k.vec)).y >> (m.*(k.vec)).z;
        return;
    }
}

const Key<HostPrim> kSphereKeys[] = {VEC(HostPrim, "O=", O), NUM(HostPrim, "R=", R), VEC(HostPrim, "De=", De), VEC(HostPrim, "Dc=", Dc)};
const Key<HostPrim> kPlaneKeys[] = {VEC(HostPrim, "N=", N), NUM(HostPrim, "R=", R), VEC(HostPrim, "Dx=", Dx), VEC(HostPrim, "Dy=", Dy)};
const Key<HostPrim> kSquareKeys[] = {VEC(HostPrim, "O=", O), VEC(HostPrim, "Dx=", Dx), VEC(HostPrim, "Dy=", Dy)};
const Key<HostPrim> kCylinderKeys[] = {VEC(HostPrim, "O1=", O1), VEC(HostPrim, "O2=", O2), NUM(HostPrim, "R=", R)};
const Key<HostPrim> kBezierKeys[] = {VEC(HostPrim, "O1=", O1), VEC(HostPrim, "O2=", O2)};
const Key<HostPrim> kTriangleKeys[] = {VEC(HostPrim, "O1=", O1), VEC(HostPrim, "O2=", O2), VEC(HostPrim, "O3=", O3)};
const Key<HostPrim> kMeshKeys[] = {VEC(HostPrim, "O=", O), VEC(HostPrim, "rotation=", rotation), VEC(HostPrim, "scale=", scale)};
const Key<HostLight> kPointLightKeys[] = {VEC(HostLight, "O=", O), VEC(HostLight, "color=", color)};
const Key<HostLight> kSquareLightKeys[] = {VEC(HostLight, "O=", O), VEC(HostLight, "Dx=", Dx), VEC(HostLight, "Dy=", Dy), VEC(HostLight, "color=", color)};
const Key<HostLight> kSphereLightKeys[] = {VEC(HostLight, "O=", O), NUM(HostLight, "R=", R), VEC(HostLight, "color=", color)};

// One line of a primitive block.  Hooks: a Bezier's axis frame is made when its far end is read and every "P=" adds a
// control point (CR/primitive.cpp:260-281); a triangle's normal is made when its third corner is read (:316-332); a
// mesh is loaded when its file is named, with the offset / rotation / scale read so far (:344-365), so those keys must
// come before "file=".
bool primitive_input(rt_host_scene& hs, HostPrim& p, const std::string& var, std::stringstream& fin, std::string& err) {
    switch (p.type) {
        case RT_PRIM_SPHERE: read_key(p, kSphereKeys, var, fin); break;
        case RT_PRIM_PLANE: read_key(p, kPlaneKeys, var, fin); break;
        case RT_PRIM_SQUARE: read_key(p, kSquareKeys, var, fin); break;
        case RT_PRIM_CYLINDER: read_key(p, kCylinderKeys, var, fin); break;
        case RT_PRIM_BEZIER:
            if (read_key(p, kBezierKeys, var, fin)) {
                if (var == "O2=") p.N = unit(p.O2 - p.O1), p.Nx = vertical(p.N), p.Ny = cross(p.N, p.Nx);
            } else if (var == "P=") {
                double z = 0, r = 0;
                fin >> z >> r;
                p.degree++;
                p.R_c = std::max(p.R_c, r);
                p.bzZ.push_back(z), p.bzR.push_back(r);
            }
            break;
        case RT_PRIM_TRIANGLE:
            if (read_key(p, kTriangleKeys, var, fin) && var == "O3=") p.N = unit(cross(p.O2 - p.O1, p.O3 - p.O1));
            break;
        case RT_PRIM_MESH:
            if (!read_key(p, kMeshKeys, var, fin) && var == "file=") {
                std::string file;
                fin >> file;
                const std::string path = resolve(hs, file);
                const bool packed = path.size() > 4 && path.compare(path.size() - 4, 4, ".rtm") == 0;
                const Xform xf(p.rotation, p.scale, p.O);
                HostMesh hm;
                if (!(packed ? load_rtm(path, xf, hm, err) : load_obj(path, xf, hm, err))) return false;
                p.mesh = add_mesh(hs, hm);
            }
            break;
        default: break;
    }
    read_material_key(hs, p.mat, var, fin);
    return true;
}

void light_input(HostLight& l, const std::string& var, std::stringstream& fin) {
    if (l.type == RT_LIGHT_POINT) read_key(l, kPointLightKeys, var, fin);
    else if (l.type == RT_LIGHT_SQUARE) read_key(l, kSquareLightKeys, var, fin);
    else if (l.type == RT_LIGHT_SPHERE) read_key(l, kSphereLightKeys, var, fin);
}

struct HostCamera {
    V3 O{0, 0, 0}, N{0, 1, 0}, Dx, Dy;
    double lens_W = 0.88, lens_H = 0.88;
    int W = 420, H = 420;
    double shade_quality = 4, drefl_quality = 4;
    void input(const std::string& var, std::stringstream& fin);
    void initialize() {
        N = unit(N);
        Dx = vertical(N);
        Dy = cross(Dx, N);
        Dx = Dx * lens_W / 2;
        Dy = Dy * lens_H / 2;
    }
};

const Key<HostCamera> kCameraKeys[] = {VEC(HostCamera, "O=", O), VEC(HostCamera, "N=", N), NUM(HostCamera, "lens_W=", lens_W),
                                      NUM(HostCamera, "lens_H=", lens_H), INT(HostCamera, "image_W=", W), INT(HostCamera, "image_H=", H),
                                      NUM(HostCamera, "shade_quality=", shade_quality), NUM(HostCamera, "drefl_quality=", drefl_quality)};
void HostCamera::input(const std::string& var, std::stringstream& fin) {
    if (read_key(*this, kCameraKeys, var, fin) && var == "N=") N = unit(N);  // CR/camera.cpp:63-66
}

// ---- blocks of a scene file (CR/raytracer.cpp:203-271) ----------------------------------------------
// A file is a sequence of blocks "<kind> [<type>] ... end".  Words between blocks are read one at a time: a word that
// starts with '#' is dropped (that one word only, so a comment containing the word "camera" opens a camera block, as in
// the reference), any other word that is not a block kind is dropped too.  The rest of the header line (up to 1024
// characters) is skipped; then every line is "key= values" until a line whose first word is "end".  A primitive or
// light of a type the loader does not know still has its block consumed.
struct Name {
    const char* word;
    int id;
};
enum { BLOCK_BACKGROUND, BLOCK_CAMERA, BLOCK_PRIMITIVE, BLOCK_LIGHT };
const Name kBlockKinds[] = {{"background", BLOCK_BACKGROUND}, {"camera", BLOCK_CAMERA}, {"primitive", BLOCK_PRIMITIVE}, {"light", BLOCK_LIGHT}};
const Name kPrimitiveTypes[] = {{"sphere", RT_PRIM_SPHERE},     {"plane", RT_PRIM_PLANE},   {"square", RT_PRIM_SQUARE}, {"cylinder", RT_PRIM_CYLINDER},
                                {"bezier", RT_PRIM_BEZIER},     {"triangle", RT_PRIM_TRIANGLE}, {"mesh", RT_PRIM_MESH}};
const Name kLightTypes[] = {{"point", RT_LIGHT_POINT}, {"square", RT_LIGHT_SQUARE}, {"sphere", RT_LIGHT_SPHERE}};
template <size_t N>
int find_name(const Name (&table)[N], const std::string& word) {
    for (const Name& n : table)
        if (word == n.word) return n.id;
    return -1;
}
struct BlockHeader {
    int kind = -1;
    int type = -1;  // primitive / light type, -1 if the block has none or the type is unknown
};
bool next_block(std::istream& in, BlockHeader& h) {
    std::string word;
    while (in >> word) {
        if (word[0] == '#') continue;
        h.kind = find_name(kBlockKinds, word);
        if (h.kind < 0) continue;
        h.type = -1;
        if (h.kind == BLOCK_PRIMITIVE || h.kind == BLOCK_LIGHT) {
            std::string type;
            in >> type;
            h.type = h.kind == BLOCK_PRIMITIVE ? find_name(kPrimitiveTypes, type) : find_name(kLightTypes, type);
        }
        in.ignore(1024, '\n');
        return true;
    }
    return false;
}
// fn(key, rest of the line) for every line of the block; fn returns false to stop reading
template <class F>
void each_key(std::istream& in, F fn) {
    std::string line;
    while (std::getline(in, line, '\n')) {
        std::stringstream rest(line);
        std::string key;
        rest >> key;
        if (!key.empty() && key[0] == '#') continue;
        if (key == "end") break;
        if (!fn(key, rest)) break;
    }
}

rt_primitive to_pod(const HostPrim& p, std::string& err) {
    rt_primitive q;
    std::memset(&q, 0, sizeof(q));
    q.type = p.type;
    q.is_light = p.is_light ? 1 : 0;
    q.material = p.mat;
    switch (p.type) {
        case RT_PRIM_SPHERE:
            q.u.sphere.O = pod(p.O), q.u.sphere.De = pod(p.De), q.u.sphere.Dc = pod(p.Dc), q.u.sphere.R = p.R;
            break;
        case RT_PRIM_PLANE:
            q.u.plane.N = pod(p.N), q.u.plane.Dx = pod(p.Dx), q.u.plane.Dy = pod(p.Dy), q.u.plane.R = p.R;
            break;
        case RT_PRIM_SQUARE:
            q.u.square.O = pod(p.O), q.u.square.Dx = pod(p.Dx), q.u.square.Dy = pod(p.Dy);
            break;
        case RT_PRIM_CYLINDER:
            q.u.cylinder.O1 = pod(p.O1), q.u.cylinder.O2 = pod(p.O2), q.u.cylinder.R = p.R;
            break;
        case RT_PRIM_BEZIER:
            q.u.bezier.O1 = pod(p.O1), q.u.bezier.O2 = pod(p.O2), q.u.bezier.N = pod(p.N);
            q.u.bezier.Nx = pod(p.Nx), q.u.bezier.Ny = pod(p.Ny), q.u.bezier.R_c = p.R_c;
            q.u.bezier.degree = p.degree;
            if (p.bzR.size() > RT_BEZIER_MAX_CTRL) {
                err = "bezier primitive has more than 10 control points";
                break;
            }
            for (size_t i = 0; i < p.bzR.size(); i++) q.u.bezier.R[i] = p.bzR[i], q.u.bezier.Z[i] = p.bzZ[i];
            break;
        case RT_PRIM_TRIANGLE:
            q.u.triangle.O1 = pod(p.O1), q.u.triangle.O2 = pod(p.O2), q.u.triangle.O3 = pod(p.O3), q.u.triangle.N = pod(p.N);
            break;
        case RT_PRIM_MESH:
            q.u.mesh.mesh = p.mesh;
            break;
    }
    return q;
}

void finalize(rt_host_scene& hs) {
    for (size_t i = 0; i < hs.textures.size(); i++) hs.textures[i].rgb = hs.texture_data[i].data();
    rt_scene_desc& d = hs.desc;
    d.n_lights = (int32_t)hs.lights.size();
    d.n_primitives = (int32_t)hs.prims.size();
    d.n_textures = (int32_t)hs.textures.size();
    d.n_meshes = (int32_t)hs.meshes.size();
    d.n_nodes = (uint32_t)hs.nodes.size();
    d.n_leaves = (uint32_t)hs.leaves.size();
    d.lights = hs.lights.data();
    d.primitives = hs.prims.data();
    d.textures = hs.textures.data();
    d.meshes = hs.meshes.data();
    d.nodes = hs.nodes.data();
    d.leaves = hs.leaves.data();
}

}  // namespace

extern "C" int rt_host_scene_load(const char* scene_path, const char* asset_dir, rt_host_scene** out) {
    if (!scene_path || !out) return rt_fail(RT_ERR_INVALID, "rt_host_scene_load: null argument");
    *out = nullptr;
    std::ifstream fin(scene_path);
    if (!fin) return rt_fail(RT_ERR_IO, std::string("cannot open scene file ") + scene_path);
    std::unique_ptr<rt_host_scene> hs(new rt_host_scene);
    if (asset_dir) hs->asset_dir = asset_dir;

    V3 bg_top, bg_bottom;
    HostCamera cam;
    std::vector<HostPrim> prims;    // file order
    std::vector<HostLight> lights;  // file order
    std::string err;

    BlockHeader block;
    while (next_block(fin, block)) {
        HostPrim* prim = nullptr;
        HostLight* light = nullptr;
        if (block.kind == BLOCK_PRIMITIVE && block.type >= 0) {
            prims.emplace_back();
            prims.back().type = block.type;
            prim = &prims.back();
        } else if (block.kind == BLOCK_LIGHT && block.type >= 0) {
            lights.emplace_back();
            lights.back().type = block.type;
            light = &lights.back();
        }
        bool ok = true;
        each_key(fin, [&](const std::string& key, std::stringstream& values) {
            switch (block.kind) {
                case BLOCK_BACKGROUND:  // CR/raytracer.cpp:240-254: one colour, or a gradient whose bottom defaults to white
                    if (key == "color=") read_v3(values, bg_top), bg_bottom = bg_top;
                    else if (key == "top=") read_v3(values, bg_top), bg_bottom = V3(1.0, 1.0, 1.0);
                    else if (key == "bottom=") read_v3(values, bg_bottom);
                    break;
                case BLOCK_CAMERA: cam.input(key, values); break;
                case BLOCK_PRIMITIVE:
                    if (prim) ok = primitive_input(*hs, *prim, key, values, err);
                    break;
                case BLOCK_LIGHT:
                    if (light) light_input(*light, key, values);
                    break;
            }
            return ok;
        });
        if (!ok) return rt_fail(RT_ERR_IO, err);
    }
    cam.initialize();

    // List order (CR/raytracer.cpp:180-193,221-224,232-235): primitives and lights are
    // prepended, so both lists run in reverse file order; then one primitive per area light
    // is prepended while walking the light list, which leaves them in file order up front.
    std::vector<HostPrim> plist;
    std::vector<int> light_prim_of(lights.size(), -1);
    for (size_t i = 0; i < lights.size(); i++) {
        const HostLight& l = lights[i];
        if (l.type == RT_LIGHT_POINT) continue;
        HostPrim p;
        p.is_light = true;
        p.mat.color = pod(l.color);
        if (l.type == RT_LIGHT_SQUARE) {
            p.type = RT_PRIM_SQUARE;
            p.O = l.O, p.Dx = l.Dx, p.Dy = l.Dy;
        } else {
            p.type = RT_PRIM_SPHERE;
            p.O = l.O, p.R = l.R;
        }
        light_prim_of[i] = (int)plist.size();
        plist.push_back(p);
    }
    for (size_t i = prims.size(); i-- > 0;) plist.push_back(prims[i]);
    for (const HostPrim& p : plist) {
        hs->prims.push_back(to_pod(p, err));
        if (!err.empty()) return rt_fail(RT_ERR_LIMIT, err);
    }
    for (size_t i = lights.size(); i-- > 0;) {
        const HostLight& l = lights[i];
        rt_light q;
        std::memset(&q, 0, sizeof(q));
        q.type = l.type;
        q.light_primitive = light_prim_of[i];
        q.color = pod(l.color), q.O = pod(l.O), q.Dx = pod(l.Dx), q.Dy = pod(l.Dy), q.R = l.R;
        hs->lights.push_back(q);
    }

    rt_scene_desc& d = hs->desc;
    d.background_top = pod(bg_top);
    d.background_bottom = pod(bg_bottom);
    d.camera.O = pod(cam.O), d.camera.N = pod(cam.N), d.camera.Dx = pod(cam.Dx), d.camera.Dy = pod(cam.Dy);
    d.camera.W = cam.W, d.camera.H = cam.H;
    d.camera.shade_quality = cam.shade_quality, d.camera.drefl_quality = cam.drefl_quality;
    d.camera.lens_W = cam.lens_W, d.camera.lens_H = cam.lens_H;
    finalize(*hs);
    // Missing or unreadable textures: the reference's CUDA path renders the primitive untextured and its CPU path
    // crashes (CR/bmp.cpp:133-150 indexes an empty image).  Default here: untextured plus a warning; with
    // RT_AMD_STRICT_ASSETS=1 in the environment the load fails instead, naming the first such file.
    if (!hs->warnings.empty()) {
        const char* strict = std::getenv("RT_AMD_STRICT_ASSETS");
        if (strict && strict[0] == '1') return rt_fail(RT_ERR_IO, hs->warnings.front());
    }
    *out = hs.release();
    return RT_OK;
}

extern "C" int rt_host_scene_set_viewport(rt_host_scene* hs, int width, int height) {
    if (!hs || width <= 0 || height <= 0) return rt_fail(RT_ERR_INVALID, "rt_host_scene_set_viewport: bad argument");
    hs->desc.camera.W = width;
    hs->desc.camera.H = height;
    return RT_OK;
}

extern "C" const rt_scene_desc* rt_host_scene_desc(const rt_host_scene* hs) { return hs ? &hs->desc : nullptr; }

extern "C" int rt_host_scene_warning_count(const rt_host_scene* hs) { return hs ? (int)hs->warnings.size() : 0; }
extern "C" const char* rt_host_scene_warning(const rt_host_scene* hs, int i) {
    return (hs && i >= 0 && (size_t)i < hs->warnings.size()) ? hs->warnings[i].c_str() : nullptr;
}

extern "C" void rt_host_scene_free(rt_host_scene* hs) { delete hs; }

extern "C" int rt_bmp_save_rgba(const char* path, const uint8_t* rgba, int width, int height) {
    if (!path || !rgba || width <= 0 || height <= 0) return rt_fail(RT_ERR_INVALID, "rt_bmp_save_rgba: bad argument");
    // Rows are padded to a multiple of four bytes, as the format requires and as this library's own reader expects
    // (CR/BmpSave.cpp:5-44 writes them unpadded, which only is a valid file when width * 3 is a multiple of four; for such
    // widths - every size the reference uses - the two writers produce the same bytes).
    const size_t row = ((size_t)width * 3 + 3) & ~(size_t)3, bytes = row * (size_t)height;
    if (bytes + 54 > 0xFFFFFFFFull) return rt_fail(RT_ERR_LIMIT, "rt_bmp_save_rgba: image too large for a BMP file");
    const uint32_t data = (uint32_t)bytes;
    unsigned char hdr[54];
    std::memset(hdr, 0, sizeof(hdr));
    auto p32 = [&](int o, uint32_t v) { hdr[o] = v & 255, hdr[o + 1] = (v >> 8) & 255, hdr[o + 2] = (v >> 16) & 255, hdr[o + 3] = (v >> 24) & 255; };
    hdr[0] = 'B', hdr[1] = 'M';
    p32(2, 54u + data);
    p32(10, 54u);
    p32(14, 40u);
    p32(18, (uint32_t)width);
    p32(22, (uint32_t)height);
    hdr[26] = 1;
    hdr[28] = 24;
    p32(34, data);
    std::vector<uint8_t> bgr(bytes, 0);
    for (size_t y = 0; y < (size_t)height; y++)
        for (size_t x = 0; x < (size_t)width; x++) {
            const uint8_t* px = rgba + 4 * (y * (size_t)width + x);
            uint8_t* out = &bgr[y * row + 3 * x];
            out[0] = px[2], out[1] = px[1], out[2] = px[0];
        }
    FILE* f = std::fopen(path, "wb");
    if (!f) return rt_fail(RT_ERR_IO, std::string("cannot open for writing: ") + path);
    bool ok = std::fwrite(hdr, 1, 54, f) == 54 && std::fwrite(bgr.data(), 1, bgr.size(), f) == bgr.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : rt_fail(RT_ERR_IO, std::string("short write: ") + path);
}
