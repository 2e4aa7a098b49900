// TEST INFRASTRUCTURE ONLY — driver that links the reference's own CPU sources (compiled where
// they lie under /root/reference by oracle/Makefile, never copied) and dumps what they compute,
// so that tools/make_golden.py can turn it into the committed fixtures under tests/golden/.
//
//   ref_dump parse  <scene.txt> <out.json> [bvh_out.bin]   parsed scene as JSON (doubles as C99
//                                               hex floats); optional canonical BVH dump
//   ref_dump render <scene.txt> <out.bin>       per-pixel nearest hit + pass-1 colour of
//                                               Raytracer::RayTracing (CR/raytracer.cpp:136-178)
//   ref_dump texture <file.bmp> <n> <out.bin>   Bmp::GetSmoothColor on an n x n lattice
//   ref_dump run <scene.txt> <out.bmp>          Raytracer::Run() timed (both passes), JSON on stderr
//
// Run it with the directory holding the scene's textures / meshes as working directory: the
// reference resolves asset paths against the process's cwd.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "raytracer.h"

static FILE* g_out = nullptr;  // JSON goes to a file: the reference prints progress text on stdout
#define P(...) std::fprintf(g_out, __VA_ARGS__)

static void jd(const char* k, double v, bool comma = true) { P("\"%s\":\"%a\"%s", k, v, comma ? "," : ""); }
static void jv(const char* k, const Vector3& v, bool comma = true) { P("\"%s\":[\"%a\",\"%a\",\"%a\"]%s", k, v.x, v.y, v.z, comma ? "," : ""); }
static void jc(const char* k, const Color& v, bool comma = true) { P("\"%s\":[\"%a\",\"%a\",\"%a\"]%s", k, v.r, v.g, v.b, comma ? "," : ""); }

struct BvhStats {
    size_t nodes = 0, leaves = 0;
    int depth = 0;
};

// canonical record per node in the reference's recursive visit order:
// [is_leaf, min xyz, max xyz, O1 xyz, O2 xyz, O3 xyz] = 16 doubles
static void walk(MeshBoundingBox* n, int depth, BvhStats& st, FILE* out, const Vector3* rootMin, const Vector3* rootMax) {
    if (!n) return;
    st.nodes++;
    if (depth > st.depth) st.depth = depth;
    double rec[16];
    std::memset(rec, 0, sizeof(rec));
    bool leaf = (n->left == nullptr && n->right == nullptr);
    rec[0] = leaf ? 1.0 : 0.0;
    Vector3 mn = n->min, mx = n->max;
    if (depth == 0 && !leaf) {  // the renderer tests the vertex bbox at the root (CR/Renderer_CUDA.cpp:234-235)
        mn = *rootMin;
        mx = *rootMax;
    }
    rec[1] = mn.x, rec[2] = mn.y, rec[3] = mn.z, rec[4] = mx.x, rec[5] = mx.y, rec[6] = mx.z;
    if (leaf && n->triangle) {
        st.leaves++;
        Triangle* t = n->triangle;
        rec[7] = t->O1.x, rec[8] = t->O1.y, rec[9] = t->O1.z;
        rec[10] = t->O2.x, rec[11] = t->O2.y, rec[12] = t->O2.z;
        rec[13] = t->O3.x, rec[14] = t->O3.y, rec[15] = t->O3.z;
    }
    if (out) std::fwrite(rec, sizeof(double), 16, out);
    walk(n->left, depth + 1, st, out, rootMin, rootMax);
    walk(n->right, depth + 1, st, out, rootMin, rootMax);
}

static int cmd_parse(const char* scene, const char* json_out, const char* bvh_out) {
    g_out = std::fopen(json_out, "w");
    if (!g_out) return 2;
    Raytracer rt;
    rt.SetInput(scene);
    rt.CreateAll();
    FILE* bo = bvh_out ? std::fopen(bvh_out, "wb") : nullptr;
    std::map<Primitive*, int> index;
    int n = 0;
    for (Primitive* p = rt.scene.GetPrimitiveHead(); p; p = p->GetNext()) index[p] = n++;

    P("{");
    jc("background_top", rt.background_color_top);
    jc("background_bottom", rt.background_color_bottom);
    Camera* c = rt.camera;
    P("\"camera\":{");
    jv("O", c->O), jv("N", c->N), jv("Dx", c->Dx), jv("Dy", c->Dy);
    P("\"W\":%d,\"H\":%d,", c->W, c->H);
    jd("shade_quality", c->shade_quality), jd("drefl_quality", c->drefl_quality), jd("lens_W", c->lens_W), jd("lens_H", c->lens_H, false);
    P("},\"lights\":[");
    bool first = true;
    for (Light* l = rt.light_head; l; l = l->GetNext()) {
        P("%s{", first ? "" : ",");
        first = false;
        int type = dynamic_cast<PointLight*>(l) ? 0 : (dynamic_cast<SquareLight*>(l) ? 1 : 2);
        P("\"type\":%d,\"light_primitive\":%d,", type, l->lightPrimitive ? index[l->lightPrimitive] : -1);
        jc("color", l->color), jv("O", l->O);
        if (SquareLight* s = dynamic_cast<SquareLight*>(l)) jv("Dx", s->Dx), jv("Dy", s->Dy);
        if (SphereLight* s = dynamic_cast<SphereLight*>(l)) jd("R", s->R);
        P("\"_\":0}");
    }
    P("],\"primitives\":[");
    first = true;
    for (Primitive* p = rt.scene.GetPrimitiveHead(); p; p = p->GetNext()) {
        P("%s{", first ? "" : ",");
        first = false;
        Material* m = p->GetMaterial();
        P("\"is_light\":%d,", p->IsLightPrimitive() ? 1 : 0);
        jc("color", m->color), jc("absor", m->absor), jd("refl", m->refl), jd("refr", m->refr), jd("diff", m->diff), jd("spec", m->spec);
        jd("rindex", m->rindex), jd("drefl", m->drefl);
        int tw = 0, th = 0;
        if (m->texture && m->texture->ima) tw = m->texture->GetW(), th = m->texture->GetH();
        P("\"texture_w\":%d,\"texture_h\":%d,", tw, th);
        if (Sphere* s = dynamic_cast<Sphere*>(p)) {
            P("\"type\":0,");
            jv("O", s->O), jv("De", s->De), jv("Dc", s->Dc), jd("R", s->R, false);
        } else if (Plane* s = dynamic_cast<Plane*>(p)) {
            P("\"type\":1,");
            jv("N", s->N), jv("Dx", s->Dx), jv("Dy", s->Dy), jd("R", s->R, false);
        } else if (Square* s = dynamic_cast<Square*>(p)) {
            P("\"type\":2,");
            jv("O", s->O), jv("Dx", s->Dx), jv("Dy", s->Dy, false);
        } else if (Cylinder* s = dynamic_cast<Cylinder*>(p)) {
            P("\"type\":3,");
            jv("O1", s->O1), jv("O2", s->O2), jd("R", s->R, false);
        } else if (Bezier* s = dynamic_cast<Bezier*>(p)) {
            P("\"type\":4,");
            jv("O1", s->O1), jv("O2", s->O2), jv("N", s->N), jv("Nx", s->Nx), jv("Ny", s->Ny), jd("R_c", s->R_c);
            P("\"degree\":%d,\"R\":[", s->degree);
            for (size_t i = 0; i < s->R.size(); i++) P("%s\"%a\"", i ? "," : "", s->R[i]);
            P("],\"Z\":[");
            for (size_t i = 0; i < s->Z.size(); i++) P("%s\"%a\"", i ? "," : "", s->Z[i]);
            P("]");
        } else if (Triangle* s = dynamic_cast<Triangle*>(p)) {
            P("\"type\":5,");
            jv("O1", s->O1), jv("O2", s->O2), jv("O3", s->O3), jv("N", s->N, false);
        } else if (Mesh* s = dynamic_cast<Mesh*>(p)) {
            P("\"type\":6,");
            BvhStats st;
            walk(s->root, 0, st, bo, &s->min, &s->max);
            jv("min", s->min), jv("max", s->max);
            if (s->root) jv("root_min", s->root->min), jv("root_max", s->root->max);
            P("\"triangles\":%zu,\"bvh_nodes\":%zu,\"bvh_leaves\":%zu,\"bvh_depth\":%d", s->triangles.size(), st.nodes, st.leaves, st.depth);
        }
        P("}");
    }
    P("]}\n");
    std::fclose(g_out);
    if (bo) std::fclose(bo);
    return 0;
}

static int cmd_render(const char* scene, const char* out) {
    Raytracer rt;
    rt.SetInput(scene);
    rt.CreateAll();
    std::map<Primitive*, int> index;
    int n = 0;
    for (Primitive* p = rt.scene.GetPrimitiveHead(); p; p = p->GetNext()) index[p] = n++;
    int H = rt.GetH(), W = rt.GetW();
    FILE* f = std::fopen(out, "wb");
    if (!f) return 2;
    int32_t hdr[2] = {W, H};
    std::fwrite(hdr, 4, 2, f);
    Vector3 O = rt.camera->GetO();
    std::vector<int32_t> hit((size_t)W * H);
    std::vector<double> dist((size_t)W * H), rgb((size_t)W * H * 3);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            Vector3 V = rt.camera->Emit(i, j);
            CollidePrimitive cp = rt.scene.FindNearestPrimitiveGetCollide(O, V);
            size_t k = (size_t)i * W + j;
            hit[k] = cp.isCollide ? index[cp.collide_primitive] : -1;
            dist[k] = cp.dist;
            int hash = 0;
            Color c = rt.RayTracing(O, V, 1, &hash);
            rgb[3 * k] = c.r, rgb[3 * k + 1] = c.g, rgb[3 * k + 2] = c.b;
        }
    std::fwrite(hit.data(), 4, hit.size(), f);
    std::fwrite(dist.data(), 8, dist.size(), f);
    std::fwrite(rgb.data(), 8, rgb.size(), f);
    std::fclose(f);
    return 0;
}

static int cmd_texture(const char* bmp, int n, const char* out) {
    Bmp b;
    b.Input(bmp);
    if (!b.ima) return 2;
    FILE* f = std::fopen(out, "wb");
    if (!f) return 2;
    int32_t hdr[3] = {b.GetW(), b.GetH(), n};
    std::fwrite(hdr, 4, 3, f);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            // irregular lattice that also leaves [0,1) so the wrap branches are exercised
            double u = -0.25 + 1.5 * (i + 0.37) / n, v = -0.25 + 1.5 * (j + 0.61) / n;
            Color c = b.GetSmoothColor(u, v);
            double rec[5] = {u, v, c.r, c.g, c.b};
            std::fwrite(rec, 8, 5, f);
        }
    std::fclose(f);
    return 0;
}

static int cmd_run(const char* scene, const char* out) {
    Raytracer rt;
    rt.SetInput(scene);
    rt.SetOutput(out);
    auto t0 = std::chrono::steady_clock::now();
    rt.Run();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "{\"seconds\":%.6f,\"W\":%d,\"H\":%d}\n", s, rt.GetW(), rt.GetH());
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && !std::strcmp(argv[1], "parse")) return cmd_parse(argv[2], argv[3], argc > 4 ? argv[4] : nullptr);
    if (argc >= 4 && !std::strcmp(argv[1], "render")) return cmd_render(argv[2], argv[3]);
    if (argc >= 5 && !std::strcmp(argv[1], "texture")) return cmd_texture(argv[2], std::atoi(argv[3]), argv[4]);
    if (argc >= 4 && !std::strcmp(argv[1], "run")) return cmd_run(argv[2], argv[3]);
    std::fprintf(stderr, "usage: ref_dump parse|render|texture|run ...\n");
    return 1;
}
