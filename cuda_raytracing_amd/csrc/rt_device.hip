// Render kernels for gfx950 and the device half of the C ABI (include/rt_amd.h).
//
// Execution model (wave64, no LDS, no MFMA - this is a pointer walk, not a contraction):
//   * a persistent grid: every wave pops tasks from one atomic counter until it runs dry;
//   * a task = the samples of a small group of pixels, one (pixel, sample) per lane, so the 64
//     lanes of a wave trace 64 paths at once and the image is resolved in-wave: per-sample
//     colours are summed in sample order through lane reads (the reference adds them in that
//     order inside one thread, CR/RayTracer.cu:74-84, and double addition is not associative);
//   * the recursion of CR/Scene.cu:23-58 becomes a loop over an explicit frame stack (<= 10
//     frames per lane, in scratch) that visits reflect / refract children in the reference's
//     order, so each path consumes its random draws in the same sequence;
//   * shading is wave-cooperative: the shadow rays of all lanes that need shading (one per
//     point light, 16*shade_quality per area light, CR/Shade.cu:23-113) are dealt out over all
//     64 lanes, 64 at a time.  Rays of one shaded point fly to the same small lamp, so
//     neighbouring lanes walk the same BVH nodes.  Visible-sample counts are integers, so the
//     sum does not depend on which lane traced which ray.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "rt_internal.h"
#include "rt_trace.hpp"

static_assert(sizeof(rt_bvh_node) == 64 && sizeof(rt_bvh_leaf) == 96, "BVH record sizes are part of the ABI");
static_assert(RT_SPEC_POWER == 20, "the specular power is expanded by hand in diffusion_coop");
static_assert(sizeof(DevShade) == 216 && sizeof(DevShade) % 4 == 0, "shading record");
static_assert(sizeof(DevNode4) == 128 && sizeof(DevGeom) == 192, "device record sizes");
static_assert(sizeof(rt_primitive) == 408 && sizeof(rt_light) == 112 && sizeof(rt_scene_desc) == 256 && sizeof(rt_settings) == 40 &&
                  sizeof(rt_stats) == 80,
              "record sizes are part of the ABI (the Python binding mirrors them)");

struct Tile {
    int32_t x0, y0, w, h;
};

// Work decomposition.  Pixels of a tile are numbered in 8x8 blocks ("slots" 0..63 inside a block, row-major);
// blocks are numbered tile after tile.  Path g of a launch is sample g % spp of slot (g / spp) % 64 of block
// g / (64 * spp): consecutive paths are consecutive samples of one pixel, then the next pixel of the block.
// Every wave keeps 64 paths in flight, one per lane, and draws the next unstarted paths from one counter
// for the whole launch, so all waves stay full until the launch runs dry.  The colour of a finished path goes
// to path_out[((block * spp + sample) * 64 + slot) * 3]; resolve_kernel adds the samples of a pixel afterwards.
struct BlockRec {  // one 8x8 block: its pixel origin, and how much of it lies inside its tile
    uint16_t x0, y0;
    uint8_t w, h;  // 1..8
    uint16_t pad_;
};
static_assert(sizeof(BlockRec) == 8, "block record");
struct RenderParams {
    DevScene sc;
    int32_t tex_w, tex_h, spp, dof;
    double focal, aperture;
    double inv_block_paths, inv_spp;  // reciprocals (rounded up a little) for the path index decode
    uint32_t k0, k1, flags;
    uint32_t n_blocks, n_paths;  // of this launch: n_paths = n_blocks * 64 * spp < 2^30
    uint32_t range_paths;        // paths per range (a whole number of blocks); eight ranges cover the launch
    const BlockRec* blocks;
    uint32_t* queue;   // eight words: paths handed out of each range
    uint8_t* out;      // tex_w * tex_h * 4, row 0 = bottom
    double* path_out;  // n_paths * 3 doubles: colours of finished paths
    unsigned long long* counters;
};
#ifndef RT_LAUNCH_PATHS
#define RT_LAUNCH_PATHS (1ull << 28)  // most paths of one launch (their colours take 24 bytes each: 6.4 GB); larger calls are cut into several launches
#endif
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 32  // idle lanes that trigger a refill: the set-up of new paths is long, and running it for one lane costs the same
                          // (measured ms/frame: 8: 52.0, 16: 49.1, 24: 47.7, 32: 47.1, 40: 47.3, 48: 47.8)
#endif

enum { CTR_SECONDARY = 0, CTR_SHADOW, CTR_NODES, CTR_TRIS, CTR_SHADED, CTR_TEXFETCH, CTR_COUNT,
       // lane-utilisation probes, only counted in -DRT_UTIL_COUNTERS builds (printed by rt_render_* to stderr)
       CTR_U_SHADOW_SLOTS = CTR_COUNT, CTR_U_SHADOW_ITEMS, CTR_U_TRACE_SLOTS, CTR_U_TRACE_ACTIVE, CTR_U_PROBES, CTR_ALL = CTR_U_PROBES + 12 };

// ---- camera: CR/RayTracer.cu:52-61, CR/camera.cpp:57-59 -------------------------------------------
DEV d3 emit(const DevScene& sc, double x, double y) {
    return sc.camN + sc.camDy * (2 * y / sc.cam_h - 1) + sc.camDx * (2 * x / sc.cam_w - 1);
}
DEV void map_pixel(const RenderParams& P, int sx, int sy, int& x, int& y) {
    if (P.flags & RT_F_PIXEL_MAP_TRUNC) {
        x = (int)((double)sx / P.tex_w * P.sc.cam_w);
        y = (int)((double)sy / P.tex_h * P.sc.cam_h);
    } else {
        x = (int)(((long long)sx * P.sc.cam_w) / P.tex_w);
        y = (int)(((long long)sy * P.sc.cam_h) / P.tex_h);
    }
}
DEV d3 clamp1(d3 c) {  // Color::Confine, CR/color.cpp:53-57 / CR/Scene.cu:54-56
    if (c.x > 1) c.x = 1;
    if (c.y > 1) c.y = 1;
    if (c.z > 1) c.z = 1;
    return c;
}
DEV uint32_t to_u8(double c) {
    double v = c * 255;
    if (!(v >= 0)) return 0;
    if (v > 255) return 255;
    return (uint32_t)(int)v;
}

// ---- CalnDiffusion for every lane that needs it, shadow rays dealt over the wave ---------------
// CR/Scene_Material.cu:29-66 + CR/Shade.cu:23-113.  Must be called by all 64 lanes.
template <bool STRICT>
DEV d3 diffusion_coop(const RenderParams& P, const DevShade* shade, unsigned long long mask, bool need, const Hit& h, uint32_t& draw,
                      uint32_t pixel, uint32_t smp, Counters& ctr, uint32_t* stk) {
    const DevScene& sc = P.sc;
    const int lane = lane_id();
    d3 color = D3(0, 0, 0), ret = D3(0, 0, 0);
    double m_diff = 0.0, m_spec = 0.0;
    if (need) {
        const DevShade& m = shade[h.prim];
        color = D3(m.color);
        m_diff = m.diff, m_spec = m.spec;
#ifndef RT_ABLATE_TEXTURE  // timing experiments only
        if (m.texture >= 0) color = mulv(color, texture_color(sc, P.flags, m, h, ctr));
#endif
        ctr.shaded++;
    }
    RT_LAP(ctr, PF_SHADE_SETUP);
    uint32_t off = 0;
    for (int l = 0; l < sc.n_lights; l++) {
        const rt_light RT_CONST& L = LIGHTS(sc)[l];
        const bool point = L.type == RT_LIGHT_POINT;
        const int n_l = point ? 1 : sc.shade_n;
        // Which lanes have shadow rays worth tracing for this lamp.  In reference-order mode all of
        // them.  Otherwise not those whose result cannot matter (lamp behind the surface: the shade
        // is only used when N.L > 1e-6, CR/Scene_Material.cu:52) and not those with nothing that
        // could stand between the point and the lamp: every sample is then visible.
        unsigned long long cand = ~0ull;
        bool work = need;
        int cnt = 0;
        double dt = 0.0;  // N.L of CR/Scene_Material.cu:50-51
        if (need) {
            ctr.rays_shadow += (uint32_t)n_l;
            dt = dot(unit(d3{L.O.x, L.O.y, L.O.z} - h.C), h.N);
        }
        if (!STRICT && need) {
            if (!(dt > RT_EPS)) {
                work = false;
            } else {
                unsigned long long full, skip = 0ull;  // primitives the shadow test leaves out for this point and lamp
                if ((P.flags & RT_F_SHADOW_SELF_SKIP) && h.prim < 64) skip |= 1ull << h.prim;
                if (L.light_primitive >= 0 && L.light_primitive < 64) skip |= 1ull << L.light_primitive;
                cand = light_candidates(sc, P.flags, h.C, L, l, skip, full) & ~skip;
                if (cand == 0ull) work = false, cnt = n_l;
                else if ((full & cand) != 0ull) work = false, cnt = 0;  // in the umbra of one primitive: no sample is visible
            }
        }
        RT_LAP(ctr, PF_CAND);
        // slots are dealt to the points that may be shadowed by a mesh first, so that the lanes of a chunk
        // walk the BVH together instead of waiting for the few that do
        const bool heavy = work && (STRICT || (cand & sc.mesh_mask) != 0ull);
        const unsigned long long wmask = __ballot(work), hmask = __ballot(heavy), lmask = wmask & ~hmask;
        const int nwork = __popcll(wmask), nheavy = __popcll(hmask);
        const int my_rank = heavy ? rank_in(hmask) : work ? nheavy + rank_in(lmask) : nwork + rank_in(~wmask);
        const int owner_tbl = __builtin_amdgcn_ds_permute(my_rank << 2, lane);  // lane r learns which lane owns slot r
        const int total = nwork * n_l;
        const float inv_nl = 1.0f / (float)n_l;
        for (int base = 0; base < total; base += 64) {
            const int item = base + lane;
            const bool act = item < total;
#ifdef RT_UTIL_COUNTERS
            ctr.u_shadow_slots++;
            ctr.u_shadow_items += act ? 1u : 0u;
#endif
            // r = item / n_l (item < 2^22): estimate in single precision, then fix the estimate by one
            int r = (int)((float)item * inv_nl);
            if ((r + 1) * n_l <= item) r++;
            if (r * n_l > item) r--;
            if (!act) r = 0;
            const int si = item - r * n_l;
            const int owner = __shfl(owner_tbl, r, 64);
            const d3 C = shfl_d3(h.C, owner);
            const int crashed = __shfl(h.prim, owner, 64);
            const uint32_t opix = __shfl(pixel, owner, 64), osmp = __shfl(smp, owner, 64), odraw = __shfl(draw, owner, 64);
            const unsigned long long ocand = STRICT ? ~0ull : __shfl(cand, owner, 64);
            bool clear = false;
#ifdef RT_UTIL_COUNTERS
            if (act && !STRICT && RT_PROBE_KINDS == 3) {  // which kinds of primitives make this ray necessary
                bool k0 = false, k1 = false, k2 = false;
                for (int j = 0; j < sc.n_prims && j < 64; j++) {
                    if (!((ocand >> j) & 1ull)) continue;
                    const int ty = GEOM(sc)[j].type;
                    if (ty == RT_PRIM_MESH) k2 = true;
                    else if (ty == RT_PRIM_CYLINDER || ty == RT_PRIM_BEZIER) k1 = true;
                    else k0 = true;
                }
                ctr.u[6] += k0, ctr.u[8] += k1, ctr.u[10] += k2;
#ifndef RT_PAIR_STATS
                ctr.u[7] += (k0 && !k1 && !k2), ctr.u[9] += (k1 && !k0 && !k2), ctr.u[11] += (k2 && !k0 && !k1);
#endif
            }
#endif
            RT_LAP(ctr, PF_CHUNK_SHFL);
            if (act) {
                d3 target = d3{L.O.x, L.O.y, L.O.z};
#ifdef RT_ABLATE_RNG
                if (false) {
#else
                if (!point) {
#endif
                    float fa, fb;
                    frand_pair(P.k0, P.k1, opix, osmp, odraw + off + 2u * (uint32_t)si, fa, fb);
                    target = light_sample(L, fa, fb);
                }
                RT_LAP(ctr, PF_RNG);
                clear = shadow_clear<STRICT>(sc, P.flags, C, target, crashed, L.light_primitive, ocand, ctr, stk);
            }
            RT_LAP(ctr, PF_SH_PRIMS);
            const unsigned long long cm = __ballot(clear);
            if (work) {
                int lo = my_rank * n_l - base, hi = lo + n_l;
                if (lo < 0) lo = 0;
                if (hi > 64) hi = 64;
                if (hi > lo) {
                    unsigned long long bits = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
                    cnt += __popcll(cm & bits);
                }
            }
        }
        RT_LAP(ctr, PF_CHUNK_SHFL);
#ifdef RT_PAIR_STATS
        if (heavy) { ctr.u[7] += 1; ctr.u[9] += (cnt == n_l) ? 1u : 0u; ctr.u[11] += (cnt == 0) ? 1u : 0u; }
#endif
        if (need && dt > RT_EPS) {
            double shade = point ? (cnt ? 1.0 : 0.0) : (double)cnt / n_l;
            if (!(shade < RT_EPS)) {
                const d3 Lc = d3{L.color.x, L.color.y, L.color.z};
                if (m_diff > RT_EPS) {
                    double diff = m_diff * dt * shade;
                    ret = ret + mulv(color, Lc) * diff;
                }
                if (m_spec > RT_EPS) {
                    // pow(N.L, 20) of CR/Scene_Material.cu:60 by squaring (a few ulp from libm's, far below one colour level)
                    const double p2 = dt * dt, p4 = p2 * p2, p5 = p4 * dt, p10 = p5 * p5;
                    double spec = m_spec * (p10 * p10) * shade;
                    ret = ret + mulv(color, Lc) * spec;
                }
            }
        }
        if (!point) off += 2u * (uint32_t)n_l;
        RT_LAP(ctr, PF_SHADE_MATH);
    }
    if (need) draw += off;
    return ret;
}

// ---- one path per lane: CR/Scene.cu:23-58 + CR/Scene_Material.cu:68-118 as a loop ----------------
enum { ST_REFL = 1, ST_REFR = 2, ST_PENDING_REFR = 4, ST_FRONT = 8 };

// q = n / d for n < 2^31, with inv a double slightly above 1 / d: the estimate is exact or one too large
DEV uint32_t div_by(uint32_t n, uint32_t d, double inv) {
    uint32_t q = (uint32_t)((double)n * inv);
    if (q * d > n) q--;
    return q;
}

// All paths of a launch.  The wave keeps up to 64 paths in flight, one per lane; lanes whose path has ended wait
// until enough of them are idle, then take the next unstarted paths of the launch together ("path regeneration").
template <bool STRICT>
DEV void trace_stream(const RenderParams& P, const DevShade* shade, Counters& ctr, uint32_t* stk) {
    const DevScene& sc = P.sc;
    // frame k = the hit at depth k+1 that is waiting for a child ray to return
    double f_acc[RT_MAX_DEPTH][3], f_C[RT_MAX_DEPTH][3], f_N[RT_MAX_DEPTH][3], f_V[RT_MAX_DEPTH][3], f_dist[RT_MAX_DEPTH];
    uint32_t f_prim[RT_MAX_DEPTH];  // primitive index | stage bits << 28
    uint32_t top = 0;  // f_prim[sp - 1], kept in a register: the material lookup of a pop need not wait for the stack read
    int sp = 0;
    uint32_t my_path = 0;  // where this lane's colour goes in path_out
    bool tracing = false, drained = false;
    // The paths of a launch form eight contiguous ranges, each with a queue head of its own.  Waves of blocks b, b + 8, ...
    // tend to share an XCD and its L2: they start on one range, so that one L2 sees one part of the image.
    uint32_t range = blockIdx.x & 7u, ranges_done = 0;
    uint32_t pixel = 0, smp = 0, draw = 0, no_log = 0;
    d3 o = D3(0, 0, 0), d = D3(0, 0, 1);
    const uint32_t block_paths = 64u * (uint32_t)P.spp;

    for (;;) {
        // ---- idle lanes start the next paths of the launch (CR/RayTracer.cu:52-72) ----
        const unsigned long long idle = __ballot(!tracing);
        if (!drained && (__popcll(idle) >= RT_REFILL_MIN || idle == ~0ull)) {
            const uint32_t n_new = (uint32_t)__popcll(idle);
            uint32_t base = 0, end = 0;
            for (;;) {  // this wave's range first; the ranges after it once theirs are used up
                const uint32_t first = range * P.range_paths;
                end = first + P.range_paths < P.n_paths ? first + P.range_paths : P.n_paths;
                if (first < end) {
                    if (lane_id() == 0) base = atomicAdd(P.queue + range, n_new);
                    base = first + (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base < end) break;
                }
                range = (range + 1u) & 7u;
                if (++ranges_done == 8) {
                    drained = true;
                    base = end = 0;
                    break;
                }
            }
            const uint32_t g = base + (uint32_t)rank_in(idle);
            if (!tracing && g < end) {
                const uint32_t blk = div_by(g, block_paths, P.inv_block_paths), r = g - blk * block_paths;
                const uint32_t slot = div_by(r, (uint32_t)P.spp, P.inv_spp), s = r - slot * (uint32_t)P.spp;
                const BlockRec rec = P.blocks[blk];
                if ((slot & 7u) < rec.w && (slot >> 3) < rec.h) {
                    const int px = rec.x0 + (int)(slot & 7u), py = rec.y0 + (int)(slot >> 3);
                    int x, y;
                    map_pixel(P, px, py, x, y);
                    const d3 dc = unit(emit(sc, x, y));
                    pixel = (uint32_t)py * (uint32_t)P.tex_w + (uint32_t)px;
                    smp = s;
                    my_path = (blk * (uint32_t)P.spp + s) * 64u + slot;
                    sp = 0, draw = 0, tracing = true;
                    o = sc.camO, d = dc;
                    if (P.spp == 1) {
                        if (!(P.flags & RT_F_NORMALIZE_PRIMARY)) d = emit(sc, x, y);
                    } else {
                        float fu, fv;
                        frand_pair(P.k0, P.k1, pixel, smp, 0u, fu, fv);
                        draw = 2;
                        double u = x + (2.0f * fu - 1.0f);
                        double v = y + (2.0f * fv - 1.0f);
                        d = emit(sc, u, v);
                        if (P.flags & RT_F_NORMALIZE_PRIMARY) d = unit(d);
                    }
                    if (P.dof) {  // modifyRayForDepthOfField, CR/RayTracer.cu:31-38
                        const d3 focal = sc.camO + dc * P.focal;
                        float fa, fb;
                        frand_pair(P.k0, P.k1, pixel, smp, draw, fa, fb);
                        draw += 2;
                        double rx = (fa - 0.5) * P.aperture / 10.0;
                        double ry = (fb - 0.5) * P.aperture / 10.0;
                        o = o + (sc.camDx * rx + sc.camDy * ry);
                        d = unit(focal - o);
                    }
                }
            }
        }
        if (__ballot(tracing) == 0ull) {
            if (drained) break;
            continue;  // every slot just handed out lies outside its tile
        }
        RT_LAP(ctr, PF_TASK);

        Hit h;
        h.prim = -1, h.leaf = -1, h.dist = RT_BIG_DIST, h.front = false, h.N = D3(0, 0, 0), h.C = D3(0, 0, 0);
        bool surface = false;
        d3 result = D3(0, 0, 0);
#ifdef RT_UTIL_COUNTERS
        ctr.u_trace_slots++;
        ctr.u_trace_active += (tracing && sp + 1 <= RT_MAX_DEPTH) ? 1u : 0u;
#endif
        if (tracing && sp + 1 <= RT_MAX_DEPTH) {
            if (sp > 0) ctr.rays_secondary++;
            h = scene_closest<STRICT, false>(sc, P.flags, o, d, ctr, nullptr, 0, no_log, stk);
            if (h.prim < 0) {
                d3 c = unit(d);
                double t = 0.5 * (((P.flags & RT_F_SKY_Y) ? c.y : c.z) + 1.0);
                result = clamp1(sc.bg_bottom * (1.0 - t) + sc.bg_top * t);
            } else if (shade[h.prim].is_light) {
                result = clamp1(D3(shade[h.prim].color));
            } else {
                surface = true;
            }
        }
        RT_LAP(ctr, PF_CLOSEST_PRIMS);
        bool need_shade = false;
        if (surface) {
            const DevShade& m = shade[h.prim];
            need_shade = m.diff > RT_EPS || m.spec > RT_EPS;
#ifdef RT_ABLATE_SHADING  // timing experiments only: paths without any local shading (same rays, wrong colours)
            need_shade = false;
#endif
        }
        d3 local = D3(0, 0, 0);
        const unsigned long long mask = __ballot(need_shade);
        if (mask != 0ull) local = diffusion_coop<STRICT>(P, shade, mask, need_shade, h, draw, pixel, smp, ctr, stk);

        if (tracing) {
            bool have_result = !surface;
            if (surface) {
                const DevShade& m = shade[h.prim];
                d3 acc = D3(0, 0, 0);
                if (need_shade) acc = acc + local;
                const bool has_refl = m.refl > RT_EPS, has_refr = m.refr > RT_EPS;
                if (has_refl || has_refr) {
                    const int k = sp;
                    f_acc[k][0] = acc.x, f_acc[k][1] = acc.y, f_acc[k][2] = acc.z;
                    if (has_refl && has_refr) {  // only a refraction that waits for the reflection subtree needs the hit again
                        f_C[k][0] = h.C.x, f_C[k][1] = h.C.y, f_C[k][2] = h.C.z;
                        f_N[k][0] = h.N.x, f_N[k][1] = h.N.y, f_N[k][2] = h.N.z;
                        f_V[k][0] = d.x, f_V[k][1] = d.y, f_V[k][2] = d.z;
                    }
                    if (has_refr) f_dist[k] = h.dist;  // absorption on the way out
                    uint32_t st = h.front ? ST_FRONT : 0;
                    d3 nd;
                    if (has_refl) {  // CalnReflection, CR/Scene_Material.cu:68-99
                        st |= ST_REFL | (has_refr ? ST_PENDING_REFR : 0);
                        d3 rv = reflect(d, h.N);
                        const int dep = sp + 1;
                        if (m.drefl < RT_EPS || dep > RT_MAX_DREFL_DEP) {
                            nd = rv;
                        } else {
                            d3 bx = vertical(rv);
                            d3 by = unit(cross(rv, bx));
                            bx = bx * m.drefl;
                            by = by * m.drefl;
                            float fx, fy;  // GetBlur, CR/Scene_Material.cu:19-27
                            frand_pair(P.k0, P.k1, pixel, smp, draw, fx, fy);
                            draw += 2;
                            double x = fx, y = fy;
                            double sn, cs;
                            det_sincos(y, sn, cs);
                            double bxk = x * cs, byk = x * sn;
                            nd = rv + bx * bxk + by * byk;
                        }
                    } else {  // CalnRefraction, CR/Scene_Material.cu:101-118
                        st |= ST_REFR;
                        double n = m.rindex;
                        if (h.front) n = 1.0 / n;
                        nd = refract(d, h.N, n);
                    }
                    top = (uint32_t)h.prim | st << 28;
                    f_prim[k] = top;
                    sp++;
                    o = h.C;
                    d = nd;
                } else {
                    result = clamp1(acc);
                    have_result = true;
                }
            }
            RT_LAP(ctr, PF_SPAWN);
            // hand `result` to the frames that wait for it
            while (have_result) {
                if (sp == 0) {  // the path is complete
                    double* dst = P.path_out + (size_t)my_path * 3;
                    dst[0] = result.x, dst[1] = result.y, dst[2] = result.z;
                    tracing = false;
                    break;
                }
                const int k = sp - 1;
                const uint32_t fp = top, st = fp >> 28;
                const DevShade& m = shade[fp & 0x0fffffffu];
                d3 contrib;
                if (st & ST_REFL) {
                    contrib = mulv(result, D3(m.color)) * m.refl;
                } else if (st & ST_FRONT) {
                    contrib = result * m.refr;
                } else {
                    d3 ab = D3(m.absor) * -f_dist[k];
                    d3 tr = D3(rt_exp(ab.x), rt_exp(ab.y), rt_exp(ab.z));
                    contrib = mulv(result, tr) * m.refr;
                }
                d3 acc = D3(f_acc[k][0], f_acc[k][1], f_acc[k][2]) + contrib;
                if ((st & ST_REFL) && (st & ST_PENDING_REFR)) {
                    f_acc[k][0] = acc.x, f_acc[k][1] = acc.y, f_acc[k][2] = acc.z;
                    top = (fp & 0x0fffffffu) | ((st & ST_FRONT) | ST_REFR) << 28;
                    f_prim[k] = top;
                    double n = m.rindex;
                    if (st & ST_FRONT) n = 1.0 / n;
                    d3 V = D3(f_V[k][0], f_V[k][1], f_V[k][2]), N = D3(f_N[k][0], f_N[k][1], f_N[k][2]);
                    o = D3(f_C[k][0], f_C[k][1], f_C[k][2]);
                    d = refract(V, N, n);
                    have_result = false;  // the refraction child is traced next (depth k+2 = sp+1)
                } else {
                    result = clamp1(acc);
                    sp--;
                    if (sp > 0) top = f_prim[sp - 1];
                }
            }
        }
        RT_LAP(ctr, PF_FRAME);
    }
}

// ---- the render kernel: CR/RayTracer.cu:40-94 --------------------------------------------------
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 3  // register budget: 512 / RT_WAVES_PER_SIMD VGPRs per lane; measured best (2: 100 ms, 3: 96 ms, 4: 113 ms per frame)
#endif
// SHADE_CAP > 0: the shading table (at most SHADE_CAP primitives) is copied to LDS and read from there.  A template
// parameter rather than a run-time choice so that the reads compile to LDS instructions, not to flat ones; two
// capacities because a smaller LDS footprint measures 1 % faster on small scenes.
template <bool STRICT, int SHADE_CAP>
__global__ __launch_bounds__(256, RT_WAVES_PER_SIMD) void render_kernel(RenderParams P) {
    const int lane = lane_id();
    // per-lane traversal stacks of the block's four waves: [wave][entry][lane]
    __shared__ uint32_t s_stack[STRICT ? 1 : 4 * RT_STACK * 64];
    uint32_t* stk = STRICT ? nullptr : s_stack + (threadIdx.x >> 6) * (RT_STACK * 64) + lane;
    __shared__ DevShade s_shade[SHADE_CAP > 0 ? SHADE_CAP : 1];
    const DevShade* shade = P.sc.shade;
    if (SHADE_CAP > 0) {  // every hit reads a few fields of its record, one after another
        const int words = P.sc.n_prims * (int)(sizeof(DevShade) / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(P.sc.shade);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_shade);
        for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
        __syncthreads();
        shade = s_shade;
    }
    Counters ctr;
    ctr.rays_secondary = ctr.rays_shadow = ctr.nodes = ctr.tris = ctr.shaded = ctr.texfetch = 0;
#ifdef RT_UTIL_COUNTERS
    ctr.u_shadow_slots = ctr.u_shadow_items = ctr.u_trace_slots = ctr.u_trace_active = 0;
    for (int i = 0; i < 12; i++) ctr.u[i] = 0;
    __shared__ unsigned long long s_prof[4][1 + PF_COUNT];
    ctr.prof = s_prof[threadIdx.x >> 6];
    if (lane < 1 + PF_COUNT) ctr.prof[lane] = lane == 0 ? __builtin_readcyclecounter() : 0ull;
#endif
    trace_stream<STRICT>(P, shade, ctr, stk);
#ifdef RT_UTIL_COUNTERS
    RT_LAP(ctr, PF_TASK);
    if (lane < PF_COUNT) atomicAdd(&P.counters[CTR_ALL + lane], ctr.prof[1 + lane]);
    unsigned long long v[CTR_ALL] = {ctr.rays_secondary, ctr.rays_shadow, ctr.nodes, ctr.tris, ctr.shaded, ctr.texfetch,
                                     ctr.u_shadow_slots, ctr.u_shadow_items, ctr.u_trace_slots, ctr.u_trace_active,
                                     ctr.u[0], ctr.u[1], ctr.u[2], ctr.u[3], ctr.u[4], ctr.u[5], ctr.u[6], ctr.u[7], ctr.u[8], ctr.u[9], ctr.u[10], ctr.u[11]};
    const int n_ctr = CTR_ALL;
#else
    unsigned long long v[CTR_COUNT] = {ctr.rays_secondary, ctr.rays_shadow, ctr.nodes, ctr.tris, ctr.shaded, ctr.texfetch};
    const int n_ctr = CTR_COUNT;
#endif
#pragma unroll
    for (int i = 0; i < n_ctr; i++) {
        unsigned long long s = wave_sum_u64(v[i]);
        if (lane == 0 && s) atomicAdd(&P.counters[i], s);
    }
}

// ---- pixel sums: CR/RayTracer.cu:74-89 ---------------------------------------------------------
// One wave per 8x8 block, lane = slot.  The samples of a pixel are added in sample order, because the reference
// adds them in that order inside one thread and double addition is not associative.
__global__ __launch_bounds__(256) void resolve_kernel(RenderParams P) {
    const uint32_t blk = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t slot = threadIdx.x & 63u;
    if (blk >= P.n_blocks) return;
    const BlockRec rec = P.blocks[blk];
    if ((slot & 7u) >= rec.w || (slot >> 3) >= rec.h) return;
    const double* src = P.path_out + ((size_t)blk * (size_t)P.spp * 64u + slot) * 3;
    d3 acc = D3(0, 0, 0);
    for (int k = 0; k < P.spp; k++, src += 64 * 3) acc = acc + D3(src[0], src[1], src[2]);
    if (P.spp > 1) acc = acc / (double)P.spp;
    const int px = rec.x0 + (int)(slot & 7u), py = rec.y0 + (int)(slot >> 3);
    const uint32_t rgba = to_u8(acc.x) | (to_u8(acc.y) << 8) | (to_u8(acc.z) << 16) | 0xFF000000u;
    *reinterpret_cast<uint32_t*>(P.out + ((size_t)py * (size_t)P.tex_w + (size_t)px) * 4) = rgba;
}

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

// ---- debug kernel: primary ray of every pixel, reference-order walk with a visit log ------------
struct ProbeParams {
    DevScene sc;
    int32_t tex_w, tex_h, x0, y0, w, h;
    uint32_t flags;
    int32_t visit_cap;
    int32_t* hit_prim;
    double* hit_t;
    int32_t* hit_leaf;
    uint32_t* visit_log;
    uint32_t* visit_count;
};

__global__ __launch_bounds__(256) void probe_kernel(ProbeParams Q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q.w * Q.h) return;
    RenderParams P;
    P.sc = Q.sc, P.tex_w = Q.tex_w, P.tex_h = Q.tex_h, P.flags = Q.flags;
    int x, y;
    map_pixel(P, Q.x0 + i % Q.w, Q.y0 + i / Q.w, x, y);
    d3 d = emit(Q.sc, x, y);
    if (Q.flags & RT_F_NORMALIZE_PRIMARY) d = unit(d);
    Counters ctr;
    uint32_t n = 0;
    uint32_t* log = Q.visit_log ? Q.visit_log + (size_t)i * Q.visit_cap : nullptr;
    Hit h = log ? scene_closest<true, true>(Q.sc, Q.flags, Q.sc.camO, d, ctr, log, (uint32_t)Q.visit_cap, n, nullptr)
                : scene_closest<true, false>(Q.sc, Q.flags, Q.sc.camO, d, ctr, nullptr, 0, n, nullptr);
    Q.hit_prim[i] = h.prim;
    Q.hit_t[i] = h.dist;
    Q.hit_leaf[i] = h.leaf;
    if (Q.visit_count) Q.visit_count[i] = n;
}

// ================================= host side ====================================================
static thread_local hipStream_t g_stream = nullptr;

#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return rt_fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct rt_scene {
    int device = 0;
    DevScene dev{};
    void* slab = nullptr;
    size_t slab_bytes = 0;
    uint32_t* queue = nullptr;             // work-queue head
    unsigned long long* counters = nullptr;  // CTR_COUNT work counters
    uint8_t* image = nullptr;              // staging image for host destinations
    size_t image_bytes = 0;
    double* path_out = nullptr;            // colours of the finished paths of one launch
    size_t path_out_bytes = 0;
    BlockRec* blocks = nullptr;            // 8x8 blocks of the tiles of the last call, in queue order
    size_t blocks_cap = 0, n_blocks = 0;
    std::vector<Tile> last_tiles;
    int cus = 0, blocks_per_cu_fast = 0, blocks_per_cu_strict = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int desc_cam_w = 0, desc_cam_h = 0;
    float tree_build_ms = 0.0f;  // device time of the four-wide tree build, 0 if the host built the trees
};

extern "C" int rt_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return rt_fail(RT_ERR_NOGPU, "no HIP device visible");
    if (device < 0 || device >= n) return rt_fail(RT_ERR_INVALID, "rt_init: device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rt_fail(RT_ERR_NOGPU, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    return RT_OK;
}

extern "C" int rt_set_stream(void* hip_stream) {
    g_stream = (hipStream_t)hip_stream;
    return RT_OK;
}

static bool link_ok(const rt_scene_desc* d, uint32_t from_pre, uint32_t ref, bool allow_end) {
    if (ref == RT_REF_END) return allow_end;
    uint32_t pre;
    if (ref & RT_REF_LEAF) {
        uint32_t i = ref & ~RT_REF_LEAF;
        if (i >= d->n_leaves) return false;
        pre = d->leaves[i].preorder;
    } else {
        if (ref >= d->n_nodes) return false;
        pre = d->nodes[ref].preorder;
    }
    return pre > from_pre;  // links only go forward in pre-order, so every walk terminates
}

static int validate_desc(const rt_scene_desc* d) {
    if (!d) return rt_fail(RT_ERR_INVALID, "scene description is null");
    if (d->n_primitives < 0 || d->n_lights < 0 || d->n_textures < 0 || d->n_meshes < 0)
        return rt_fail(RT_ERR_INVALID, "negative count in scene description");
    if ((d->n_primitives && !d->primitives) || (d->n_lights && !d->lights) || (d->n_textures && !d->textures) ||
        (d->n_meshes && !d->meshes) || (d->n_nodes && !d->nodes) || (d->n_leaves && !d->leaves))
        return rt_fail(RT_ERR_INVALID, "null array in scene description");
    if ((int)(16 * d->camera.shade_quality) < 1) return rt_fail(RT_ERR_INVALID, "camera shade_quality must give at least one light sample (16*shade_quality >= 1)");
    if (d->camera.W <= 0 || d->camera.H <= 0) return rt_fail(RT_ERR_INVALID, "camera W/H must be positive");
    for (int i = 0; i < d->n_textures; i++)
        if (d->textures[i].width <= 0 || d->textures[i].height <= 0 || !d->textures[i].rgb) return rt_fail(RT_ERR_INVALID, "bad texture");
    for (int i = 0; i < d->n_primitives; i++) {
        const rt_primitive& p = d->primitives[i];
        if (p.type < RT_PRIM_SPHERE || p.type > RT_PRIM_MESH) return rt_fail(RT_ERR_INVALID, "unknown primitive type");
        if (p.material.texture >= d->n_textures) return rt_fail(RT_ERR_INVALID, "texture index out of range");
        if (p.type == RT_PRIM_MESH && (p.u.mesh.mesh < 0 || p.u.mesh.mesh >= d->n_meshes)) return rt_fail(RT_ERR_INVALID, "mesh index out of range");
        if (p.type == RT_PRIM_BEZIER && (p.u.bezier.degree > RT_BEZIER_MAX_CTRL - 1)) return rt_fail(RT_ERR_LIMIT, "bezier degree too large");
    }
    for (int i = 0; i < d->n_lights; i++) {
        const rt_light& l = d->lights[i];
        if (l.type < RT_LIGHT_POINT || l.type > RT_LIGHT_SPHERE) return rt_fail(RT_ERR_INVALID, "unknown light type");
        if (l.light_primitive >= d->n_primitives) return rt_fail(RT_ERR_INVALID, "light primitive index out of range");
    }
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        const rt_bvh_node& n = d->nodes[i];
        if (!link_ok(d, n.preorder, n.left, false) || !link_ok(d, n.preorder, n.right, false) || !link_ok(d, n.preorder, n.skip, true)) return rt_fail(RT_ERR_INVALID, "BVH node link out of range or not forward");
    }
    for (uint32_t i = 0; i < d->n_leaves; i++)
        if (!link_ok(d, d->leaves[i].preorder, d->leaves[i].skip, true)) return rt_fail(RT_ERR_INVALID, "BVH leaf link out of range or not forward");
    for (int i = 0; i < d->n_meshes; i++) {
        uint32_t r = d->meshes[i].root;
        if (r == RT_REF_END) continue;
        if ((r & RT_REF_LEAF) ? ((r & ~RT_REF_LEAF) >= d->n_leaves) : (r >= d->n_nodes)) return rt_fail(RT_ERR_INVALID, "mesh root out of range");
    }
    return RT_OK;
}


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
    std::vector<TreeJob> jobs;
};
static int g_tree_builder = 0;  // 0 = host (binned surface-area heuristic), 1 = device (Morton order, balanced)
static bool g_tree_builder_set = false;

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
        if (usable && T.nodes.size() < (1u << 30)) T.mesh_root4[mi] = (int32_t)first;
        else T.nodes.resize(first);
    }
}

static void build_shade(const rt_primitive& p, const DevPrimX& x, DevShade& o) {
    using namespace hostmath;
    std::memset(&o, 0, sizeof(o));
    const rt_material& m = p.material;
    o.color[0] = m.color.x, o.color[1] = m.color.y, o.color[2] = m.color.z;
    o.absor[0] = m.absor.x, o.absor[1] = m.absor.y, o.absor[2] = m.absor.z;
    o.diff = m.diff, o.spec = m.spec, o.refl = m.refl, o.refr = m.refr, o.rindex = m.rindex, o.drefl = m.drefl;
    o.type = p.type, o.texture = m.texture, o.is_light = p.is_light;
    o.tex_w = x.tex_w, o.tex_h = x.tex_h, o.tex_off = x.tex_off;
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
    for (int l = 0; l < d->n_lights; l++) {
        const h3 Lc = H(d->lights[l].O);
        const double Rl = lamp_r[l];
        for (int iz = 0; iz < RT_CGRID; iz++)
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
                            const double smax = (len + rho + Rl) * (1.0 + 1e-6), pad = 1e-9 * (1.0 + g.f[MSH_SCALE]);
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
                            if (hdot(e, e) > rr * rr * (1.0 + 1e-9)) may = false;
                            if (may && (g.type == RT_PRIM_CYLINDER || g.type == RT_PRIM_BEZIER) && g.f[CAP_R] < 1e100) {
                                const double reach = g.f[CAP_R] + Rl + rho + 1e-9 * (1.0 + len);
                                const h3 P0{g.f[CAP_P0], g.f[CAP_P0 + 1], g.f[CAP_P0 + 2]}, P1{g.f[CAP_P1], g.f[CAP_P1 + 1], g.f[CAP_P1 + 2]};
                                double ss, tt;
                                const double dd2 = host_seg_seg_dist2(A, Lc, P0, P1, ss, tt);
                                if (dd2 > reach * reach * (1.0 + 1e-6)) may = false;
                                if (may && g.type == RT_PRIM_CYLINDER) {  // umbra of the cylinder for every point of the cell (see light_candidates)
                                    const double Rc = g.f[CAP_R] * (1.0 - 2e-6) - 2e-9, alen = g.f[CYL_LEN];
                                    const double mrg = Rl + rho + 1e-5 * (1.0 + alen + Rc);
                                    const h3 ax{g.f[CYL_DN], g.f[CYL_DN + 1], g.f[CYL_DN + 2]}, a0 = A - P0, a1 = A - P1;
                                    const double ta = std::fmin(std::fmax(hdot(a0, ax), 0.0), alen);
                                    const h3 ea = a0 - ax * ta;
                                    const double far = std::sqrt(std::fmax(hdot(a0, a0), hdot(a1, a1))) + rho;
                                    const double lim = len - rho - Rl - 2e-6 - g.f[CAP_R], deep = Rc - Rl - rho - 1e-6;
                                    if (tt * alen > mrg && (1.0 - tt) * alen > mrg && deep > 0.0 && dd2 < deep * deep * (1.0 - 1e-6) && lim > 0.0 && far < lim * (1.0 - 1e-6) &&
                                        std::sqrt(hdot(ea, ea)) - rho > (g.f[CAP_R] + 2e-6) * (1.0 + 1e-6))
                                        full = true;
                                }
                            }
                            if (may && g.type == RT_PRIM_SPHERE) {  // umbra of the sphere for every point of the cell
                                const double rb = g.f[BND_R], r_lo = rb * (1.0 - 2e-6) - 2e-9, dA = hmodule(w);
                                const double dline = std::sqrt(hdot(e, e)) + Rl + rho;
                                if (r_lo * r_lo - dline * dline > 2e-6 * (1.0 + r_lo * r_lo) && dline < r_lo && dA + rho + rb < len - rho - Rl - 2e-6 && dA - rho - rb > 2e-6)
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

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->slab) (void)hipFree(s->slab);
    if (s->queue) (void)hipFree(s->queue);
    if (s->counters) (void)hipFree(s->counters);
    if (s->image) (void)hipFree(s->image);
    if (s->path_out) (void)hipFree(s->path_out);
    if (s->blocks) (void)hipFree(s->blocks);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
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

extern "C" double rt_scene_tree_build_ms(const rt_scene* s) { return s ? (double)s->tree_build_ms : 0.0; }

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

extern "C" int rt_scene_create(const rt_scene_desc* d, rt_scene** out) {
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_scene_create: null out pointer");
    *out = nullptr;
    int rc = validate_desc(d);
    if (rc) return rc;
    if (d->n_primitives >= (1 << 28)) return rt_fail(RT_ERR_LIMIT, "2^28 primitives or more (the path state keeps the index in 28 bits)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return rt_fail(RT_ERR_NOGPU, "no HIP device visible");
    rt_scene* s = new rt_scene;
    struct Guard {
        rt_scene* s;
        ~Guard() { if (s) rt_scene_destroy(s); }
    } guard{s};
    HIPCHK(hipGetDevice(&s->device));

    // one slab: lights | geometry | Bezier segments | primitives | primitive extras | nodes | four-wide nodes | parents | leaves | texels, one copy
    std::vector<DevPrimX> px((size_t)d->n_primitives);
    std::vector<DevShade> shade((size_t)d->n_primitives);
    std::vector<DevGeom> geom((size_t)d->n_primitives), segs;
    std::vector<uint32_t> tex_off((size_t)d->n_textures);
    Tree4 T;
    build_tree4(d, T, tree_builder_choice() == 1);
    size_t texel_bytes = 0;
    for (int i = 0; i < d->n_textures; i++) {
        tex_off[i] = (uint32_t)texel_bytes;
        texel_bytes += align_up((size_t)d->textures[i].width * d->textures[i].height * 3, 16);
    }
    if (texel_bytes >= (1ull << 32)) return rt_fail(RT_ERR_LIMIT, "more than 4 GiB of textures");
    for (int i = 0; i < d->n_primitives; i++) {
        const rt_primitive& p = d->primitives[i];
        DevPrimX& x = px[i];
        x.tex_w = x.tex_h = 0, x.tex_off = 0, x.pad_ = 0;
        if (p.material.texture >= 0) {
            const rt_texture& t = d->textures[p.material.texture];
            x.tex_w = t.width, x.tex_h = t.height, x.tex_off = tex_off[p.material.texture];
        }
        build_geom(d, p, geom[i], segs, T);
        build_shade(p, x, shade[i]);
    }
    size_t off_l = 0;
    size_t off_g = align_up(off_l + sizeof(rt_light) * d->n_lights, 256);
    size_t off_s = align_up(off_g + sizeof(DevGeom) * d->n_primitives, 256);
    size_t off_p = align_up(off_s + sizeof(DevGeom) * segs.size(), 256);
    size_t off_x = align_up(off_p + sizeof(rt_primitive) * d->n_primitives, 256);
    size_t off_sh = align_up(off_x + sizeof(DevPrimX) * d->n_primitives, 256);
    size_t off_n = align_up(off_sh + sizeof(DevShade) * d->n_primitives, 256);
    size_t off_n2 = align_up(off_n + sizeof(rt_bvh_node) * (size_t)d->n_nodes, 256);
    size_t off_np = align_up(off_n2 + sizeof(DevNode4) * T.nodes.size(), 256);
    size_t off_lp = align_up(off_np + sizeof(uint32_t) * (size_t)d->n_nodes, 256);
    size_t off_f = align_up(off_lp + sizeof(uint32_t) * (size_t)d->n_leaves, 256);
    std::vector<unsigned long long> cgrid;
    double cg_lo[3] = {0, 0, 0}, cg_inv[3] = {0, 0, 0};
    build_cand_grid(d, geom, cgrid, cg_lo, cg_inv);
    size_t off_cg = align_up(off_f + sizeof(rt_bvh_leaf) * (size_t)d->n_leaves, 256);
    size_t off_t = align_up(off_cg + sizeof(unsigned long long) * cgrid.size(), 256);
    s->slab_bytes = align_up(off_t + texel_bytes, 256) + 256;
    std::vector<uint8_t> host(s->slab_bytes, 0);
    if (d->n_lights) std::memcpy(&host[off_l], d->lights, sizeof(rt_light) * d->n_lights);
    if (d->n_primitives) std::memcpy(&host[off_g], geom.data(), sizeof(DevGeom) * d->n_primitives);
    if (!segs.empty()) std::memcpy(&host[off_s], segs.data(), sizeof(DevGeom) * segs.size());
    if (d->n_primitives) std::memcpy(&host[off_p], d->primitives, sizeof(rt_primitive) * d->n_primitives);
    if (d->n_primitives) std::memcpy(&host[off_x], px.data(), sizeof(DevPrimX) * d->n_primitives);
    if (d->n_primitives) std::memcpy(&host[off_sh], shade.data(), sizeof(DevShade) * d->n_primitives);
    if (d->n_nodes) std::memcpy(&host[off_n], d->nodes, sizeof(rt_bvh_node) * (size_t)d->n_nodes);
    if (!T.nodes.empty()) std::memcpy(&host[off_n2], T.nodes.data(), sizeof(DevNode4) * T.nodes.size());
    if (d->n_nodes) std::memcpy(&host[off_np], T.node_parent.data(), sizeof(uint32_t) * (size_t)d->n_nodes);
    if (d->n_leaves) std::memcpy(&host[off_lp], T.leaf_parent.data(), sizeof(uint32_t) * (size_t)d->n_leaves);
    if (d->n_leaves) std::memcpy(&host[off_f], d->leaves, sizeof(rt_bvh_leaf) * (size_t)d->n_leaves);
    if (!cgrid.empty()) std::memcpy(&host[off_cg], cgrid.data(), sizeof(unsigned long long) * cgrid.size());
    for (int i = 0; i < d->n_textures; i++)
        std::memcpy(&host[off_t + tex_off[i]], d->textures[i].rgb, (size_t)d->textures[i].width * d->textures[i].height * 3);
    HIPCHK(hipMalloc(&s->slab, s->slab_bytes));
    HIPCHK(hipMemcpy(s->slab, host.data(), s->slab_bytes, hipMemcpyHostToDevice));
    uint8_t* base = (uint8_t*)s->slab;
    rc = run_tree_jobs(T, (DevNode4*)(base + off_n2), (const rt_bvh_leaf*)(base + off_f), &s->tree_build_ms);
    if (rc) return rc;
    DevScene& v = s->dev;
    v.bg_top = d3{d->background_top.x, d->background_top.y, d->background_top.z};
    v.bg_bottom = d3{d->background_bottom.x, d->background_bottom.y, d->background_bottom.z};
    v.camO = d3{d->camera.O.x, d->camera.O.y, d->camera.O.z};
    v.camN = d3{d->camera.N.x, d->camera.N.y, d->camera.N.z};
    v.camDx = d3{d->camera.Dx.x, d->camera.Dx.y, d->camera.Dx.z};
    v.camDy = d3{d->camera.Dy.x, d->camera.Dy.y, d->camera.Dy.z};
    v.cam_w = d->camera.W, v.cam_h = d->camera.H;
    s->desc_cam_w = d->camera.W, s->desc_cam_h = d->camera.H;
    v.shade_n = (int)(16 * d->camera.shade_quality);
    v.n_lights = d->n_lights, v.n_prims = d->n_primitives;
    v.mesh_mask = 0;
    for (int i = 0; i < d->n_primitives && i < 64; i++)
        if (d->primitives[i].type == RT_PRIM_MESH) v.mesh_mask |= 1ull << i;
    v.lights = (const rt_light*)(base + off_l);
    v.geom = (const DevGeom*)(base + off_g);
    v.segs = (const DevGeom*)(base + off_s);
    v.prims = (const rt_primitive*)(base + off_p);
    v.primx = (const DevPrimX*)(base + off_x);
    v.shade = (const DevShade*)(base + off_sh);
    v.nodes = (const rt_bvh_node*)(base + off_n);
    v.nodes4 = (const DevNode4*)(base + off_n2);
    v.node_parent = (const uint32_t*)(base + off_np);
    v.leaf_parent = (const uint32_t*)(base + off_lp);
    v.leaves = (const rt_bvh_leaf*)(base + off_f);
    v.texels = base + off_t;
    v.cand_grid = cgrid.empty() ? nullptr : (const unsigned long long*)(base + off_cg);
    for (int k = 0; k < 3; k++) v.cg_lo[k] = cg_lo[k], v.cg_inv[k] = cg_inv[k];

    HIPCHK(hipMalloc(&s->queue, 256));
    HIPCHK(hipMalloc(&s->counters, sizeof(unsigned long long) * (CTR_ALL + 16)));
    HIPCHK(hipEventCreate(&s->ev0));
    HIPCHK(hipEventCreate(&s->ev1));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, s->device));
    s->cus = prop.multiProcessorCount;
    // the kernels are built for RT_WAVES_PER_SIMD resident waves per SIMD = that many 256-thread blocks per CU
    s->blocks_per_cu_fast = s->blocks_per_cu_strict = RT_WAVES_PER_SIMD;
    guard.s = nullptr;
    *out = s;
    return RT_OK;
}

static bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

static int check_common(rt_scene* s, const rt_settings* st, int tex_w, int tex_h) {
    if (!s || !st) return rt_fail(RT_ERR_INVALID, "null scene or settings");
    if (tex_w <= 0 || tex_h <= 0 || tex_w > 32768 || tex_h > 32768) return rt_fail(RT_ERR_INVALID, "image size out of range");
    if (st->spp < 1 || st->spp > 65536) return rt_fail(RT_ERR_INVALID, "samples per pixel must be in [1, 65536]");
    return RT_OK;
}

// Render the given tiles into `dst` (device image).  Fills kernel time and counters.
static int launch_tiles(rt_scene* s, const rt_settings* st, int tex_w, int tex_h, int cam_w, int cam_h, const std::vector<Tile>& tiles,
                        uint8_t* dst, rt_stats* stats) {
    // The 8x8 blocks of the tiles in Z-order: blocks that follow each other in the queue are neighbours in the image,
    // and each of the eight ranges of the queue is a compact region (its rays meet the same part of the scene in L2).
    // The list is kept from call to call: a renderer asks for the same tiles frame after frame.
    uint64_t pixels = 0;
    for (const Tile& t : tiles) pixels += (uint64_t)t.w * t.h;
    const bool same_tiles = s->last_tiles.size() == tiles.size() &&
                            (tiles.empty() || std::memcmp(s->last_tiles.data(), tiles.data(), tiles.size() * sizeof(Tile)) == 0);
    if (!same_tiles) {
        auto spread = [](uint32_t v) {
            v &= 0xFFFFu;
            v = (v | (v << 8)) & 0x00FF00FFu, v = (v | (v << 4)) & 0x0F0F0F0Fu, v = (v | (v << 2)) & 0x33333333u, v = (v | (v << 1)) & 0x55555555u;
            return v;
        };
        std::vector<std::pair<uint32_t, BlockRec>> keyed;
        for (const Tile& t : tiles) {
            for (int by = 0; by < t.h; by += 8)
                for (int bx = 0; bx < t.w; bx += 8) {
                    BlockRec r;
                    r.x0 = (uint16_t)(t.x0 + bx), r.y0 = (uint16_t)(t.y0 + by);
                    r.w = (uint8_t)std::min(8, t.w - bx), r.h = (uint8_t)std::min(8, t.h - by), r.pad_ = 0;
                    keyed.push_back({spread((uint32_t)r.x0 >> 3) | (spread((uint32_t)r.y0 >> 3) << 1), r});
                }
            if (keyed.size() >= (1ull << 24)) return rt_fail(RT_ERR_LIMIT, "too many pixels in one call");
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        std::vector<BlockRec> recs(keyed.size());
        for (size_t i = 0; i < keyed.size(); i++) recs[i] = keyed[i].second;
        if (recs.size() > s->blocks_cap) {
            if (s->blocks) (void)hipFree(s->blocks);
            s->blocks = nullptr, s->blocks_cap = 0;
            const size_t cap = recs.size() + 1024;
            HIPCHK(hipMalloc(&s->blocks, cap * sizeof(BlockRec)));
            s->blocks_cap = cap;
        }
        s->last_tiles.clear();  // stays empty if the copy fails
        if (!recs.empty()) HIPCHK(hipMemcpy(s->blocks, recs.data(), recs.size() * sizeof(BlockRec), hipMemcpyHostToDevice));
        s->last_tiles = tiles;
        s->n_blocks = recs.size();
    }
    const uint64_t blocks = s->n_blocks;
    const bool strict = (st->flags & RT_F_STRICT_TRAVERSAL) != 0;
    const int bpc = strict ? s->blocks_per_cu_strict : s->blocks_per_cu_fast;
    // blocks per launch: the colours of all paths of a launch are kept until its pixel sums are taken
    const uint64_t block_paths = 64ull * (uint64_t)st->spp;
    uint64_t launch_paths = RT_LAUNCH_PATHS;
    if (const char* e = std::getenv("RT_AMD_LAUNCH_PATHS")) {  // smaller launches on request (tests of the multi-launch path, small-memory runs)
        const unsigned long long v = std::strtoull(e, nullptr, 10);
        if (v >= 64 && v < launch_paths) launch_paths = v;
    }
    const uint64_t chunk_blocks = std::max<uint64_t>(1, launch_paths / block_paths);
    const size_t out_bytes = (size_t)(std::min(blocks, chunk_blocks) * block_paths) * 3 * sizeof(double);
    if (s->path_out_bytes < out_bytes) {
        if (s->path_out) (void)hipFree(s->path_out);
        s->path_out = nullptr, s->path_out_bytes = 0;
        HIPCHK(hipMalloc(&s->path_out, out_bytes));
        s->path_out_bytes = out_bytes;
    }
    HIPCHK(hipMemsetAsync(s->counters, 0, sizeof(unsigned long long) * (CTR_ALL + 16), g_stream));

    RenderParams P;
    P.sc = s->dev;
    P.sc.cam_w = cam_w > 0 ? cam_w : s->desc_cam_w;
    P.sc.cam_h = cam_h > 0 ? cam_h : s->desc_cam_h;
    P.tex_w = tex_w, P.tex_h = tex_h, P.spp = st->spp, P.dof = st->dof_enabled ? 1 : 0;
    P.focal = st->focal_distance, P.aperture = st->aperture;
    P.inv_block_paths = 1.0 / (double)block_paths * (1.0 + 1e-12), P.inv_spp = 1.0 / (double)st->spp * (1.0 + 1e-12);
    P.k0 = (uint32_t)st->seed, P.k1 = (uint32_t)(st->seed >> 32), P.flags = st->flags;
    P.queue = s->queue, P.out = dst, P.counters = s->counters, P.path_out = s->path_out;

    HIPCHK(hipEventRecord(s->ev0, g_stream));
    for (uint64_t b0 = 0; b0 < blocks; b0 += chunk_blocks) {
        const uint64_t nb = std::min(chunk_blocks, blocks - b0);
        P.blocks = s->blocks + b0, P.n_blocks = (uint32_t)nb, P.n_paths = (uint32_t)(nb * block_paths);
        P.range_paths = (uint32_t)(((nb + 7) / 8) * block_paths);
        HIPCHK(hipMemsetAsync(s->queue, 0, 256, g_stream));
        // persistent grid: every wave draws paths until the launch runs dry
        uint64_t grid = (uint64_t)s->cus * bpc;
        const uint64_t want_blocks = ((uint64_t)P.n_paths + 255) / 256;
        if (grid > want_blocks) grid = want_blocks;
        if (strict) hipLaunchKernelGGL((render_kernel<true, 0>), dim3((unsigned)grid), dim3(256), 0, g_stream, P);
        else if (P.sc.n_prims <= 16) hipLaunchKernelGGL((render_kernel<false, 16>), dim3((unsigned)grid), dim3(256), 0, g_stream, P);
        else if (P.sc.n_prims <= RT_SHADE_LDS) hipLaunchKernelGGL((render_kernel<false, RT_SHADE_LDS>), dim3((unsigned)grid), dim3(256), 0, g_stream, P);
        else hipLaunchKernelGGL((render_kernel<false, 0>), dim3((unsigned)grid), dim3(256), 0, g_stream, P);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, g_stream, P);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(s->ev1, g_stream));
    unsigned long long c[CTR_ALL + 16];
    HIPCHK(hipMemcpyAsync(c, s->counters, sizeof(c), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
#ifdef RT_UTIL_COUNTERS
    std::fprintf(stderr, "[util] shadow lanes used %.1f%% (%llu of %llu slots), closest-hit lanes used %.1f%% (%llu of %llu slots)\n",
                 100.0 * c[CTR_U_SHADOW_ITEMS] / (double)(c[CTR_U_SHADOW_SLOTS] ? c[CTR_U_SHADOW_SLOTS] : 1), c[CTR_U_SHADOW_ITEMS], c[CTR_U_SHADOW_SLOTS],
                 100.0 * c[CTR_U_TRACE_ACTIVE] / (double)(c[CTR_U_TRACE_SLOTS] ? c[CTR_U_TRACE_SLOTS] : 1), c[CTR_U_TRACE_ACTIVE], c[CTR_U_TRACE_SLOTS]);
    static const char* const probe_name[6] = {"shadow mesh walks", "closest mesh walks", "shadow interior steps", "shadow leaf steps",
                                              "closest interior steps", "closest leaf steps"};
    static const char* const sect[PF_COUNT] = {"task setup / resolve", "frame pop / hand-back", "closest: analytic prims", "closest: mesh walk", "shade setup + texture",
                                               "light candidates", "chunk shuffles + counts", "rng + light sample", "shadow: analytic prims", "shadow: mesh walk",
                                               "shade math", "child ray set-up"};
    unsigned long long tot = 0;
    for (int k = 0; k < PF_COUNT; k++) tot += c[CTR_ALL + k];
    for (int k = 0; k < PF_COUNT; k++)
        std::fprintf(stderr, "[time] %-26s %5.1f%%\n", sect[k], 100.0 * c[CTR_ALL + k] / (double)(tot ? tot : 1));
    std::fprintf(stderr, "[util] shadow rays traced because of: sphere/plane/square %llu (only: %llu), cylinder/bezier %llu (only: %llu), mesh %llu (only: %llu)\n",
                 c[CTR_U_PROBES + 6], c[CTR_U_PROBES + 7], c[CTR_U_PROBES + 8], c[CTR_U_PROBES + 9], c[CTR_U_PROBES + 10], c[CTR_U_PROBES + 11]);
    for (int k = 0; k < RT_PROBE_KINDS; k++) {
        const unsigned long long lanes = c[CTR_U_PROBES + 2 * k], slots = c[CTR_U_PROBES + 2 * k + 1];
        std::fprintf(stderr, "[util] %-24s lanes %.1f%% (%llu lane executions, %llu wave executions)\n", probe_name[k],
                     100.0 * lanes / (double)(slots ? slots : 1), lanes, slots / 64);
    }
#endif
    if (stats) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        stats->kernel_ms = ms;
        stats->pixels = pixels;
        stats->rays_primary = pixels * (uint64_t)st->spp;
        stats->rays_secondary = c[CTR_SECONDARY], stats->rays_shadow = c[CTR_SHADOW];
        stats->nodes_visited = c[CTR_NODES], stats->tris_tested = c[CTR_TRIS];
        stats->shaded_hits = c[CTR_SHADED], stats->texture_fetches = c[CTR_TEXFETCH];
    }
    return RT_OK;
}

static int ensure_image(rt_scene* s, size_t bytes) {
    if (s->image_bytes >= bytes) return RT_OK;
    if (s->image) (void)hipFree(s->image);
    s->image = nullptr, s->image_bytes = 0;
    HIPCHK(hipMalloc(&s->image, bytes));
    s->image_bytes = bytes;
    return RT_OK;
}

static int render_common(rt_scene* s, const rt_settings* st, int tex_w, int tex_h, int cam_w, int cam_h, const std::vector<Tile>& tiles,
                         uint8_t* rgba, rt_stats* stats) {
    auto t0 = std::chrono::steady_clock::now();
    if (!rgba) return rt_fail(RT_ERR_INVALID, "null output image");
    HIPCHK(hipSetDevice(s->device));
    const size_t bytes = (size_t)tex_w * tex_h * 4;
    int rc;
    if (is_device_ptr(rgba)) {
        rc = launch_tiles(s, st, tex_w, tex_h, cam_w, cam_h, tiles, rgba, stats);
        if (rc) return rc;
    } else {
        rc = ensure_image(s, bytes);
        if (rc) return rc;
        rc = launch_tiles(s, st, tex_w, tex_h, cam_w, cam_h, tiles, s->image, stats);
        if (rc) return rc;
        // copy back only the rows of the rendered rectangles; other pixels of the caller's image stay
        for (const Tile& t : tiles) {
            const size_t o = ((size_t)t.y0 * tex_w + t.x0) * 4;
            HIPCHK(hipMemcpy2DAsync(rgba + o, (size_t)tex_w * 4, s->image + o, (size_t)tex_w * 4, (size_t)t.w * 4, (size_t)t.h,
                                    hipMemcpyDeviceToHost, g_stream));
        }
        HIPCHK(hipStreamSynchronize(g_stream));
    }
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

extern "C" int rt_render_tile(rt_scene* s, const rt_settings* st, int tex_w, int tex_h, int cam_w, int cam_h, int x0, int y0, int w,
                              int h, uint8_t* rgba, rt_stats* stats) {
    int rc = check_common(s, st, tex_w, tex_h);
    if (rc) return rc;
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > tex_w || y0 + h > tex_h)
        return rt_fail(RT_ERR_INVALID, "rectangle does not lie inside the image");
    std::vector<Tile> tiles{Tile{x0, y0, w, h}};
    return render_common(s, st, tex_w, tex_h, cam_w, cam_h, tiles, rgba, stats);
}

extern "C" int rt_render_frame(rt_scene* s, const rt_settings* st, int tex_w, int tex_h, int cam_w, int cam_h, int tile, int part,
                               int n_parts, uint8_t* rgba, rt_stats* stats) {
    int rc = check_common(s, st, tex_w, tex_h);
    if (rc) return rc;
    if (tile <= 0 || (tile & 7) || n_parts <= 0 || part < 0 || part >= n_parts)
        return rt_fail(RT_ERR_INVALID, "tile must be a positive multiple of 8 and 0 <= part < n_parts");
    std::vector<Tile> tiles;
    const int tx = (tex_w + tile - 1) / tile, ty = (tex_h + tile - 1) / tile;
    for (int j = 0; j < ty; j++)
        for (int i = 0; i < tx; i++) {
            if ((j * tx + i) % n_parts != part) continue;
            Tile t{i * tile, j * tile, std::min(tile, tex_w - i * tile), std::min(tile, tex_h - j * tile)};
            tiles.push_back(t);
        }
    if (n_parts == 1) tiles.assign(1, Tile{0, 0, tex_w, tex_h});  // one rectangle: no partial 8x8 blocks at tile seams
    if (tiles.empty()) {
        if (stats) std::memset(stats, 0, sizeof(*stats));
        return RT_OK;
    }
    return render_common(s, st, tex_w, tex_h, cam_w, cam_h, tiles, rgba, stats);
}

extern "C" int rt_trace_primary(rt_scene* s, const rt_settings* st, int tex_w, int tex_h, int cam_w, int cam_h, int x0, int y0, int w,
                                int h, int32_t* hit_prim, double* hit_t, int32_t* hit_leaf, uint32_t* visit_log, uint32_t* visit_count,
                                int visit_cap) {
    int rc = check_common(s, st, tex_w, tex_h);
    if (rc) return rc;
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > tex_w || y0 + h > tex_h)
        return rt_fail(RT_ERR_INVALID, "rectangle does not lie inside the image");
    if (!hit_prim || !hit_t || !hit_leaf) return rt_fail(RT_ERR_INVALID, "null output array");
    if (visit_log && visit_cap <= 0) return rt_fail(RT_ERR_INVALID, "visit_cap must be positive when a visit log is requested");
    HIPCHK(hipSetDevice(s->device));
    const size_t n = (size_t)w * h;
    const size_t b_prim = align_up(n * 4, 256), b_t = align_up(n * 8, 256), b_leaf = align_up(n * 4, 256), b_cnt = align_up(n * 4, 256);
    const size_t b_log = visit_log ? align_up(n * (size_t)visit_cap * 4, 256) : 0;
    uint8_t* buf = nullptr;
    HIPCHK(hipMalloc(&buf, b_prim + b_t + b_leaf + b_cnt + b_log));
    ProbeParams Q;
    Q.sc = s->dev;
    Q.sc.cam_w = cam_w > 0 ? cam_w : s->desc_cam_w;
    Q.sc.cam_h = cam_h > 0 ? cam_h : s->desc_cam_h;
    Q.tex_w = tex_w, Q.tex_h = tex_h, Q.x0 = x0, Q.y0 = y0, Q.w = w, Q.h = h, Q.flags = st->flags, Q.visit_cap = visit_cap;
    Q.hit_t = (double*)buf;
    Q.hit_prim = (int32_t*)(buf + b_t);
    Q.hit_leaf = (int32_t*)(buf + b_t + b_prim);
    Q.visit_count = (uint32_t*)(buf + b_t + b_prim + b_leaf);
    Q.visit_log = visit_log ? (uint32_t*)(buf + b_t + b_prim + b_leaf + b_cnt) : nullptr;
    hipLaunchKernelGGL(probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, Q);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(hit_t, Q.hit_t, n * 8, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hit_prim, Q.hit_prim, n * 4, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hit_leaf, Q.hit_leaf, n * 4, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess && visit_count) e = hipMemcpyAsync(visit_count, Q.visit_count, n * 4, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess && visit_log) e = hipMemcpyAsync(visit_log, Q.visit_log, n * (size_t)visit_cap * 4, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    (void)hipFree(buf);
    if (e != hipSuccess) return rt_fail(RT_ERR_DEVICE, std::string("rt_trace_primary: ") + hipGetErrorString(e));
    return RT_OK;
}
