// Render kernels for gfx950 and the device half of the C ABI (include/rt_amd.h).
//
// Execution model (wave64, no MFMA - this is a pointer walk, not a contraction):
//   * a persistent grid: every wave keeps 64 paths in flight, one per lane, and draws the next unstarted
//     paths of the launch from a queue (eight ranges of Z-ordered 8x8 pixel blocks) until it runs dry;
//   * the colour of a finished path goes to a per-path slot in HBM and resolve_kernel adds the samples of
//     a pixel in sample order (the reference adds them in that order inside one thread,
//     CR/RayTracer.cu:74-84, and double addition is not associative);
//   * the recursion of CR/Scene.cu:23-58 becomes a loop over an explicit frame stack (<= 10
//     frames per lane, in scratch) that visits reflect / refract children in the reference's
//     order, so each path consumes its random draws in the same sequence;
//   * shading is wave-cooperative: the shadow rays of all lanes that need shading (one per
//     point light, 16*shade_quality per area light, CR/Shade.cu:23-113) are dealt out over all
//     64 lanes, 64 at a time.  Rays of one shaded point fly to the same small lamp, so
//     neighbouring lanes walk the same BVH nodes.  Visible-sample counts are integers, so the
//     sum does not depend on which lane traced which ray;
//   * LDS holds the per-lane traversal stacks of the four-wide tree walk and the shading table.
// Scene preparation on the host (geometry records, four-wide trees, candidate grid) is in
// rt_scene_build.hpp, the device tree builder in rt_tree_device.hpp; both are part of this translation unit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"
#include "rt_internal.h"
#include "rt_trace.hpp"

static_assert(sizeof(rt_bvh_node) == 64 && sizeof(rt_bvh_leaf) == 96, "BVH record sizes are part of the ABI");
static_assert(RT_SPEC_POWER == 20, "the specular power is expanded by hand in diffusion_coop");
static_assert(sizeof(DevShade) == 216 && sizeof(DevShade) % 4 == 0, "shading record");
static_assert(sizeof(DevNode4) == 128 && sizeof(DevGeom) == 192, "device record sizes");
static_assert(sizeof(rt_primitive) == 408 && sizeof(rt_light) == 112 && sizeof(rt_scene_desc) == 256 && sizeof(rt_settings) == 40 &&
                  sizeof(rt_stats) == 88,
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
#define RPREF const RenderParams RT_CONST&
// the kernel's first argument, read in place (the by-value parameter of the same type is what the launch fills in)
#define KERNARG(T) (*(const T RT_CONST*)(unsigned long long)__builtin_amdgcn_kernarg_segment_ptr())
#ifndef RT_LAUNCH_PATHS
#define RT_LAUNCH_PATHS (1ull << 28)  // most paths of one launch (their colours take 24 bytes each: 6.4 GB); larger calls are cut into several launches
#endif
#ifndef RT_SHARE_CLOSEST
#define RT_SHARE_CLOSEST 0  // 1: the shared variant also shares its closest-hit walks (mesh_closest_shared).  Halves the steps of those
                           // walks, but what it costs in registers around them is as much: c4_lucy 43.7 -> 44.8, c5_final 201 -> 197 ms/frame
#endif
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 32  // idle lanes that trigger a refill: the set-up of new paths is long, and running it for one lane costs the same
                          // (measured ms/frame: 8: 52.0, 16: 49.1, 24: 47.7, 32: 47.1, 40: 47.3, 48: 47.8)
#endif

enum { CTR_SECONDARY = 0, CTR_SHADOW, CTR_NODES, CTR_TRIS, CTR_SHADED, CTR_TEXFETCH, CTR_TRACED, CTR_COUNT,
       // lane-utilisation probes, only counted in -DRT_UTIL_COUNTERS builds (printed by rt_render_* to stderr)
       CTR_U_SHADOW_SLOTS = CTR_COUNT, CTR_U_SHADOW_ITEMS, CTR_U_TRACE_SLOTS, CTR_U_TRACE_ACTIVE, CTR_U_PROBES, CTR_ALL = CTR_U_PROBES + 12 };

// ---- camera: CR/RayTracer.cu:52-61, CR/camera.cpp:57-59 -------------------------------------------
DEV d3 emit(SCREF sc, double x, double y) {
    return D3(sc.camN) + D3(sc.camDy) * (2 * y / sc.cam_h - 1) + D3(sc.camDx) * (2 * x / sc.cam_w - 1);
}
DEV void map_pixel(uint32_t flags, int tex_w, int tex_h, int cam_w, int cam_h, int sx, int sy, int& x, int& y) {
    if (flags & RT_F_PIXEL_MAP_TRUNC) {
        x = (int)((double)sx / tex_w * cam_w);
        y = (int)((double)sy / tex_h * cam_h);
    } else {
        x = (int)(((long long)sx * cam_w) / tex_w);
        y = (int)(((long long)sy * cam_h) / tex_h);
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

// `pp` = the address of the launch parameters again, as a value the compiler knows nothing about: loads through it cannot be
// moved above this point, so what a stage needs of the parameters is loaded (scalar loads, cached) when the stage begins
#ifndef RT_X_OUTCOME
#define RT_X_OUTCOME 0
#endif
#define RT_FRESH()                   \
    do {                             \
        pp = &P_in;                  \
        asm volatile("" : "+s"(pp)); \
    } while (0)

// ---- CalnDiffusion for every lane that needs it, shadow rays dealt over the wave ---------------
// CR/Scene_Material.cu:29-66 + CR/Shade.cu:23-113.  Must be called by all 64 lanes.
template <bool STRICT, bool SHARE, bool CONE, bool MESH4>
DEV d3 diffusion_coop(RPREF P_in, const DevShade* shade_tbl, unsigned long long mask, bool need, const Hit& h, uint32_t& draw,
                      uint32_t pixel, uint32_t smp, Counters& ctr, uint32_t* stk, uint32_t* xch) {
    const RenderParams RT_CONST* pp;
    RT_FRESH();
#define P (*pp)
#define sc (pp->sc)
#define L (LIGHTS(sc)[l])  // the lamp's record, read where it lies
    const int lane = lane_id();
    d3 color = D3(0, 0, 0), ret = D3(0, 0, 0);
    if (need) {
        const DevShade& m = shade_tbl[h.prim];
        color = D3(m.color);
        if (m.texture >= 0) color = mulv(color, texture_color(sc, P.flags, m, h, ctr));
        ctr.shaded++;
    }
    RT_LAP(ctr, PF_SHADE_SETUP);
    uint32_t off = 0;
    for (int l = 0; l < sc.n_lights; l++) {
        RT_FRESH();
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
        if (!STRICT) {
            // (every lane goes through here, the ones with nothing to ask too: the unions below are taken over the whole wave)
            const bool ask = need && dt > RT_EPS;
            if (need && !ask) work = false;
            unsigned long long full = 0ull, skip = 0ull, cell = 0ull;  // skip: the primitives the shadow test leaves out for this point and lamp
            if (ask) {
                if ((P.flags & RT_F_SHADOW_SELF_SKIP) && h.prim < 64) skip |= 1ull << h.prim;
                if (L.light_primitive >= 0 && L.light_primitive < 64) skip |= 1ull << L.light_primitive;
                cell = candidate_cell(sc, h.C, l, skip, full);
                if (full != 0ull) work = false, cnt = 0, cell = 0ull;  // the whole cell lies in the umbra of one primitive: no sample is visible
            }
            const unsigned long long wave_cells = wave_or_u64(cell);
            if (ask && full == 0ull) {
                cand = light_candidates(sc, P.flags, h.C, L, skip, cell, wave_cells, full) & ~skip;
                if (cand == 0ull) work = false, cnt = n_l;
                else if ((full & cand) != 0ull) work = false, cnt = 0;  // in the umbra of one primitive: no sample is visible
            }
            // A mesh's root box is mostly empty: before a point's samples go to the tracer because of a mesh, look at the boxes
            // further down its tree (mesh_cone_may).  Not for point lamps: one ray each, the look would cost as much as the ray.
            if (CONE && stk != nullptr && !point && sc.n_prims <= 64) {
                const unsigned long long mine = work ? cand & sc.mesh_mask : 0ull;
                const unsigned long long meshes = wave_or_u64(mine);
                if (meshes != 0ull) {
                    RT_FRESH();
                    const d3 Dl = d3{L.O.x, L.O.y, L.O.z} - h.C;
                    const double Rl = lamp_reach(L);
                    const double smax = (P.flags & RT_F_MESH_RAW_DIR) ? (module(Dl) + Rl) * (1.0 + 1e-6) : 1.0;  // as in light_candidates
                    for (unsigned long long todo = meshes; todo != 0ull; todo &= todo - 1ull) {
                        const int j = __builtin_ctzll(todo);
                        const DevGeom RT_CONST& g = GEOM(sc)[j];
                        if (((mine >> j) & 1ull) && g.aux2 && !mesh_cone_may(sc, g, h.C, Dl, Rl, smax, stk)) cand &= ~(1ull << j);
                    }
                    if (work && cand == 0ull) work = false, cnt = n_l;
                }
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
        ctr.traced_wave += (uint32_t)total;  // the shadow rays of this lamp that are intersected with the scene (same value in every lane)
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
            // (the pruned tests read the owner's candidate mask, which already leaves out the shaded primitive and the lamp's own)
            const int crashed = (STRICT || sc.n_prims > 64) ? __shfl(h.prim, owner, 64) : -1;
            const uint32_t opix = __shfl(pixel, owner, 64), osmp = __shfl(smp, owner, 64), odraw = __shfl(draw, owner, 64);
            const unsigned long long ocand = STRICT ? ~0ull : __shfl(cand, owner, 64);
            bool clear = false;
#ifdef RT_UTIL_COUNTERS
            if (act && !STRICT && RT_PROBE_KINDS == 3 && !RT_X_OUTCOME) {  // which kinds of primitives make this ray necessary
                bool k0 = false, k1 = false, k2 = false;
                for (int j = 0; j < sc.n_prims && j < 64; j++) {
                    if (!((ocand >> j) & 1ull)) continue;
                    const int ty = GEOM(sc)[j].type;
                    if (ty == RT_PRIM_MESH) k2 = true;
                    else if (ty == RT_PRIM_CYLINDER || ty == RT_PRIM_BEZIER) k1 = true;
                    else k0 = true;
                }
                ctr.u[6] += k0, ctr.u[8] += k1, ctr.u[10] += k2;
                ctr.u[7] += (k0 && !k1 && !k2), ctr.u[9] += (k1 && !k0 && !k2), ctr.u[11] += (k2 && !k0 && !k1);
            }
#endif
            RT_LAP(ctr, PF_CHUNK_SHFL);
#ifndef RT_X_NOFRESH226
            RT_FRESH();
#endif
            d3 target = d3{L.O.x, L.O.y, L.O.z};
            if (act) {
                if (!point) {
                    float fa, fb;
                    frand_pair(P.k0, P.k1, opix, osmp, odraw + off + 2u * (uint32_t)si, fa, fb);
                    target = light_sample(L, fa, fb);
                }
            }
            RT_LAP(ctr, PF_RNG);
            // the primitives some ray of this chunk has to be tested against: the tests below loop over these only
            const unsigned long long chunk_prims = STRICT ? ~0ull : wave_or_u64(act ? ocand : 0ull);
            if (SHARE && stk != nullptr) {
                clear = shadow_clear_shared(sc, P.flags, act, C, target, ocand, chunk_prims, ctr, stk, xch);
            } else if (act) {
                clear = shadow_clear<STRICT, MESH4>(sc, P.flags, C, target, crashed, L.light_primitive, ocand, chunk_prims, ctr, stk);
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
#if defined(RT_UTIL_COUNTERS) && RT_X_OUTCOME
        if (heavy && !STRICT) ctr.u[6]++, ctr.u[7] += (cnt == n_l), ctr.u[8] += (cnt == 0), ctr.u[9] += (uint32_t)cnt, ctr.u[10] += (uint32_t)n_l;
        if (work && !heavy && !STRICT) ctr.u[11] += (cnt == n_l);
#endif
        RT_LAP(ctr, PF_CHUNK_SHFL);
        RT_FRESH();
        if (need && dt > RT_EPS) {
            double shade = point ? (cnt ? 1.0 : 0.0) : (double)cnt / n_l;
            if (!(shade < RT_EPS)) {
                const d3 Lc = d3{L.color.x, L.color.y, L.color.z};
                // (the coefficients are read from the table again for every lamp: two registers fewer alive across the shadow rays)
                const double m_diff = shade_tbl[h.prim].diff, m_spec = shade_tbl[h.prim].spec;
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
#undef P
#undef sc
#undef L
}

// ---- one path per lane: CR/Scene.cu:23-58 + CR/Scene_Material.cu:68-118 as a loop ----------------
enum { ST_REFL = 1, ST_REFR = 2, ST_PENDING_REFR = 4, ST_FRONT = 8 };

// q = n / d for n < 2^31, with inv a double slightly above 1 / d: the estimate is exact or one too large
DEV uint32_t div_by(uint32_t n, uint32_t d, double inv) {
    uint32_t q = (uint32_t)((double)n * inv);
    if (q * d > n) q--;
    return q;
}

// colour sum of frame k in the LDS part of the frame stack (six words; see trace_stream)
DEV void st_frame_acc(uint32_t* fl, int k, d3 a) {
    uint32_t* p = fl + k * 7 * 64;
    p[0] = (uint32_t)__double2loint(a.x), p[64] = (uint32_t)__double2hiint(a.x);
    p[128] = (uint32_t)__double2loint(a.y), p[192] = (uint32_t)__double2hiint(a.y);
    p[256] = (uint32_t)__double2loint(a.z), p[320] = (uint32_t)__double2hiint(a.z);
}
DEV d3 ld_frame_acc(const uint32_t* fl, int k) {
    const uint32_t* p = fl + k * 7 * 64;
    return D3(__hiloint2double((int)p[64], (int)p[0]), __hiloint2double((int)p[192], (int)p[128]), __hiloint2double((int)p[320], (int)p[256]));
}

// All paths of a launch.  The wave keeps up to 64 paths in flight, one per lane; lanes whose path has ended wait
// until enough of them are idle, then take the next unstarted paths of the launch together ("path regeneration").
// FLDS = how many frames of the frame stack (the two at its bottom: depth 1 and 2) keep their colour sum and primitive word in
// LDS instead of scratch; `fl` = this lane's column of that area, word w of frame k at fl[(k * 7 + w) * 64].  Most paths never get
// deeper, so most pushes and pops then stay on the chip (scratch is per-lane memory behind L2: every frame written there is
// HBM traffic, and a pop waits for it to come back).
#ifndef RT_FRAME_LDS
#define RT_FRAME_LDS 2
#endif
#ifndef RT_FRAME_LDS_LEAN
#define RT_FRAME_LDS_LEAN 4  // in the kernels without traversal stacks
#endif
template <bool STRICT, bool SHARE, int FLDS, bool CONE, bool MESH4>
DEV void trace_stream(RPREF P_in, const DevShade* shade, Counters& ctr, uint32_t* stk, uint32_t* xch, uint32_t* fl) {
    // The launch parameters and the scene record are read from the kernel-argument segment where they are needed, through a
    // pointer the compiler has to take as new at the start of every stage (RT_FRESH): otherwise all ~100 words of them are
    // loaded at the kernel's entry, kept for its whole life, and - there being no scalar registers for that - parked in lanes
    // of vector registers, from where every use fetches them with a vector instruction.
    const RenderParams RT_CONST* pp = &P_in;
#define P (*pp)
#define sc (pp->sc)
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

    for (;;) {
        // ---- idle lanes start the next paths of the launch (CR/RayTracer.cu:52-72) ----
        const unsigned long long idle = __ballot(!tracing);
        if (!drained && (__popcll(idle) >= RT_REFILL_MIN || idle == ~0ull)) {
            RT_FRESH();
            const uint32_t block_paths = 64u * (uint32_t)P.spp;
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
                    map_pixel(P.flags, P.tex_w, P.tex_h, sc.cam_w, sc.cam_h, px, py, x, y);
                    const d3 dc = unit(emit(sc, x, y));
                    pixel = (uint32_t)py * (uint32_t)P.tex_w + (uint32_t)px;
                    smp = s;
                    my_path = (blk * (uint32_t)P.spp + s) * 64u + slot;
                    sp = 0, draw = 0, tracing = true;
                    o = D3(sc.camO), d = dc;
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
                        const d3 focal = D3(sc.camO) + dc * P.focal;
                        float fa, fb;
                        frand_pair(P.k0, P.k1, pixel, smp, draw, fa, fb);
                        draw += 2;
                        double rx = (fa - 0.5) * P.aperture / 10.0;
                        double ry = (fb - 0.5) * P.aperture / 10.0;
                        o = o + (D3(sc.camDx) * rx + D3(sc.camDy) * ry);
                        d = unit(focal - o);
                    }
                }
            }
        }
        if (__ballot(tracing) == 0ull) {
            if (drained) break;
            continue;  // every slot just handed out lies outside its tile
        }
        RT_LAP(ctr, PF_REFILL);

        Hit h;
        h.prim = -1, h.leaf = -1, h.dist = RT_BIG_DIST, h.front = false, h.N = D3(0, 0, 0), h.C = D3(0, 0, 0);
        bool surface = false;
        d3 result = D3(0, 0, 0);
#ifdef RT_UTIL_COUNTERS
        ctr.u_trace_slots++;
        ctr.u_trace_active += (tracing && sp + 1 <= RT_MAX_DEPTH) ? 1u : 0u;
#endif
        RT_FRESH();
        if (tracing && sp + 1 <= RT_MAX_DEPTH) {
            if (sp > 0) ctr.rays_secondary++;
            h = scene_closest<STRICT, false, MESH4>(sc, P.flags, o, d, ctr, nullptr, 0, no_log, stk, RT_SHARE_CLOSEST && SHARE ? xch : nullptr);
            if (h.prim < 0) {
                d3 c = unit(d);
                double t = 0.5 * (((P.flags & RT_F_SKY_Y) ? c.y : c.z) + 1.0);
                result = clamp1(D3(sc.bg_bottom) * (1.0 - t) + D3(sc.bg_top) * t);
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
        }
        d3 local = D3(0, 0, 0);
        const unsigned long long mask = __ballot(need_shade);
        RT_FRESH();
        if (mask != 0ull) local = diffusion_coop<STRICT, SHARE, CONE, MESH4>(P, shade, mask, need_shade, h, draw, pixel, smp, ctr, stk, xch);

        RT_FRESH();
        if (tracing) {
            bool have_result = !surface;
            if (surface) {
                const DevShade& m = shade[h.prim];
                d3 acc = D3(0, 0, 0);
                if (need_shade) acc = acc + local;
                const bool has_refl = m.refl > RT_EPS, has_refr = m.refr > RT_EPS;
                if (has_refl || has_refr) {
                    const int k = sp;
                    if (FLDS > 0 && k < FLDS) st_frame_acc(fl, k, acc);
                    else f_acc[k][0] = acc.x, f_acc[k][1] = acc.y, f_acc[k][2] = acc.z;
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
                    if (FLDS > 0 && k < FLDS) fl[(k * 7 + 6) * 64] = top;
                    else f_prim[k] = top;
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
                const bool in_lds = FLDS > 0 && k < FLDS;
                d3 acc = (in_lds ? ld_frame_acc(fl, k) : D3(f_acc[k][0], f_acc[k][1], f_acc[k][2])) + contrib;
                if ((st & ST_REFL) && (st & ST_PENDING_REFR)) {
                    top = (fp & 0x0fffffffu) | ((st & ST_FRONT) | ST_REFR) << 28;
                    if (in_lds) st_frame_acc(fl, k, acc), fl[(k * 7 + 6) * 64] = top;
                    else f_acc[k][0] = acc.x, f_acc[k][1] = acc.y, f_acc[k][2] = acc.z, f_prim[k] = top;
                    double n = m.rindex;
                    if (st & ST_FRONT) n = 1.0 / n;
                    d3 V = D3(f_V[k][0], f_V[k][1], f_V[k][2]), N = D3(f_N[k][0], f_N[k][1], f_N[k][2]);
                    o = D3(f_C[k][0], f_C[k][1], f_C[k][2]);
                    d = refract(V, N, n);
                    have_result = false;  // the refraction child is traced next (depth k+2 = sp+1)
                } else {
                    result = clamp1(acc);
                    sp--;
                    if (sp > 0) top = (FLDS > 0 && sp - 1 < FLDS) ? fl[((sp - 1) * 7 + 6) * 64] : f_prim[sp - 1];
                }
            }
        }
        RT_LAP(ctr, PF_FRAME);
    }
#undef P
#undef sc
}

// ---- the render kernel: CR/RayTracer.cu:40-94 --------------------------------------------------
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 3  // register budget: 512 / RT_WAVES_PER_SIMD VGPRs per lane; measured best (2: 100 ms, 3: 96 ms, 4: 113 ms per frame)
#endif
// SHADE_CAP > 0: the shading table (at most SHADE_CAP primitives) is copied to LDS and read from there.  A template
// parameter rather than a run-time choice so that the reads compile to LDS instructions, not to flat ones; two
// capacities because a smaller LDS footprint measures 1 % faster on small scenes.
// SHARE: shadow rays walk the meshes with mesh_anyhit_shared (idle lanes take over parts of long walks).  Faster where
// such walks dominate (c5_final: 265 -> 205 ms/frame), slower elsewhere because of the registers it takes (c4_lucy:
// 44 -> 49 ms), so the host picks the variant per scene by timing both on part of a frame.
// WAVES: resident waves per SIMD the kernel is built for (register budget 512 / WAVES).  Three is best wherever the varied
// path code dominates (at four it spills three times as much: c4_lucy 44 -> 57 ms/frame); four pays where shadow rays
// walking meshes dominate, because that loop waits on dependent node loads (c5_final at 32 spp: 182 -> 169 ms/frame).
template <bool STRICT, int SHADE_CAP, bool SHARE, int WAVES = RT_WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES) void render_kernel(RenderParams) {
    // The kernels with an LDS shading table and per-lane shadow walks are launched only for scenes in which no mesh has a
    // four-wide tree (launch_tiles): the four-wide walks and their LDS stacks are left out of them.
    constexpr bool MESH4 = !STRICT && (SHARE || SHADE_CAP == 0);
    // frames kept in LDS: only where three blocks still fit a CU's 160 KB with them (small shading table and three-waves build, or no stacks)
    constexpr int FLDS = STRICT ? 0 : (SHADE_CAP > 0 && !MESH4) ? RT_FRAME_LDS_LEAN : (SHADE_CAP == 16 && WAVES == 3) ? RT_FRAME_LDS : 0;
    __shared__ uint32_t s_frames[FLDS > 0 ? 4 * FLDS * 7 * 64 : 1];
    uint32_t* fl = FLDS > 0 ? s_frames + (threadIdx.x >> 6) * (FLDS * 7 * 64) + (threadIdx.x & 63) : nullptr;
    RPREF P = KERNARG(RenderParams);
    const int lane = lane_id();
    // per-lane traversal stacks of the block's four waves: [wave][entry][lane]
    __shared__ uint32_t s_stack[MESH4 ? 4 * RT_STACK_ROWS * 64 : 1];
    uint32_t* stk = MESH4 ? s_stack + (threadIdx.x >> 6) * (RT_STACK_ROWS * 64) + lane : nullptr;
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
    ctr.traced_wave = 0;
#ifdef RT_UTIL_COUNTERS
    ctr.u_shadow_slots = ctr.u_shadow_items = ctr.u_trace_slots = ctr.u_trace_active = 0;
    for (int i = 0; i < 12; i++) ctr.u[i] = 0;
    __shared__ unsigned long long s_prof[4][1 + PF_COUNT];
    ctr.prof = s_prof[threadIdx.x >> 6];
    if (lane < 1 + PF_COUNT) ctr.prof[lane] = lane == 0 ? __builtin_readcyclecounter() : 0ull;
#endif
    // flags and exchange arrays of the shared walks (192 words per wave), plus the per-ray hit records of the shared closest-hit walk
    constexpr int XCH_WORDS = RT_SHARE_CLOSEST ? 448 : 192;
    __shared__ __attribute__((aligned(8))) uint32_t s_xch[SHARE ? 4 * XCH_WORDS : 2];
    uint32_t* xch = SHARE ? s_xch + (threadIdx.x >> 6) * XCH_WORDS : nullptr;
    // the look down the meshes' trees before shadow rays are dealt (mesh_cone_may): in the build for mesh-heavy scenes only.
    // Elsewhere it saves less than the registers it takes cost (c4_lucy 35.0 -> 36.3 ms/frame with it, c5_final 137 -> 118).
    constexpr bool CONE = !STRICT && RT_CONE_WALK && WAVES == 4;
    trace_stream<STRICT, SHARE, FLDS, CONE, MESH4>(P, shade, ctr, stk, xch, fl);
#ifdef RT_UTIL_COUNTERS
    RT_LAP(ctr, PF_REFILL);
    if (lane < PF_COUNT) atomicAdd(&P.counters[CTR_ALL + lane], ctr.prof[1 + lane]);
    unsigned long long v[CTR_ALL] = {ctr.rays_secondary, ctr.rays_shadow, ctr.nodes, ctr.tris, ctr.shaded, ctr.texfetch, 0ull,
                                     ctr.u_shadow_slots, ctr.u_shadow_items, ctr.u_trace_slots, ctr.u_trace_active,
                                     ctr.u[0], ctr.u[1], ctr.u[2], ctr.u[3], ctr.u[4], ctr.u[5], ctr.u[6], ctr.u[7], ctr.u[8], ctr.u[9], ctr.u[10], ctr.u[11]};
    const int n_ctr = CTR_ALL;
#else
    unsigned long long v[CTR_COUNT] = {ctr.rays_secondary, ctr.rays_shadow, ctr.nodes, ctr.tris, ctr.shaded, ctr.texfetch, 0ull};
    const int n_ctr = CTR_COUNT;
#endif
#pragma unroll
    for (int i = 0; i < n_ctr; i++) {
        unsigned long long s = wave_sum_u64(v[i]);
        if (lane == 0 && s) atomicAdd(&P.counters[i], s);
    }
    if (lane == 0 && ctr.traced_wave) atomicAdd(&P.counters[CTR_TRACED], (unsigned long long)ctr.traced_wave);
}

// ---- pixel sums: CR/RayTracer.cu:74-89 ---------------------------------------------------------
// One wave per 8x8 block, lane = slot.  The samples of a pixel are added in sample order, because the reference
// adds them in that order inside one thread and double addition is not associative.
__global__ __launch_bounds__(256) void resolve_kernel(RenderParams) {
    RPREF P = KERNARG(RenderParams);
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

#ifdef RT_EXPERIMENTS
#include "rt_defer.hpp"  // the same render as three lean kernels: measured slower everywhere, kept for A/B runs only
#endif

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
    SCREF sc = KERNARG(ProbeParams).sc;
    int x, y;
    map_pixel(Q.flags, Q.tex_w, Q.tex_h, sc.cam_w, sc.cam_h, Q.x0 + i % Q.w, Q.y0 + i / Q.w, x, y);
    d3 d = emit(sc, x, y);
    if (Q.flags & RT_F_NORMALIZE_PRIMARY) d = unit(d);
    Counters ctr;
    uint32_t n = 0;
    uint32_t* log = Q.visit_log ? Q.visit_log + (size_t)i * Q.visit_cap : nullptr;
    Hit h = log ? scene_closest<true, true>(sc, Q.flags, Q.sc.camO, d, ctr, log, (uint32_t)Q.visit_cap, n, nullptr)
                : scene_closest<true, false>(sc, Q.flags, Q.sc.camO, d, ctr, nullptr, 0, n, nullptr);
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

// Device memory a render call works in besides the scene, the image and the caller's buffers.
struct Workspace {
    uint32_t* queue = nullptr;               // 64 words: heads of the eight ranges of the path queue
    unsigned long long* counters = nullptr;  // CTR_ALL + 16 work counters
    BlockRec* blocks = nullptr;              // the 8x8 blocks of the call, in queue order
    size_t blocks_cap = 0;
    double* path_out = nullptr;              // colours of the finished paths of one launch
    size_t path_out_bytes = 0;
    bool fixed = false;                      // true: the buffers above are all there is (nothing may be allocated or grown)
#ifdef RT_EXPERIMENTS
    // the three-kernel render (rt_defer.hpp): node records, shadow jobs per lamp, allocation counters, last node per path
    NodeRec* nodes = nullptr;
    JobRec* jobs = nullptr;
    uint32_t* alloc = nullptr;
    uint32_t* last_node = nullptr;
    size_t nodes_cap = 0, jobs_cap = 0, last_cap = 0;
#endif
};

struct rt_scene {
    int device = 0;
    DevScene dev{};
    void* slab = nullptr;
    size_t slab_bytes = 0;
    Workspace ws;                          // queue heads, counters, block list, path colours: allocated at first use, grown on demand
    uint8_t* image = nullptr;              // staging image for host destinations
    size_t image_bytes = 0;
    int shared_walk = 0;                   // shadow mesh walks shared by the wave (render_kernel<.., SHARE = true>): 0 = no, 1 = yes
    int waves = RT_WAVES_PER_SIMD;         // kernel build the scene gets: RT_WAVES_PER_SIMD, or 4 for mesh-heavy scenes (shared variant only)
#ifdef RT_EXPERIMENTS
    int defer = 0;                         // pruned renders go through the three-kernel path (rt_defer.hpp) instead of the one kernel
    uint32_t launch_id = 0;                // tag of the last launch of the three-kernel render
    int defer_fallbacks = 0;               // calls that had to be rendered again by the one kernel because a record buffer was too small
#endif
    size_t n_blocks = 0;                   // blocks in ws.blocks: those of last_tiles
    std::vector<Tile> last_tiles;
    int cus = 0, blocks_per_cu_fast = 0, blocks_per_cu_strict = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int desc_cam_w = 0, desc_cam_h = 0;
    float tree_build_ms = 0.0f;  // device time of the four-wide tree build, 0 if the host built the trees
    std::vector<int32_t> tree_roots;               // per mesh: root of its four-wide tree, -1 = none (rt_scene_check_trees)
    std::vector<std::vector<uint32_t>> tree_tris;  // per mesh with such a tree: leaf indices of its triangles
    size_t n_nodes4 = 0, n_leaves = 0;
};

// ---- fixed workspace of rt_render_tile_noalloc ----------------------------------------------------
// One array per GPU the library runs on, held until the process ends: queue heads | counters | block list | path colours.
#define RT_NOALLOC_BLOCKS 16384u                     // 8x8 pixel blocks of one call (1 M pixels)
#define RT_NOALLOC_PATHS (1u << 19)                  // paths of one launch; a call with more is cut into several launches
#define RT_NOALLOC_OFF_COUNTERS 256u
#define RT_NOALLOC_OFF_BLOCKS 1024u
#define RT_NOALLOC_OFF_PATHS (RT_NOALLOC_OFF_BLOCKS + RT_NOALLOC_BLOCKS * 8u)
#define RT_NOALLOC_BYTES (RT_NOALLOC_OFF_PATHS + RT_NOALLOC_PATHS * 24u)  // 12.1 MiB
static_assert(sizeof(unsigned long long) * (CTR_ALL + 16) <= RT_NOALLOC_OFF_BLOCKS - RT_NOALLOC_OFF_COUNTERS, "counters fit their slot");
static_assert(RT_NOALLOC_OFF_PATHS % 256 == 0 && sizeof(BlockRec) == 8, "workspace layout");
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];

#define RT_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static unsigned char* g_scratch_ptr[RT_MAX_DEVICES];  // per device: where that GPU's copy of `scratch` lives (null = not looked up yet)
static std::mutex g_scratch_use[RT_MAX_DEVICES];      // per device: held by the one call that is working in that GPU's copy

// Address of the current device's copy of `scratch`.  The array's name is only a handle on the host; every use of the
// array's memory from host code (kernel arguments, memsets, copies) goes through the pointer returned here.
static int scratch_of_device(int device, unsigned char** out) {
    if (device < 0 || device >= RT_MAX_DEVICES) return rt_fail(RT_ERR_LIMIT, "device index beyond RT_MAX_DEVICES");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        int cur = -1;
        HIPCHK(hipGetDevice(&cur));
        if (cur != device) return rt_fail(RT_ERR_INVALID, "scratch_of_device: the device is not current");
        void* ptr = nullptr;
        HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return rt_fail(RT_ERR_DEVICE, "hipGetSymbolAddress returned a null address for the workspace");
        g_scratch_ptr[device] = (unsigned char*)ptr;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

extern "C" int rt_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return rt_fail(RT_ERR_NOGPU, "no HIP device visible");
    if (device < 0 || device >= n) return rt_fail(RT_ERR_INVALID, "rt_init: device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rt_fail(RT_ERR_NOGPU, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    unsigned char* ws = nullptr;
    return scratch_of_device(device, &ws);  // this GPU's fixed workspace (rt_render_tile_noalloc), looked up once
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


#include "rt_scene_build.hpp"

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->slab) (void)hipFree(s->slab);
    if (s->ws.queue) (void)hipFree(s->ws.queue);
    if (s->ws.counters) (void)hipFree(s->ws.counters);
    if (s->image) (void)hipFree(s->image);
#ifdef RT_EXPERIMENTS
    if (s->ws.nodes) (void)hipFree(s->ws.nodes);
    if (s->ws.jobs) (void)hipFree(s->ws.jobs);
    if (s->ws.alloc) (void)hipFree(s->ws.alloc);
    if (s->ws.last_node) (void)hipFree(s->ws.last_node);
#endif
    if (s->ws.path_out) (void)hipFree(s->ws.path_out);
    if (s->ws.blocks) (void)hipFree(s->ws.blocks);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

#include "rt_tree_device.hpp"

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

    // one slab: lights | geometry | Bezier segments | shading table | nodes | four-wide nodes | parents | leaves | candidate grid | texels, one copy
    std::vector<DevShade> shade((size_t)d->n_primitives);
    std::vector<DevGeom> geom((size_t)d->n_primitives), segs;
    std::vector<uint32_t> tex_off((size_t)d->n_textures);
    Tree4 T;
    const int tree_builder = tree_builder_choice();
    build_tree4(d, T, tree_builder != 0);
    size_t texel_bytes = 0;
    for (int i = 0; i < d->n_textures; i++) {
        tex_off[i] = (uint32_t)texel_bytes;
        texel_bytes += align_up((size_t)d->textures[i].width * d->textures[i].height * 3, 16);
    }
    if (texel_bytes >= (1ull << 32)) return rt_fail(RT_ERR_LIMIT, "more than 4 GiB of textures");
    for (int i = 0; i < d->n_primitives; i++) {
        const rt_primitive& p = d->primitives[i];
        int tw = 0, th = 0;
        uint32_t toff = 0;
        if (p.material.texture >= 0) {
            const rt_texture& t = d->textures[p.material.texture];
            tw = t.width, th = t.height, toff = tex_off[p.material.texture];
        }
        build_geom(d, p, geom[i], segs, T);
        build_shade(p, tw, th, toff, shade[i]);
    }
    size_t off_l = 0;
    size_t off_g = align_up(off_l + sizeof(rt_light) * d->n_lights, 256);
    size_t off_s = align_up(off_g + sizeof(DevGeom) * d->n_primitives, 256);
    size_t off_sh = align_up(off_s + sizeof(DevGeom) * segs.size(), 256);
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
    rc = run_tree_jobs(T, (DevNode4*)(base + off_n2), (const rt_bvh_leaf*)(base + off_f), &s->tree_build_ms, tree_builder == 2);
    if (rc) return rc;
    s->tree_roots = T.mesh_root4, s->tree_tris = std::move(T.mesh_tris), s->n_nodes4 = T.nodes.size(), s->n_leaves = (size_t)d->n_leaves;
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
    v.mesh_scale_max = 0.0;
    for (int i = 0; i < d->n_primitives; i++)
        if (geom[i].type == RT_PRIM_MESH && geom[i].aux2) v.mesh_scale_max = std::fmax(v.mesh_scale_max, geom[i].f[MSH_SCALE]);
    v.lights = (const rt_light*)(base + off_l);
    v.geom = (const DevGeom*)(base + off_g);
    v.segs = (const DevGeom*)(base + off_s);
    v.shade = (const DevShade*)(base + off_sh);
    v.nodes = (const rt_bvh_node*)(base + off_n);
    v.nodes4 = (const DevNode4*)(base + off_n2);
    v.node_parent = (const uint32_t*)(base + off_np);
    v.leaf_parent = (const uint32_t*)(base + off_lp);
    v.leaves = (const rt_bvh_leaf*)(base + off_f);
    v.texels = base + off_t;
    v.cand_grid = cgrid.empty() ? nullptr : (const unsigned long long*)(base + off_cg);
    for (int k = 0; k < 3; k++) v.cg_lo[k] = cg_lo[k], v.cg_inv[k] = cg_inv[k];

    HIPCHK(hipMalloc(&s->ws.queue, 256));
    HIPCHK(hipMalloc(&s->ws.counters, sizeof(unsigned long long) * (CTR_ALL + 16)));
    HIPCHK(hipEventCreate(&s->ev0));
    HIPCHK(hipEventCreate(&s->ev1));
    // Which variant of the shadow mesh walk the scene gets.  Sharing pays where shadow rays walk meshes at all (c5_final:
    // 265 -> 198 ms per frame) and costs nothing measurable where they rarely do (c4_lucy: 43.3 vs 43.2 ms), so the rule is
    // static: scenes with a four-wide tree get the shared variant.  RT_AMD_SHARED_WALK=0|1 overrides it.
    s->shared_walk = (v.mesh_scale_max > 0.0 && v.n_prims <= RT_SHADE_LDS) ? 1 : 0;
    if (const char* e = std::getenv("RT_AMD_SHARED_WALK"))
        if ((e[0] == '0' || e[0] == '1') && v.n_prims <= RT_SHADE_LDS) s->shared_walk = e[0] - '0';
#ifdef RT_EXPERIMENTS
    // The three-kernel render needs room for every lamp in a node record and the primitive index in 24 bits.
    {
        const bool can = d->n_lights <= RT_DEFER_LAMPS && d->n_primitives < (1 << 24) && v.shade_n < (1 << 16);
        s->defer = 0;
        if (const char* e = std::getenv("RT_AMD_DEFER"))
            if (e[0] == '1' && can) s->defer = 1;
    }
#endif
    // Four waves per SIMD where shadow walks over meshes carry the frame: several meshes, or one large one (more triangles than
    // 64 Ki).  RT_AMD_WAVES=3|4 overrides the rule.
    {
        int meshes = 0;
        for (int i = 0; i < d->n_primitives; i++)
            if (geom[i].type == RT_PRIM_MESH && geom[i].aux2) meshes++;
        // (the four-wave build exists for the small shading table only: with the large one, four blocks do not fit a CU's LDS)
        const bool can4 = s->shared_walk && v.n_prims <= 16;
        s->waves = (can4 && (meshes >= 2 || d->n_leaves > 65536u)) ? 4 : RT_WAVES_PER_SIMD;
        if (const char* e = std::getenv("RT_AMD_WAVES"))
            if (e[0] == '3' || (e[0] == '4' && can4)) s->waves = e[0] - '0';
    }
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

// The 8x8 blocks of the tiles in Z-order: blocks that follow each other in the queue are neighbours in the image,
// and each of the eight ranges of the queue is a compact region (its rays meet the same part of the scene in L2).
static int blocks_of(const std::vector<Tile>& tiles, std::vector<BlockRec>& recs) {
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
    recs.resize(keyed.size());
    for (size_t i = 0; i < keyed.size(); i++) recs[i] = keyed[i].second;
    return RT_OK;
}

// Render the given tiles into `dst` (device image).  Fills kernel time and counters.
#if defined(RT_EXPERIMENTS) && !defined(RT_DEFER_PATHS)
#define RT_DEFER_PATHS (1ull << 25)  // most paths of one launch of the three-kernel render (node records + jobs: about 0.9 KB per path)
#endif
static int launch_tiles(rt_scene* s, Workspace& ws, const rt_settings* st, int tex_w, int tex_h, int cam_w, int cam_h,
                        const std::vector<Tile>& tiles, uint8_t* dst, rt_stats* stats, bool allow_defer = true) {
    uint64_t pixels = 0;
    for (const Tile& t : tiles) pixels += (uint64_t)t.w * t.h;
    // The block list of the scene's own workspace is kept from call to call: a renderer asks for the same tiles frame after frame.
    const bool same_tiles = !ws.fixed && s->last_tiles.size() == tiles.size() &&
                            (tiles.empty() || std::memcmp(s->last_tiles.data(), tiles.data(), tiles.size() * sizeof(Tile)) == 0);
    uint64_t blocks = s->n_blocks;
    if (!same_tiles) {
        std::vector<BlockRec> recs;
        int rc = blocks_of(tiles, recs);
        if (rc) return rc;
        if (recs.size() > ws.blocks_cap) {
            if (ws.fixed) return rt_fail(RT_ERR_LIMIT, "rectangle has more 8x8 pixel blocks than the fixed workspace holds");
            s->last_tiles.clear(), s->n_blocks = 0;  // the cached list goes with its buffer (a failed allocation below must not leave a cache hit)
            if (ws.blocks) (void)hipFree(ws.blocks);
            ws.blocks = nullptr, ws.blocks_cap = 0;
            const size_t cap = recs.size() + 1024;
            HIPCHK(hipMalloc(&ws.blocks, cap * sizeof(BlockRec)));
            ws.blocks_cap = cap;
        }
        if (!ws.fixed) s->last_tiles.clear();  // stays empty if the copy fails
        if (!recs.empty()) HIPCHK(hipMemcpyAsync(ws.blocks, recs.data(), recs.size() * sizeof(BlockRec), hipMemcpyHostToDevice, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));  // `recs` goes out of scope
        blocks = recs.size();
        if (!ws.fixed) s->last_tiles = tiles, s->n_blocks = recs.size();
    }
    const bool strict = (st->flags & RT_F_STRICT_TRAVERSAL) != 0;
    const bool four = !strict && s->shared_walk == 1 && s->waves == 4;
    const int bpc = strict ? s->blocks_per_cu_strict : four ? 4 : s->blocks_per_cu_fast;
    // blocks per launch: the colours of all paths of a launch are kept until its pixel sums are taken
    const uint64_t block_paths = 64ull * (uint64_t)st->spp;
    uint64_t launch_paths = RT_LAUNCH_PATHS;
    if (const char* e = std::getenv("RT_AMD_LAUNCH_PATHS")) {  // smaller launches on request (tests of the multi-launch path, small-memory runs)
        const unsigned long long v = std::strtoull(e, nullptr, 10);
        if (v >= 64 && v < launch_paths) launch_paths = v;
    }
#ifdef RT_EXPERIMENTS
    const bool defer = allow_defer && !strict && !ws.fixed && s->defer == 1;
    if (defer) launch_paths = std::min<uint64_t>(launch_paths, RT_DEFER_PATHS);
#else
    (void)allow_defer;
#endif
    if (ws.fixed) {
        const uint64_t cap = ws.path_out_bytes / (3 * sizeof(double));
        if (cap < block_paths) return rt_fail(RT_ERR_LIMIT, "too many samples per pixel for the fixed workspace (one 8x8 block must fit)");
        launch_paths = std::min(launch_paths, cap);
    }
    const uint64_t chunk_blocks = std::max<uint64_t>(1, launch_paths / block_paths);
    const size_t out_bytes = (size_t)(std::min(blocks, chunk_blocks) * block_paths) * 3 * sizeof(double);
    if (ws.path_out_bytes < out_bytes) {
        if (ws.fixed) return rt_fail(RT_ERR_LIMIT, "fixed workspace too small");  // not reached: chunk_blocks was cut to fit
        if (ws.path_out) (void)hipFree(ws.path_out);
        ws.path_out = nullptr, ws.path_out_bytes = 0;
        HIPCHK(hipMalloc(&ws.path_out, out_bytes));
        ws.path_out_bytes = out_bytes;
    }
    HIPCHK(hipMemsetAsync(ws.counters, 0, sizeof(unsigned long long) * (CTR_ALL + 16), g_stream));
#ifdef RT_EXPERIMENTS
    DeferParams D{};
    if (defer) {
        // record buffers for the largest launch of this call: six nodes per path (a path has one node per ray it traces), and per
        // lamp 1.25 jobs per path; a call that needs more is rendered again by the one kernel (the overflow word tells)
        const uint64_t paths = std::min(blocks, chunk_blocks) * block_paths;
        // (every wave of the path kernel may leave most of one allocation chunk unused when it ends)
        const uint64_t path_waves = 4 * std::min<uint64_t>((uint64_t)s->cus * RT_PATH_WAVES, (paths + 255) / 256);
        size_t want_nodes = (size_t)(6 * paths + path_waves * RT_NODE_CHUNK + 65536);
        if (const char* e = std::getenv("RT_AMD_DEFER_NODES")) {  // a fixed node capacity (tests of the fall-back to the one kernel)
            const unsigned long long v = std::strtoull(e, nullptr, 10);
            if (v >= 1) want_nodes = (size_t)v, ws.nodes_cap = std::min(ws.nodes_cap, want_nodes);
        }
        size_t want_jobs = (size_t)(3 * paths + path_waves * RT_JOB_CHUNK + 65536);  // per lamp: at most one job per shaded node
        want_jobs = std::min<size_t>(want_jobs, (size_t)(0x7FFFFFFFull / (uint64_t)std::max(1, s->dev.shade_n)));
        if (ws.nodes_cap < want_nodes) {
            if (ws.nodes) (void)hipFree(ws.nodes);
            ws.nodes = nullptr, ws.nodes_cap = 0;
            HIPCHK(hipMalloc(&ws.nodes, want_nodes * sizeof(NodeRec)));
            ws.nodes_cap = want_nodes;
        }
        if (ws.jobs_cap < want_jobs) {
            if (ws.jobs) (void)hipFree(ws.jobs);
            ws.jobs = nullptr, ws.jobs_cap = 0;
            const size_t bytes = want_jobs * (size_t)std::max(1, s->dev.n_lights) * sizeof(JobRec);
            HIPCHK(hipMalloc(&ws.jobs, bytes));
            HIPCHK(hipMemsetAsync(ws.jobs, 0, bytes, g_stream));  // no slot carries a launch tag yet
            ws.jobs_cap = want_jobs;
        }
        if (ws.last_cap < paths) {
            if (ws.last_node) (void)hipFree(ws.last_node);
            ws.last_node = nullptr, ws.last_cap = 0;
            HIPCHK(hipMalloc(&ws.last_node, (size_t)paths * sizeof(uint32_t)));
            ws.last_cap = (size_t)paths;
        }
        if (!ws.alloc) HIPCHK(hipMalloc(&ws.alloc, 256));
        D.nodes = ws.nodes, D.jobs = ws.jobs, D.alloc = ws.alloc, D.last_node = ws.last_node;
        D.nodes_cap = (uint32_t)std::min<size_t>(std::min(ws.nodes_cap, want_nodes), 0xFFFFFFF0u), D.jobs_cap = (uint32_t)ws.jobs_cap;
        D.overflow = ws.counters + (CTR_ALL + 15);
    }
#endif

    RenderParams P;
    P.sc = s->dev;
    P.sc.cam_w = cam_w > 0 ? cam_w : s->desc_cam_w;
    P.sc.cam_h = cam_h > 0 ? cam_h : s->desc_cam_h;
    P.tex_w = tex_w, P.tex_h = tex_h, P.spp = st->spp, P.dof = st->dof_enabled ? 1 : 0;
    P.focal = st->focal_distance, P.aperture = st->aperture;
    P.inv_block_paths = 1.0 / (double)block_paths * (1.0 + 1e-12), P.inv_spp = 1.0 / (double)st->spp * (1.0 + 1e-12);
    P.k0 = (uint32_t)st->seed, P.k1 = (uint32_t)(st->seed >> 32), P.flags = st->flags;
    P.queue = ws.queue, P.out = dst, P.counters = ws.counters, P.path_out = ws.path_out;

    const bool share = !strict && s->shared_walk == 1;
    const bool tree4 = s->dev.mesh_scale_max > 0.0;  // some mesh has a four-wide tree
    // one launch: the blocks [first, first + nb) of the list, traced and resolved
    auto launch = [&](uint64_t first, uint64_t nb) -> int {
        P.blocks = ws.blocks + first, P.n_blocks = (uint32_t)nb, P.n_paths = (uint32_t)(nb * block_paths);
        P.range_paths = (uint32_t)(((nb + 7) / 8) * block_paths);
        HIPCHK(hipMemsetAsync(ws.queue, 0, 256, g_stream));
        // persistent grid: every wave draws paths until the launch runs dry
        uint64_t grid = (uint64_t)s->cus * bpc;
        const uint64_t want_blocks = ((uint64_t)P.n_paths + 255) / 256;
        if (grid > want_blocks) grid = want_blocks;
#ifdef RT_EXPERIMENTS
        if (defer) {
            // path kernel (no shadow rays) -> shadow kernel (every job) -> fold kernel (colours), see rt_defer.hpp
            HIPCHK(hipMemsetAsync(ws.alloc, 0, 256, g_stream));
            if (++s->launch_id == 0) s->launch_id = 1;
            D.launch_id = s->launch_id;
            HIPCHK(hipMemsetAsync(ws.last_node, 0xFF, (size_t)P.n_paths * sizeof(uint32_t), g_stream));
            uint64_t pg = std::min<uint64_t>((uint64_t)s->cus * RT_PATH_WAVES, want_blocks);
            const dim3 gp((unsigned)pg), t(256);
            if (P.sc.n_prims <= 16) hipLaunchKernelGGL((path_kernel<16>), gp, t, 0, g_stream, P, D);
            else if (P.sc.n_prims <= RT_SHADE_LDS) hipLaunchKernelGGL((path_kernel<RT_SHADE_LDS>), gp, t, 0, g_stream, P, D);
            else hipLaunchKernelGGL((path_kernel<0>), gp, t, 0, g_stream, P, D);
            HIPCHK(hipGetLastError());
            const dim3 gs((unsigned)((uint64_t)s->cus * RT_SHADOW_WAVES));
            if (s->dev.mesh_scale_max > 0.0 && P.sc.n_prims <= 64) hipLaunchKernelGGL((shadow_kernel<true>), gs, t, 0, g_stream, P, D);  // (the shared test reads 64-bit candidate masks)
            else hipLaunchKernelGGL((shadow_kernel<false>), gs, t, 0, g_stream, P, D);
            HIPCHK(hipGetLastError());
            const dim3 gf((unsigned)std::min<uint64_t>(want_blocks, (uint64_t)s->cus * 8));
            hipLaunchKernelGGL(fold_kernel, gf, t, 0, g_stream, P, D);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, g_stream, P);
            HIPCHK(hipGetLastError());
            return RT_OK;
        }
#endif
        const dim3 g((unsigned)grid), t(256);
        if (strict) hipLaunchKernelGGL((render_kernel<true, 0, false>), g, t, 0, g_stream, P);
        else if (P.sc.n_prims <= 16 && four) hipLaunchKernelGGL((render_kernel<false, 16, true, 4>), g, t, 0, g_stream, P);
        else if (P.sc.n_prims <= 16 && share) hipLaunchKernelGGL((render_kernel<false, 16, true>), g, t, 0, g_stream, P);
        // (per-lane shadow walks and a four-wide tree - only under RT_AMD_SHARED_WALK=0 - go to the general kernel below)
        else if (P.sc.n_prims <= 16 && !tree4) hipLaunchKernelGGL((render_kernel<false, 16, false>), g, t, 0, g_stream, P);
        else if (P.sc.n_prims <= RT_SHADE_LDS && share) hipLaunchKernelGGL((render_kernel<false, RT_SHADE_LDS, true>), g, t, 0, g_stream, P);
        else if (P.sc.n_prims <= RT_SHADE_LDS && !tree4) hipLaunchKernelGGL((render_kernel<false, RT_SHADE_LDS, false>), g, t, 0, g_stream, P);
        else hipLaunchKernelGGL((render_kernel<false, 0, false>), g, t, 0, g_stream, P);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, g_stream, P);
        HIPCHK(hipGetLastError());
        return RT_OK;
    };
    HIPCHK(hipEventRecord(s->ev0, g_stream));
    for (uint64_t b0 = 0; b0 < blocks; b0 += chunk_blocks) {
        int rc = launch(b0, std::min(chunk_blocks, blocks - b0));
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(s->ev1, g_stream));
    unsigned long long c[CTR_ALL + 16];
    HIPCHK(hipMemcpyAsync(c, ws.counters, sizeof(c), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
#ifdef RT_EXPERIMENTS
    if (defer && c[CTR_ALL + 15] != 0ull) {  // a record buffer was too small for this call: the one kernel renders it
        s->defer_fallbacks++;
        return launch_tiles(s, ws, st, tex_w, tex_h, cam_w, cam_h, tiles, dst, stats, false);
    }
#endif
#ifdef RT_UTIL_COUNTERS
    std::fprintf(stderr, "[util] shadow lanes used %.1f%% (%llu of %llu slots), closest-hit lanes used %.1f%% (%llu of %llu slots)\n",
                 100.0 * c[CTR_U_SHADOW_ITEMS] / (double)(c[CTR_U_SHADOW_SLOTS] ? c[CTR_U_SHADOW_SLOTS] : 1), c[CTR_U_SHADOW_ITEMS], c[CTR_U_SHADOW_SLOTS],
                 100.0 * c[CTR_U_TRACE_ACTIVE] / (double)(c[CTR_U_TRACE_SLOTS] ? c[CTR_U_TRACE_SLOTS] : 1), c[CTR_U_TRACE_ACTIVE], c[CTR_U_TRACE_SLOTS]);
    static const char* const probe_name[6] = {"shadow mesh walks", "closest mesh walks", "shadow interior steps", "shadow leaf steps",
                                              "closest interior steps", "closest leaf steps"};
    static const char* const sect[PF_COUNT] = {"new paths (refill)", "frame pop / hand-back", "closest: analytic prims", "closest: mesh walk", "shade setup + texture",
                                               "light candidates", "chunk shuffles + counts", "rng + light sample", "shadow: analytic prims", "shadow: mesh walk",
                                               "shade math", "child ray set-up"};
    unsigned long long tot = 0;
    for (int k = 0; k < PF_COUNT; k++) tot += c[CTR_ALL + k];
    for (int k = 0; k < PF_COUNT; k++)
        std::fprintf(stderr, "[time] %-26s %5.1f%%\n", sect[k], 100.0 * c[CTR_ALL + k] / (double)(tot ? tot : 1));
#ifndef RT_PROBE_STEPS
    std::fprintf(stderr, "[util] shadow rays traced because of: sphere/plane/square %llu (only: %llu), cylinder/bezier %llu (only: %llu), mesh %llu (only: %llu)\n",
                 c[CTR_U_PROBES + 6], c[CTR_U_PROBES + 7], c[CTR_U_PROBES + 8], c[CTR_U_PROBES + 9], c[CTR_U_PROBES + 10], c[CTR_U_PROBES + 11]);
#endif
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
        stats->rays_shadow_traced = c[CTR_TRACED];
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
        rc = launch_tiles(s, s->ws, st, tex_w, tex_h, cam_w, cam_h, tiles, rgba, stats);
        if (rc) return rc;
    } else {
        rc = ensure_image(s, bytes);
        if (rc) return rc;
        rc = launch_tiles(s, s->ws, st, tex_w, tex_h, cam_w, cam_h, tiles, s->image, stats);
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

// rt_render_tile for callers that may not allocate device memory: the same kernels, with the queue heads, counters, block
// list and path colours in this GPU's copy of `scratch` instead of buffers the scene owns and grows.  Nothing here calls
// hipMalloc / hipFree.  The array keeps the previous call's contents: the queue heads and counters are cleared per launch,
// the block list is rewritten, and path colours are read only where this call wrote them.  Calls of this process that overlap on
// one GPU (other threads, other scenes) take turns: a per-device mutex is held from the first write to the array until the
// call's last kernel has finished.  Not for a capturing stream: the call copies the block list in, reads the counters back and
// synchronises the stream.
extern "C" int rt_render_tile_noalloc(rt_scene* s, const rt_settings* st, int tex_w, int tex_h, int cam_w, int cam_h, int x0, int y0, int w,
                                      int h, uint8_t* rgba_device, rt_stats* stats) {
    auto t0 = std::chrono::steady_clock::now();
    int rc = check_common(s, st, tex_w, tex_h);
    if (rc) return rc;
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > tex_w || y0 + h > tex_h)
        return rt_fail(RT_ERR_INVALID, "rectangle does not lie inside the image");
    if (!rgba_device) return rt_fail(RT_ERR_INVALID, "null output image");
    const uint64_t n_blocks = (uint64_t)((w + 7) / 8) * (uint64_t)((h + 7) / 8);
    if (n_blocks > RT_NOALLOC_BLOCKS)
        return rt_fail(RT_ERR_LIMIT, "rt_render_tile_noalloc: the rectangle has " + std::to_string(n_blocks) + " blocks of 8x8 pixels, the fixed workspace holds " +
                                         std::to_string(RT_NOALLOC_BLOCKS) + " (about 1 M pixels); render it in parts");
    if ((uint64_t)st->spp * 64u > RT_NOALLOC_PATHS)
        return rt_fail(RT_ERR_LIMIT, "rt_render_tile_noalloc: at most " + std::to_string(RT_NOALLOC_PATHS / 64u) + " samples per pixel fit the fixed workspace");
    HIPCHK(hipSetDevice(s->device));
    if (!is_device_ptr(rgba_device)) return rt_fail(RT_ERR_INVALID, "rt_render_tile_noalloc: the image must be device memory (a host image needs a staging buffer)");
    unsigned char* base = nullptr;
    rc = scratch_of_device(s->device, &base);
    if (rc) return rc;
    Workspace ws;
    ws.fixed = true;
    ws.queue = reinterpret_cast<uint32_t*>(base);
    ws.counters = reinterpret_cast<unsigned long long*>(base + RT_NOALLOC_OFF_COUNTERS);
    ws.blocks = reinterpret_cast<BlockRec*>(base + RT_NOALLOC_OFF_BLOCKS), ws.blocks_cap = RT_NOALLOC_BLOCKS;
    ws.path_out = reinterpret_cast<double*>(base + RT_NOALLOC_OFF_PATHS), ws.path_out_bytes = (size_t)RT_NOALLOC_PATHS * 24u;
    std::vector<Tile> tiles{Tile{x0, y0, w, h}};
    {
        std::lock_guard<std::mutex> turn(g_scratch_use[s->device]);  // launch_tiles returns after the stream has been synchronised
        rc = launch_tiles(s, ws, st, tex_w, tex_h, cam_w, cam_h, tiles, rgba_device, stats);
    }
    if (rc) return rc;
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
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

#include "rt_multi.hpp"
