// Device-side intersectors, stackless BVH traversal and shading.
//
// What each function computes is the reference's (citations per function; CR/ =
// CUDA_RayTracing/ in the reference tree); how it is organised is not: primitives and lights
// are wave-uniform records fetched through the scalar cache, the per-mesh BVH is walked without
// a stack through skip links, and shadow rays are spread over the lanes of the wave.
#ifndef RT_TRACE_HPP
#define RT_TRACE_HPP
#include "rt_math.hpp"

// -DRT_EXPERIMENTS puts the measurement scaffolding into the library: the render as three lean kernels (rt_defer.hpp) and, with
// -DRT_UTIL_COUNTERS on top of it, section timers and lane-use probes inside the render kernel.  tools/ab.sh builds such
// libraries for timing runs; the shipped library has neither.
#if defined(RT_UTIL_COUNTERS) && !defined(RT_EXPERIMENTS)
#error "RT_UTIL_COUNTERS instruments the render kernel: build it with -DRT_EXPERIMENTS as well (tools/ab.sh)"
#endif

// Wave-uniform tables (primitive geometry, lights) are read through the constant address space:
// with a uniform index the compiler then fetches them with scalar loads into SGPRs, which keeps
// the vector registers for per-lane ray state.
#define RT_CONST __attribute__((address_space(4)))
DEV d3 D3(const d3 RT_CONST& v) { return d3{v.x, v.y, v.z}; }  // a vector out of a record that is read in place

// Intersection data of one top-level primitive (or of one cylinder segment of a Bezier stack),
// with everything that does not depend on the ray computed once on the host: the same IEEE
// double operations the reference performs per ray (CR/Collision.cu), so the values are
// bit-identical, just not recomputed billions of times.
struct DevGeom {  // 192 bytes
    int32_t type;
    uint32_t aux0;  // mesh: BVH root ref; bezier: first segment in DevScene::segs
    int32_t aux1;   // bezier: number of segments (two more records follow them: curve frame, curve coefficients); mesh: mesh index
    int32_t aux2;   // mesh: 1 + index of its root DevNode4 if the four-wide walk may be used, else 0
    double f[22];
};
// field map of DevGeom::f
enum {
    SPH_O = 0, SPH_R2 = 3,
    PLN_N = 0, PLN_NR = 3,                                              // unit normal, unit normal * R
    SQR_O = 0, SQR_N = 3, SQR_DX = 6, SQR_DY = 9, SQR_LX = 12, SQR_LY = 13,
    CYL_O1 = 0, CYL_O2 = 3, CYL_DN = 6, CYL_LEN = 9, CYL_R2 = 10, CYL_NTOP = 11, CYL_NBOT = 14,
    TRI_V1 = 0, TRI_E1 = 3, TRI_E2 = 6, TRI_N = 9,
    MSH_SCALE = 0, MSH_BMIN = 1, MSH_BMAX = 4,                          // culling scale, root box (= the root node's box)
    BND_C = 17, BND_R = 20,                                             // bounding ball (cylinder, bezier, segments)
    CAP_P0 = 0, CAP_P1 = 3, CAP_R = 21                                  // bounding capsule of a cylinder / Bezier primitive: axis segment, radius
};

// Node of the pruned walk: the binary tree of the reference collapsed two levels at a time, so a
// node holds up to four children (the grandchildren of a binary node; a leaf child of the binary
// node stays a child) with their boxes in single precision, rounded outward.  The boxes only steer
// the walk - they decide which triangles get tested, never a hit: triangles are tested in double
// with the reference's arithmetic, and a hit is kept only if the reference's own double-precision
// box tests would have led to that triangle (see mesh_walk4).  Child refs: node4 index, or
// RT_REF_LEAF | leaf index, or RT_REF_END for an unused slot.
#ifndef RT_STACK
#define RT_STACK 32  // per-lane traversal stack entries (LDS); a mesh whose collapsed tree needs more uses the stackless walk
#endif
#define RT_STACK_ROWS RT_STACK
struct DevNode4 {    // 128 bytes
    float lo[3][4], hi[3][4];  // [axis][child]
    uint32_t child[4];
    uint32_t pad_[4];
};

// Everything shading needs to know about a primitive, indexed per lane at a hit: material, texture
// and the frame the texture coordinates are taken in, with the ray-independent parts of
// CR/Primitive.cu:16-83 computed once on the host (same IEEE operations).  The kernel keeps the table
// in LDS when the scene has at most RT_SHADE_LDS primitives.
#ifndef RT_CGRID
#define RT_CGRID 64
#endif
#ifndef RT_SHADE_LDS
#define RT_SHADE_LDS 64
#endif
struct DevShade {  // 216 bytes
    double color[3], absor[3], diff, spec, refl, refr, rindex, drefl;
    double tf[12];  // sphere: O, De, Dc, Dc x De | plane: Dx, Dy, Dx.Dx, Dy.Dy | square: O, Dx, Dy, Dx.Dx, Dy.Dy
                    // cylinder: O1, O2, unit(O2 - O1), |O2 - O1|, R
    int32_t type, texture, is_light, tex_w, tex_h;
    uint32_t tex_off;  // byte offset of the texture in the texel pool
};
enum { TF_SPH_O = 0, TF_SPH_DE = 3, TF_SPH_DC = 6, TF_SPH_X = 9, TF_PLN_DX = 0, TF_PLN_DY = 3, TF_PLN_DXX = 6, TF_PLN_DYY = 7,
       TF_SQR_O = 0, TF_SQR_DX = 3, TF_SQR_DY = 6, TF_SQR_DXX = 9, TF_SQR_DYY = 10, TF_CYL_O1 = 0, TF_CYL_O2 = 3, TF_CYL_AX = 6,
       TF_CYL_LEN = 9, TF_CYL_R = 10 };

struct DevScene {
    d3 bg_top, bg_bottom, camO, camN, camDx, camDy;
    int32_t cam_w, cam_h, shade_n, n_lights, n_prims;
    unsigned long long mesh_mask;  // bit j set: primitive j (< 64) is a mesh
    double mesh_scale_max;         // largest MSH_SCALE of the meshes that have a four-wide tree
    // Coarse answer to "which primitives can stand between a point and lamp l": a grid of RT_CGRID^3 cells over the
    // scene's bounded part, one 64-bit primitive mask per cell and lamp (null if the scene has more than 64 primitives).
    const unsigned long long* cand_grid;  // [lamp][z][y][x]
    double cg_lo[3], cg_inv[3];           // cell index = floor((p - cg_lo) * cg_inv)
    const rt_light* lights;        // read as RT_CONST
    const DevGeom* geom;           // read as RT_CONST, one per primitive
    const DevGeom* segs;           // read as RT_CONST, Bezier cylinder segments
    const DevShade* shade;         // global copy of the shading table
    const rt_bvh_node* nodes;
    const DevNode4* nodes4;        // collapsed tree of the pruned walk
    const uint32_t* node_parent;   // per rt_bvh_node: parent node (RT_REF_END at a root)
    const uint32_t* leaf_parent;   // per rt_bvh_leaf: parent node (RT_REF_END if the leaf is a root)
    const rt_bvh_leaf* leaves;
    const uint8_t* texels;
};
// the scene record is read where it lies, in the kernel-argument segment (constant address space: scalar loads)
#define SCREF const DevScene RT_CONST&
#define GEOM(sc) ((const DevGeom RT_CONST*)(sc).geom)
#define SEGS(sc) ((const DevGeom RT_CONST*)(sc).segs)
#define LIGHTS(sc) ((const rt_light RT_CONST*)(sc).lights)

struct Hit {  // CR/Primitive.cuh:107-113
    d3 N, C;
    double dist;
    int32_t prim;   // -1 = no hit
    int32_t leaf;   // leaf pre-order index inside the mesh, -1 otherwise
    bool front;
};

struct Counters {  // per-lane work counters, flushed once per wave
    uint32_t rays_secondary, rays_shadow, nodes, tris, shaded, texfetch;
    uint32_t traced_wave;  // shadow rays of the whole wave that were intersected with the scene: wave-uniform, lane 0 flushes it
#ifdef RT_UTIL_COUNTERS
    uint32_t u_shadow_slots, u_shadow_items, u_trace_slots, u_trace_active;
    uint32_t u[12];  // pairs (lane executions, 64 x wave executions): shadow mesh calls, closest mesh calls, then interior / leaf steps of each
    unsigned long long* prof;  // this wave's LDS record: [0] = time of the last lap, [1 + section] = cycles spent per section
#endif
};
// sections of the per-wave time profile (util builds)
enum { PF_REFILL = 0, PF_FRAME, PF_CLOSEST_PRIMS, PF_CLOSEST_MESH, PF_SHADE_SETUP, PF_CAND, PF_CHUNK_SHFL, PF_RNG, PF_SH_PRIMS, PF_SH_MESH,
       PF_SHADE_MATH, PF_SPAWN, PF_COUNT };
// lane-utilisation probe: counts this lane, and 64 slots once per wave-level execution
#ifdef RT_UTIL_COUNTERS
#ifdef RT_PROBE_STEPS  // all six step probes instead of the classes of traced shadow rays (they share counters)
#define RT_PROBE_KINDS 6
#else
#define RT_PROBE_KINDS 3
#endif
#define RT_PROBE(ctr, k)                                                                            \
    do {                                                                                            \
        if ((k) < RT_PROBE_KINDS) {                                                                 \
            (ctr).u[2 * (k)]++;                                                                     \
            if (lane_id() == (int)__ffsll((long long)__ballot(true)) - 1) (ctr).u[2 * (k) + 1] += 64u; \
        }                                                                                           \
    } while (0)
// RT_LAP(ctr, s): the time since the previous lap of this wave is charged to section s
#define RT_LAP(ctr, s)                                                                              \
    do {                                                                                            \
        const unsigned long long now_ = __builtin_readcyclecounter();                               \
        if (lane_id() == (int)__ffsll((long long)__ballot(true)) - 1) {                             \
            (ctr).prof[1 + (s)] += now_ - (ctr).prof[0];                                            \
            (ctr).prof[0] = now_;                                                                   \
        }                                                                                           \
    } while (0)
#else
#define RT_PROBE(ctr, k) do { } while (0)
#define RT_LAP(ctr, s) do { } while (0)
#endif

DEV d3 G3(const DevGeom RT_CONST& g, int i) { return d3{g.f[i], g.f[i + 1], g.f[i + 2]}; }

// The reference's intersectors compare squared lengths built from (ray origin - primitive), so their rounding error grows with
// the square of that distance: for a ray that starts 1e5 away from a cylinder of radius 0.1 the quadratic of CR/Collision.cu:3-107
// is off by ~1e-5 in the squared distances it compares, more than the 1e-6 relative margins of the bounding volumes, and what
// it reports for a grazing ray is noise.  Every test that predicts an intersector's answer from geometry (bounding balls and
// capsules, umbra tests) therefore keeps this much clear of the boundary: 1e-13 (~450 units of double rounding) times the squared
// distance between origin and primitive.  (Found by the randomised scenes: a ray from a wall sphere of radius 1e5 towards a lamp
// behind a thin cylinder.)
DEV double isect_noise(double dist2) { return 1e-13 * dist2; }

// Conservative pre-test: false only if the ray (unit direction V) certainly misses the ball, or
// the ball lies wholly behind the origin, or (t_max given) wholly beyond t_max.  The ball is
// inflated on the host and the comparisons keep clear of the intersectors' own rounding noise, so
// primitives that fail it report "no hit" exactly as their full test would.
DEV bool ball_may_hit(const DevGeom RT_CONST& g, d3 o, d3 V, double t_max) {
    d3 P = o - G3(g, BND_C);
    double r = g.f[BND_R];
    double b = -dot(P, V), pp = dot(P, P), r2 = r * r;
    r2 = r2 + isect_noise(pp + r2);  // the squared radius, widened by the intersectors' rounding noise at this distance
    double disc = b * b - pp + r2;
    if (disc < 0) return false;
    if (b < 0 && pp > r2) return false;
    double far = b - t_max;
    if (far > 0 && far * far > disc && pp > r2) return false;
    return true;
}

// ---- analytic primitives -------------------------------------------------------------------
// Each returns the hit distance or RT_BIG_DIST.  FULL = also produce C, N, front (closest-hit
// rays); shadow rays only need the distance.  V is the unit ray direction, computed once per
// ray by the caller (every reference intersector starts with V = normalize(direction)).

// CR/Collision.cu:155-176, CR/primitive.cpp:109-138
template <bool FULL>
DEV double sphere_hit(const DevGeom RT_CONST& g, d3 o, d3 V, Hit& h) {
    d3 O = G3(g, SPH_O), P = o - O;
    double b = -dot(P, V);
    double det = b * b - dot(P, P) + g.f[SPH_R2];
    if (!(det > RT_EPS)) return RT_BIG_DIST;
    det = sqrt(det);
    double x1 = b - det, x2 = b + det;
    if (x2 < RT_EPS) return RT_BIG_DIST;
    bool front = x1 > RT_EPS;
    double dist = front ? x1 : x2;
    if (FULL) {
        h.front = front;
        h.C = o + V * dist;
        h.N = unit(h.C - O);
        if (!front) h.N = -h.N;
    }
    return dist;
}

// CR/Collision.cu:177-196, CR/primitive.cpp:159-175
template <bool FULL>
DEV double plane_hit(const DevGeom RT_CONST& g, d3 o, d3 V, Hit& h) {
    d3 N = G3(g, PLN_N);
    double den = dot(N, V);
    if (!(fabs(den) >= RT_EPS)) return RT_BIG_DIST;
    double t = dot(G3(g, PLN_NR) - o, N) / den;
    if (!(t >= RT_EPS)) return RT_BIG_DIST;
    if (FULL) {
        h.C = o + V * t;
        h.front = den < 0;
        h.N = h.front ? N : -N;
    }
    return t;
}

// CR/Collision.cu:197-256, CR/primitive.cpp:192-227
template <bool FULL>
DEV double square_hit(const DevGeom RT_CONST& g, d3 o, d3 V, Hit& h) {
    d3 O = G3(g, SQR_O), N = G3(g, SQR_N);
    double den = dot(N, V);
    if (fabs(den) < RT_EPS) return RT_BIG_DIST;
    double t = dot(O - o, N) / den;
    if (t < RT_EPS) return RT_BIG_DIST;
    d3 P = o + V * t;
    double x2 = fabs(dot(P - O, G3(g, SQR_DX))), y2 = fabs(dot(P - O, G3(g, SQR_DY)));
    if (x2 > g.f[SQR_LX] || y2 > g.f[SQR_LY]) return RT_BIG_DIST;
    if (FULL) {
        h.front = den < 0;
        h.C = P;
        h.N = h.front ? N : -N;
    }
    return t;
}

// CR/Collision.cu:3-107.  front = false marks a body hit, true a cap hit (the texture lookup
// reuses it that way, CR/Primitive.cu:56).
template <bool FULL>
DEV double cylinder_hit(const DevGeom RT_CONST& g, d3 o, d3 V, Hit& h) {
    d3 O1 = G3(g, CYL_O1), O2 = G3(g, CYL_O2);
    {
        d3 m = o - O1, dN = G3(g, CYL_DN);
        d3 mP = cross(m, dN), vP = cross(V, dN);
        double a = dot(vP, vP), b = 2.0 * dot(vP, mP), c = dot(mP, mP) - g.f[CYL_R2];
        double det = b * b - 4 * a * c;
        if (det >= 0) {
            det = sqrt(det);
            double t0 = (-b - det) / (2.0 * a), t1 = (-b + det) / (2.0 * a);
            double t = fmin(t0, t1);
            if (t < 0) t = fmax(t0, t1);
            if (t >= 0) {
                d3 P = o + V * t;
                double u = dot(P - O1, dN);
                if (u >= 0 && u <= g.f[CYL_LEN]) {
                    if (FULL) {
                        h.C = P;
                        h.front = false;
                        h.N = unit(P - (O1 + dN * u));
                    }
                    return t;
                }
            }
        }
    }
#pragma unroll
    for (int cap = 0; cap < 2; cap++) {  // cap at O2 first, then cap at O1
        d3 N = cap == 0 ? G3(g, CYL_NTOP) : G3(g, CYL_NBOT);
        d3 Q0 = cap == 0 ? O2 : O1;
        double den = dot(N, V);
        if (fabs(den) >= RT_EPS) {
            double t = dot(Q0 - o, N) / den;
            if (t >= RT_EPS) {
                d3 P = o + V * t, Q = P - Q0;
                if (dot(Q, Q) <= g.f[CYL_R2]) {
                    if (FULL) {
                        h.C = P;
                        h.front = true;
                        h.N = den < 0 ? N : -N;
                    }
                    return t;
                }
            }
        }
    }
    return RT_BIG_DIST;
}

// CR/Collision_Bezier.cu:200-240: the profile as a stack of cylinders, including the way
// `closest` trails one accepted hit behind.
template <bool FULL>
DEV double bezier_stacked_hit(SCREF sc, const DevGeom RT_CONST& g, d3 o, d3 V, double t_max, Hit& h) {
    double closest = 1e10, cur = RT_BIG_DIST;
    for (int i = 0; i < g.aux1; i++) {
        const DevGeom RT_CONST& seg = SEGS(sc)[g.aux0 + i];
        if (!ball_may_hit(seg, o, V, t_max)) continue;
        Hit tmp;
        double t = cylinder_hit<FULL>(seg, o, V, tmp);
        if (t < RT_BIG_DIST && t < closest) {
            closest = cur;
            cur = t;
            if (FULL) h.C = tmp.C, h.N = tmp.N, h.front = tmp.front;
        }
    }
    return cur;
}

// Bezier surface of revolution proper (RT_F_BEZIER_STACKED cleared): the (z, r) profile is the
// Bezier curve of all control points, swept around the axis; the surface is open (no caps).  The
// reference only ever approximates it by the cylinder stack above (its own attempt at the curve,
// CR/Collision_Bezier.cu:3-198, is dead code), so this intersector has no reference counterpart.
// In the axis frame the ray is p + t v.  The curve parameter u is found by Newton's method from
// RT_BZ_SEEDS starting values; t comes from whichever equation is the better conditioned: for a ray
// mostly along the axis t(u) = (z(u) - pz)/vz and |(p + t v)_xy|^2 = r(u)^2 is solved; for a ray
// mostly across it t(u) is the near (then far) crossing of the cylinder of radius r(u) and
// z(u) = pz + t(u) vz is solved.  A candidate counts only if both equations hold to 1e-9.
// Records: SEGS[aux0 + aux1] holds the frame (BZN_*), the next one the power-basis coefficients
// (host-computed), both fetched with scalar loads.  Only + - * / sqrt: reproducible bit for bit.
#define RT_BZ_SEEDS 8
#define RT_BZ_ITERS 12
enum { BZN_O1 = 0, BZN_AX = 3, BZN_NX = 6, BZN_NY = 9, BZN_CZ = 0, BZN_CR = RT_BEZIER_MAX_CTRL };
DEV void bz_horner(const DevGeom RT_CONST& co, int base, int n, double u, double& val, double& der) {
    double v = co.f[base + n], dv = 0.0;
    for (int k = n - 1; k >= 0; k--) {
        dv = dv * u + v;
        v = v * u + co.f[base + k];
    }
    val = v, der = dv;
}
template <bool FULL>
DEV double bezier_newton_hit(SCREF sc, const DevGeom RT_CONST& g, d3 o, d3 V, Hit& h) {
    const DevGeom RT_CONST& fr = SEGS(sc)[g.aux0 + g.aux1];
    const DevGeom RT_CONST& co = SEGS(sc)[g.aux0 + g.aux1 + 1];
    const int n = fr.aux1;
    if (n < 1) return RT_BIG_DIST;
    const d3 ax = G3(fr, BZN_AX), Nx = G3(fr, BZN_NX), Ny = G3(fr, BZN_NY), q = o - G3(fr, BZN_O1);
    const double pz = dot(q, ax), px = dot(q, Nx), py = dot(q, Ny), vz = dot(V, ax), vx = dot(V, Nx), vy = dot(V, Ny);
    const double qa = vx * vx + vy * vy, qb = px * vx + py * vy, qc = px * px + py * py;
    const bool radial = qa > vz * vz;
    double best_t = RT_BIG_DIST, best_u = 0.0;
    for (int pass = 0; pass < (radial ? 2 : 1); pass++) {
        if (pass == 1 && best_t < RT_BIG_DIST) break;  // every near-side crossing precedes every far-side one
        const double sgn = pass == 0 ? -1.0 : 1.0;
        for (int s = 0; s < RT_BZ_SEEDS; s++) {
            double u = (s + 0.5) / RT_BZ_SEEDS, z, r, dz, dr;
            for (int it = 0; it < RT_BZ_ITERS; it++) {
                bz_horner(co, BZN_CZ, n, u, z, dz);
                bz_horner(co, BZN_CR, n, u, r, dr);
                double F, dF;
                if (radial) {
                    double disc = qb * qb - qa * (qc - r * r), sq = 0.0, dt = 0.0;
                    if (disc > 0.0) {
                        sq = sqrt(disc);
                        if (sq > 1e-12) dt = sgn * (r * dr) / sq;
                    }
                    double t = (-qb + sgn * sq) / qa;
                    F = z - pz - t * vz;
                    dF = dz - vz * dt;
                } else {
                    double t = (z - pz) / vz, qx = px + t * vx, qy = py + t * vy;
                    F = qx * qx + qy * qy - r * r;
                    dF = 2.0 * (qx * vx + qy * vy) * (dz / vz) - 2.0 * r * dr;
                }
                if (fabs(dF) < 1e-14) break;
                double du = F / dF;
                u = u - du;
                if (u < 0.0) u = 0.0;
                if (u > 1.0) u = 1.0;
                if (fabs(du) < 1e-13) break;
            }
            bz_horner(co, BZN_CZ, n, u, z, dz);
            bz_horner(co, BZN_CR, n, u, r, dr);
            double t = RT_BIG_DIST;
            if (radial) {
                double disc = qb * qb - qa * (qc - r * r);
                if (disc >= 0.0) {
                    double tt = (-qb + sgn * sqrt(disc)) / qa;
                    if (fabs(z - pz - tt * vz) <= 1e-9 * (1.0 + fabs(z) + fabs(pz))) t = tt;
                }
            } else {
                double tt = (z - pz) / vz, qx = px + tt * vx, qy = py + tt * vy;
                if (fabs(qx * qx + qy * qy - r * r) <= 1e-9 * (1.0 + r * r)) t = tt;
            }
            if (t > RT_EPS && t < best_t) best_t = t, best_u = u;
        }
    }
    if (FULL && best_t < RT_BIG_DIST) {
        double z, r, dz, dr;
        bz_horner(co, BZN_CZ, n, best_u, z, dz);
        bz_horner(co, BZN_CR, n, best_u, r, dr);
        double qx = px + best_t * vx, qy = py + best_t * vy;
        d3 erho = unit(Nx * qx + Ny * qy);
        d3 Nn = unit(erho * dz - ax * dr);
        h.C = o + V * best_t;
        h.front = dot(V, Nn) < 0;
        h.N = h.front ? Nn : -Nn;
    }
    return best_t;
}

// CR/Collision.cu:110-149,267-282 (standalone triangle; the normal is not flipped; d is the
// direction as given or normalised, see RT_F_MESH_RAW_DIR)
template <bool FULL>
DEV double triangle_hit(const DevGeom RT_CONST& g, d3 o, d3 d, Hit& h) {
    d3 V1 = G3(g, TRI_V1), e1 = G3(g, TRI_E1), e2 = G3(g, TRI_E2);
    d3 hh = cross(d, e2);
    double a = dot(e1, hh);
    if (fabs(a) < RT_EPS) return RT_BIG_DIST;
    double f = 1.0 / a;
    d3 s = o - V1;
    double u = f * dot(s, hh);
    if (u < 0.0 || u > 1.0) return RT_BIG_DIST;
    d3 q = cross(s, e1);
    double v = f * dot(d, q);
    if (v < 0.0 || u + v > 1.0) return RT_BIG_DIST;
    double t = f * dot(e2, q);
    if (!(t > RT_EPS)) return RT_BIG_DIST;
    if (FULL) {
        h.C = o + d * t;
        h.N = G3(g, TRI_N);
        h.front = dot(d, h.N) < 0;
    }
    return t;
}

// ---- mesh: stackless BVH walk ----------------------------------------------------------------
// CR/Collision_Mesh.cu:3-17 slab test, :19-63 Moeller-Trumbore, :65-80 depth-first order
// (left subtree, then right subtree, both always).  The reference recurses; here every item
// carries the ref of the next item in pre-order after its subtree ("skip"), so "box missed"
// and "leaf done" are one pointer step and no stack exists.
//
// STRICT walks exactly the reference's sequence.  Otherwise two prunings apply, neither of
// which can change the result:
//   * a subtree whose box is entered beyond `t_limit` (the distance that still matters: the
//     best hit so far for closest-hit rays, the light distance for shadow rays) or lies wholly
//     behind the ray origin is skipped - every triangle inside is farther than t_limit or
//     behind.  The slab distances carry rounding error, so the comparison is relaxed by
//     `margin`, a bound on that error for this ray;
//   * ANYHIT stops at the first triangle closer than t_limit (a shadow ray only asks whether
//     one exists).
// Ties keep the reference's winner: leaves are met in the same pre-order and a later leaf
// replaces the best only if strictly closer (CR/Collision_Mesh.cu:71).
struct MeshResult {
    double t;       // closest accepted t (RT_BIG_DIST if none)
    uint32_t leaf;  // ref of the winning leaf
};

template <bool STRICT, bool ANYHIT, bool LOG>
DEV MeshResult mesh_walk(SCREF sc, uint32_t root, d3 o, d3 d, double t_limit, double cull_scale, Counters& ctr,
                         uint32_t* log, uint32_t log_cap, uint32_t& log_n, uint32_t log_tag) {
    MeshResult res{RT_BIG_DIST, RT_REF_END};
    const double t_exact = t_limit;  // ANYHIT: the caller's occlusion test is t_exact - t > 1e-6
    d3 inv = D3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    double margin = 0.0;
    if (!STRICT) {
        // |error| of a slab distance <= ~4 ulp * (|box| + |o|) * |1/d|; infinite 1/d (an exactly
        // axis-parallel ray) yields exact +-inf / NaN slab values and needs no margin.
        double m = 0.0;
        if (fabs(inv.x) < 1e300) m = fmax(m, fabs(inv.x));
        if (fabs(inv.y) < 1e300) m = fmax(m, fabs(inv.y));
        if (fabs(inv.z) < 1e300) m = fmax(m, fabs(inv.z));
        margin = 1e-13 * (cull_scale + fabs(o.x) + fabs(o.y) + fabs(o.z)) * m;
        t_limit = t_limit + margin + 1e-9 * fabs(t_limit);
    }
    uint32_t ref = root;
    while (ref != RT_REF_END) {
        // interior items until every lane that is still walking stands on a leaf
        while (!(ref & RT_REF_LEAF)) {
            const rt_bvh_node& n = sc.nodes[ref];
            ctr.nodes++;
            if (LOG) {
                if (log_n < log_cap) log[log_n] = log_tag | n.preorder;
                log_n++;
            }
            double t0x = (n.bmin[0] - o.x) * inv.x, t1x = (n.bmax[0] - o.x) * inv.x;
            double t0y = (n.bmin[1] - o.y) * inv.y, t1y = (n.bmax[1] - o.y) * inv.y;
            double t0z = (n.bmin[2] - o.z) * inv.z, t1z = (n.bmax[2] - o.z) * inv.z;
            double tminmax = fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmin(t0z, t1z));
            double tmaxmin = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmax(t0z, t1z));
            bool enter = tminmax <= tmaxmin;
            if (!STRICT) enter = enter && !(tminmax > t_limit) && !(tmaxmin < -margin);
            ref = enter ? n.left : n.skip;
        }
        if (ref == RT_REF_END) break;
        const rt_bvh_leaf& lf = sc.leaves[ref & ~RT_REF_LEAF];
        ctr.nodes++;
        ctr.tris++;
        if (LOG) {
            if (log_n < log_cap) log[log_n] = log_tag | lf.preorder;
            log_n++;
        }
        uint32_t here = ref;
        ref = lf.skip;
        d3 O1 = D3(lf.O1), e1 = D3(lf.O2) - O1, e2 = D3(lf.O3) - O1;
        d3 p = cross(d, e2);
        double det = dot(e1, p);
        if (fabs(det) < RT_EPS) continue;
        double idet = 1.0 / det;
        d3 tv = o - O1;
        double u = dot(tv, p) * idet;
        if (u < 0.0 || u > 1.0) continue;
        d3 q = cross(tv, e1);
        double v = dot(d, q) * idet;
        if (v < 0.0 || u + v > 1.0) continue;
        double t0 = dot(e2, q) * idet;
        if (t0 <= RT_EPS) continue;
        if (t0 < res.t) {
            res.t = t0;
            res.leaf = here;
            if (!STRICT) {
                if (ANYHIT) {
                    if (t_exact - t0 > RT_EPS) break;  // this triangle already decides the caller's test
                } else {
                    double lim = t0 + margin + 1e-9 * fabs(t0);
                    if (lim < t_limit) t_limit = lim;
                }
            }
        }
    }
    return res;
}

// Would the reference's depth-first walk (CR/Collision_Mesh.cu:65-80) have reached this leaf?  It does
// iff the slab test of every ancestor passes; same arithmetic as mesh_walk<true, ...>.
DEV bool reference_reaches_leaf(SCREF sc, uint32_t leaf_index, d3 o, d3 inv) {
    uint32_t n = sc.leaf_parent[leaf_index];
    while (n != RT_REF_END) {
        const rt_bvh_node& nd = sc.nodes[n];
        double t0x = (nd.bmin[0] - o.x) * inv.x, t1x = (nd.bmax[0] - o.x) * inv.x;
        double t0y = (nd.bmin[1] - o.y) * inv.y, t1y = (nd.bmax[1] - o.y) * inv.y;
        double t0z = (nd.bmin[2] - o.z) * inv.z, t1z = (nd.bmax[2] - o.z) * inv.z;
        double tminmax = fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmin(t0z, t1z));
        double tmaxmin = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmax(t0z, t1z));
        if (!(tminmax <= tmaxmin)) return false;
        n = sc.node_parent[n];
    }
    return true;
}

// Pruned walk over the four-wide tree with a per-lane stack in LDS (`stk` points at this lane's
// column; entry k lives at stk[k * 64]).  It returns exactly what the reference-order walk returns:
//   * the single-precision boxes are supersets of the reference's boxes and every comparison is
//     relaxed by M, a bound on the rounding error of the slab distances of this ray, so a child is
//     skipped only if the ray misses the reference's box, or the box lies wholly behind the origin,
//     or wholly beyond t_limit - no triangle the reference would accept within t_limit is lost;
//   * the superset may reach a triangle the reference never tests, because one of its own box tests
//     fails by a rounding error.  The host has checked that every box of this mesh contains the boxes
//     below it, so a hit point well inside its triangle's bounds is well inside every ancestor box
//     and those tests pass; only a hit within a hair of its triangle's bounds re-runs the ancestors'
//     tests in double (reference_reaches_leaf);
//   * ties go to the smaller pre-order index, which is the leaf the reference meets first.
// Rays whose slab distances do not fit single precision take the stackless double-precision walk.
#ifndef RT_ANYHIT_SORTED
#define RT_ANYHIT_SORTED 0
#endif
// Closest-hit walks pick each child's entry and exit planes by the sign of the ray direction (one fused multiply-add and
// a three-way max / min per child instead of min / max per axis).  The any-hit walk keeps the min / max form: it sits at
// the register peak of the kernel, and the six plane offsets cost more in spills there than they save (measured on
// c4_lucy: both 47.1, any-hit only 46.6, closest only 44.5, neither 45.2 ms/frame).
#ifndef RT_SIGN_SELECT
#define RT_SIGN_SELECT(anyhit) (!(anyhit))
#endif
template <bool ANYHIT>
DEV MeshResult mesh_walk4(SCREF sc, const DevGeom RT_CONST& g, d3 o, d3 d, double t_limit, Counters& ctr, uint32_t* stk) {
    const d3 inv = D3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    const double S = g.f[MSH_SCALE];
    const double mm = fmax(fmax((S + fabs(o.x)) * fabs(inv.x), (S + fabs(o.y)) * fabs(inv.y)), (S + fabs(o.z)) * fabs(inv.z));
    if (!(mm < 1e30) || !(fabs(inv.x) < 1e30) || !(fabs(inv.y) < 1e30) || !(fabs(inv.z) < 1e30)) {
        uint32_t dummy = 0;
        return mesh_walk<false, ANYHIT, false>(sc, g.aux0, o, d, t_limit, S, ctr, nullptr, 0, dummy, 0);
    }
    MeshResult res{RT_BIG_DIST, RT_REF_END};
    {   // root box (CR/Renderer_CUDA.cpp:234-235: the vertex bounds) from the scalar registers: the reference's
        // own test, plus "wholly beyond t_limit" and "wholly behind", relaxed by the rounding error of the distances
        ctr.nodes++;
        const double margin = 1e-13 * mm;
        double t0x = (g.f[MSH_BMIN] - o.x) * inv.x, t1x = (g.f[MSH_BMAX] - o.x) * inv.x;
        double t0y = (g.f[MSH_BMIN + 1] - o.y) * inv.y, t1y = (g.f[MSH_BMAX + 1] - o.y) * inv.y;
        double t0z = (g.f[MSH_BMIN + 2] - o.z) * inv.z, t1z = (g.f[MSH_BMAX + 2] - o.z) * inv.z;
        double tn = fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmin(t0z, t1z));
        double tf = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmax(t0z, t1z));
        if (!(tn <= tf) || tn > t_limit + margin + 1e-9 * fabs(t_limit) || tf < -margin) return res;
    }
    uint32_t res_pre = 0xFFFFFFFFu;
    const double t_exact = t_limit;
    // slab distance of plane p on axis k: fma(p, ik, ck); |error| <= 2^-22 (S + |o_k|) |1/d_k| <= M
    const float ix = (float)inv.x, iy = (float)inv.y, iz = (float)inv.z;
    const float cx = (float)(-(o.x * inv.x)), cy = (float)(-(o.y * inv.y)), cz = (float)(-(o.z * inv.z));
    const float M = (float)(mm * 2.4e-7) * 1.000001f + 1e-30f;
    float tl = (float)t_limit;  // +inf if beyond single precision
    tl = tl + fabsf(tl) * 2.4e-7f;
    RT_PROBE(ctr, ANYHIT ? 0 : 1);
    uint32_t ref = (uint32_t)(g.aux2 - 1);
    int sp = 0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    static_assert(offsetof(DevNode4, hi) == 48 && offsetof(DevNode4, child) == 96, "node layout");
    const char* nbase = reinterpret_cast<const char*>(sc.nodes4);
    const uint32_t onx = ix >= 0.0f ? 0u : 48u, ony = (iy >= 0.0f ? 0u : 48u) + 16u, onz = (iz >= 0.0f ? 0u : 48u) + 32u;
    const uint32_t ofx = 48u - onx, ofy = 80u - ony, ofz = 112u - onz;
    for (;;) {
        while (!(ref & RT_REF_LEAF)) {
            ctr.nodes++;
            RT_PROBE(ctr, ANYHIT ? 2 : 4);
            float key[4];
            uint32_t rf[4];
            if (RT_SIGN_SELECT(ANYHIT)) {
            // the planes the ray enters through (lo where it runs up the axis, hi where it runs down) and leaves through
                const uint32_t nb = ref * (uint32_t)sizeof(DevNode4);
                const f4 nx = *reinterpret_cast<const f4*>(nbase + (nb + onx)), ny = *reinterpret_cast<const f4*>(nbase + (nb + ony)),
                         nz = *reinterpret_cast<const f4*>(nbase + (nb + onz));
                const f4 fx = *reinterpret_cast<const f4*>(nbase + (nb + ofx)), fy = *reinterpret_cast<const f4*>(nbase + (nb + ofy)),
                         fz = *reinterpret_cast<const f4*>(nbase + (nb + ofz));
                const u4 ch = *reinterpret_cast<const u4*>(nbase + (nb + 96u));
    #pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float a = fmaxf(fmaxf(__builtin_fmaf(nx[c], ix, cx), __builtin_fmaf(ny[c], iy, cy)), __builtin_fmaf(nz[c], iz, cz)) - M;
                    const float b = fminf(fminf(__builtin_fmaf(fx[c], ix, cx), __builtin_fmaf(fy[c], iy, cy)), __builtin_fmaf(fz[c], iz, cz)) + M;
                    rf[c] = ch[c];
                    const bool ok = rf[c] != RT_REF_END && a <= b && !(a > tl) && !(b < 0.0f);
                    key[c] = ok ? a : __builtin_inff();
                }
            } else {
                const DevNode4& n = sc.nodes4[ref];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float t0x = __builtin_fmaf(n.lo[0][c], ix, cx), t1x = __builtin_fmaf(n.hi[0][c], ix, cx);
                    const float t0y = __builtin_fmaf(n.lo[1][c], iy, cy), t1y = __builtin_fmaf(n.hi[1][c], iy, cy);
                    const float t0z = __builtin_fmaf(n.lo[2][c], iz, cz), t1z = __builtin_fmaf(n.hi[2][c], iz, cz);
                    const float a = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)) - M;
                    const float b = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)) + M;
                    rf[c] = n.child[c];
                    const bool ok = rf[c] != RT_REF_END && a <= b && !(a > tl) && !(b < 0.0f);
                    key[c] = ok ? a : __builtin_inff();
                }
            }
            if (!ANYHIT || RT_ANYHIT_SORTED) {  // nearest first
#define RT_CE(i, j)                                                  \
    {                                                                \
        const bool sw = key[j] < key[i];                             \
        const float ka = sw ? key[j] : key[i], kb = sw ? key[i] : key[j]; \
        const uint32_t ra = sw ? rf[j] : rf[i], rb = sw ? rf[i] : rf[j];  \
        key[i] = ka, key[j] = kb, rf[i] = ra, rf[j] = rb;            \
    }
                RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#undef RT_CE
                if (key[3] < __builtin_inff() && sp < RT_STACK) stk[sp * 64] = rf[3], sp++;
                if (key[2] < __builtin_inff() && sp < RT_STACK) stk[sp * 64] = rf[2], sp++;
                if (key[1] < __builtin_inff() && sp < RT_STACK) stk[sp * 64] = rf[1], sp++;
                if (key[0] < __builtin_inff()) ref = rf[0];
                else ref = sp > 0 ? stk[(--sp) * 64] : RT_REF_END;
            } else {
                uint32_t next = RT_REF_END;
                bool have = false;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (key[c] < __builtin_inff()) {
                        if (!have) next = rf[c], have = true;
                        else if (sp < RT_STACK) stk[sp * 64] = rf[c], sp++;
                    }
                }
                ref = have ? next : (sp > 0 ? stk[(--sp) * 64] : RT_REF_END);
            }
        }
        if (ref == RT_REF_END) break;
        const uint32_t here = ref, leaf_index = ref & ~RT_REF_LEAF;
        const rt_bvh_leaf& lf = sc.leaves[leaf_index];
        ctr.tris++;
        RT_PROBE(ctr, ANYHIT ? 3 : 5);
        ref = sp > 0 ? stk[(--sp) * 64] : RT_REF_END;
        const d3 O1 = D3(lf.O1), O2 = D3(lf.O2), O3 = D3(lf.O3), e1 = O2 - O1, e2 = O3 - O1;
        d3 p = cross(d, e2);
        double det = dot(e1, p);
        if (fabs(det) < RT_EPS) continue;
        double idet = 1.0 / det;
        d3 tv = o - O1;
        double u = dot(tv, p) * idet;
        if (u < 0.0 || u > 1.0) continue;
        d3 q = cross(tv, e1);
        double v = dot(d, q) * idet;
        if (v < 0.0 || u + v > 1.0) continue;
        double t0 = dot(e2, q) * idet;
        if (t0 <= RT_EPS) continue;
        const bool wanted = ANYHIT ? (t_exact - t0 > RT_EPS) : (t0 < res.t || (t0 == res.t && lf.preorder < res_pre));
        if (!wanted) continue;
        {   // the reference must have reached this leaf
            const d3 P = o + d * t0;
            const double eps = 1e-9 * (S + fabs(o.x) + fabs(o.y) + fabs(o.z) + fabs(P.x) + fabs(P.y) + fabs(P.z));
            const bool inside = P.x - fmin(O1.x, fmin(O2.x, O3.x)) > eps && fmax(O1.x, fmax(O2.x, O3.x)) - P.x > eps &&
                                P.y - fmin(O1.y, fmin(O2.y, O3.y)) > eps && fmax(O1.y, fmax(O2.y, O3.y)) - P.y > eps &&
                                P.z - fmin(O1.z, fmin(O2.z, O3.z)) > eps && fmax(O1.z, fmax(O2.z, O3.z)) - P.z > eps;
            if (!inside && !reference_reaches_leaf(sc, leaf_index, o, inv)) continue;
        }
        res.t = t0;
        res.leaf = here;
        res_pre = lf.preorder;
        if (ANYHIT) break;
        tl = (float)t0;
        tl = tl + fabsf(tl) * 2.4e-7f;
    }
    return res;
}

// ---- any-hit walk shared by the wave ------------------------------------------------------------
// Shadow rays of one wave end their walks at very different times, and a lane that is done would idle until the
// longest walk of the wave ends.  Here an idle lane takes over part of a busy lane's walk instead: the busy lane
// gives away the subtree on top of its stack, the idle lane copies the ray and walks that subtree.  Whoever finds
// an occluding triangle raises the ray's flag; the answer for a ray is its flag, whichever lanes worked on it.
// The set of subtrees walked for a ray is the same as in mesh_walk4<true> (less what a raised flag cuts short), and
// every triangle goes through the same tests, so the answer is the same.
// Must be called by all lanes of the wave that are active at the call site; `want` = this lane has a ray to test.
// `xch` = 3 * 64 words of LDS of this wave: flags, and two exchange arrays.
// The shared any-hit walk picks entry and exit planes by the sign of the ray direction as well, with the six plane offsets
// derived from the sign bits once per round of steps instead of being kept in registers (headline 34.5 -> 34.4 ms/frame,
// c5_final at 32 spp 117.6 -> 115.8; 0: the min / max form).
#ifndef RT_ANYHIT_SHARED_SIGN
#define RT_ANYHIT_SHARED_SIGN 1
#endif
#ifndef RT_STEAL_MIN
#define RT_STEAL_MIN 8  // idle lanes that make a round of hand-overs worth its cost (c5_final ms/frame: 4: 201, 8: 201, 12: 205, 16: 211)
#endif
#ifndef RT_SHARE_STEPS
#define RT_SHARE_STEPS 2  // interior steps between two looks at who is idle (1: 207, 2: 201, 4: 211)
#endif
// Mesh roots a ray may put on its stack before the shared walk starts.  A four-wide tree is at most RT_STACK / 3 levels
// deep and a walk pushes up to three siblings per level, so the walk of one tree needs 3 * (RT_STACK / 3) entries; the
// roots that wait below it must fit in what is left, or a push would be dropped and an occluder missed.  Further
// candidate meshes of the same ray are walked by the lane on its own.
#define RT_SHARED_ROOTS (RT_STACK - 3 * (RT_STACK / 3) + 1)  // the root being walked has left the stack
static_assert(RT_SHARED_ROOTS >= 1 && (RT_SHARED_ROOTS - 1) + 3 * (RT_STACK / 3) <= RT_STACK, "roots + one tree walk fit the stack");
// On entry the lane's stack (sp entries) holds the roots of the meshes its ray may enter; `S` = the largest mesh scale
// of the scene (it only widens the error bounds).  The ray is passed by value and overwritten by hand-overs.
DEV bool mesh_anyhit_shared(SCREF sc, double S, int sp, d3 o, d3 d, double t_limit, Counters& ctr, uint32_t* stk, uint32_t* xch) {
    const int lane = lane_id();
    uint32_t* flag = xch;
    uint32_t* x_lane = xch + 64;
    uint32_t* x_ref = xch + 128;
    const bool enter = sp > 0;
    if (__ballot(enter) == 0ull) return false;
    flag[lane] = 0u;
    // per-ray constants of the single-precision slab test (see mesh_walk4)
    float ix, iy, iz, cx, cy, cz, M, tl;
    {
        const d3 inv = D3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
        ix = (float)inv.x, iy = (float)inv.y, iz = (float)inv.z;
        cx = (float)(-(o.x * inv.x)), cy = (float)(-(o.y * inv.y)), cz = (float)(-(o.z * inv.z));
        const double mm = fmax(fmax((S + fabs(o.x)) * fabs(inv.x), (S + fabs(o.y)) * fabs(inv.y)), (S + fabs(o.z)) * fabs(inv.z));
        M = (float)(mm * 2.4e-7) * 1.000001f + 1e-30f;
        tl = (float)t_limit;
        tl = tl + fabsf(tl) * 2.4e-7f;
    }
    uint32_t ref = enter ? stk[(--sp) * 64] : RT_REF_END;
    int ray = lane;  // the lane whose ray this lane is working on
    RT_PROBE(ctr, 0);
    for (;;) {
        // a ray that somebody has found occluded needs no more work
        if (ref != RT_REF_END && flag[ray] != 0u) ref = RT_REF_END, sp = 0;
        const unsigned long long busy = __ballot(ref != RT_REF_END);
        if (busy == 0ull) break;
        // ---- hand-overs: the k-th idle lane takes the top of the stack of the k-th lane that has one ----
        const unsigned long long givers = __ballot(ref != RT_REF_END && sp > 0);
        const unsigned long long idlers = __ballot(ref == RT_REF_END);  // of the lanes that are here: some may not have been called
        const int n_idle = __popcll(idlers), n_give = __popcll(givers);
        if (n_idle >= RT_STEAL_MIN && n_give > 0) {
            const bool giver = ref != RT_REF_END && sp > 0;
            if (giver) {
                const int k = rank_in(givers);
                if (k < n_idle) {
                    x_lane[k] = (uint32_t)lane;
                    x_ref[k] = stk[(--sp) * 64];
                }
            }
            const bool idle = ref == RT_REF_END;
            const int k = rank_in(idlers);
            const bool take = idle && k < n_give;
            const int from = take ? (int)x_lane[k] : lane;
            // the ray, from the lane that gives
            o = shfl_d3(o, from), d = shfl_d3(d, from);
            t_limit = shfl_d(t_limit, from);
            ix = __shfl(ix, from, 64), iy = __shfl(iy, from, 64), iz = __shfl(iz, from, 64);
            cx = __shfl(cx, from, 64), cy = __shfl(cy, from, 64), cz = __shfl(cz, from, 64);
            M = __shfl(M, from, 64), tl = __shfl(tl, from, 64);
            ray = __shfl(ray, from, 64);
            if (take) ref = x_ref[k];
        }
        // ---- a few interior steps ----
#if RT_ANYHIT_SHARED_SIGN
        // the planes the ray enters through (lo where it runs up the axis, hi where it runs down) and leaves through
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const char* nbase = reinterpret_cast<const char*>(sc.nodes4);
        const uint32_t onx = (uint32_t)(__float_as_int(ix) >> 31) & 48u, ony = ((uint32_t)(__float_as_int(iy) >> 31) & 48u) + 16u,
                       onz = ((uint32_t)(__float_as_int(iz) >> 31) & 48u) + 32u;
        const uint32_t ofx = 48u - onx, ofy = 80u - ony, ofz = 112u - onz;
#endif
#pragma unroll 1
        for (int step = 0; step < RT_SHARE_STEPS; step++) {
            const bool interior = ref != RT_REF_END && !(ref & RT_REF_LEAF);
            if (__ballot(interior) == 0ull) break;
            if (interior) {
                ctr.nodes++;
                RT_PROBE(ctr, 2);
                // all four boxes first (one batch of loads), then the choice of where to go
                bool ok[4];
                uint32_t rf[4];
#if RT_ANYHIT_SHARED_SIGN
                const uint32_t nb = ref * (uint32_t)sizeof(DevNode4);
                const f4 nx = *reinterpret_cast<const f4*>(nbase + (nb + onx)), ny = *reinterpret_cast<const f4*>(nbase + (nb + ony)),
                         nz = *reinterpret_cast<const f4*>(nbase + (nb + onz));
                const f4 fx = *reinterpret_cast<const f4*>(nbase + (nb + ofx)), fy = *reinterpret_cast<const f4*>(nbase + (nb + ofy)),
                         fz = *reinterpret_cast<const f4*>(nbase + (nb + ofz));
                const u4 ch = *reinterpret_cast<const u4*>(nbase + (nb + 96u));
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float a = fmaxf(fmaxf(__builtin_fmaf(nx[c], ix, cx), __builtin_fmaf(ny[c], iy, cy)), __builtin_fmaf(nz[c], iz, cz)) - M;
                    const float b = fminf(fminf(__builtin_fmaf(fx[c], ix, cx), __builtin_fmaf(fy[c], iy, cy)), __builtin_fmaf(fz[c], iz, cz)) + M;
                    rf[c] = ch[c];
                    ok[c] = rf[c] != RT_REF_END && a <= b && !(a > tl) && !(b < 0.0f);
                }
#else
                const DevNode4& n = sc.nodes4[ref];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float t0x = __builtin_fmaf(n.lo[0][c], ix, cx), t1x = __builtin_fmaf(n.hi[0][c], ix, cx);
                    const float t0y = __builtin_fmaf(n.lo[1][c], iy, cy), t1y = __builtin_fmaf(n.hi[1][c], iy, cy);
                    const float t0z = __builtin_fmaf(n.lo[2][c], iz, cz), t1z = __builtin_fmaf(n.hi[2][c], iz, cz);
                    const float a = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)) - M;
                    const float b = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)) + M;
                    rf[c] = n.child[c];
                    ok[c] = rf[c] != RT_REF_END && a <= b && !(a > tl) && !(b < 0.0f);
                }
#endif
                uint32_t next = RT_REF_END;
                bool have = false;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (ok[c]) {
                        if (!have) next = rf[c], have = true;
                        else if (sp < RT_STACK) stk[sp * 64] = rf[c], sp++;
                    }
                }
                ref = have ? next : (sp > 0 ? stk[(--sp) * 64] : RT_REF_END);
            }
        }
        // ---- one triangle for the lanes that stand at a leaf ----
        if (ref != RT_REF_END && (ref & RT_REF_LEAF)) {
            const uint32_t leaf_index = ref & ~RT_REF_LEAF;
            const rt_bvh_leaf& lf = sc.leaves[leaf_index];
            ctr.tris++;
            RT_PROBE(ctr, 3);
            ref = sp > 0 ? stk[(--sp) * 64] : RT_REF_END;
            const d3 O1 = D3(lf.O1), O2 = D3(lf.O2), O3 = D3(lf.O3), e1 = O2 - O1, e2 = O3 - O1;
            bool hit = false;
            d3 p = cross(d, e2);
            double det = dot(e1, p);
            if (!(fabs(det) < RT_EPS)) {
                double idet = 1.0 / det;
                d3 tv = o - O1;
                double u = dot(tv, p) * idet;
                if (!(u < 0.0 || u > 1.0)) {
                    d3 q = cross(tv, e1);
                    double v = dot(d, q) * idet;
                    if (!(v < 0.0 || u + v > 1.0)) {
                        double t0 = dot(e2, q) * idet;
                        if (!(t0 <= RT_EPS) && t_limit - t0 > RT_EPS) {
                            // the reference must have reached this leaf (see mesh_walk4)
                            const d3 P = o + d * t0;
                            const double eps = 1e-9 * (S + fabs(o.x) + fabs(o.y) + fabs(o.z) + fabs(P.x) + fabs(P.y) + fabs(P.z));
                            const bool inside = P.x - fmin(O1.x, fmin(O2.x, O3.x)) > eps && fmax(O1.x, fmax(O2.x, O3.x)) - P.x > eps &&
                                                P.y - fmin(O1.y, fmin(O2.y, O3.y)) > eps && fmax(O1.y, fmax(O2.y, O3.y)) - P.y > eps &&
                                                P.z - fmin(O1.z, fmin(O2.z, O3.z)) > eps && fmax(O1.z, fmax(O2.z, O3.z)) - P.z > eps;
                            hit = inside || reference_reaches_leaf(sc, leaf_index, o, D3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z));
                        }
                    }
                }
            }
            if (hit) {
                flag[ray] = 1u;
                ref = RT_REF_END, sp = 0;
            }
        }
    }
    return enter && flag[lane] != 0u;
}

// ---- closest-hit walk shared by the wave ----------------------------------------------------------
// The same hand-overs as mesh_anyhit_shared.  Every lane that works on a ray prunes with the ray's best distance so far,
// kept in LDS, and an accepted triangle is written to the ray's record by one lane at a time (distance, then pre-order
// index: the rule of mesh_walk4), so the record ends up holding what a single lane would have found.
// Must be called by all lanes that are active at the call site.  `xch` = 448 words of LDS of this wave.
DEV MeshResult mesh_closest_shared(SCREF sc, const DevGeom RT_CONST& g, d3 o, d3 d, double t_limit, Counters& ctr, uint32_t* stk,
                                   uint32_t* xch) {
    const int lane = lane_id();
    uint32_t* x_lane = xch + 64;
    uint32_t* x_ref = xch + 128;
    double* rec_t = reinterpret_cast<double*>(xch + 192);
    uint32_t* rec_pre = xch + 320;
    uint32_t* rec_leaf = xch + 384;
    const double S = g.f[MSH_SCALE];
    MeshResult res{RT_BIG_DIST, RT_REF_END};
    bool enter = true;
    float ix, iy, iz, cx, cy, cz, M, tl;
    {
        const d3 inv = D3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
        const double mm = fmax(fmax((S + fabs(o.x)) * fabs(inv.x), (S + fabs(o.y)) * fabs(inv.y)), (S + fabs(o.z)) * fabs(inv.z));
        if (!(mm < 1e30) || !(fabs(inv.x) < 1e30) || !(fabs(inv.y) < 1e30) || !(fabs(inv.z) < 1e30)) {
            uint32_t dummy = 0;  // slab distances beyond single precision: this ray walks on its own, in double
            res = mesh_walk<false, false, false>(sc, g.aux0, o, d, t_limit, S, ctr, nullptr, 0, dummy, 0);
            enter = false;
        } else {
            ctr.nodes++;
            const double margin = 1e-13 * mm;
            double t0x = (g.f[MSH_BMIN] - o.x) * inv.x, t1x = (g.f[MSH_BMAX] - o.x) * inv.x;
            double t0y = (g.f[MSH_BMIN + 1] - o.y) * inv.y, t1y = (g.f[MSH_BMAX + 1] - o.y) * inv.y;
            double t0z = (g.f[MSH_BMIN + 2] - o.z) * inv.z, t1z = (g.f[MSH_BMAX + 2] - o.z) * inv.z;
            double tn = fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmin(t0z, t1z));
            double tf = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmax(t0z, t1z));
            if (!(tn <= tf) || tn > t_limit + margin + 1e-9 * fabs(t_limit) || tf < -margin) enter = false;
        }
        ix = (float)inv.x, iy = (float)inv.y, iz = (float)inv.z;
        cx = (float)(-(o.x * inv.x)), cy = (float)(-(o.y * inv.y)), cz = (float)(-(o.z * inv.z));
        M = (float)(mm * 2.4e-7) * 1.000001f + 1e-30f;
        tl = (float)t_limit;
        tl = tl + fabsf(tl) * 2.4e-7f;
    }
    if (__ballot(enter) == 0ull) return res;
    rec_t[lane] = RT_BIG_DIST, rec_pre[lane] = 0xFFFFFFFFu, rec_leaf[lane] = RT_REF_END;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const char* nbase = reinterpret_cast<const char*>(sc.nodes4);
    uint32_t ref = enter ? (uint32_t)(g.aux2 - 1) : RT_REF_END;
    int sp = 0;
    int ray = lane;  // the lane whose ray this lane is working on
    RT_PROBE(ctr, 1);
    for (;;) {
        if (ref != RT_REF_END) {  // prune with what anybody has found for this ray
            const float best = (float)rec_t[ray];
            tl = fminf(tl, best + fabsf(best) * 2.4e-7f);
        }
        const unsigned long long busy = __ballot(ref != RT_REF_END);
        if (busy == 0ull) break;
        // ---- hand-overs: the k-th idle lane takes the top of the stack of the k-th lane that has one ----
        const unsigned long long givers = __ballot(ref != RT_REF_END && sp > 0);
        const unsigned long long idlers = __ballot(ref == RT_REF_END);  // of the lanes that are here: some may not have been called
        const int n_idle = __popcll(idlers), n_give = __popcll(givers);
        if (n_idle >= RT_STEAL_MIN && n_give > 0) {
            const bool giver = ref != RT_REF_END && sp > 0;
            if (giver) {
                const int k = rank_in(givers);
                if (k < n_idle) {
                    x_lane[k] = (uint32_t)lane;
                    x_ref[k] = stk[(--sp) * 64];
                }
            }
            const bool idle = ref == RT_REF_END;
            const int k = rank_in(idlers);
            const bool take = idle && k < n_give;
            const int from = take ? (int)x_lane[k] : lane;
            o = shfl_d3(o, from), d = shfl_d3(d, from);
            ix = __shfl(ix, from, 64), iy = __shfl(iy, from, 64), iz = __shfl(iz, from, 64);
            cx = __shfl(cx, from, 64), cy = __shfl(cy, from, 64), cz = __shfl(cz, from, 64);
            M = __shfl(M, from, 64), tl = __shfl(tl, from, 64);
            ray = __shfl(ray, from, 64);
            if (take) ref = x_ref[k];
        }
        // the planes the ray enters through (lo where it runs up the axis, hi where it runs down) and leaves through
        const uint32_t onx = (uint32_t)(__float_as_int(ix) >> 31) & 48u, ony = ((uint32_t)(__float_as_int(iy) >> 31) & 48u) + 16u,
                       onz = ((uint32_t)(__float_as_int(iz) >> 31) & 48u) + 32u;
        const uint32_t ofx = 48u - onx, ofy = 80u - ony, ofz = 112u - onz;
        // ---- a few interior steps, nearest child first ----
#pragma unroll 1
        for (int step = 0; step < RT_SHARE_STEPS; step++) {
            const bool interior = ref != RT_REF_END && !(ref & RT_REF_LEAF);
            if (__ballot(interior) == 0ull) break;
            if (interior) {
                ctr.nodes++;
                RT_PROBE(ctr, 4);
                float key[4];
                uint32_t rf[4];
                const uint32_t nb = ref * (uint32_t)sizeof(DevNode4);
                const f4 nx = *reinterpret_cast<const f4*>(nbase + (nb + onx)), ny = *reinterpret_cast<const f4*>(nbase + (nb + ony)),
                         nz = *reinterpret_cast<const f4*>(nbase + (nb + onz));
                const f4 fx = *reinterpret_cast<const f4*>(nbase + (nb + ofx)), fy = *reinterpret_cast<const f4*>(nbase + (nb + ofy)),
                         fz = *reinterpret_cast<const f4*>(nbase + (nb + ofz));
                const u4 ch = *reinterpret_cast<const u4*>(nbase + (nb + 96u));
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float a = fmaxf(fmaxf(__builtin_fmaf(nx[c], ix, cx), __builtin_fmaf(ny[c], iy, cy)), __builtin_fmaf(nz[c], iz, cz)) - M;
                    const float b = fminf(fminf(__builtin_fmaf(fx[c], ix, cx), __builtin_fmaf(fy[c], iy, cy)), __builtin_fmaf(fz[c], iz, cz)) + M;
                    rf[c] = ch[c];
                    const bool ok = rf[c] != RT_REF_END && a <= b && !(a > tl) && !(b < 0.0f);
                    key[c] = ok ? a : __builtin_inff();
                }
#define RT_CE(i, j)                                                  \
    {                                                                \
        const bool sw = key[j] < key[i];                             \
        const float ka = sw ? key[j] : key[i], kb = sw ? key[i] : key[j]; \
        const uint32_t ra = sw ? rf[j] : rf[i], rb = sw ? rf[i] : rf[j];  \
        key[i] = ka, key[j] = kb, rf[i] = ra, rf[j] = rb;            \
    }
                RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#undef RT_CE
                if (key[3] < __builtin_inff() && sp < RT_STACK) stk[sp * 64] = rf[3], sp++;
                if (key[2] < __builtin_inff() && sp < RT_STACK) stk[sp * 64] = rf[2], sp++;
                if (key[1] < __builtin_inff() && sp < RT_STACK) stk[sp * 64] = rf[1], sp++;
                if (key[0] < __builtin_inff()) ref = rf[0];
                else ref = sp > 0 ? stk[(--sp) * 64] : RT_REF_END;
            }
        }
        // ---- one triangle for the lanes that stand at a leaf ----
        bool hit = false;
        double hit_t = 0.0;
        uint32_t hit_pre = 0, hit_ref = 0;
        if (ref != RT_REF_END && (ref & RT_REF_LEAF)) {
            const uint32_t here = ref, leaf_index = ref & ~RT_REF_LEAF;
            const rt_bvh_leaf& lf = sc.leaves[leaf_index];
            ctr.tris++;
            RT_PROBE(ctr, 5);
            ref = sp > 0 ? stk[(--sp) * 64] : RT_REF_END;
            const d3 O1 = D3(lf.O1), O2 = D3(lf.O2), O3 = D3(lf.O3), e1 = O2 - O1, e2 = O3 - O1;
            d3 p = cross(d, e2);
            double det = dot(e1, p);
            if (!(fabs(det) < RT_EPS)) {
                double idet = 1.0 / det;
                d3 tv = o - O1;
                double u = dot(tv, p) * idet;
                if (!(u < 0.0 || u > 1.0)) {
                    d3 q = cross(tv, e1);
                    double v = dot(d, q) * idet;
                    if (!(v < 0.0 || u + v > 1.0)) {
                        double t0 = dot(e2, q) * idet;
                        const double cur = rec_t[ray];
                        if (!(t0 <= RT_EPS) && (t0 < cur || (t0 == cur && lf.preorder < rec_pre[ray]))) {
                            // the reference must have reached this leaf (see mesh_walk4)
                            const d3 P = o + d * t0;
                            const double eps = 1e-9 * (S + fabs(o.x) + fabs(o.y) + fabs(o.z) + fabs(P.x) + fabs(P.y) + fabs(P.z));
                            const bool inside = P.x - fmin(O1.x, fmin(O2.x, O3.x)) > eps && fmax(O1.x, fmax(O2.x, O3.x)) - P.x > eps &&
                                                P.y - fmin(O1.y, fmin(O2.y, O3.y)) > eps && fmax(O1.y, fmax(O2.y, O3.y)) - P.y > eps &&
                                                P.z - fmin(O1.z, fmin(O2.z, O3.z)) > eps && fmax(O1.z, fmax(O2.z, O3.z)) - P.z > eps;
                            hit = inside || reference_reaches_leaf(sc, leaf_index, o, D3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z));
                            hit_t = t0, hit_pre = lf.preorder, hit_ref = here;
                        }
                    }
                }
            }
        }
        // accepted triangles go to their rays' records one lane at a time: two lanes may have found one for the same ray
        for (unsigned long long hm = __ballot(hit); hm != 0ull; hm &= hm - 1ull) {
            if (lane == (int)__ffsll((long long)hm) - 1) {
                const double cur = rec_t[ray];
                if (hit_t < cur || (hit_t == cur && hit_pre < rec_pre[ray])) rec_t[ray] = hit_t, rec_pre[ray] = hit_pre, rec_leaf[ray] = hit_ref;
            }
        }
    }
    if (enter) res.t = rec_t[lane], res.leaf = rec_leaf[lane];
    return res;
}

// CR/Collision_Mesh.cu:45-60: hit record of the winning leaf (normal as CR/primitive.cpp:457 stores it)
DEV void mesh_fill_hit(SCREF sc, uint32_t leafref, d3 o, d3 d, double t0, Hit& h) {
    const rt_bvh_leaf& lf = sc.leaves[leafref & ~RT_REF_LEAF];
    d3 O1 = D3(lf.O1), e1 = D3(lf.O2) - O1, e2 = D3(lf.O3) - O1;
    h.C = o + d * t0;
    d3 N = unit(cross(e1, e2));
    h.front = dot(d, N) < 0;
    h.N = h.front ? N : -N;
    h.leaf = (int32_t)lf.preorder;
}

// ---- closest hit over the scene: CR/Scene.cu:3-21, CR/Scene.cpp:27-38 ---------------------
// MESH4 = false: built for scenes in which no mesh has a four-wide tree (the per-lane stack `stk` is then unused and may be null)
template <bool STRICT, bool LOG, bool MESH4 = true>
DEV Hit scene_closest(SCREF sc, uint32_t flags, d3 o, d3 d, Counters& ctr, uint32_t* log, uint32_t log_cap,
                      uint32_t& log_n, uint32_t* stk, uint32_t* xch = nullptr) {
    Hit best;
    best.prim = -1, best.leaf = -1, best.dist = RT_BIG_DIST, best.front = false;
    best.N = D3(0, 0, 0), best.C = D3(0, 0, 0);
    const bool later_wins = (flags & RT_F_TIE_LATER_WINS) != 0;
    const d3 V = unit(d);                                   // what every intersector normalises to
    const d3 dm = (flags & RT_F_MESH_RAW_DIR) ? d : V;      // triangle / mesh direction
    for (int i = 0; i < sc.n_prims; i++) {
        const DevGeom RT_CONST& g = GEOM(sc)[i];
        Hit tmp;
        tmp.leaf = -1;
        double t = RT_BIG_DIST;
        // a hit can only replace the best one if it is closer (or within 1e-6 under later-wins)
        const double lim = later_wins ? best.dist + RT_EPS : best.dist;
        switch (g.type) {
            case RT_PRIM_SPHERE: t = sphere_hit<true>(g, o, V, tmp); break;
            case RT_PRIM_PLANE: t = plane_hit<true>(g, o, V, tmp); break;
            case RT_PRIM_SQUARE: t = square_hit<true>(g, o, V, tmp); break;
            case RT_PRIM_CYLINDER:
                if (ball_may_hit(g, o, V, lim)) t = cylinder_hit<true>(g, o, V, tmp);
                break;
            case RT_PRIM_BEZIER:
                if (!(flags & RT_F_BEZIER_STACKED)) {
                    if (ball_may_hit(g, o, V, lim)) t = bezier_newton_hit<true>(sc, g, o, V, tmp);
                } else if (ball_may_hit(g, o, V, RT_BIG_DIST)) {
                    t = bezier_stacked_hit<true>(sc, g, o, V, RT_BIG_DIST, tmp);
                }
                break;
            case RT_PRIM_TRIANGLE: t = triangle_hit<true>(g, o, dm, tmp); break;
            case RT_PRIM_MESH: {
                RT_LAP(ctr, PF_CLOSEST_PRIMS);
                MeshResult r = (!STRICT && MESH4 && g.aux2 && stk && xch)
                                   ? mesh_closest_shared(sc, g, o, dm, lim, ctr, stk, xch)
                               : (!STRICT && MESH4 && g.aux2 && stk)
                                   ? mesh_walk4<false>(sc, g, o, dm, lim, ctr, stk)
                                   : mesh_walk<STRICT, false, LOG>(sc, g.aux0, o, dm, lim, g.f[MSH_SCALE], ctr, log, log_cap, log_n,
                                                                   (uint32_t)g.aux1 << 24);
                t = r.t;
                if (t < RT_BIG_DIST) mesh_fill_hit(sc, r.leaf, o, dm, t, tmp);
                RT_LAP(ctr, PF_CLOSEST_MESH);
                break;
            }
            default: break;
        }
        if (t < RT_BIG_DIST) {
            bool take = later_wins ? (t - best.dist < RT_EPS) : (t < best.dist);
            if (take) {
                best.N = tmp.N, best.C = tmp.C, best.front = tmp.front, best.leaf = tmp.leaf;
                best.dist = t;
                best.prim = i;
            }
        }
    }
    return best;
}

// ---- which primitives can shadow a point from a lamp at all ------------------------------------
// For a shaded point A and a lamp (centre Lc, every sample within Rl of Lc), every shadow ray of
// that lamp runs inside the capsule of radius Rl around the segment A..Lc, and inside the cone with
// apex A that touches the ball (Lc, Rl).  A primitive whose bounding ball stays clear of the capsule
// or of the cone (or a plane that has A and the whole lamp strictly on one side) reports "no
// occlusion" for every sample, so its test can be skipped for all of them; bit j of the result is
// cleared for such primitives.  Radii are inflated so that "clear" holds with rounding to spare.
// Under RT_F_MESH_RAW_DIR a mesh / triangle hit counts up to ray parameter |L - A|
// (CR/Collision_Mesh.cu:19-63 does not normalise the direction), i.e. far beyond the lamp, so their
// capsule is stretched accordingly; the cone needs no stretching, it is unbounded.
// reciprocal and reciprocal square root good to ~1e-13 (hardware estimate + one Newton step): for the
// conservative tests below, whose margins are 1e-9 and wider; never on a path that decides a hit
DEV double approx_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
DEV double approx_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x * y;
    return __builtin_fma(y, __builtin_fma(-h, y, 0.5), y);
}

// squared distance between the segments p1..q1 and p2..q2 (closest points by the usual clamped solve); good to
// ~1e-12 of the lengths involved, which the callers' margins cover
DEV double seg_seg_dist2(d3 p1, d3 q1, d3 p2, d3 q2, double& s_out, double& t_out) {
    const d3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const double a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2);
    const double tiny = 1e-200;
    const double ia = a > tiny ? approx_rcp(a) : 0.0, ie = e > tiny ? approx_rcp(e) : 0.0;
    const double denom = a * e - b * b;
    double s = denom > 1e-12 * a * e && denom > tiny ? fmin(fmax((b * f - c * e) * approx_rcp(denom), 0.0), 1.0) : 0.0;
    if (!(a > tiny)) s = 0.0;
    double t = (b * s + f) * ie;
    if (t < 0.0) t = 0.0, s = fmin(fmax(-c * ia, 0.0), 1.0);
    else if (t > 1.0) t = 1.0, s = fmin(fmax((b - c) * ia, 0.0), 1.0);
    const d3 g = (p1 + d1 * s) - (p2 + d2 * t);
    s_out = s, t_out = t;
    return dot(g, g);
}

// ---- can a mesh stand between a point and a lamp at all? ----------------------------------------------
// Every shadow segment from A to a point T of the lamp (|T - Lc| <= Rl, D = Lc - A), continued to parameter smax, lies in
//     { A + s D + e : 0 <= s <= smax, |e| <= s Rl },
// and such a point can be inside a box [lo, hi] only if, on every axis k, lo_k - s Rl <= A_k + s D_k <= hi_k + s Rl: two
// half-lines in s per axis.  The walk below goes down the mesh's four-wide tree with that test (single precision, boxes and
// bounds padded by far more than the rounding errors, constraints whose slope is too close to zero left out - leaving one out
// only widens the answer) and answers "may" as soon as a box RT_CONE_LEVELS levels down, or a triangle's box, passes it;
// "cannot" only when no such box is left.  A mesh that cannot is dropped from the point's candidates for this lamp, and a
// point without candidates has every sample visible without a ray being traced.  Most points near a mesh's bounding box
// are not near the mesh (c5_final: four out of five points that had to trace shadow rays because of a mesh's root box found
// every sample visible).  `stk` = this lane's traversal stack (free while candidates are collected).
#ifndef RT_CONE_WALK
#define RT_CONE_WALK 1
#endif
#ifndef RT_CONE_LEVELS
#define RT_CONE_LEVELS 5
#endif
#ifndef RT_CONE_STEPS
#define RT_CONE_STEPS 48  // interior nodes looked at before the walk gives up and answers "may"
#endif
#ifdef RT_X_CONE_NOINLINE
static __device__ __attribute__((noinline))
#else
DEV
#endif
bool mesh_cone_may(SCREF sc, const DevGeom RT_CONST& g, d3 A, d3 D, double Rl, double smax, uint32_t* stk) {
    uint32_t ref = (uint32_t)(g.aux2 - 1);
    const float S = (float)g.f[MSH_SCALE];
    const float Af[3] = {(float)A.x, (float)A.y, (float)A.z}, Df[3] = {(float)D.x, (float)D.y, (float)D.z};
    const float R = (float)Rl * 1.00001f + 1e-30f, sm = (float)smax * 1.0001f;
    const float reach = S + fabsf(Af[0]) + fabsf(Af[1]) + fabsf(Af[2]);
    if (!(reach + sm * (fabsf(Df[0]) + fabsf(Df[1]) + fabsf(Df[2]) + R) < 1e30f)) return true;  // (also NaN)
    const float pad = 2e-6f * reach + 1e-30f;
    // per axis: the plane that bounds s from below (m_n, k_n) and the one that bounds it from above (m_f, k_f)
    float m_n[3], k_n[3], m_f[3], k_f[3];
    uint32_t o_n[3], o_f[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const bool up = Df[k] >= 0.0f;
        const float thr = 1e-2f * (fabsf(Df[k]) + R) + 1e-30f;
        // entering: s (D_k + R) >= lo - A going up, s (D_k - R) <= hi - A going down (slope of the same sign as D_k, never small
        // unless both are); leaving: s (D_k - R) <= hi - A going up, s (D_k + R) >= lo - A going down: bounds s from above
        // only where the cone grows more slowly than its axis moves
        const float a_n = up ? Df[k] + R : Df[k] - R, a_f = up ? Df[k] - R : Df[k] + R;
        const float p_n = up ? Af[k] + pad : Af[k] - pad, p_f = up ? Af[k] - pad : Af[k] + pad;
        const bool has_n = fabsf(a_n) > thr, has_f = up ? a_f > thr : a_f < -thr;
        m_n[k] = has_n ? 1.0f / a_n : 0.0f, k_n[k] = has_n ? -(p_n * m_n[k]) : -__builtin_inff();
        m_f[k] = has_f ? 1.0f / a_f : 0.0f, k_f[k] = has_f ? -(p_f * m_f[k]) : __builtin_inff();
        o_n[k] = (up ? 0u : 48u) + 16u * (uint32_t)k, o_f[k] = (up ? 48u : 0u) + 16u * (uint32_t)k;
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const char* nbase = reinterpret_cast<const char*>(sc.nodes4);
    int sp = 0, level = 0, steps = 0;
    if (ref & RT_REF_LEAF) return true;
    for (;;) {
        const uint32_t nb = ref * (uint32_t)sizeof(DevNode4);
        const f4 nx = *reinterpret_cast<const f4*>(nbase + (nb + o_n[0])), ny = *reinterpret_cast<const f4*>(nbase + (nb + o_n[1])),
                 nz = *reinterpret_cast<const f4*>(nbase + (nb + o_n[2]));
        const f4 fx = *reinterpret_cast<const f4*>(nbase + (nb + o_f[0])), fy = *reinterpret_cast<const f4*>(nbase + (nb + o_f[1])),
                 fz = *reinterpret_cast<const f4*>(nbase + (nb + o_f[2]));
        const u4 ch = *reinterpret_cast<const u4*>(nbase + (nb + 96u));
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float lo_s = fmaxf(fmaxf(__builtin_fmaf(nx[c], m_n[0], k_n[0]), __builtin_fmaf(ny[c], m_n[1], k_n[1])), __builtin_fmaf(nz[c], m_n[2], k_n[2]));
            const float hi_s = fminf(fminf(__builtin_fmaf(fx[c], m_f[0], k_f[0]), __builtin_fmaf(fy[c], m_f[1], k_f[1])), __builtin_fmaf(fz[c], m_f[2], k_f[2]));
            // (written so that a NaN passes)
            const bool ok = ch[c] != RT_REF_END && !(lo_s > hi_s + fabsf(hi_s) * 2e-4f + 1e-30f) && !(hi_s < 0.0f) && !(lo_s > sm);
            if (ok) {
                if ((ch[c] & RT_REF_LEAF) || level + 1 >= RT_CONE_LEVELS || ch[c] >= (1u << 24) || sp >= RT_STACK) return true;
                stk[sp * 64] = ch[c] | (uint32_t)(level + 1) << 24, sp++;
            }
        }
        if (sp == 0) return false;
        if (++steps >= RT_CONE_STEPS) return true;
        const uint32_t e = stk[(--sp) * 64];
        ref = e & 0xFFFFFFu, level = (int)(e >> 24);
    }
}

#ifndef RT_CONE_MESH_PASSES
#define RT_CONE_MESH_PASSES 2
#endif
#ifndef RT_CONE_SQUARE
#define RT_CONE_SQUARE 1
#endif
#ifndef RT_CONE_BALL
#define RT_CONE_BALL 1
#endif
// The grid's answer for the cell A lies in: the primitives that may stand between its points and the lamp (all of them where
// the grid has no answer), less `skip`; `full` gets those that hide the whole lamp from every point of the cell.
DEV unsigned long long candidate_cell(SCREF sc, d3 A, int lamp, unsigned long long skip, unsigned long long& full) {
    full = 0ull;
    unsigned long long cell = ~0ull;
    if (sc.n_prims <= 64 && sc.cand_grid) {
        const double gx = (A.x - sc.cg_lo[0]) * sc.cg_inv[0], gy = (A.y - sc.cg_lo[1]) * sc.cg_inv[1], gz = (A.z - sc.cg_lo[2]) * sc.cg_inv[2];
        if (gx >= 0.0 && gy >= 0.0 && gz >= 0.0 && gx < (double)RT_CGRID && gy < (double)RT_CGRID && gz < (double)RT_CGRID) {
            const int ix = (int)gx, iy = (int)gy, iz = (int)gz;
            const ulonglong2 cm = reinterpret_cast<const ulonglong2*>(sc.cand_grid)[(((size_t)lamp * RT_CGRID + iz) * RT_CGRID + iy) * RT_CGRID + ix];
            cell = cm.x;
            full = cm.y & ~skip;
        }
    }
    return cell & ~skip;
}

// radius of the ball around L.O that holds every sample point of the lamp, rounded up
DEV double lamp_reach(const rt_light RT_CONST& L) {
    double Rl = 0.0;
    if (L.type == RT_LIGHT_SPHERE) Rl = fabs(L.R);
    else if (L.type == RT_LIGHT_SQUARE) Rl = module(d3{L.Dx.x, L.Dx.y, L.Dx.z}) + module(d3{L.Dy.x, L.Dy.y, L.Dy.z});
    return Rl * (1.0 + 1e-6) + 1e-9;
}

// The primitives that may stand between A and the lamp, for a point whose cell mask is `cell` (candidate_cell; the caller has
// dealt with cells in an umbra).  `wave_cells` = the union of `cell` over the lanes that call (the same in every lane): only those
// primitives are looked at.  `full` gets the primitives that, each on its own, hide the whole lamp from A (spheres and cylinders
// only: the point lies in their umbra, every sample is occluded, none needs tracing).
DEV unsigned long long light_candidates(SCREF sc, uint32_t flags, d3 A, const rt_light RT_CONST& L, unsigned long long skip,
                                        unsigned long long cell, unsigned long long wave_cells, unsigned long long& full) {
    full = 0ull;
    if (sc.n_prims > 64) return ~0ull;
    if (cell == 0ull) return 0ull;
    const double Rl = lamp_reach(L);
    const d3 Lc = d3{L.O.x, L.O.y, L.O.z}, D = Lc - A;
    const double L2 = dot(D, D);
    if (!(L2 > 1e-200) || !(L2 < 1e200) || !(Rl < 1e100)) return ~0ull;
    const double inv_len = approx_rsqrt(L2), inv = inv_len * inv_len, len = L2 * inv_len;
    const double s_raw = (len + Rl) * (1.0 + 1e-6), r_raw = Rl * s_raw + 1e-9;  // stretched capsule (raw-direction meshes)
    const bool raw = (flags & RT_F_MESH_RAW_DIR) != 0;
    // cone: sin and cos^2 of the half-angle, and its radius per unit of the segment parameter s (distance s * len from A)
    const bool cone = len > 1.001 * Rl;
    const double sin_a = cone ? Rl * inv_len : 0.0, cos2_a = 1.0 - sin_a * sin_a;
    const double cone_k = cone ? Rl * approx_rsqrt(cos2_a) * (1.0 + 1e-6) : 0.0;
    // per-axis reciprocals of the segment direction for the box clips (0 marks an axis the segment does not move along)
    const double Aa[3] = {A.x, A.y, A.z}, Da[3] = {D.x, D.y, D.z};
    double iD[3];
#pragma unroll
    for (int k = 0; k < 3; k++) iD[k] = fabs(Da[k]) > 1e-200 ? approx_rcp(Da[k]) : 0.0;
    unsigned long long m = 0;
    const unsigned long long in_scene = sc.n_prims >= 64 ? ~0ull : (1ull << sc.n_prims) - 1ull;
    for (unsigned long long todo = wave_cells & in_scene; todo != 0ull; todo &= todo - 1ull) {
        const int j = __builtin_ctzll(todo);
        if (__ballot(((cell >> j) & 1ull) != 0ull && (full & ~skip) == 0ull) == 0ull) continue;  // no lane still needs this primitive
        const DevGeom RT_CONST& g = GEOM(sc)[j];
        bool may = ((cell >> j) & 1ull) != 0ull;
        if (!may) {
        } else if (g.type == RT_PRIM_PLANE) {
            const d3 N = G3(g, PLN_N), Q = G3(g, PLN_NR);
            const double sa = dot(A - Q, N), sl = dot(Lc - Q, N);
            const double tol = 1e-7 * (1.0 + fabs(sa) + fabs(sl));
            if ((sa > tol && sl > Rl + tol) || (sa < -tol && sl < -(Rl + tol))) may = false;
        } else if (g.type == RT_PRIM_MESH && g.aux0 != RT_REF_END) {
            // segment A + D*s, 0 <= s <= smax, against the root box grown by the radius of the region; the cone
            // is narrower than the capsule near A, so the radius is lowered to the cone's at the far end of the
            // overlap found so far and the clip repeated
            const double smax = raw ? s_raw : 1.0, pad = 1e-9 * (1.0 + g.f[MSH_SCALE] + fabs(A.x) + fabs(A.y) + fabs(A.z));
            double rc = (raw ? r_raw : Rl) + pad, thi = smax;
#pragma unroll
            for (int pass = 0; pass < RT_CONE_MESH_PASSES && may; pass++) {
                double tlo = 0.0;
                thi = smax;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double lo = g.f[MSH_BMIN + k] - rc, hi = g.f[MSH_BMAX + k] + rc;
                    if (iD[k] != 0.0) {
                        const double t1 = (lo - Aa[k]) * iD[k], t2 = (hi - Aa[k]) * iD[k];
                        tlo = fmax(tlo, fmin(t1, t2));
                        thi = fmin(thi, fmax(t1, t2));
                    } else if (Aa[k] < lo || Aa[k] > hi) {
                        thi = -1.0;
                    }
                }
                if (tlo > thi * (1.0 + 1e-9) + 1e-12) may = false;
                if (!cone) break;
                const double rc2 = thi * (1.0 + 1e-9) * cone_k + 1e-12 * cone_k + pad;
                if (!(rc2 < rc)) break;
                rc = rc2;
            }
        } else {
            const bool stretched = raw && (g.type == RT_PRIM_MESH || g.type == RT_PRIM_TRIANGLE);
            const double smax = stretched ? s_raw : 1.0, rc = stretched ? r_raw : Rl;
            const d3 w = G3(g, BND_C) - A;
            const double spu = dot(w, D) * inv, sp = fmin(fmax(spu, 0.0), smax);
            const d3 e = w - D * sp;
            const double rb = g.f[BND_R], rr = rb + rc, ee = dot(e, e);
            // (2 (|w|^2 + rb^2) bounds the squared distance from A to any point of the primitive; taken afresh at each use: one
            // register pair fewer alive across the tests)
#define noise isect_noise(2.0 * (dot(w, w) + rb * rb))
            if (ee > rr * rr * (1.0 + 1e-9) + noise) may = false;
            if (g.type == RT_PRIM_SPHERE) {
                // a sphere around the whole capsule: every shadow ray starts and ends inside it and meets its surface
                // only beyond the lamp sample, which is no occlusion (CR/Shade.cu:75: dist - hit.dist > 1e-6)
                const d3 wl = w - D;  // centre seen from the lamp
                const double reach = sqrt(fmax(dot(w, w), dot(wl, wl))) + Rl, r_in = rb * (1.0 - 2e-6) - 2e-9 - 1e-6 * (1.0 + len);
                if (reach < r_in) may = false;
            }
            if (may && (g.type == RT_PRIM_CYLINDER || g.type == RT_PRIM_BEZIER)) {
                // long thin primitives: their bounding ball is mostly empty, the capsule around their axis is not
                const double reach = g.f[CAP_R] + Rl + 1e-9 * (1.0 + len);
                const d3 P0 = G3(g, CAP_P0), P1 = G3(g, CAP_P1);
                double ss, tt;
                const double dd2 = seg_seg_dist2(A, Lc, P0, P1, ss, tt);
                if (dd2 > reach * reach * (1.0 + 1e-6) + noise) may = false;
                if (may && g.type == RT_PRIM_CYLINDER) {
                    // Umbra of the solid cylinder: the axis of the shadow rays passes so deep inside it (radially and away
                    // from both ends) that every ray to the lamp goes through its interior, all of it lies nearer than the
                    // lamp, and A is outside: CR/Collision.cu:3-107 then reports a hit in front of the lamp for every sample.
                    const double Rc = g.f[CAP_R] * (1.0 - 2e-6) - 2e-9, alen = g.f[CYL_LEN];
                    const double m = Rl + 1e-5 * (1.0 + alen + Rc);
                    const d3 a0 = A - P0, a1 = A - P1, ax = G3(g, CYL_DN);
                    const double ta = fmin(fmax(dot(a0, ax), 0.0), alen);
                    const d3 ea = a0 - ax * ta;
                    const double far2 = fmax(dot(a0, a0), dot(a1, a1));
                    const double lim = len - Rl - 2e-6 - g.f[CAP_R];
                    const double deep = Rc - Rl - 1e-6;
                    if (tt * alen > m && (1.0 - tt) * alen > m && deep > 0.0 && dd2 + noise < deep * deep * (1.0 - 1e-6) && lim > 0.0 &&
                        far2 < lim * lim * (1.0 - 1e-6) && dot(ea, ea) > (g.f[CAP_R] + 2e-6) * (g.f[CAP_R] + 2e-6) * (1.0 + 1e-6) + noise)
                        full |= 1ull << j;
                }
            }
            if (may && g.type == RT_PRIM_SPHERE) {
                // Umbra of a sphere (CR/Collision.cu:155-176): every ray to the lamp passes its centre at less than the radius
                // (with 1e-6 to spare in the discriminant), the sphere lies wholly in front of the lamp, A wholly outside
                const double r_lo = rb * (1.0 - 2e-6) - 2e-9, dA2 = dot(w, w);
                const double dA = dA2 * approx_rsqrt(fmax(dA2, 1e-300)), dline = ee * approx_rsqrt(fmax(ee, 1e-300)) + Rl;
                if (r_lo * r_lo - dline * dline > 2e-6 * (1.0 + r_lo * r_lo) + noise && dA + rb < len - Rl - 2e-6 && dA - rb > 2e-6) full |= 1ull << j;
            }
            if (RT_CONE_BALL && cone && spu > 0.0 && spu <= smax) {
                // beside the segment: e is the offset of the ball's centre from the axis, and its distance from the
                // cone's surface is |e| cos(a) - (distance along the axis) sin(a) = |e| cos(a) - spu * Rl
                const double q = rb * (1.0 + 1e-6) + 1e-9 * (1.0 + len) + spu * Rl;
                if (ee * cos2_a > q * q * (1.0 + 1e-9) + noise) may = false;
            }
            if (RT_CONE_SQUARE && may && g.type == RT_PRIM_SQUARE) {
                // A square fills little of its bounding ball.  A shadow segment A + s (D + e), |e| <= Rl, meets the square's plane
                // where hA + s (hL - hA + N.e) = 0 (hA, hL: heights of A and of the lamp's centre over the plane): not at all
                // if A and the whole lamp lie on one side, else for s between hA / (hA - hL -+ Rl); there its coordinates
                // along Dx, Dy (CR/Collision.cu:219-224 compares |(P - O).Dx| with Dx.Dx) lie within s Rl |Dx| of the axis point's.
                const d3 O = G3(g, SQR_O), N = G3(g, SQR_N);
                const double hA = dot(A - O, N), hL = dot(Lc - O, N);
                const double tol = 1e-7 * (1.0 + fabs(hA) + fabs(hL));
                if ((hA > tol && hL > Rl + tol) || (hA < -tol && hL < -(Rl + tol))) {
                    may = false;
                } else {
                    const double dlo = (hA - hL) - Rl, dhi = (hA - hL) + Rl;
                    if (dlo > tol || dhi < -tol) {  // (else some segments run along the plane: no bound on s)
                        const double s1 = hA / dlo, s2 = hA / dhi;
                        const double slo = fmax(fmin(s1, s2) * (1.0 - 1e-9) - 1e-9, 0.0), shi = fmin(fmax(s1, s2) * (1.0 + 1e-9) + 1e-9, 1.0 + 1e-9);
                        if (slo > shi) {
                            may = false;  // the plane is met before A or beyond the lamp
                        } else {
                            const d3 w0 = A - O, Dx = G3(g, SQR_DX), Dy = G3(g, SQR_DY);
                            const double lx = g.f[SQR_LX], ly = g.f[SQR_LY];
                            const double uA = dot(w0, Dx), uD = dot(D, Dx), vA = dot(w0, Dy), vD = dot(D, Dy);
                            const double ru = shi * Rl * sqrt(lx), rv = shi * Rl * sqrt(ly);
                            const double u1 = uA + slo * uD, u2 = uA + shi * uD, v1 = vA + slo * vD, v2 = vA + shi * vD;
                            const double eu = 1e-9 * (lx + fabs(uA) + fabs(uD)) + 1e-12, ev = 1e-9 * (ly + fabs(vA) + fabs(vD)) + 1e-12;
                            if (fmin(u1, u2) - ru > lx + eu || fmax(u1, u2) + ru < -(lx + eu) || fmin(v1, v2) - rv > ly + ev || fmax(v1, v2) + rv < -(ly + ev))
                                may = false;
                        }
                    }
                }
            }
        }
#undef noise
        if (may) m |= 1ull << j;
    }
    return m;
}

// ---- one shadow ray: CR/Shade.cu:36-46,64-75,94-105, CR/Light.cpp:35-40,67-78 ----------------
// true = nothing between C and target.  A primitive occludes if dist - its_hit_distance > 1e-6.
// `cand` = the primitives this ray has to be tested against (pruned mode, scenes of at most 64 primitives: the shaded one and the
// lamp's own are already left out of it), `wave_prims` = the union of `cand` over the rays of the wave (the same in every lane):
// the loop visits only those.  Reference-order mode and scenes of more than 64 primitives test every primitive but the skipped.
template <bool STRICT, bool MESH4 = true>
DEV bool shadow_clear(SCREF sc, uint32_t flags, d3 C, d3 target, int crashed, int light_prim, unsigned long long cand,
                      unsigned long long wave_prims, Counters& ctr, uint32_t* stk) {
    const d3 Vv = target - C;
    const double dist = module(Vv);
    const d3 V = Vv / dist;                                  // = normalize(Vv)
    const d3 dm = (flags & RT_F_MESH_RAW_DIR) ? Vv : V;
    uint32_t dummy_n = 0;
    const bool self_skip = (flags & RT_F_SHADOW_SELF_SKIP) != 0;
    const bool masked = !STRICT && sc.n_prims <= 64;
    unsigned long long todo = masked ? wave_prims : ~0ull;
    for (int j = 0; j < sc.n_prims; j++) {
        if (masked) {
            // primitives that cannot lie between any lane's shaded point and its lamp are skipped by the whole wave
            if (todo == 0ull) break;
            j = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            if (!((cand >> j) & 1ull)) continue;
        } else {
            if (self_skip && j == crashed) continue;
            if (j == light_prim) continue;
        }
        const DevGeom RT_CONST& g = GEOM(sc)[j];
        Hit tmp;
        double t = RT_BIG_DIST;
        switch (g.type) {
            case RT_PRIM_SPHERE: t = sphere_hit<false>(g, C, V, tmp); break;
            case RT_PRIM_PLANE: t = plane_hit<false>(g, C, V, tmp); break;
            case RT_PRIM_SQUARE: t = square_hit<false>(g, C, V, tmp); break;
            case RT_PRIM_CYLINDER:
                if (ball_may_hit(g, C, V, dist)) t = cylinder_hit<false>(g, C, V, tmp);
                break;
            case RT_PRIM_BEZIER:
                // the stack's `closest` bookkeeping can return a far segment's distance, so segments
                // are not pruned by distance here, only by direction
                if (!(flags & RT_F_BEZIER_STACKED)) {
                    if (ball_may_hit(g, C, V, dist)) t = bezier_newton_hit<false>(sc, g, C, V, tmp);
                } else if (ball_may_hit(g, C, V, RT_BIG_DIST)) {
                    t = bezier_stacked_hit<false>(sc, g, C, V, RT_BIG_DIST, tmp);
                }
                break;
            case RT_PRIM_TRIANGLE: t = triangle_hit<false>(g, C, dm, tmp); break;
            case RT_PRIM_MESH:
                RT_LAP(ctr, PF_SH_PRIMS);
                t = (!STRICT && MESH4 && g.aux2 && stk) ? mesh_walk4<true>(sc, g, C, dm, dist, ctr, stk).t
                                               : mesh_walk<STRICT, true, false>(sc, g.aux0, C, dm, dist, g.f[MSH_SCALE], ctr, nullptr, 0, dummy_n, 0).t;
                RT_LAP(ctr, PF_SH_MESH);
                break;
            default: break;
        }
        if (dist - t > RT_EPS) return false;
    }
    return true;
}

// shadow_clear for the pruned mode, called by all 64 lanes (`act` = this lane has a ray): the meshes are walked with
// mesh_anyhit_shared, so lanes without a ray, or whose ray is already known to be occluded, help the others.
// (scenes of at most 64 primitives only: `cand` and `wave_prims` as in shadow_clear)
DEV bool shadow_clear_shared(SCREF sc, uint32_t flags, bool act, d3 C, d3 target, unsigned long long cand, unsigned long long wave_prims,
                             Counters& ctr, uint32_t* stk, uint32_t* xch) {
    const d3 Vv = target - C;
    const double dist = module(Vv);
    const d3 V = Vv / dist;  // = normalize(Vv)
    const d3 dm = (flags & RT_F_MESH_RAW_DIR) ? Vv : V;
    uint32_t dummy_n = 0;
    bool alive = act;  // has a ray that nothing has occluded so far
    int sp = 0;        // roots of the meshes the ray may enter, on this lane's stack
    for (unsigned long long todo = wave_prims; todo != 0ull; todo &= todo - 1ull) {
        const int j = __builtin_ctzll(todo);
        const bool mine = alive && ((cand >> j) & 1ull);
        if (__ballot(mine) == 0ull) continue;
        if (!mine) continue;
        const DevGeom RT_CONST& g = GEOM(sc)[j];
        Hit tmp;
        double t = RT_BIG_DIST;
        switch (g.type) {
            case RT_PRIM_SPHERE: t = sphere_hit<false>(g, C, V, tmp); break;
            case RT_PRIM_PLANE: t = plane_hit<false>(g, C, V, tmp); break;
            case RT_PRIM_SQUARE: t = square_hit<false>(g, C, V, tmp); break;
            case RT_PRIM_CYLINDER:
                if (ball_may_hit(g, C, V, dist)) t = cylinder_hit<false>(g, C, V, tmp);
                break;
            case RT_PRIM_BEZIER:
                if (!(flags & RT_F_BEZIER_STACKED)) {
                    if (ball_may_hit(g, C, V, dist)) t = bezier_newton_hit<false>(sc, g, C, V, tmp);
                } else if (ball_may_hit(g, C, V, RT_BIG_DIST)) {
                    t = bezier_stacked_hit<false>(sc, g, C, V, RT_BIG_DIST, tmp);
                }
                break;
            case RT_PRIM_TRIANGLE: t = triangle_hit<false>(g, C, dm, tmp); break;
            case RT_PRIM_MESH: {
                // the root box test of mesh_walk4; rays whose slab distances do not fit single precision walk on their own
                const d3 inv = D3(1.0 / dm.x, 1.0 / dm.y, 1.0 / dm.z);
                const double S = g.f[MSH_SCALE];
                const double mm = fmax(fmax((S + fabs(C.x)) * fabs(inv.x), (S + fabs(C.y)) * fabs(inv.y)), (S + fabs(C.z)) * fabs(inv.z));
                if (!g.aux2 || sp >= RT_SHARED_ROOTS || !(mm < 1e30) || !(fabs(inv.x) < 1e30) || !(fabs(inv.y) < 1e30) || !(fabs(inv.z) < 1e30)) {
                    // (also: a mesh without a four-wide tree, a ray with more candidate meshes than roots are kept for)
                    t = mesh_walk<false, true, false>(sc, g.aux0, C, dm, dist, S, ctr, nullptr, 0, dummy_n, 0).t;
                } else {
                    ctr.nodes++;
                    const double margin = 1e-13 * mm;
                    double t0x = (g.f[MSH_BMIN] - C.x) * inv.x, t1x = (g.f[MSH_BMAX] - C.x) * inv.x;
                    double t0y = (g.f[MSH_BMIN + 1] - C.y) * inv.y, t1y = (g.f[MSH_BMAX + 1] - C.y) * inv.y;
                    double t0z = (g.f[MSH_BMIN + 2] - C.z) * inv.z, t1z = (g.f[MSH_BMAX + 2] - C.z) * inv.z;
                    double tn = fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmin(t0z, t1z));
                    double tf = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmax(t0z, t1z));
                    if (!(!(tn <= tf) || tn > dist + margin + 1e-9 * fabs(dist) || tf < -margin)) stk[sp * 64] = (uint32_t)(g.aux2 - 1), sp++;
                }
                break;
            }
            default: break;
        }
        if (dist - t > RT_EPS) alive = false;
    }
    // all meshes in one walk, shared by the wave (every lane gets here: no lane leaves the loop above early)
    if (!alive) sp = 0;
    RT_LAP(ctr, PF_SH_PRIMS);
    const bool occluded = mesh_anyhit_shared(sc, sc.mesh_scale_max, sp, C, dm, dist, ctr, stk, xch);
    RT_LAP(ctr, PF_SH_MESH);
    return alive && !occluded;
}

// ---- textures: CR/bmp.cpp:133-150 (double taps), CR/Primitive.cu:85-115 (8-bit taps) --------
DEV d3 smooth_color(SCREF sc, uint32_t flags, const DevShade& px, double u, double v) {
    const int H = px.tex_h, W = px.tex_w;
    const bool u8 = (flags & RT_F_TEX_U8_TAPS) != 0;
    double U = (u - (u8 ? (double)floorf((float)u) : floor(u))) * H;
    double Vv = (v - (u8 ? (double)floorf((float)v) : floor(v))) * W;
    int U1 = u8 ? (int)floorf((float)(U - RT_EPS)) : (int)floor(U - RT_EPS), U2 = U1 + 1;
    int V1 = u8 ? (int)floorf((float)(Vv - RT_EPS)) : (int)floor(Vv - RT_EPS), V2 = V1 + 1;
    double ru = U2 - U, rv = V2 - Vv;
    if (U1 < 0) U1 = H - 1;
    if (U2 == H) U2 = 0;
    if (V1 < 0) V1 = W - 1;
    if (V2 == W) V2 = 0;
    if (U1 < 0 || U1 >= H) U1 = 0;  // finite u,v never get here; a NaN must not read outside the texture
    if (U2 < 0 || U2 >= H) U2 = 0;
    if (V1 < 0 || V1 >= W) V1 = 0;
    if (V2 < 0 || V2 >= W) V2 = 0;
    // one 4-byte read per tap (its three channels and one byte of the neighbour; the slab is padded past the last texel)
    const uint8_t* base = sc.texels + px.tex_off;
    uint32_t q11, q12, q21, q22;
    __builtin_memcpy(&q11, base + ((size_t)U1 * W + V1) * 3, 4);
    __builtin_memcpy(&q12, base + ((size_t)U1 * W + V2) * 3, 4);
    __builtin_memcpy(&q21, base + ((size_t)U2 * W + V1) * 3, 4);
    __builtin_memcpy(&q22, base + ((size_t)U2 * W + V2) * 3, 4);
    const uint32_t t11[3] = {q11 & 255u, (q11 >> 8) & 255u, (q11 >> 16) & 255u}, t12[3] = {q12 & 255u, (q12 >> 8) & 255u, (q12 >> 16) & 255u};
    const uint32_t t21[3] = {q21 & 255u, (q21 >> 8) & 255u, (q21 >> 16) & 255u}, t22[3] = {q22 & 255u, (q22 >> 8) & 255u, (q22 >> 16) & 255u};
    // CR/bmp.cpp:147-148 weights tap (U1,V2) by rat_U (1 - rat_V) and tap (U2,V1) by (1 - rat_U) rat_V; the .cu fetch indexes a
    // transposed copy of the texture (CR/Renderer_CUDA.cpp:149-152), so in CR/Primitive.cu:109-110 the two taps swap weights
    double w1a = ru, w1b = rv, w2a = ru, w2b = 1 - rv, w3a = 1 - ru, w3b = rv, w4a = 1 - ru, w4b = 1 - rv;
    if (u8) w2a = 1 - ru, w2b = rv, w3a = ru, w3b = 1 - rv;
    double out[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (u8) {
            double s = 0;
            s += (double)(uint8_t)(int)((double)(uint8_t)(int)(t11[k] * w1a) * w1b);
            s += (double)(uint8_t)(int)((double)(uint8_t)(int)(t12[k] * w2a) * w2b);
            s += (double)(uint8_t)(int)((double)(uint8_t)(int)(t21[k] * w3a) * w3b);
            s += (double)(uint8_t)(int)((double)(uint8_t)(int)(t22[k] * w4a) * w4b);
            out[k] = s / 256.0;
        } else {
            double s = 0;
            s = s + t11[k] / 256.0 * w1a * w1b;
            s = s + t12[k] / 256.0 * w2a * w2b;
            s = s + t21[k] / 256.0 * w3a * w3b;
            s = s + t22[k] / 256.0 * w4a * w4b;
            out[k] = s;
        }
    }
    return D3(out[0], out[1], out[2]);
}

// CR/Primitive.cu:16-83, CR/primitive.cpp:140-147,177-181,229-233
DEV d3 T3(const DevShade& p, int i) { return d3{p.tf[i], p.tf[i + 1], p.tf[i + 2]}; }
DEV d3 texture_color(SCREF sc, uint32_t flags, const DevShade& p, const Hit& c, Counters& ctr) {
    double u, v;
    switch (p.type) {
        case RT_PRIM_SPHERE: {
            d3 I = unit(c.C - T3(p, TF_SPH_O));
            // acos / sin of CR/Primitive.cu:24-33 through the compact routines; sin(acos(x)) as sqrt(1 - x^2)
            const double x = fmin(fmax(-dot(I, T3(p, TF_SPH_DE)), -1.0), 1.0);
            double a = rt_acos(x);
            double b = rt_acos(fmin(fmax(dot(I, T3(p, TF_SPH_DC)) / sqrt(1.0 - x * x), -1.0), 1.0));
            u = a / M_PI, v = b / 2 / M_PI;
            if (dot(I, T3(p, TF_SPH_X)) < 0) v = 1 - v;
            break;
        }
        case RT_PRIM_PLANE: {
            u = dot(c.C, T3(p, TF_PLN_DX)) / p.tf[TF_PLN_DXX];
            v = dot(c.C, T3(p, TF_PLN_DY)) / p.tf[TF_PLN_DYY];
            break;
        }
        case RT_PRIM_SQUARE: {
            d3 b = (flags & RT_F_SQUARE_UV_NO_ORIGIN) ? c.C : c.C - T3(p, TF_SQR_O);
            u = dot(b, T3(p, TF_SQR_DX)) / p.tf[TF_SQR_DXX] / 2 + 0.5;
            v = dot(b, T3(p, TF_SQR_DY)) / p.tf[TF_SQR_DYY] / 2 + 0.5;
            break;
        }
        case RT_PRIM_CYLINDER: {
            d3 O1 = T3(p, TF_CYL_O1), O2 = T3(p, TF_CYL_O2);
            if (!c.front) {
                d3 cp = c.C - O1;
                double ang = rt_atan2(cp.z, cp.x);  // atan2 of CR/Primitive.cu:58, see rt_atan2
                if (ang < 0) ang += 2 * M_PI;
                u = ang / (2 * M_PI);
                v = dot(c.C - O1, T3(p, TF_CYL_AX)) / p.tf[TF_CYL_LEN];
            } else {
                d3 cc = module(c.C - O1) < module(c.C - O2) ? O1 : O2;
                d3 l = c.C - cc;
                u = l.x / (2.0 * p.tf[TF_CYL_R]) + 0.5;
                v = l.z / (2.0 * p.tf[TF_CYL_R]) + 0.5;
            }
            break;
        }
        default:  // Bezier / triangle / mesh have no texture mapping: Color() (CR/primitive.cpp:290-295,339-342,372-375)
            return D3(0, 0, 0);
    }
    ctr.texfetch++;
    return smooth_color(sc, flags, p, u, v);
}

// Point on a light for sample draws (a, b): CR/Shade.cu:3-21
DEV d3 light_sample(const rt_light RT_CONST& L, float fa, float fb) {
    if (L.type == RT_LIGHT_SQUARE) {
        double a = (double)(2 * fa) - 1.0, b = (double)(2 * fb) - 1.0;
        return d3{L.O.x, L.O.y, L.O.z} + d3{L.Dx.x, L.Dx.y, L.Dx.z} * a + d3{L.Dy.x, L.Dy.y, L.Dy.z} * b;
    }
    double z = 2.0 * fa - 1.0;
    double th = fb * 2.0 * M_PI;
    double r = sqrt(1.0 - z * z), sn, cs;
    det_sincos(th, sn, cs);
    double x = r * cs, y = r * sn;
    return d3{L.O.x, L.O.y, L.O.z} + D3(L.R * x, L.R * y, L.R * z);
}

#endif
