/*
 * cpu_ref.c — TEST INFRASTRUCTURE ONLY.  CPU restatement (the "oracle") of the reference's
 * render path, in plain C, double precision, recursive exactly like the reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the product (librt_amd.so, cuda_raytracing_amd/) never does.
 *
 * Parity pinning: in RT_MODE_CPU_REF this file must reproduce, bit for bit, what the
 * reference's own CPU renderer (CR/raytracer.cpp, CR/primitive.cpp, CR/Light.cpp, compiled
 * by oracle/Makefile into oracle/_ref/ref_dump) computes for sphere / plane / square scenes
 * with point lights: nearest-hit primitive, hit distance and pass-1 colour per pixel.  Those
 * values are committed under tests/golden/render/ and checked by tests/test_oracle_pinned.py.
 * Cylinder, Bezier, triangle, mesh/BVH, sphere and square lights, depth of field and
 * multi-sample jitter exist only in the reference's CUDA files, which cannot be built here
 * (no nvcc); for those this file restates the .cu source line by line (citations below) and
 * parity with the reference is "unpinned" beyond code review.  The reference's CUDA RNG is
 * undefined behaviour (CR/RayTracer.cu:9-21: shared-memory curand state from another launch),
 * so random draws come from Philox4x32-10 keyed by (seed; pixel, sample, draw index) instead.
 *
 * Arithmetic: operation order follows the reference expression by expression; compile with
 * -ffp-contract=off.  Places where the .cu rounds through float (`float t`, sqrtf in length(),
 * sinf/cosf) are evaluated in double here, as the reference's CPU path does.  sin/cos of angles
 * that steer rays come from det_sincos() below rather than libm (see there).
 */
#include "rt_amd.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef rt_vec3 v3;

static inline v3 V3(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulk(v3 a, double k) { return V3(a.x * k, a.y * k, a.z * k); }
static inline v3 divk(v3 a, double k) { return V3(a.x / k, a.y / k, a.z / k); }
static inline v3 mulv(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline double module(v3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); } /* CR/vector3.cpp:98-101 */
static inline v3 unit(v3 a) { return divk(a, module(a)); }                            /* CR/vector3.cpp:128-131 */
/* CR/vector3.cpp:142-156, helper_math.h:1873-1883 */
static inline v3 vertical(v3 a) {
    v3 r = cross(a, V3(0, 0, 1));
    if (fabs(r.x) < RT_EPS && fabs(r.y) < RT_EPS && fabs(r.z) < RT_EPS) return V3(1, 0, 0);
    return unit(r);
}
/* CR/vector3.cpp:162-165, helper_math.h:1789-1792 */
static inline v3 reflect(v3 i, v3 n) { return sub(i, mulk(n, 2 * dot(n, i))); }
/* CR/vector3.cpp:167-173, helper_math.h:1894-1900 */
static inline v3 refract(v3 Vv, v3 N, double n) {
    Vv = unit(Vv);
    double cosI = -dot(N, Vv), cosT2 = 1 - (n * n) * (1 - cosI * cosI);
    if (cosT2 > RT_EPS) return add(mulk(Vv, n), mulk(N, n * cosI - sqrt(cosT2)));
    return reflect(Vv, N);
}

/* ---- Philox4x32-10 (Salmon et al., SC'11), counter = (pixel, sample, block, 0), key = seed */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
}

typedef struct {
    uint32_t k0, k1, pixel, sample, next; /* next = index of the next draw of this (pixel, sample) */
    uint32_t block, buf[4];
} rng_t;

static void rng_init(rng_t* g, uint64_t seed, uint32_t pixel, uint32_t sample) {
    g->k0 = (uint32_t)seed, g->k1 = (uint32_t)(seed >> 32);
    g->pixel = pixel, g->sample = sample, g->next = 0, g->block = 0xFFFFFFFFu;
}
/* uniform float in (0, 1], the mapping of curand_uniform that frand() returns (CR/RayTracer.cu:17-21) */
static float frand(rng_t* g) {
    uint32_t i = g->next++;
    if ((i >> 2) != g->block) {
        g->block = i >> 2;
        g->buf[0] = g->pixel, g->buf[1] = g->sample, g->buf[2] = g->block, g->buf[3] = 0;
        philox4x32_10(g->buf, g->k0, g->k1);
    }
    return (float)g->buf[i & 3] * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}
/* exported for the known-answer test */
void orc_philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1) { philox4x32_10(ctr, k0, k1); }
float orc_frand_at(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t draw) {
    rng_t g;
    rng_init(&g, seed, pixel, sample);
    g.next = draw;
    return frand(&g);
}


/* ---- sin/cos of the angles that steer rays (sphere-light samples, fuzzy reflection) ----------
 * An explicit polynomial (Cody-Waite reduction by pi/2 + the fdlibm kernel polynomials), not
 * libm, so that the device code - which evaluates the same expressions in the same order - gets
 * bit-identical ray directions and therefore identical visibility decisions.  Accurate to about
 * one ulp for the arguments that occur (0 <= x <= 2*pi). */
static void det_sincos(double x, double* sn, double* cs) {
    int k = (int)(x * 6.36619772367581382433e-01 + 0.5);        /* nearest multiple of pi/2 */
    double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    double z = r * r;
    double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    double s = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    double c = 1.0 - (0.5 * z - z * pc);
    switch (k & 3) {
        case 0: *sn = s, *cs = c; break;
        case 1: *sn = c, *cs = -s; break;
        case 2: *sn = -s, *cs = -c; break;
        default: *sn = -c, *cs = s; break;
    }
}
void orc_det_sincos(double x, double* sn, double* cs) { det_sincos(x, sn, cs); }

/* ---- per-call context -------------------------------------------------------------------- */
typedef struct {
    const rt_scene_desc* sc;
    uint32_t flags;
    int cam_w, cam_h;
    int shade_n; /* int(16 * shade_quality), CR/Scene_Material.cu:42 */
    rt_stats st;
    /* optional visit log of the ray being traced */
    uint32_t* log;
    uint32_t log_cap, log_n;
} ctx_t;

typedef struct { /* CR/Primitive.cuh:107-113 */
    int hit;
    int prim;
    v3 N, C;
    double dist;
    int front;
    int leaf; /* leaf pre-order index inside the mesh, or -1 */
} col_t;

static col_t col_init(void) {
    col_t c;
    memset(&c, 0, sizeof(c));
    c.prim = -1, c.dist = RT_BIG_DIST, c.leaf = -1;
    return c;
}

/* CR/Collision.cu:3-107 */
static void cylinder_intersect(v3 O1, v3 O2, double R, v3 o, v3 d, col_t* c) {
    v3 Vv = unit(d);
    {
        v3 dd = sub(O2, O1), m = sub(o, O1);
        double r2 = R * R;
        v3 dN = unit(dd);
        v3 mP = cross(m, dN), vP = cross(Vv, dN);
        double a = dot(vP, vP), b = 2.0 * dot(vP, mP), cc = dot(mP, mP) - r2;
        double det = b * b - 4 * a * cc;
        if (det >= 0) {
            det = sqrt(det);
            double t0 = (-b - det) / (2.0 * a), t1 = (-b + det) / (2.0 * a);
            double t = fmin(t0, t1);
            if (t < 0) t = fmax(t0, t1);
            if (t >= 0) {
                v3 P = add(o, mulk(Vv, t));
                double u = dot(sub(P, O1), dN);
                if (u >= 0 && u <= module(dd)) {
                    c->dist = t, c->C = P, c->hit = 1, c->front = 0; /* front = 0 marks the body */
                    c->N = unit(sub(P, add(O1, mulk(dN, u))));
                    return;
                }
            }
        }
    }
    for (int cap = 0; cap < 2; cap++) { /* top cap (at O2) first, then bottom cap (at O1) */
        v3 N = cap == 0 ? unit(sub(O1, O2)) : unit(sub(O2, O1));
        v3 Q0 = cap == 0 ? O2 : O1;
        double den = dot(N, Vv);
        if (fabs(den) >= RT_EPS) {
            double t = dot(sub(Q0, o), N) / den;
            if (t >= RT_EPS) {
                v3 P = add(o, mulk(Vv, t)), Q = sub(P, Q0);
                if (dot(Q, Q) <= R * R) {
                    c->dist = t, c->C = P, c->front = 1, c->hit = 1;
                    c->N = den < 0 ? N : neg(N);
                    return;
                }
            }
        }
    }
}

/* CR/Collision_Mesh.cu:3-17 */
static int box_hit(v3 o, v3 d, const double* mn, const double* mx) {
    v3 inv = V3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    v3 t0 = mulv(sub(V3(mn[0], mn[1], mn[2]), o), inv), t1 = mulv(sub(V3(mx[0], mx[1], mx[2]), o), inv);
    v3 tmin = V3(fmin(t0.x, t1.x), fmin(t0.y, t1.y), fmin(t0.z, t1.z));
    v3 tmax = V3(fmax(t0.x, t1.x), fmax(t0.y, t1.y), fmax(t0.z, t1.z));
    double tminmax = fmax(fmax(tmin.x, tmin.y), tmin.z), tmaxmin = fmin(fmin(tmax.x, tmax.y), tmax.z);
    return tminmax <= tmaxmin;
}

/* CR/Collision_Mesh.cu:19-63; the stored normal is rebuilt as CR/primitive.cpp:457 made it */
static int leaf_hit(v3 o, v3 d, const rt_bvh_leaf* lf, col_t* tmp) {
    v3 e1 = sub(lf->O2, lf->O1), e2 = sub(lf->O3, lf->O1);
    v3 p = cross(d, e2);
    double det = dot(e1, p);
    if (fabs(det) < RT_EPS) return 0;
    double inv = 1.0 / det;
    v3 t = sub(o, lf->O1);
    double u = dot(t, p) * inv;
    if (u < 0.0 || u > 1.0) return 0;
    v3 q = cross(t, e1);
    double v = dot(d, q) * inv;
    if (v < 0.0 || u + v > 1.0) return 0;
    double t0 = dot(e2, q) * inv;
    if (t0 <= RT_EPS) return 0;
    tmp->hit = 1;
    tmp->C = add(o, mulk(d, t0));
    v3 N = unit(cross(e1, e2));
    if (dot(d, N) < 0) tmp->front = 1, tmp->N = N;
    else tmp->front = 0, tmp->N = neg(N);
    tmp->dist = t0;
    tmp->leaf = (int)lf->preorder;
    return 1;
}

/* CR/Collision_Mesh.cu:65-80: unordered, unculled depth-first walk, left then right */
static int mesh_walk(ctx_t* cx, uint32_t mesh_tag, uint32_t ref, v3 o, v3 d, col_t* c) {
    if (ref == RT_REF_END) return 0;
    cx->st.nodes_visited++;
    if (ref & RT_REF_LEAF) {
        const rt_bvh_leaf* lf = &cx->sc->leaves[ref & ~RT_REF_LEAF];
        if (cx->log) {
            if (cx->log_n < cx->log_cap) cx->log[cx->log_n] = mesh_tag | lf->preorder;
            cx->log_n++;
        }
        cx->st.tris_tested++;
        col_t tmp = col_init();
        if (leaf_hit(o, d, lf, &tmp) && tmp.dist < c->dist) *c = tmp;
        return c->hit;
    }
    const rt_bvh_node* n = &cx->sc->nodes[ref];
    if (cx->log) {
        if (cx->log_n < cx->log_cap) cx->log[cx->log_n] = mesh_tag | n->preorder;
        cx->log_n++;
    }
    if (!box_hit(o, d, n->bmin, n->bmax)) return 0;
    int f1 = mesh_walk(cx, mesh_tag, n->left, o, d, c);
    int f2 = mesh_walk(cx, mesh_tag, n->right, o, d, c);
    return f1 || f2;
}


/* ---- Bezier surface of revolution by Newton iteration (RT_F_BEZIER_STACKED cleared) ---------------
 * The reference's live code approximates the surface by stacked cylinders (CR/Collision_Bezier.cu:
 * 200-240); its abandoned attempt (:3-198) reduced the problem to a profile curve in 2-D and refined
 * a parameter guess numerically.  This is that idea done properly, for the surface
 *     S(u, phi) = O1 + z(u) N + r(u) (cos phi Nx + sin phi Ny),   0 <= u <= 1,
 * where (z, r)(u) is the Bezier curve of ALL control points P= z r (degree = #points - 1) and
 * (N, Nx, Ny) the frame the loader stores (CR/primitive.cpp:265-271).  The ray in that frame is
 * p + t v.  One unknown, u, is found by Newton's method from BZ_SEEDS starting points (BZ_ITERS
 * steps each, u clamped to [0,1]); the other equation supplies t, chosen for conditioning:
 *   - ray mostly along the axis (vz^2 >= vx^2+vy^2): t(u) = (z(u) - pz)/vz, solve |(p+tv)_xy|^2 = r(u)^2;
 *   - ray mostly across it: t(u) = near (then far) crossing of the cylinder of radius r(u) around
 *     the axis, solve z(u) = pz + t(u) vz.  Every near-side crossing comes before every far-side
 *     one, so the far side is searched only when the near side yields nothing.
 * A candidate counts only if both equations hold to 1e-9 (relative).  The smallest
 * t > 1e-6 wins; the normal is z'(u) e_rho - r'(u) N, turned to face the ray.  The surface is open
 * (no end caps) and has no texture mapping.  No counterpart runs in the reference, so parity for this
 * intersector is GPU-versus-this-file only ("unpinned").  The polynomial is evaluated in the power
 * basis by Horner's rule; only + - * / and sqrt are used, so the device reproduces it bit for bit. */
#define BZ_SEEDS 8
#define BZ_ITERS 12
static void bezier_power_basis(const double* P, int n, double* c) { /* c[j] = C(n,j) sum_i (-1)^(j-i) C(j,i) P[i] */
    double binom[RT_BEZIER_MAX_CTRL][RT_BEZIER_MAX_CTRL];
    for (int a = 0; a <= n; a++) {
        binom[a][0] = 1.0;
        for (int b = 1; b <= a; b++) binom[a][b] = (b == a) ? 1.0 : binom[a - 1][b - 1] + binom[a - 1][b];
    }
    for (int j = 0; j <= n; j++) {
        double s = 0.0;
        for (int i = 0; i <= j; i++) s += (((j - i) & 1) ? -1.0 : 1.0) * binom[j][i] * P[i];
        c[j] = binom[n][j] * s;
    }
}
static void horner2(const double* c, int n, double u, double* val, double* der) {
    double v = c[n], dv = 0.0;
    for (int k = n - 1; k >= 0; k--) {
        dv = dv * u + v;
        v = v * u + c[k];
    }
    *val = v, *der = dv;
}
static void bezier_newton_intersect(const rt_primitive* p, v3 o, v3 d, col_t* c) {
    const int n = p->u.bezier.degree;
    if (n < 1 || n > RT_BEZIER_MAX_CTRL - 1) return;
    double cz[RT_BEZIER_MAX_CTRL], cr[RT_BEZIER_MAX_CTRL];
    bezier_power_basis(p->u.bezier.Z, n, cz);
    bezier_power_basis(p->u.bezier.R, n, cr);
    const v3 Vv = unit(d), ax = p->u.bezier.N, Nx = p->u.bezier.Nx, Ny = p->u.bezier.Ny, q = sub(o, p->u.bezier.O1);
    const double pz = dot(q, ax), px = dot(q, Nx), py = dot(q, Ny), vz = dot(Vv, ax), vx = dot(Vv, Nx), vy = dot(Vv, Ny);
    const double qa = vx * vx + vy * vy, qb = px * vx + py * vy, qc = px * px + py * py;
    const int radial = qa > vz * vz;
    double best_t = RT_BIG_DIST, best_u = 0.0;
    for (int pass = 0; pass < (radial ? 2 : 1); pass++) {
        if (pass == 1 && best_t < RT_BIG_DIST) break; /* every near-side crossing precedes every far-side one */
        const double sgn = pass == 0 ? -1.0 : 1.0;
        for (int g = 0; g < BZ_SEEDS; g++) {
            double u = (g + 0.5) / BZ_SEEDS, z, r, dz, dr;
            for (int it = 0; it < BZ_ITERS; it++) {
                horner2(cz, n, u, &z, &dz);
                horner2(cr, n, u, &r, &dr);
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
            horner2(cz, n, u, &z, &dz);
            horner2(cr, n, u, &r, &dr);
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
    if (best_t < RT_BIG_DIST) {
        double z, r, dz, dr;
        horner2(cz, n, best_u, &z, &dz);
        horner2(cr, n, best_u, &r, &dr);
        double qx = px + best_t * vx, qy = py + best_t * vy;
        v3 erho = unit(add(mulk(Nx, qx), mulk(Ny, qy)));
        v3 Nn = unit(sub(mulk(erho, dz), mulk(ax, dr)));
        c->hit = 1, c->dist = best_t, c->C = add(o, mulk(Vv, best_t));
        c->front = dot(Vv, Nn) < 0;
        c->N = c->front ? Nn : neg(Nn);
    }
}
void orc_bezier_power_basis(const double* P, int n, double* c) { bezier_power_basis(P, n, c); }

/* CR/Collision.cu:151-291 (+ CR/primitive.cpp:109-138,159-175,192-227 for the CPU twins) */
static int prim_intersect(ctx_t* cx, int pi, v3 o, v3 d, col_t* c) {
    const rt_primitive* p = &cx->sc->primitives[pi];
    switch (p->type) {
        case RT_PRIM_SPHERE: {
            v3 Vv = unit(d), P = sub(o, p->u.sphere.O);
            double b = -dot(P, Vv), R = p->u.sphere.R;
            double det = b * b - dot(P, P) + R * R;
            if (det > RT_EPS) {
                det = sqrt(det);
                double x1 = b - det, x2 = b + det;
                if (x2 < RT_EPS) return 0;
                c->front = (x1 > RT_EPS);
                c->dist = c->front ? x1 : x2;
                c->C = add(o, mulk(Vv, c->dist));
                c->N = unit(sub(c->C, p->u.sphere.O));
                if (!c->front) c->N = neg(c->N);
                c->hit = 1;
            }
            break;
        }
        case RT_PRIM_PLANE: {
            v3 Vv = unit(d), N = unit(p->u.plane.N);
            double den = dot(N, Vv);
            if (fabs(den) >= RT_EPS) {
                double t = dot(sub(mulk(N, p->u.plane.R), o), N) / den;
                if (t >= RT_EPS) {
                    c->dist = t;
                    c->C = add(o, mulk(Vv, t));
                    c->front = den < 0;
                    c->N = c->front ? N : neg(N);
                    c->hit = 1;
                }
            }
            break;
        }
        case RT_PRIM_SQUARE: {
            v3 Vv = unit(d), Dx = p->u.square.Dx, Dy = p->u.square.Dy, O = p->u.square.O;
            v3 N = unit(cross(Dx, Dy));
            double den = dot(N, Vv);
            if (fabs(den) < RT_EPS) return 0;
            double t = dot(sub(O, o), N) / den;
            if (t < RT_EPS) return 0;
            v3 P = add(o, mulk(Vv, t));
            double lx = dot(Dx, Dx), ly = dot(Dy, Dy);
            double x2 = fabs(dot(sub(P, O), Dx)), y2 = fabs(dot(sub(P, O), Dy));
            if (x2 > lx || y2 > ly) return 0;
            c->dist = t, c->front = den < 0, c->C = P, c->N = c->front ? N : neg(N), c->hit = 1;
            break;
        }
        case RT_PRIM_CYLINDER:
            cylinder_intersect(p->u.cylinder.O1, p->u.cylinder.O2, p->u.cylinder.R, o, d, c);
            break;
        case RT_PRIM_BEZIER: { /* CR/Collision_Bezier.cu:200-240, `closest` bookkeeping included */
            if (!(cx->flags & RT_F_BEZIER_STACKED)) {
                bezier_newton_intersect(p, o, d, c);
                break;
            }
            double closest = 1e10;
            for (int i = 0; i < p->u.bezier.degree; i++) {
                v3 axis = unit(sub(p->u.bezier.O2, p->u.bezier.O1)), p1, p2;
                double R;
                if (i == 0) {
                    p1 = p->u.bezier.O1;
                    p2 = add(p->u.bezier.O1, mulk(axis, p->u.bezier.Z[i]));
                    R = p->u.bezier.R_c;
                } else {
                    p1 = add(p->u.bezier.O1, mulk(axis, p->u.bezier.Z[i - 1]));
                    p2 = add(p->u.bezier.O1, mulk(axis, p->u.bezier.Z[i]));
                    R = p->u.bezier.R[i];
                }
                col_t tmp = col_init();
                cylinder_intersect(p1, p2, R, o, d, &tmp);
                if (tmp.hit && tmp.dist < closest) {
                    closest = c->dist;
                    *c = tmp;
                }
            }
            break;
        }
        case RT_PRIM_TRIANGLE: { /* CR/Collision.cu:110-149,267-282 */
            v3 dd = (cx->flags & RT_F_MESH_RAW_DIR) ? d : unit(d);
            v3 e1 = sub(p->u.triangle.O2, p->u.triangle.O1), e2 = sub(p->u.triangle.O3, p->u.triangle.O1);
            v3 h = cross(dd, e2);
            double a = dot(e1, h);
            if (fabs(a) < RT_EPS) return 0;
            double f = 1.0 / a;
            v3 s = sub(o, p->u.triangle.O1);
            double u = f * dot(s, h);
            if (u < 0.0 || u > 1.0) return 0;
            v3 q = cross(s, e1);
            double v = f * dot(dd, q);
            if (v < 0.0 || u + v > 1.0) return 0;
            double t = f * dot(e2, q);
            if (t > RT_EPS) {
                c->dist = t;
                c->C = add(o, mulk(dd, t));
                c->N = unit(cross(e1, e2));
                c->hit = 1;
                c->front = dot(dd, c->N) < 0;
            }
            break;
        }
        case RT_PRIM_MESH: { /* CR/Collision_Mesh.cu:82-89 */
            v3 dd = (cx->flags & RT_F_MESH_RAW_DIR) ? d : unit(d);
            int mi = p->u.mesh.mesh;
            if (mi >= 0) mesh_walk(cx, (uint32_t)mi << 24, cx->sc->meshes[mi].root, o, dd, c);
            break;
        }
        default:
            break;
    }
    if (c->hit) c->prim = pi;
    return c->hit;
}

/* CR/Scene.cu:3-21 (tie rule :14) and CR/Scene.cpp:27-38 (tie rule :33) */
static col_t scene_intersect(ctx_t* cx, v3 o, v3 d) {
    col_t best = col_init();
    for (int i = 0; i < cx->sc->n_primitives; i++) {
        col_t tmp = col_init();
        if (prim_intersect(cx, i, o, d, &tmp)) {
            int take = (cx->flags & RT_F_TIE_LATER_WINS) ? (tmp.dist - best.dist < RT_EPS) : (tmp.dist < best.dist);
            if (take) best = tmp;
        }
    }
    return best;
}

/* CR/bmp.cpp:133-150 (double taps) and CR/Primitive.cu:85-115 (8-bit truncated taps, floorf, weights of the transposed copy) */
static v3 smooth_color(ctx_t* cx, const rt_texture* tx, double u, double v) {
    cx->st.texture_fetches++;
    int H = tx->height, W = tx->width;
    int u8 = (cx->flags & RT_F_TEX_U8_TAPS) != 0;
    double U = (u - (u8 ? (double)floorf((float)u) : floor(u))) * H;
    double Vv = (v - (u8 ? (double)floorf((float)v) : floor(v))) * W;
    int U1 = u8 ? (int)floorf((float)(U - RT_EPS)) : (int)floor(U - RT_EPS), U2 = U1 + 1;
    int V1 = u8 ? (int)floorf((float)(Vv - RT_EPS)) : (int)floor(Vv - RT_EPS), V2 = V1 + 1;
    double ru = U2 - U, rv = V2 - Vv;
    if (U1 < 0) U1 = H - 1;
    if (U2 == H) U2 = 0;
    if (V1 < 0) V1 = W - 1;
    if (V2 == W) V2 = 0;
    /* indices stay in range for finite u, v; clamp defensively so a NaN cannot read outside */
    if (U1 < 0 || U1 >= H) U1 = 0;
    if (U2 < 0 || U2 >= H) U2 = 0;
    if (V1 < 0 || V1 >= W) V1 = 0;
    if (V2 < 0 || V2 >= W) V2 = 0;
    const uint8_t* t11 = tx->rgb + ((size_t)U1 * W + V1) * 3;
    const uint8_t* t12 = tx->rgb + ((size_t)U1 * W + V2) * 3;
    const uint8_t* t21 = tx->rgb + ((size_t)U2 * W + V1) * 3;
    const uint8_t* t22 = tx->rgb + ((size_t)U2 * W + V2) * 3;
    /* Weights of the taps (U1,V2) and (U2,V1).  CR/bmp.cpp:147-148: ima[U1][V2] * rat_U * (1 - rat_V) and
     * ima[U2][V1] * (1 - rat_U) * rat_V.  The .cu fetch reads a transposed copy (texture[col * H + row] = ima[row][col],
     * CR/Renderer_CUDA.cpp:149-152), so CR/Primitive.cu:109 is ima[U2][V1] * rat_U * (1 - rat_V) and :110 is
     * ima[U1][V2] * (1 - rat_U) * rat_V: the two taps carry each other's weights there. */
    double w1a = ru, w1b = rv, w2a = ru, w2b = 1 - rv, w3a = 1 - ru, w3b = rv, w4a = 1 - ru, w4b = 1 - rv;
    if (u8) w2a = 1 - ru, w2b = rv, w3a = ru, w3b = 1 - rv;
    double out[3];
    for (int k = 0; k < 3; k++) {
        if (u8) {
            double s = 0;
            s += (double)(uint8_t)((double)(uint8_t)(t11[k] * w1a) * w1b);
            s += (double)(uint8_t)((double)(uint8_t)(t12[k] * w2a) * w2b);
            s += (double)(uint8_t)((double)(uint8_t)(t21[k] * w3a) * w3b);
            s += (double)(uint8_t)((double)(uint8_t)(t22[k] * w4a) * w4b);
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
    return V3(out[0], out[1], out[2]);
}

/* test hook: one bilinear fetch in either tap mode */
void orc_smooth_color(const rt_texture* tx, uint32_t flags, double u, double v, double out[3]) {
    ctx_t cx;
    memset(&cx, 0, sizeof cx);
    cx.flags = flags;
    v3 c = smooth_color(&cx, tx, u, v);
    out[0] = c.x, out[1] = c.y, out[2] = c.z;
}

/* CR/Primitive.cu:16-83 and CR/primitive.cpp:140-147,177-181,229-233 */
static v3 texture_color(ctx_t* cx, const col_t* c) {
    const rt_primitive* p = &cx->sc->primitives[c->prim];
    const rt_texture* tx = &cx->sc->textures[p->material.texture];
    double u, v;
    switch (p->type) {
        case RT_PRIM_SPHERE: {
            v3 I = unit(sub(c->C, p->u.sphere.O));
            double a = acos(-dot(I, p->u.sphere.De));
            double b = acos(fmin(fmax(dot(I, p->u.sphere.Dc) / sin(a), -1.0), 1.0));
            u = a / M_PI, v = b / 2 / M_PI;
            if (dot(I, cross(p->u.sphere.Dc, p->u.sphere.De)) < 0) v = 1 - v;
            break;
        }
        case RT_PRIM_PLANE:
            u = dot(c->C, p->u.plane.Dx) / dot(p->u.plane.Dx, p->u.plane.Dx);
            v = dot(c->C, p->u.plane.Dy) / dot(p->u.plane.Dy, p->u.plane.Dy);
            break;
        case RT_PRIM_SQUARE: {
            v3 b = (cx->flags & RT_F_SQUARE_UV_NO_ORIGIN) ? c->C : sub(c->C, p->u.square.O);
            u = dot(b, p->u.square.Dx) / dot(p->u.square.Dx, p->u.square.Dx) / 2 + 0.5;
            v = dot(b, p->u.square.Dy) / dot(p->u.square.Dy, p->u.square.Dy) / 2 + 0.5;
            break;
        }
        case RT_PRIM_CYLINDER: {
            v3 O1 = p->u.cylinder.O1, O2 = p->u.cylinder.O2;
            if (!c->front) { /* body */
                v3 cp = sub(c->C, O1);
                double ang = atan2(cp.z, cp.x);
                if (ang < 0) ang += 2 * M_PI;
                u = ang / (2 * M_PI);
                v = dot(sub(c->C, O1), unit(sub(O2, O1))) / module(sub(O2, O1));
            } else { /* caps */
                v3 cc = module(sub(c->C, O1)) < module(sub(c->C, O2)) ? O1 : O2;
                v3 l = sub(c->C, cc);
                u = l.x / (2.0 * p->u.cylinder.R) + 0.5;
                v = l.z / (2.0 * p->u.cylinder.R) + 0.5;
            }
            break;
        }
        default: /* Bezier / triangle / mesh: Color() on the CPU (CR/primitive.cpp:290-295,339-342,372-375) */
            return V3(0, 0, 0);
    }
    return smooth_color(cx, tx, u, v);
}

/* one shadow ray: is `target` visible from C?  CR/Shade.cu:36-46,64-75,94-105, CR/Light.cpp:35-40,67-78 */
static int shadow_clear(ctx_t* cx, v3 C, v3 target, int crashed, int light_prim) {
    cx->st.rays_shadow++;
    v3 Vv = sub(target, C);
    double dist = module(Vv);
    for (int j = 0; j < cx->sc->n_primitives; j++) {
        if ((cx->flags & RT_F_SHADOW_SELF_SKIP) && j == crashed) continue;
        if (j == light_prim) continue;
        col_t tmp = col_init();
        prim_intersect(cx, j, C, Vv, &tmp);
        if (dist - tmp.dist > RT_EPS) return 0;
    }
    return 1;
}

/* CR/Shade.cu:23-113 */
static double caln_shade(ctx_t* cx, rng_t* g, v3 C, int crashed, const rt_light* L) {
    int n = cx->shade_n;
    double shade = 0.0;
    switch (L->type) {
        case RT_LIGHT_POINT:
            return shadow_clear(cx, C, L->O, crashed, -1) ? 1.0 : 0.0;
        case RT_LIGHT_SQUARE:
            for (int i = 0; i < n; i++) {
                double a = (double)(2 * frand(g)) - 1.0; /* first draw scales Dx (CR/Shade.cu:5) */
                double b = (double)(2 * frand(g)) - 1.0;
                v3 ro = add(add(L->O, mulk(L->Dx, a)), mulk(L->Dy, b));
                shade += shadow_clear(cx, C, ro, crashed, L->light_primitive);
            }
            return shade / n;
        case RT_LIGHT_SPHERE:
            for (int i = 0; i < n; i++) { /* CR/Shade.cu:8-21 */
                double z = 2.0 * frand(g) - 1.0;
                double th = frand(g) * 2.0 * M_PI;
                double r = sqrt(1.0 - z * z), sn, cs;
                det_sincos(th, &sn, &cs);
                double x = r * cs, y = r * sn;
                v3 ro = add(L->O, V3(L->R * x, L->R * y, L->R * z));
                shade += shadow_clear(cx, C, ro, crashed, L->light_primitive);
            }
            return shade / n;
    }
    return 0.0;
}

/* CR/Scene_Material.cu:29-66, CR/raytracer.cpp:26-68 */
static v3 caln_diffusion(ctx_t* cx, rng_t* g, const col_t* c) {
    const rt_primitive* p = &cx->sc->primitives[c->prim];
    cx->st.shaded_hits++;
    v3 color = p->material.color;
    if (p->material.texture >= 0) color = mulv(color, texture_color(cx, c));
    v3 ret = V3(0, 0, 0);
    for (int i = 0; i < cx->sc->n_lights; i++) {
        const rt_light* L = &cx->sc->lights[i];
        double shade = caln_shade(cx, g, c->C, c->prim, L);
        if (shade < RT_EPS) continue;
        v3 R = unit(sub(L->O, c->C));
        double dt = dot(R, c->N);
        if (dt > RT_EPS) {
            if (p->material.diff > RT_EPS) {
                double diff = p->material.diff * dt * shade;
                ret = add(ret, mulk(mulv(color, L->color), diff));
            }
            if (p->material.spec > RT_EPS) {
                double spec = p->material.spec * pow(dt, RT_SPEC_POWER) * shade;
                ret = add(ret, mulk(mulv(color, L->color), spec));
            }
        }
    }
    return ret;
}

static v3 trace_ray(ctx_t* cx, rng_t* g, v3 o, v3 d, int depth);

/* CR/Scene_Material.cu:68-99 (one fuzzy sample; the CPU's 16-sample cone at
 * CR/raytracer.cpp:81-113 draws from std::random_device and is not reproducible) */
static v3 caln_reflection(ctx_t* cx, rng_t* g, const col_t* c, v3 rv, int dep) {
    const rt_primitive* p = &cx->sc->primitives[c->prim];
    rv = reflect(rv, c->N);
    if (p->material.drefl < RT_EPS || dep > RT_MAX_DREFL_DEP)
        return mulk(mulv(trace_ray(cx, g, c->C, rv, dep + 1), p->material.color), p->material.refl);
    v3 bx = vertical(rv);
    v3 by = unit(cross(rv, bx));
    bx = mulk(bx, p->material.drefl);
    by = mulk(by, p->material.drefl);
    double x = frand(g), y = frand(g); /* CR/Scene_Material.cu:19-27 */
    double sn, cs;
    det_sincos(y, &sn, &cs);
    double bxk = x * cs, byk = x * sn;
    v3 fz = add(add(rv, mulk(bx, bxk)), mulk(by, byk));
    return mulk(mulv(trace_ray(cx, g, c->C, fz, dep + 1), p->material.color), p->material.refl);
}

/* CR/Scene_Material.cu:101-118, CR/raytracer.cpp:116-134 */
static v3 caln_refraction(ctx_t* cx, rng_t* g, const col_t* c, v3 rv, int dep) {
    const rt_primitive* p = &cx->sc->primitives[c->prim];
    double n = p->material.rindex;
    if (c->front) n = 1.0 / n;
    rv = refract(rv, c->N, n);
    v3 rcol = trace_ray(cx, g, c->C, rv, dep + 1);
    if (c->front) return mulk(rcol, p->material.refr);
    v3 ab = mulk(p->material.absor, -c->dist);
    v3 tr = V3(exp(ab.x), exp(ab.y), exp(ab.z));
    return mulk(mulv(rcol, tr), p->material.refr);
}

/* CR/Scene.cu:23-58, CR/raytracer.cpp:136-178 */
static v3 trace_ray(ctx_t* cx, rng_t* g, v3 o, v3 d, int depth) {
    v3 color = V3(0, 0, 0);
    if (depth > RT_MAX_DEPTH) return color;
    if (depth == 1) cx->st.rays_primary++;
    else cx->st.rays_secondary++;
    col_t c = scene_intersect(cx, o, d);
    if (c.hit) {
        const rt_primitive* p = &cx->sc->primitives[c.prim];
        if (p->is_light) color = p->material.color;
        else {
            if (p->material.diff > RT_EPS || p->material.spec > RT_EPS) color = add(color, caln_diffusion(cx, g, &c));
            if (p->material.refl > RT_EPS) color = add(color, caln_reflection(cx, g, &c, d, depth));
            if (p->material.refr > RT_EPS) color = add(color, caln_refraction(cx, g, &c, d, depth));
        }
    } else {
        v3 cdir = unit(d);
        double t = 0.5 * (((cx->flags & RT_F_SKY_Y) ? cdir.y : cdir.z) + 1.0);
        color = add(mulk(cx->sc->background_bottom, 1.0 - t), mulk(cx->sc->background_top, t));
    }
    if (color.x > 1) color.x = 1; /* Color::Confine, CR/color.cpp:53-57 */
    if (color.y > 1) color.y = 1;
    if (color.z > 1) color.z = 1;
    return color;
}

/* pixel -> camera pixel, CR/RayTracer.cu:55-56 */
static void map_pixel(const ctx_t* cx, int sx, int sy, int tex_w, int tex_h, int* x, int* y) {
    if (cx->flags & RT_F_PIXEL_MAP_TRUNC) {
        *x = (int)((double)sx / tex_w * cx->cam_w);
        *y = (int)((double)sy / tex_h * cx->cam_h);
    } else {
        *x = (int)(((int64_t)sx * cx->cam_w) / tex_w);
        *y = (int)(((int64_t)sy * cx->cam_h) / tex_h);
    }
}

/* CR/camera.cpp:57-59 / CR/RayTracer.cu:60 */
static v3 emit(const ctx_t* cx, double x, double y) {
    const rt_camera* c = &cx->sc->camera;
    return add(add(c->N, mulk(c->Dy, 2 * y / cx->cam_h - 1)), mulk(c->Dx, 2 * x / cx->cam_w - 1));
}

/* CR/RayTracer.cu:31-38 */
static void depth_of_field(const ctx_t* cx, rng_t* g, const rt_settings* s, v3 focal, v3* o, v3* d) {
    double rx = (frand(g) - 0.5) * s->aperture / 10.0;
    double ry = (frand(g) - 0.5) * s->aperture / 10.0;
    v3 off = add(mulk(cx->sc->camera.Dx, rx), mulk(cx->sc->camera.Dy, ry));
    *o = add(*o, off);
    *d = unit(sub(focal, *o));
}

static int to_u8(double c) {
    double v = c * 255;
    if (!(v >= 0)) return 0; /* negative or NaN */
    if (v > 255) return 255;
    return (int)v;
}

/* CR/RayTracer.cu:40-94 for one pixel; returns the averaged colour */
static v3 render_pixel(ctx_t* cx, const rt_settings* s, int sx, int sy, int tex_w, int tex_h) {
    int x, y;
    map_pixel(cx, sx, sy, tex_w, tex_h, &x, &y);
    const rt_camera* cam = &cx->sc->camera;
    uint32_t pixel = (uint32_t)sy * (uint32_t)tex_w + (uint32_t)sx;
    v3 dc = unit(emit(cx, x, y));
    v3 focal = add(cam->O, mulk(dc, s->focal_distance));
    v3 color = V3(0, 0, 0);
    int norm = (cx->flags & RT_F_NORMALIZE_PRIMARY) != 0;
    if (s->spp == 1) {
        rng_t g;
        rng_init(&g, s->seed, pixel, 0);
        v3 o = cam->O, d = norm ? dc : emit(cx, x, y);
        if (s->dof_enabled) depth_of_field(cx, &g, s, focal, &o, &d);
        color = trace_ray(cx, &g, o, d, 1);
    } else {
        for (int i = 0; i < s->spp; i++) {
            rng_t g;
            rng_init(&g, s->seed, pixel, (uint32_t)i);
            double u = x + (2.0f * frand(&g) - 1.0f);
            double v = y + (2.0f * frand(&g) - 1.0f);
            v3 o = cam->O, d = emit(cx, u, v);
            if (norm) d = unit(d);
            if (s->dof_enabled) depth_of_field(cx, &g, s, focal, &o, &d);
            color = add(color, trace_ray(cx, &g, o, d, 1));
        }
        color = divk(color, s->spp);
    }
    return color;
}

static void stats_add(rt_stats* a, const rt_stats* b) {
    a->rays_primary += b->rays_primary, a->rays_secondary += b->rays_secondary, a->rays_shadow += b->rays_shadow;
    a->nodes_visited += b->nodes_visited, a->tris_tested += b->tris_tested, a->shaded_hits += b->shaded_hits;
    a->texture_fetches += b->texture_fetches, a->pixels += b->pixels;
    a->rays_shadow_traced += b->rays_shadow;  /* every shadow ray is traced here, as in the reference */
}

static int ctx_setup(ctx_t* cx, const rt_scene_desc* sc, const rt_settings* s, int cam_w, int cam_h) {
    memset(cx, 0, sizeof(*cx));
    cx->sc = sc, cx->flags = s->flags;
    cx->cam_w = cam_w > 0 ? cam_w : sc->camera.W;
    cx->cam_h = cam_h > 0 ? cam_h : sc->camera.H;
    cx->shade_n = (int)(16 * sc->camera.shade_quality);
    if (cx->shade_n < 1 || s->spp < 1) return RT_ERR_INVALID;
    return RT_OK;
}

/* Render a rectangle.  rgba: full tex_w*tex_h*4 image (may be NULL); rgb_f64: w*h*3 doubles of
 * the rectangle (may be NULL); n_threads <= 0 = all OpenMP threads. */
int orc_render(const rt_scene_desc* sc, const rt_settings* s, int tex_w, int tex_h, int cam_w, int cam_h, int x0, int y0,
               int w, int h, uint8_t* rgba, double* rgb_f64, rt_stats* stats, int n_threads) {
    ctx_t base;
    int rc = ctx_setup(&base, sc, s, cam_w, cam_h);
    if (rc) return rc;
    rt_stats total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        ctx_t cx = base;
#pragma omp for schedule(dynamic, 1)
        for (int j = 0; j < h; j++) {
            for (int i = 0; i < w; i++) {
                int sx = x0 + i, sy = y0 + j;
                v3 c = render_pixel(&cx, s, sx, sy, tex_w, tex_h);
                cx.st.pixels++;
                if (rgb_f64) {
                    double* o = rgb_f64 + ((size_t)j * w + i) * 3;
                    o[0] = c.x, o[1] = c.y, o[2] = c.z;
                }
                if (rgba) {
                    uint8_t* o = rgba + ((size_t)sy * tex_w + sx) * 4;
                    o[0] = (uint8_t)to_u8(c.x), o[1] = (uint8_t)to_u8(c.y), o[2] = (uint8_t)to_u8(c.z), o[3] = 255;
                }
            }
        }
#pragma omp critical
        stats_add(&total, &cx.st);
    }
    if (stats) {
        double k = stats->kernel_ms, t = stats->total_ms;
        *stats = total;
        stats->kernel_ms = k, stats->total_ms = t;
    }
    return RT_OK;
}

/* Primary ray only (sample 0, no jitter, no depth of field); outputs as rt_trace_primary. */
int orc_trace_primary(const rt_scene_desc* sc, const rt_settings* s, int tex_w, int tex_h, int cam_w, int cam_h, int x0,
                      int y0, int w, int h, int32_t* hit_prim, double* hit_t, int32_t* hit_leaf, uint32_t* visit_log,
                      uint32_t* visit_count, int visit_cap) {
    ctx_t cx;
    rt_settings s1 = *s;
    s1.spp = 1;
    int rc = ctx_setup(&cx, sc, &s1, cam_w, cam_h);
    if (rc) return rc;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            int x, y;
            size_t k = (size_t)j * w + i;
            map_pixel(&cx, x0 + i, y0 + j, tex_w, tex_h, &x, &y);
            v3 d = emit(&cx, x, y);
            if (cx.flags & RT_F_NORMALIZE_PRIMARY) d = unit(d);
            if (visit_log) cx.log = visit_log + k * (size_t)visit_cap, cx.log_cap = (uint32_t)visit_cap, cx.log_n = 0;
            col_t c = scene_intersect(&cx, sc->camera.O, d);
            if (hit_prim) hit_prim[k] = c.hit ? c.prim : -1;
            if (hit_t) hit_t[k] = c.dist;
            if (hit_leaf) hit_leaf[k] = c.leaf;
            if (visit_count) visit_count[k] = cx.log_n;
        }
    return RT_OK;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
