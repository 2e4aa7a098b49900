// Device-side double3 arithmetic and the counter-based RNG.
//
// Every expression keeps the operation order of the reference (helper_math.h:546-570,833-853,
// 1138-1158,1302-1305,1613-1616,1789-1806,1873-1900 for the CUDA path; CR/vector3.cpp:14-173 for
// the CPU path) and this file is compiled with -ffp-contract=off so no multiply-add is fused:
// hit distances and hit ids are then bit-identical with a CPU evaluation of the same formulas.
// normalize() divides by sqrt() (CR/vector3.cpp:128-131) instead of multiplying by rsqrt()
// (helper_math.h:1681-1685), whose CUDA implementation is not correctly rounded.
#ifndef RT_MATH_HPP
#define RT_MATH_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_amd.h"

struct d3 {
    double x, y, z;
};
#define DEV static __device__ __forceinline__

DEV d3 D3(double x, double y, double z) { return d3{x, y, z}; }
DEV d3 D3(const rt_vec3& v) { return d3{v.x, v.y, v.z}; }
DEV d3 D3(const double (&v)[3]) { return d3{v[0], v[1], v[2]}; }
DEV d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV d3 operator-(d3 a) { return d3{-a.x, -a.y, -a.z}; }
DEV d3 operator*(d3 a, double k) { return d3{a.x * k, a.y * k, a.z * k}; }
DEV d3 operator/(d3 a, double k) { return d3{a.x / k, a.y / k, a.z / k}; }
DEV d3 mulv(d3 a, d3 b) { return d3{a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV d3 cross(d3 a, d3 b) { return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV double module(d3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
DEV d3 unit(d3 a) { return a / module(a); }
// atan2 for texture coordinates: the quotient of the smaller by the larger magnitude, folded once more around
// tan(pi/8), then the odd polynomial fdlibm's atan uses below 7/16 (its coefficients), and the octant put back.
// A few ulp from a correctly rounded atan2 - texture coordinates are on the colour path - at a fraction of the
// instructions and registers of the general library routine, which this kernel cannot afford.
DEV double rt_atan2(double y, double x) {
    const double ax = fabs(x), ay = fabs(y), mx = fmax(ax, ay), mn = fmin(ax, ay);
    const double a = mx > 0.0 ? mn / mx : 0.0;
    const bool big = a > 0.41421356237309503;
    const double t = big ? (a - 1.0) / (a + 1.0) : a;
    const double z = t * t, w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                      w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                      w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    double r = t - t * (s1 + s2);
    if (big) r = 7.85398163397448278999e-01 + (r + 3.06161699786838301793e-17);
    if (ay > ax) r = 1.57079632679489655800e+00 - r;
    if (x < 0.0) r = 3.14159265358979311600e+00 - r;
    return y < 0.0 ? -r : r;
}
// acos and exp of the colour path, written out (fdlibm's rational / polynomial kernels, an ulp or two from
// correctly rounded) for the same reason as rt_atan2: the general library routines are too large for this kernel.
DEV double rt_asin_kernel(double z) {  // (asin(x) - x) / x for z = x*x <= 1/4
    const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 +
                     z * (-4.00555345006794114027e-02 + z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
    const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 + z * (-6.88283971605453293030e-01 +
                     z * 7.70381505559019352791e-02)));
    return p / q;
}
DEV double rt_acos(double x) {  // |x| <= 1
    const double ax = fabs(x);
    if (ax < 0.5) return 1.57079632679489655800e+00 - (x + x * rt_asin_kernel(x * x)) + 6.12323399573676603587e-17;
    const double z = fmax((1.0 - ax) * 0.5, 0.0), s = sqrt(z);
    const double a = 2.0 * (s + s * rt_asin_kernel(z));
    return x > 0.0 ? a : 3.14159265358979311600e+00 - a;
}
DEV double rt_exp(double x) {
    if (x < -745.0) return 0.0;
    if (!(x <= 709.0)) return x != x ? x : __builtin_inf();
    const double k = floor(x * 1.44269504088896338700e+00 + 0.5);
    const double hi = x - k * 6.93147180369123816490e-01, lo = k * 1.90821492927058770002e-10;
    const double r = hi - lo, t = r * r;
    const double c = r - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 +
                     t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return ldexp(y, (int)k);
}
DEV d3 vertical(d3 a) {
    d3 r = cross(a, D3(0, 0, 1));
    if (fabs(r.x) < RT_EPS && fabs(r.y) < RT_EPS && fabs(r.z) < RT_EPS) return D3(1, 0, 0);
    return unit(r);
}
DEV d3 reflect(d3 i, d3 n) { return i - n * (2 * dot(n, i)); }
DEV d3 refract(d3 V, d3 N, double n) {
    V = unit(V);
    double cosI = -dot(N, V), cosT2 = 1 - (n * n) * (1 - cosI * cosI);
    if (cosT2 > RT_EPS) return V * n + N * (n * cosI - sqrt(cosT2));
    return reflect(V, N);
}


// sin/cos of the angles that steer rays (sphere-light samples CR/Shade.cu:8-21, fuzzy reflection
// CR/Scene_Material.cu:19-27).  An explicit polynomial (Cody-Waite reduction by pi/2 + the fdlibm
// kernel polynomials) instead of the device math library: the same expressions evaluated in the
// same order on any IEEE-754 machine give the same bits, so ray directions - and with them every
// visibility decision - can be reproduced exactly by a CPU check.  About one ulp for 0 <= x <= 2*pi.
DEV void det_sincos(double x, double& sn, double& cs) {
    int k = (int)(x * 6.36619772367581382433e-01 + 0.5);
    double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    double z = r * r;
    double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    double s = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    double c = 1.0 - (0.5 * z - z * pc);
    switch (k & 3) {
        case 0: sn = s, cs = c; break;
        case 1: sn = c, cs = -s; break;
        case 2: sn = -s, cs = -c; break;
        default: sn = -c, cs = s; break;
    }
}

// ---- Philox4x32-10, counter = (pixel, sample, block, 0), key = seed ----------------------
// Replaces the curand state of CR/RayTracer.cu:9-21 (shared memory written by another launch,
// indexed by threadIdx.x only).  Draw i of a (pixel, sample) is word i%4 of block i/4, so the
// image does not depend on how pixels are spread over lanes, tiles or GPUs.
DEV void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // both halves of each product from one 32 x 32 -> 64 bit multiply (integer multiplies run at a quarter of the vector rate)
        const unsigned long long p0 = (unsigned long long)c[0] * 0xD2511F53ull, p1 = (unsigned long long)c[2] * 0xCD9E8D57ull;
        uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        uint32_t n0 = h1 ^ c[1] ^ k0, n2 = h0 ^ c[3] ^ k1;
        c[0] = n0, c[1] = l1, c[2] = n2, c[3] = l0;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
}
// uniform float in (0, 1] from 32 random bits: curand_uniform's mapping, which frand() returns
DEV float u32_to_unit(uint32_t x) { return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }

// Draws always come in pairs at even indices (jitter u,v; aperture x,y; light sample a,b;
// blur x,y), so both values of a pair live in one Philox block.
DEV void frand_pair(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t draw, float& a, float& b) {
    uint32_t c[4] = {pixel, sample, draw >> 2, 0u};
    philox4x32_10(c, k0, k1);
    bool hi = (draw & 2u) != 0;
    a = u32_to_unit(hi ? c[2] : c[0]);
    b = u32_to_unit(hi ? c[3] : c[1]);
}

// ---- wave64 helpers ------------------------------------------------------------------------
DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
DEV int rank_in(unsigned long long mask) {  // number of set bits of `mask` below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
DEV double shfl_d(double v, int src) { return __shfl(v, src, 64); }
DEV d3 shfl_d3(d3 v, int src) { return d3{__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64)}; }
// Bitwise OR over the wave, the same value in every lane and held in scalar registers.  Call it with all 64 lanes active
// (lanes with nothing to contribute pass 0): the exchanges go through fixed partner lanes, and a partner that is switched
// off passes nothing on.  Three exchanges inside each quad / half row / row of 16 lanes leave every lane with the OR of its
// row; the four rows are read out by lane.
DEV uint32_t wave_or_u32(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm:[2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);  // row_mirror
    return ((uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16)) |
           ((uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48));
}
DEV unsigned long long wave_or_u64(unsigned long long v) {
    return ((unsigned long long)wave_or_u32((uint32_t)(v >> 32)) << 32) | wave_or_u32((uint32_t)v);
}
DEV unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#endif
