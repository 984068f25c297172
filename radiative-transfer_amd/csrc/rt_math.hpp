// rt_math.hpp -- fp64 elementary functions shaped for the walk kernel (gfx950).
//
// The two transcendental calls of a loop iteration have very restricted arguments:
//   step  = -ln(xi)            xi = k / 2^52 in [0, 1)          (TakeStep, MC code.py:496)
//   cos(phi), phi = 2*pi*xi    xi in [0, 1)                     (RotateIntoLabFrame, MC code.py:615)
// so neither needs the general-purpose library routines (argument reduction for huge angles,
// subnormal handling, ...).  Both functions below are accurate to a few 1e-16 ABSOLUTE; the parity
// tests bound the effect.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef RT_ABLATION
#include "../../tools/ablation/ablation.hpp"  // timing experiments; such a build reports another ABI version
#else
namespace rt {
struct Ablation {  // the product's values of the switches the timing experiments vary
    static constexpr int kPhiloxRounds = 10;
    static constexpr bool kUniform53 = false, kPlainFma = false, kSqrtGeneral = false, kDivGeneral = false, kWaveTrace = false;
    static constexpr int kAbiTag = 0;
    static bool chunk_rule(unsigned&, unsigned&) { return false; }
    static int scatter_quorum(int product) { return product; }
    static int interleave_min(int product) { return product; }
    static void tail_rule(unsigned&, unsigned&) {}
    static bool mix_rule(double&, double&) { return false; }
};
}  // namespace rt
#endif

namespace rt {

// ---------------------------------------------------------------------------
// ln(x) for a normal x in (0, 1]; ln(0) = -inf.
//   x = 2^e * m, m in [1,2);  j = top 7 fraction bits of m;  r = m * inv_c[j] - 1, |r| < 2^-8
//   ln x = e ln2 + L[j] + log1p(r),   L[j] = -ln(inv_c[j]) for the ROUNDED inv_c[j]
// The 128-entry table {inv_c, L} (2 KiB) is staged in LDS once per workgroup.
// ---------------------------------------------------------------------------
// a*b + c with the constant addend read in place (VOP3 v_fma_f64).  The compiler's choice for a Horner
// step is the two-address v_fmac_f64, which first copies the coefficient into the destination: one extra
// 4-cycle v_mov_b64 per step.
__device__ __forceinline__ double fma_step(double a, double b, double c) {
    if constexpr (Ablation::kPlainFma) return fma(a, b, c);
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

constexpr int kLogTableEntries = 128;

struct LogTableEntry {
    double inv_c;
    double l;
};

__device__ __forceinline__ double log_unit(double x, const LogTableEntry* __restrict__ lds_table) {
    const uint32_t hi = (uint32_t)__double2hiint(x);
    const uint32_t lo = (uint32_t)__double2loint(x);
    const int e = (int)(hi >> 20) - 1023;
    const uint32_t j = (hi >> 13) & 127u;
    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), (int)lo);
    const LogTableEntry t = lds_table[j];
    const double r = fma(m, t.inv_c, -1.0);
    double p = fma_step(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma_step(r, p, 1.0 / 5.0);
    p = fma_step(r, p, -1.0 / 4.0);
    p = fma_step(r, p, 1.0 / 3.0);
    p = fma_step(r, p, -0.5);
    const double r2 = r * r;
    p = fma(r2, p, r);  // log1p(r)
    const double ed = (double)e;
    const double head = fma(ed, 6.93147180369123816490e-01, t.l);  // e * ln2_hi + L[j]
    const double tail = fma(ed, 1.90821492927058770002e-10, p);    // e * ln2_lo + log1p(r)
    const double res = head + tail;
    return x == 0.0 ? -__builtin_inf() : res;
}

// The polynomial coefficients of log_unit and cos_2pi as one array, for kernels that read them from LDS where
// they are used instead of holding fourteen register pairs of constants across their loop: [0..5] the
// logarithm's (highest degree first), [6..13] the cosine's.  Same values, same operations, same results.
constexpr int kMathCoefs = 14;
__device__ __forceinline__ double math_coef(int k) {
    switch (k) {
        case 0: return 1.0 / 7.0;
        case 1: return -1.0 / 6.0;
        case 2: return 1.0 / 5.0;
        case 3: return -1.0 / 4.0;
        case 4: return 1.0 / 3.0;
        case 5: return -0.5;
        case 6: return 4.16639726608991643254e-06;
        case 7: return -1.04582860367372637990e-04;
        case 8: return 1.92956265087559594197e-03;
        case 9: return -2.58068900255420882739e-02;
        case 10: return 2.35330630272128971114e-01;
        case 11: return -1.33526276885187988874e+00;
        case 12: return 4.05871212641673650978e+00;
        default: return -4.93480220054467899615e+00;
    }
}

__device__ __forceinline__ double log_unit(double x, const LogTableEntry* __restrict__ lds_table, const double* lds_coef) {
    const uint32_t hi = (uint32_t)__double2hiint(x);
    const uint32_t lo = (uint32_t)__double2loint(x);
    const int e = (int)(hi >> 20) - 1023;
    const uint32_t j = (hi >> 13) & 127u;
    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), (int)lo);
    const LogTableEntry t = lds_table[j];
    const double2 c01 = *reinterpret_cast<const double2*>(lds_coef);
    const double2 c23 = *reinterpret_cast<const double2*>(lds_coef + 2);
    const double2 c45 = *reinterpret_cast<const double2*>(lds_coef + 4);
    const double r = fma(m, t.inv_c, -1.0);
    double p = fma_step(r, c01.x, c01.y);
    p = fma_step(r, p, c23.x);
    p = fma_step(r, p, c23.y);
    p = fma_step(r, p, c45.x);
    p = fma_step(r, p, c45.y);
    const double r2 = r * r;
    p = fma(r2, p, r);  // log1p(r)
    const double ed = (double)e;
    const double head = fma(ed, 6.93147180369123816490e-01, t.l);  // e * ln2_hi + L[j]
    const double tail = fma(ed, 1.90821492927058770002e-10, p);    // e * ln2_lo + log1p(r)
    const double res = head + tail;
    return x == 0.0 ? -__builtin_inf() : res;
}

// MFORM: the argument is m = 1 + u; 2u = 2m - 2 is exact either way.
template <bool MFORM = false>
__device__ __forceinline__ double cos_2pi(double u, const double* lds_coef) {
    const double2 c01 = *reinterpret_cast<const double2*>(lds_coef + 6);
    const double2 c23 = *reinterpret_cast<const double2*>(lds_coef + 8);
    const double2 c45 = *reinterpret_cast<const double2*>(lds_coef + 10);
    const double2 c67 = *reinterpret_cast<const double2*>(lds_coef + 12);
    const double t = MFORM ? fma(2.0, u, -2.0) : u + u;
    const double k = rint(t);
    const double w = t - k;
    const double z = w * w;
    double r = fma_step(z, c01.x, c01.y);
    r = fma_step(z, r, c23.x);
    r = fma_step(z, r, c23.y);
    r = fma_step(z, r, c45.x);
    r = fma_step(z, r, c45.y);
    r = fma_step(z, r, c67.x);
    r = fma_step(z, r, c67.y);
    const double c = fma(z, r, 1.0);
    const uint32_t sign = (uint32_t)(int)k << 31;  // k odd: negate
    return __hiloint2double((int)((uint32_t)__double2hiint(c) ^ sign), __double2loint(c));
}

// ---------------------------------------------------------------------------
// cos(2*pi*u) for u in [0, 1).  Half-period reduction, exact in u:
//   t = 2u, k = rint(t) in {0, 1, 2}, w = t - k in [-1/2, 1/2];  cos(2 pi u) = (-1)^k cos(pi w)
//   cos(pi w) = 1 + z R(z), z = w^2 in [0, 1/4]
// R is the degree-7 Chebyshev interpolant of (cos(pi sqrt z) - 1)/z (tools/fit_cos2pi.py); absolute
// error of the result <= 3.5e-16 on the whole interval.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double cos_2pi(double u) {
    const double t = u + u;
    const double k = rint(t);
    const double w = t - k;
    const double z = w * w;
    double r = fma_step(z, 4.16639726608991643254e-06, -1.04582860367372637990e-04);
    r = fma_step(z, r, 1.92956265087559594197e-03);
    r = fma_step(z, r, -2.58068900255420882739e-02);
    r = fma_step(z, r, 2.35330630272128971114e-01);
    r = fma_step(z, r, -1.33526276885187988874e+00);
    r = fma_step(z, r, 4.05871212641673650978e+00);
    r = fma_step(z, r, -4.93480220054467899615e+00);
    const double c = fma(z, r, 1.0);
    const uint32_t sign = (uint32_t)(int)k << 31;  // k odd: negate
    return __hiloint2double((int)((uint32_t)__double2hiint(c) ^ sign), __double2loint(c));
}

// ---------------------------------------------------------------------------
// sqrt(x) for x = 0 or a normal x that needs no rescaling (2^-500 < x < 2^500): the product
// (1 - mu^2)(1 - mu'^2) of RotateIntoLabFrame lies in [0, 1] and is either 0 or >= 2^-106.
// Reciprocal-square-root seed, one coupled Goldschmidt step and two residual corrections: the
// sequence of the general fp64 square root without its range scaling.  NaN for x < 0.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double sqrt_unit(double x) {
    if constexpr (Ablation::kSqrtGeneral) return sqrt(x);
    const double r = __builtin_amdgcn_rsq(x);
    double y = x * r;
    double h = r * 0.5;
    const double e = fma(-h, y, 0.5);
    y = fma(y, e, y);
    h = fma(h, e, h);
    double d = fma(-y, y, x);
    y = fma(d, h, y);
    d = fma(-y, y, x);
    y = fma(d, h, y);
    return x == 0.0 ? 0.0 : y;
}

// ---------------------------------------------------------------------------
// n / d for normal operands whose quotient needs no rescaling (HGScatter: n = 1 - g^2 in (0, 1],
// d = 1 + g*s in (0, 2)): reciprocal seed, two Newton steps, quotient and one residual correction --
// the sequence of the general fp64 division without its range scaling and special-case fix-up.
// d = 0 gives NaN (the general division gives inf or NaN there).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double div_unit(double n, double d) {
    if constexpr (Ablation::kDivGeneral) return n / d;
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    const double q = n * r;
    const double res = fma(-d, q, n);
    return fma(res, r, q);
}

}  // namespace rt
