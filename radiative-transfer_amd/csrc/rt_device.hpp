// rt_device.hpp -- device-side building blocks of the photon random walk (gfx950).
//
// One photon lives in the registers of one lane.  The functions here are the
// per-lane pieces: the counter-based generator, the emission lookup and one
// loop iteration ("flight": step, boundary/escape tests, absorption,
// scattering).  The kernels in rt_kernels.hip wrap them in a persistent
// wave-level loop (production) or a one-photon-per-thread loop (replay).
//
// Reference semantics implemented (file:line into the reference checkout):
//   TakeStep            MC code.py:493-508
//   CheckForEscape      MC code.py:576-593   (strict >, tested twice in RT_ORDER_MC)
//   CheckBoundary       MC code.py:522-574   (<=)
//   Absorb              MC code.py:470-491   (xi > Albedo)
//   HGScatter           MC code.py:427-441
//   IsoScatter          MC code.py:414-425
//   RotateIntoLabFrame  MC code.py:595-618
//   GetZMomentum        MC code.py:360-369
//   gray-dust loop      Numpy_random_walk.py:313-396
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"
#include "rt_math.hpp"

namespace rt {

constexpr uint32_t kEmitBlock = 0x80000000u;  // Philox blocks >= this feed photon emission
constexpr double kTwoPi = 2 * 3.141592653589793;

// ---------------------------------------------------------------------------
// Philox4x32-10.  counter = (photon id lo, hi, block, atmosphere), key = seed.
// ---------------------------------------------------------------------------
struct PhiloxOut {
    uint32_t w0, w1, w2, w3;
};

__device__ __forceinline__ PhiloxOut philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                   uint32_t k1) {
#pragma unroll
    for (int r = 0; r < Ablation::kPhiloxRounds; ++r) {  // 10
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // a ^ b ^ c in one instruction: v_bitop3_b32 with the truth table of three-way xor (0x96)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// Uniform in [0,1) with 52 random bits: the top 52 bits of a:b become the fraction of a double in [1,2),
// minus 1 (exact).  Two funnel shifts and one subtraction; values are the multiples of 2^-52.
__device__ __forceinline__ double u52(uint32_t a, uint32_t b) {
    if constexpr (Ablation::kUniform53) return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
    const uint32_t hi = __builtin_amdgcn_alignbit(0x3FFu, a, 12);  // 0x3FF00000 | (a >> 12)
    const uint32_t lo = __builtin_amdgcn_alignbit(a, b, 12);       // (a << 20) | (b >> 12)
    return __hiloint2double((int)hi, (int)lo) - 1.0;
}

// The same draw before the subtraction: m = 1 + u in [1, 2).  Where a uniform only enters an affine expression
// that one fma evaluates exactly as before -- 2u = 2m - 2 (azimuth), u*n = m*n - n rounded once (emission
// bucket) -- the production kernel passes m on and saves the subtraction.  (The cosine draw, 2u - 1 = 2m - 3,
// would need the constant -3 in a register pair, and the kernel has none to spare at 5 waves per SIMD.)
__device__ __forceinline__ double m52(uint32_t a, uint32_t b) {
    const uint32_t hi = __builtin_amdgcn_alignbit(0x3FFu, a, 12);
    const uint32_t lo = __builtin_amdgcn_alignbit(a, b, 12);
    return __hiloint2double((int)hi, (int)lo);
}

// ---------------------------------------------------------------------------
// Run configuration (wave-uniform; passed by value as a kernel argument).
// ---------------------------------------------------------------------------
// Uniform in (0,1) from one word: (w + 1/2) / 2^32, exact in fp64 (RT_RNG_PHILOX32X4).
__device__ __forceinline__ double u32mid(uint32_t w) { return fma((double)w, 0x1p-32, 0x1p-33); }

// RT_RNG_PHILOX40: the four uniforms of a loop iteration cut from ONE block as 24 | 40 | 40 | 24 bits, each at the
// midpoint of its cell (never 0 or 1) and exact in fp64:
//   azimuth (w0 >> 8 + 1/2) / 2^24,  cosine ((w0 & 255) 2^32 + w1 + 1/2) / 2^40,
//   step    (w2 2^8 + (w3 >> 24) + 1/2) / 2^40,  absorption ((w3 & 0xffffff) + 1/2) / 2^24.
// The 40-bit ones fill the fraction of a double in [1, 2) (two bit operations, a funnel shift, a subtraction); the
// 24-bit ones go through an integer conversion.
__device__ __forceinline__ double u40_step(uint32_t w2, uint32_t w3) {
    const uint32_t hi = __builtin_amdgcn_alignbit(0x3FFu, w2, 12);                               // 0x3FF00000 | (w2 >> 12)
    const uint32_t lo = __builtin_amdgcn_alignbit(w2, (w3 & 0xFF000000u) | 0x00800000u, 12);     // w2 << 20 | (w3 >> 24) << 12 | 1 << 11
    return __hiloint2double((int)hi, (int)lo) - 1.0;
}
__device__ __forceinline__ double u40_cosine(uint32_t w0, uint32_t w1) {
    const uint32_t hi = __builtin_amdgcn_alignbit((w0 & 0xFFu) | 0x3FF00u, w1, 20);              // 0x3FF00000 | (w0 & 255) << 12 | w1 >> 20
    const uint32_t lo = (w1 << 12) | 0x800u;
    return __hiloint2double((int)hi, (int)lo) - 1.0;
}
__device__ __forceinline__ double u24_azimuth(uint32_t w0) { return (double)((w0 & 0xFFFFFF00u) | 0x80u) * 0x1p-32; }
__device__ __forceinline__ double u24_absorb(uint32_t w3) { return (double)((w3 << 8) | 0x80u) * 0x1p-32; }

// How the one-block streams cut a block: (azimuth, cosine) of the scattering before a flight, (step, absorb) of the flight.
template <int RNG>
struct BlockCut {
    static __device__ __forceinline__ double azimuth(const PhiloxOut& o) { return RNG == RT_RNG_PHILOX40 ? u24_azimuth(o.w0) : u32mid(o.w0); }
    static __device__ __forceinline__ double cosine(const PhiloxOut& o) { return RNG == RT_RNG_PHILOX40 ? u40_cosine(o.w0, o.w1) : u32mid(o.w1); }
    static __device__ __forceinline__ double step(const PhiloxOut& o) { return RNG == RT_RNG_PHILOX40 ? u40_step(o.w2, o.w3) : u32mid(o.w2); }
    static __device__ __forceinline__ double absorb(const PhiloxOut& o) { return RNG == RT_RNG_PHILOX40 ? u24_absorb(o.w3) : u32mid(o.w3); }
};

struct TableRec {  // one wavelength: what CreatePhotons copies into a photon row (MC code.py:403-408), with the
                   // photon's momentum weight: the quotient GetZMomentum forms on every call, 1/lambda
                   // (MC code.py:367), taken once on the host (IEEE division, as NumPy's) -- or lambda itself, which
                   // the gray loop order weights by; the host keeps one record array per loop order and a sweep
                   // reads the one of its order.  32 bytes: two 16-byte loads, and a record never straddles a
                   // 128-byte cache line (a 48-byte record with 1/(2g) stored as well did one time in four, and
                   // the extra misses cost more than the division does).
    double g, albedo, scale, weight;
};
static_assert(sizeof(TableRec) == 32, "two 16-byte loads per record");

// Walker/Vose alias table of the emission law: bucket i (one per wavelength) keeps wavelength i with
// probability `prob` and hands the rest of its 1/n share to wavelength `alias`.  One 16-byte load and no
// data-dependent loop per emission, whatever the shape of the law.
struct AliasEntry {
    double prob;
    int32_t alias;
    int32_t pad_;
};
static_assert(sizeof(AliasEntry) == 16, "one 16-byte load per entry");

struct Cfg {
    int32_t boundary, scatter, order, rand_mu, abs_only;
    int32_t n_mu_bins, n_depth_bins, max_events;
    double alpha, h;
    int32_t geometry, pad_;
    double shell_inner;  // RT_GEOM_SHELL: inner radius / thickness
    uint32_t k0, k1;
    uint64_t id_base;
    // emission tables
    const TableRec* rec;      // the record array of this sweep's loop order (TableRec::weight)
    const AliasEntry* alias;  // n_wl entries
    int32_t n_wl, mono_index;
    int32_t table_base;  // first entry of the current atmosphere's table in rec / alias (several tables back to back)
    const LogTableEntry* log_table;  // kLogTableEntries entries in global memory (copied to LDS per workgroup)
    double* escaped_mu;              // per-photon escape cosines (rt_params.escaped_mu), or null
    int32_t quorum, pad2_;           // lean kernels: scattering lanes wait until this many have gathered (see walk_kernel)
};

// Which run flags a kernel has fixed at compile time.  Fixed<...>: loop order, lower boundary, scattering law, the
// depth tally, abs_only, the geometry and the per-photon escape-cosine output are template constants, so the code of
// every other option is not in the kernel at all; Runtime: everything is read from cfg.
struct WalkKey {  // the same flags as data: what a sweep asks for, and what an instantiation serves
    int32_t order, boundary, law;
    bool depth, abs_only, shell, emu;
    int32_t rng;  // RT_RNG_*
    bool plain;   // rand_mu and monochromatic off as well (the reference's defaults)
};

template <int ORDER, int BOUNDARY, int LAW, bool DEPTH, bool ABS_ONLY = false, bool SHELL = false, bool EMU = false>
struct Fixed {
    static constexpr bool kFixed = true;
    static constexpr bool kDepth = DEPTH;
    static constexpr bool kReemit = BOUNDARY == RT_BOUNDARY_REEMIT;
    static constexpr bool kRejection = LAW == RT_SCATTER_DRAINE || LAW == RT_SCATTER_DRAINE_EXACT;
    static __device__ __forceinline__ bool gray(const Cfg&) { return ORDER == RT_ORDER_GRAY; }
    static __device__ __forceinline__ int boundary(const Cfg&) { return BOUNDARY; }
    static __device__ __forceinline__ int law(const Cfg&) { return LAW; }
    static __device__ __forceinline__ bool depth(const Cfg&) { return DEPTH; }
    static __device__ __forceinline__ bool abs_only(const Cfg&) { return ABS_ONLY; }
    static __device__ __forceinline__ bool shell(const Cfg&) { return SHELL; }
    static __device__ __forceinline__ bool emu(const Cfg&) { return EMU; }
    static WalkKey key() { return {ORDER, BOUNDARY, LAW, DEPTH, ABS_ONLY, SHELL, EMU, 0, false}; }
};
struct Runtime {
    static constexpr bool kFixed = false;
    static constexpr bool kDepth = true;
    static constexpr bool kReemit = true;
    static constexpr bool kRejection = true;
    static __device__ __forceinline__ bool gray(const Cfg& c) { return c.order == RT_ORDER_GRAY; }
    static __device__ __forceinline__ int boundary(const Cfg& c) { return c.boundary; }
    static __device__ __forceinline__ int law(const Cfg& c) { return c.scatter; }
    static __device__ __forceinline__ bool depth(const Cfg& c) { return c.n_depth_bins > 0; }
    static __device__ __forceinline__ bool abs_only(const Cfg& c) { return c.abs_only != 0; }
    static __device__ __forceinline__ bool shell(const Cfg& c) { return c.geometry == RT_GEOM_SHELL; }
    static __device__ __forceinline__ bool emu(const Cfg& c) { return c.escaped_mu != nullptr; }
    static WalkKey key() { return {-1, -1, -1, true, true, true, true, 0, false}; }
};
// the reference's production configuration (MC code.py: reflecting floor, Henyey-Greenstein, two escape tests)
typedef Fixed<RT_ORDER_MC, RT_BOUNDARY_REFLECT, RT_SCATTER_HG, false> FixedBase;

struct Photon {
    double z, mu, wh, g, albedo, zmax, inv2g, opg2, omg2, p0, e1, b;
    double dep;  // RT_GEOM_SHELL: radial momentum this photon has deposited so far
    uint32_t iter;
    int32_t n_scatter;
    int32_t kbin;  // depth bin of the position the last flight ended at (kernels with depth bins)
};

struct Acc {
    double s[RT_NSUM];
    uint32_t c[RT_NCNT];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < RT_NSUM; ++k) s[k] = 0.0;
#pragma unroll
        for (int k = 0; k < RT_NCNT; ++k) c[k] = 0u;
    }
    __device__ __forceinline__ void add(int k, double v) { s[k] += v; }
    __device__ __forceinline__ void count(int k) { c[k]++; }
    __device__ __forceinline__ void count_if(int k, bool p) { c[k] += p; }
};

// Tallies of the fast kernel: the integer tallies other than the boundary hits are counted per WAVE by the
// kernel itself (population counts of lane masks, scalar registers), so the per-lane hooks do nothing.
// (Keeping the floating-point sums in LDS slots updated with ds_add_f64 instead of registers was measured:
// 16 fewer vector registers and 10 fewer vector instructions per iteration, no change in run time.)
struct AccWave {
    double s[RT_NSUM];
    uint32_t hits, reemits;  // per lane: a photon can meet the boundary any number of times
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < RT_NSUM; ++k) s[k] = 0.0;
        hits = 0u;
        reemits = 0u;
    }
    __device__ __forceinline__ void add(int k, double v) { s[k] += v; }
    __device__ __forceinline__ void count(int k) {
        if (k == RT_CNT_BOUNDARY_HITS) hits++;
        if (k == RT_CNT_REEMITTED) reemits++;
    }
    __device__ __forceinline__ void count_if(int, bool) {}
};

// Wave-private histograms in LDS (pointers are null when the tally is off) and the
// workgroup's copy of the logarithm table.
struct Hist {
    uint32_t* mu;
    double* depth_mom;
    double* depth_en;
    const LogTableEntry* log_table;
    const double* coef;  // LDS copy of math_coef(0..kMathCoefs-1), or null (kernels that keep them in registers)
};

// ---------------------------------------------------------------------------
// Random sources.  Draws are consumed in aligned pairs:
//   pair 2k   -> (step, absorb)      pair 2k+1 -> (azimuth, cosine)
// which is the per-photon order of the reference loop (MC code.py:496,479,436-437).
// ---------------------------------------------------------------------------
struct PhiloxSrc {
    uint32_t id_lo, id_hi, atm, k0, k1;
    uint32_t blk;     // next walk block
    uint32_t n_emit;  // emissions so far
    static constexpr bool kBounded = false;
    __device__ __forceinline__ int pair(double& a, double& b) {
        const PhiloxOut o = philox4x32_10(id_lo, id_hi, blk, atm, k0, k1);
        ++blk;
        a = u52(o.w0, o.w1);
        b = u52(o.w2, o.w3);
        return 2;
    }
    __device__ __forceinline__ void emit_pair(double& a, double& b) {
        const PhiloxOut o = philox4x32_10(id_lo, id_hi, kEmitBlock + n_emit, atm, k0, k1);
        ++n_emit;
        a = u52(o.w0, o.w1);
        b = u52(o.w2, o.w3);
    }
    __device__ __forceinline__ void used(int) {}
};

struct ReplaySrc {
    const double* xi;  // this photon's row
    int32_t max_draws;
    int32_t j;       // next pair starts here
    int32_t n_used;  // uniforms logically consumed
    static constexpr bool kBounded = true;
    __device__ __forceinline__ int pair(double& a, double& b) {
        const int avail = max_draws - j;
        a = avail > 0 ? xi[j] : 0.5;
        b = avail > 1 ? xi[j + 1] : 0.5;
        j += 2;
        return avail > 2 ? 2 : (avail < 0 ? 0 : avail);
    }
    __device__ __forceinline__ void used(int n) { n_used += n; }
};

// ---------------------------------------------------------------------------
// Emission: wavelength index from the tabulated law, then the photon constants.
// P(index i) is proportional to min(max(pdf_i,0),1)  (AcceptReject, MC code.py:239-268).
// ---------------------------------------------------------------------------
// u in [0,1) -> record index (table_base + wavelength index) with P(i) proportional to the table weights:
// x = u*n, bucket = floor(x), keep the bucket's own wavelength if frac(x) < prob, else its alias.
__device__ __forceinline__ int sample_index(const Cfg& cfg, double u) {
    if (cfg.mono_index >= 0) return cfg.table_base + cfg.mono_index;
    const double x = u * (double)cfg.n_wl;
    int i = (int)x;
    i = i < cfg.n_wl ? i : cfg.n_wl - 1;  // u*n can round up to n for u just below 1
    const double frac = x - (double)i;
    const uint4 raw = *reinterpret_cast<const uint4*>(&cfg.alias[cfg.table_base + i]);  // one 16-byte load
    const double prob = __hiloint2double((int)raw.y, (int)raw.x);
    return cfg.table_base + (frac < prob ? i : (int)raw.z);
}

// Per-photon constants from the photon's row values.  1/lambda, 1/(2*g), 1+g**2 and 1-g**2 are the
// sub-expressions GetZMomentum and HGScatter evaluate for every photon on every call
// (MC code.py:367, :439); they depend on the photon only, so they are formed once here (IEEE division).
__device__ __forceinline__ void set_constants(const Cfg& cfg, double tau, double lam, double g, double albedo,
                                              double scale, double mu0, Photon& ph) {
    const double w = cfg.order == RT_ORDER_GRAY ? lam : 1.0 / lam;
    ph.wh = w * cfg.h;
    ph.g = g;
    ph.albedo = albedo;
    ph.zmax = tau * scale;
    ph.inv2g = 1.0 / (2.0 * g);
    ph.opg2 = 1.0 + g * g;
    ph.omg2 = 1.0 - g * g;
    ph.z = 0.0;
    ph.mu = mu0;
}

// 16 bytes at base + byte_off as a raw buffer load: the 32-bit offset goes to the instruction as it is, where a
// global load needs the 64-bit address formed in vector registers first (sign extension, shift, 64-bit add).
// The caller keeps byte_off + 16 below 2^32 (rt_set_tables_multi limits the tables to 2^26 entries of <= 32 bytes);
// beyond the descriptor's range the hardware returns zeros instead of faulting.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 gather16(const void* base, uint32_t byte_off) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1 /* 2^32 - 1 bytes */, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ TableRec load_record_buffer(const Cfg& cfg, int i) {
    const uint4 a = gather16(cfg.rec, (uint32_t)i * 32u), b = gather16(cfg.rec, (uint32_t)i * 32u + 16u);
    return {__hiloint2double((int)a.y, (int)a.x), __hiloint2double((int)a.w, (int)a.z),
            __hiloint2double((int)b.y, (int)b.x), __hiloint2double((int)b.w, (int)b.z)};
}

__device__ __forceinline__ TableRec load_record(const Cfg& cfg, int i) {
    const double2* p = reinterpret_cast<const double2*>(&cfg.rec[i]);  // two 16-byte loads
    const double2 a = p[0], b = p[1];
    return {a.x, a.y, b.x, b.y};
}

template <class V>
__device__ __forceinline__ void set_constants(const Cfg& cfg, double tau, const TableRec& r, double mu0, Photon& ph) {
    ph.wh = r.weight * cfg.h;
    ph.g = r.g;
    ph.albedo = r.albedo;
    ph.zmax = tau * r.scale;
    // HGScatter's 1/(2*g), MC code.py:439.  div_unit gives the IEEE quotient for every normal g (tested bit for
    // bit); g = 0 gives NaN where IEEE gives inf, and the cosine it enters, inf * 0, is NaN either way.
    ph.inv2g = div_unit(1.0, 2.0 * r.g);
    ph.opg2 = 1.0 + r.g * r.g;
    ph.omg2 = 1.0 - r.g * r.g;
    ph.z = 0.0;
    ph.mu = mu0;
}

// BUF: alias entry and record through raw buffer loads (32-bit offsets), as the lean kernels' own emission.
// PLAIN: rand_mu and monochromatic are known to be off (the kernel was chosen for that).
template <class V = Runtime, bool BUF = false, bool PLAIN = false>
__device__ __forceinline__ void emit_from_table(const Cfg& cfg, double tau, PhiloxSrc& src, Photon& ph) {
    double ua, ub;
    src.emit_pair(ua, ub);
    int i;
    if (BUF && (PLAIN || cfg.mono_index < 0)) {  // sample_index() with the entry gathered by offset
        const double x = ua * (double)cfg.n_wl;
        int b = (int)x;
        b = b < cfg.n_wl ? b : cfg.n_wl - 1;
        const uint4 raw = gather16(cfg.alias, (uint32_t)(cfg.table_base + b) * 16u);
        i = cfg.table_base + ((x - (double)b) < __hiloint2double((int)raw.y, (int)raw.x) ? b : (int)raw.z);
    } else {
        i = sample_index(cfg, ua);
    }
    const TableRec r = BUF ? load_record_buffer(cfg, i) : load_record(cfg, i);
    set_constants<V>(cfg, tau, r, (!PLAIN && cfg.rand_mu) ? ub : 1.0, ph);
}

// LEAN: the quotient by the unscaled division sequence (the IEEE quotient for the normal operands that occur: 0 <= z <= zmax)
template <bool LEAN = false>
__device__ __forceinline__ int depth_bin(const Cfg& cfg, const Photon& ph) {
    int k = (int)floor((double)cfg.n_depth_bins * (LEAN ? div_unit(ph.z, ph.zmax) : ph.z / ph.zmax));
    k = k < 0 ? 0 : k;
    return k >= cfg.n_depth_bins ? cfg.n_depth_bins - 1 : k;
}

constexpr int kMaxDraineTries = 4096;

__device__ __forceinline__ double draine_pdf(double g, double alpha, double m) {
    // Numpy_random_walk.py:124.  The power 3/2 is taken as d*sqrt(d): within 2 ulp of pow(d, 1.5), which only
    // matters to the acceptance test pdf > y when the two agree to 16 digits, and it keeps the general pow
    // routine (80 vector registers) out of the walk kernel.
    const double d = 1.0 + g * g - 2.0 * g * m;
    return 1.0 / (4.0 * 3.141592653589793) * (1.0 - g * g) / (1.0 + alpha * (1.0 + 2.0 * g * g) / 3.0) *
           (1.0 + alpha * m * m) / (d * sqrt(d));
}

// A straight flight of optical length `path` through a uniform spherical shell (RT_GEOM_SHELL): ph.z is the
// height above the inner radius r_in = shell_inner * zmax, ph.mu the cosine to the local radial direction.
// A photon that reaches the empty cavity crosses it and re-enters on the far side with the mirrored cosine
// (reflect), or ends / is replaced there.  Returns a fate or 0.  No counterpart in the reference.
template <class Reemit, class AccT>
__device__ __forceinline__ int fly_shell(const Cfg& cfg, Photon& ph, double path, int boundary, AccT& acc,
                                         Reemit&& reemit) {
    const double rin = cfg.shell_inner * ph.zmax;
    double r = rin + ph.z;
    if (ph.mu < 0.0) {
        const double sin2 = 1.0 - ph.mu * ph.mu;
        const double d2 = rin * rin - r * r * sin2;  // r_in^2 - (impact parameter)^2
        if (d2 > 0.0) {
            const double half = sin2 == 0.0 ? rin : sqrt(d2);  // half the chord through the cavity
            double d_in = -r * ph.mu - half;
            d_in = d_in > 0.0 ? d_in : 0.0;
            if (path >= d_in) {
                acc.count(RT_CNT_BOUNDARY_HITS);
                const double mu_c = half / rin;  // cosine to the radial direction on leaving the cavity
                if (boundary == RT_BOUNDARY_REFLECT) {
                    path -= d_in;
                    r = rin;
                    ph.z = 0.0;
                    ph.mu = mu_c;
                } else if (boundary == RT_BOUNDARY_ABSORB) {
                    ph.b += -mu_c * ph.wh;
                    return RT_FATE_BOUNDARY;
                } else {
                    ph.b += -mu_c * ph.wh;
                    acc.count(RT_CNT_REEMITTED);
                    reemit(ph);
                    acc.add(RT_SUM_REEMITTED, ph.mu * ph.wh);
                    return 0;  // the new photon interacts at z = 0, as in the slab (MC code.py:539-547, :700)
                }
            }
        }
    }
    if (1.0 - ph.mu * ph.mu == 0.0) {  // a radial photon stays radial, exactly (most photons before their first collision)
        const double z = ph.z + path * ph.mu;
        ph.z = z > 0.0 ? z : 0.0;
        return 0;
    }
    const double q = path * (path + 2.0 * r * ph.mu);  // r'^2 - r^2
    const double r2 = sqrt(r * r + q);
    double mu = (r * ph.mu + path) / r2;
    mu = mu > 1.0 ? 1.0 : (mu < -1.0 ? -1.0 : mu);
    double z = ph.z + q / (r2 + r);  // r' - r without cancellation
    ph.z = z > 0.0 ? z : 0.0;        // a grazing pass of the cavity can round to -1 ulp
    ph.mu = mu;
    return 0;
}

// ---------------------------------------------------------------------------
// One loop iteration of one photon, in two halves.
//   fly()      TakeStep, escape / boundary / escape tests, Absorb.  Returns the photon's
//              RT_FATE_* if its history ends here, 0 if it goes on to scatter.
//   scatter()  HGScatter / IsoScatter / Draine sampling + RotateIntoLabFrame.  Returns 0, or
//              RT_FATE_GUARD when a bounded stream runs out.
// V (Fixed<...> or Runtime) says which flags are compile-time constants; the generic form reads cfg.  `reemit` is called when a photon is re-emitted at the lower boundary.
// ---------------------------------------------------------------------------
// fly() with the optical length of the step, -ln(xi), already formed (a caller that has independent memory
// requests in flight takes the logarithm first, so that it runs under their latency).
template <class V, class Reemit, class AccT>
__device__ __forceinline__ int fly_path(const Cfg& cfg, Photon& ph, double path, double u_abs, int avail, bool bounded,
                                        AccT& acc, const Hist& hist, Reemit&& reemit, int& n_used) {
    const bool gray = V::gray(cfg);
    const int boundary = V::boundary(cfg);
    const bool depth = V::depth(cfg);
    const bool shell = V::shell(cfg);

    if (bounded && avail < 1) return RT_FATE_GUARD;
    ph.iter++;
    acc.count(RT_CNT_FLIGHTS);
    n_used += 1;

    // TakeStep
    if (shell) {
        const int f = fly_shell(cfg, ph, path, boundary, acc, reemit);
        if (f != 0) return f;
    } else {
        ph.z += path * ph.mu;
    }

    if (!shell && !gray && ph.z > ph.zmax) {  // first escape test: momentum tallied
        ph.e1 = ph.mu * ph.wh;
        return RT_FATE_ESCAPED;
    }
    if (!shell && ph.z <= 0.0) {  // CheckBoundary
        const double pb = ph.mu * ph.wh;
        acc.count(RT_CNT_BOUNDARY_HITS);
        if (boundary == RT_BOUNDARY_REFLECT) {
            ph.b -= pb;
            ph.z = -ph.z;
            ph.mu = -ph.mu;
        } else if (boundary == RT_BOUNDARY_ABSORB) {
            ph.b += pb;
            return RT_FATE_BOUNDARY;
        } else {
            ph.b += pb;
            acc.count(RT_CNT_REEMITTED);
            reemit(ph);
            acc.add(RT_SUM_REEMITTED, ph.mu * ph.wh);
        }
    }
    if (ph.z > ph.zmax) {  // second escape test (the only one in gray order and in a shell)
        if (gray || shell) {
            ph.e1 = ph.mu * ph.wh;
            return RT_FATE_ESCAPED;
        }
        acc.add(RT_SUM_ESCAPED_UNCOUNTED, ph.mu * ph.wh);
        return RT_FATE_ESCAPED_UNCOUNTED;
    }

    // Absorb
    if (bounded && avail < 2) return RT_FATE_GUARD;
    n_used += 1;
    // the bin of this position, once for whichever tally follows here: the absorption below or the scattering the
    // caller performs next (nothing moves the photon in between)
    if (depth) ph.kbin = depth_bin<V::kFixed>(cfg, ph);
    const bool absorbed = V::abs_only(cfg) ? (u_abs > 0.0) : (u_abs > ph.albedo);
    if (absorbed) {
        const double pa = ph.mu * ph.wh;
        acc.add(RT_SUM_DUST_MOMENTUM, pa);
        acc.add(RT_SUM_ABSORBED_ENERGY, ph.wh);
        if (shell) ph.dep += pa;
        if (depth) {
            atomicAdd(&hist.depth_mom[ph.kbin], pa);
            atomicAdd(&hist.depth_en[ph.kbin], ph.wh);
        }
        return RT_FATE_ABSORBED;
    }
    return 0;
}

template <class V, class Reemit, class AccT>
__device__ __forceinline__ int fly(const Cfg& cfg, Photon& ph, double u_step, double u_abs, int avail, bool bounded,
                                   AccT& acc, const Hist& hist, Reemit&& reemit, int& n_used) {
    if (bounded && avail < 1) return RT_FATE_GUARD;  // before the logarithm: its argument is not a draw then
    const double path = -log_unit(u_step, hist.log_table);
    return fly_path<V>(cfg, ph, path, u_abs, avail, bounded, acc, hist, reemit, n_used);
}

// LEAN (the production walk kernel): the cosine's polynomial coefficients are read from LDS (hist.coef), and 1 + g^2 and
// 1 - g^2 are formed here from g (the same two operations set_constants performs, so the same values) instead
// of being carried in ph: fewer registers held across the loop, same results.
// MFORM: u_phi is passed as m = 1 + u (see m52).
template <class V, bool LEAN = false, bool MFORM = false, class Src, class AccT>
__device__ __forceinline__ int scatter(const Cfg& cfg, Photon& ph, Src& src, double u_phi, double u_s, int avail2,
                                       AccT& acc, const Hist& hist) {
    constexpr bool FAST = V::kFixed;  // unscaled division / square root, one rotation formula
    const bool gray = V::gray(cfg);
    const int law = V::law(cfg);
    const bool depth = V::depth(cfg);
    const bool shell = V::shell(cfg);
    if (Src::kBounded && avail2 < 2) return RT_FATE_GUARD;
    src.used(2);
    const double mu_old = ph.mu;
    bool iso = false;
    {
        if (law == RT_SCATTER_ISO) iso = gray ? true : (ph.iter == 1u);
        if (gray && law == RT_SCATTER_HG && ph.g == 0.0) iso = true;
        // the gray script's Draine branch falls back to IsoScatter for g == 0 and alpha == 0 (Numpy_random_walk.py:388-391)
        if (law == RT_SCATTER_DRAINE && ph.g == 0.0 && cfg.alpha == 0.0) iso = true;
    }
    double mu_prime;
    bool direct = false;
    if ((law == RT_SCATTER_DRAINE || law == RT_SCATTER_DRAINE_EXACT) && !iso) {
        mu_prime = 0.0;
        bool ok = false;
        // rejection sampling of the Draine phase function, one pair (y, candidate) per try.  The bound only ends
        // a loop that could not end otherwise (a NaN asymmetry parameter accepts nothing); the slowest legitimate
        // case, the reference's sampler at g = 0.85, accepts 8 % of its tries.
        for (int tries = 0; tries < kMaxDraineTries; ++tries) {
            double y, cand;
            const int av = src.pair(y, cand);
            if (Src::kBounded && av < 2) break;
            src.used(2);
            cand = 2.0 * cand - 1.0;
            bool accept;
            if (law == RT_SCATTER_DRAINE) {
                accept = draine_pdf(ph.g, cfg.alpha, cand) > y;  // uniform candidate, the reference's test
            } else {
                if (ph.g != 0.0) {  // Henyey-Greenstein candidate (MC code.py:439); g = 0 is its isotropic limit
                    // (FAST: the unscaled division, bit for bit the IEEE quotient on this domain, as in the HG branch)
                    const double q = FAST ? div_unit(ph.omg2, 1.0 + ph.g * cand) : ph.omg2 / (1.0 + ph.g * cand);
                    cand = ph.inv2g * (ph.opg2 - q * q);
                }
                accept = 1.0 + cfg.alpha * cand * cand > y * (1.0 + cfg.alpha);
            }
            if (accept) {
                mu_prime = cand;
                ok = true;
                break;
            }
        }
        if (!ok) return RT_FATE_GUARD;
    } else {
        const double s = fma(2.0, u_s, -1.0);  // 2*u is exact, so this is (2*u) - 1 in one operation
        if (iso) {
            mu_prime = s;
            direct = true;
        } else {
            const double opg2 = LEAN ? 1.0 + ph.g * ph.g : ph.opg2;
            const double omg2 = LEAN ? 1.0 - ph.g * ph.g : ph.omg2;
            // LEAN: 1/(2g) where it is used (the same div_unit as set_constants, so the same value): most photons of a
            // thin atmosphere never scatter, and here its sequence interleaves with the division below
            const double inv2g = LEAN ? div_unit(1.0, 2.0 * ph.g) : ph.inv2g;
            const double q = FAST ? div_unit(omg2, 1.0 + ph.g * s) : omg2 / (1.0 + ph.g * s);
            mu_prime = inv2g * (opg2 - q * q);
        }
    }
    // RotateIntoLabFrame
    double mu_new;
    if (FAST) {
        // The mu == +-1 branches of the reference (MC code.py:605-612) give what the general formula gives
        // there: 1 - mu^2 is exactly 0, so the first term is -(+-0) and the sum is mu' or -mu' exactly.
        const double c = LEAN ? cos_2pi<MFORM>(u_phi, hist.coef) : cos_2pi(u_phi);
        mu_new = -sqrt_unit((1.0 - mu_old * mu_old) * (1.0 - mu_prime * mu_prime)) * c + mu_old * mu_prime;
        if (direct) mu_new = mu_prime;  // IsoScatter assigns the lab-frame cosine itself (MC code.py:741-743)
    } else if (direct || mu_old == 1.0) {
        mu_new = mu_prime;
    } else if (mu_old == -1.0) {
        mu_new = -mu_prime;
    } else {
        mu_new = -sqrt_unit((1.0 - mu_old * mu_old) * (1.0 - mu_prime * mu_prime)) * cos_2pi(u_phi) + mu_old * mu_prime;
    }
    ph.mu = mu_new;
    ph.n_scatter++;
    acc.count(RT_CNT_SCATTER);
    const double dp = (mu_old - mu_new) * ph.wh;
    acc.add(RT_SUM_DUST_MOMENTUM, dp);
    if (shell) ph.dep += dp;
    if (depth) atomicAdd(&hist.depth_mom[ph.kbin], dp);
    return 0;
}

// fly() then scatter() with the stream's own pairs: the generic kernel and the replay kernel.
template <class V, class Src, class Reemit>
__device__ __forceinline__ int flight(const Cfg& cfg, Photon& ph, Src& src, Acc& acc, const Hist& hist,
                                      Reemit&& reemit) {
    if (ph.iter >= (uint32_t)cfg.max_events) return RT_FATE_GUARD;
    double u_step, u_abs;
    const int avail = src.pair(u_step, u_abs);
    int n_used = 0;
    const int fate = fly<V>(cfg, ph, u_step, u_abs, avail, Src::kBounded, acc, hist, reemit, n_used);
    src.used(n_used);
    if (fate != 0) return fate;
    double u_phi, u_s;
    const int avail2 = src.pair(u_phi, u_s);
    return scatter<V>(cfg, ph, src, u_phi, u_s, avail2, acc, hist);
}

// Bookkeeping when a photon's history ends.
// UNIT_MU0: every photon starts with mu = 1, so its initial momentum p0 = 1 * wh is wh itself and ph.p0 need not
// be kept.
template <bool UNIT_MU0 = false, class V = Runtime, class AccT>
__device__ __forceinline__ void retire(const Cfg& cfg, const Photon& ph, int fate, AccT& acc, const Hist& hist) {
    const double p0 = UNIT_MU0 ? ph.wh : ph.p0;
    const bool shell = V::shell(cfg);
    const bool gray = V::gray(cfg) || shell;  // a shell has one escape test: every escape is logged
    acc.count_if(RT_CNT_ESCAPED, fate == RT_FATE_ESCAPED);
    acc.count_if(RT_CNT_ESCAPED_UNCOUNTED, fate == RT_FATE_ESCAPED_UNCOUNTED);
    acc.count_if(RT_CNT_ABSORBED, fate == RT_FATE_ABSORBED);
    acc.count_if(RT_CNT_BOUNDARY_ABSORBED, fate == RT_FATE_BOUNDARY);
    acc.count_if(RT_CNT_GUARD, fate == RT_FATE_GUARD);
    acc.add(RT_SUM_ESCAPED, ph.e1);
    acc.add(RT_SUM_BOUNDARY, ph.b);
    const double t = shell ? ph.dep : p0 - ph.e1 + ph.b;
    acc.add(RT_SUM_TRANSFER_SQ, t * t);
    acc.add(RT_SUM_TRANSFER_P0, t * p0);
    acc.add(RT_SUM_P0_SQ, p0 * p0);
    if (hist.mu && ((fate == RT_FATE_ESCAPED && gray) || fate == RT_FATE_ESCAPED_UNCOUNTED)) {
        int k = (int)floor((double)cfg.n_mu_bins * ph.mu);
        k = k < 0 ? 0 : k;
        k = k > cfg.n_mu_bins ? cfg.n_mu_bins : k;
        atomicAdd(&hist.mu[k], 1u);
    }
}

template <class AccT>
__device__ __forceinline__ void start_photon(Photon& ph, AccT& acc) {
    ph.p0 = ph.mu * ph.wh;
    ph.e1 = 0.0;
    ph.b = 0.0;
    ph.dep = 0.0;
    ph.iter = 0u;
    ph.n_scatter = 0;
    acc.add(RT_SUM_INITIAL, ph.p0);
    acc.count(RT_CNT_PHOTONS);
}

}  // namespace rt
