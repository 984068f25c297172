// rt_kernels.hip -- gfx950 kernels around the photon random walk and the launchers of all of them.
//
//   zero_results_kernel / finish_kernel   before and after the walk: zero the tally copies; add the chunk partials of
//                       every slab in a fixed shape, fold the tally copies.
//   replay_kernel       one photon per thread on caller-supplied uniforms.
//   emission_kernel     the wavelength index each photon id is emitted with.
//   grain_integrals_kernel, selftest_math_kernel
// The persistent walk kernel itself is a template (rt_walk.hpp) instantiated in rt_walk_*.hip; this file collects
// the instantiations into one table.
#include <algorithm>
#include <mutex>

#include "rt_walk.hpp"

namespace rt {

// Zeroes what a sweep accumulates into with atomics: the queue counter and the tally copies.
__global__ void zero_results_kernel(unsigned int* __restrict__ d_next, unsigned long long* __restrict__ a, size_t na) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && d_next) *d_next = 0u;
    for (size_t i = t; i < na; i += stride) a[i] = 0ull;
}

// What follows the walk, in one launch of 64-thread workgroups.
//   Workgroups [0, n_atm * n_slabs): one wave per (atmosphere, slab); the slabs of other ranks are set to zero.  An
//   owned slab's chunk partials are added in a fixed shape -- lane l takes the slots l, l + 64, ... in order, then the
//   lanes are combined by the butterfly -- that depends on the slab's chunk count only.  Every slot of the range was
//   written by the walk kernel (a chunk beyond the end of a short last slab holds zeros).
//   The next workgroups fold the copies of the integer tallies (counts, then the escape histogram) into the
//   caller's arrays: out[i] = sum over the copies of in[c * n + i] (integers: any order gives the same bits).
//   The last n_atm * n_slabs * ceil(2 n_depth / 64) workgroups add the depth-resolved deposition of each slab: lane l
//   owns entry l of a block of 64 of the slab's 2 n_depth entries and adds the chunks' values in chunk order.
__global__ __launch_bounds__(64) void finish_kernel(const Sweep sw, const double* __restrict__ d_chunk_sums,
                                                    double* __restrict__ d_slab_sums,
                                                    const unsigned long long* __restrict__ d_count_copies,
                                                    unsigned long long* __restrict__ d_counts, uint32_t n_counts,
                                                    const unsigned long long* __restrict__ d_hist_copies,
                                                    unsigned long long* __restrict__ d_mu_hist, uint32_t n_hist,
                                                    const double* __restrict__ d_chunk_depth,
                                                    double* __restrict__ d_slab_depth, uint32_t n_depth) {
    const int lane = threadIdx.x;
    const unsigned int n_reduce = sw.n_atm * sw.n_slabs;
    const unsigned int n_fold = (n_counts + n_hist + 63u) / 64u;
    if (blockIdx.x >= n_reduce + n_fold) {
        const unsigned int per_slab = (2u * n_depth + 63u) / 64u;
        const unsigned int b = blockIdx.x - n_reduce - n_fold;
        const unsigned int as = b / per_slab, entry = (b - as * per_slab) * 64u + (unsigned int)lane;
        if (entry >= 2u * n_depth) return;
        const unsigned int a = as / sw.n_slabs, slab = as - a * sw.n_slabs;
        const PlanEntry plan = sw.plan[a];
        double s = 0.0;
        if (slab % sw.world == sw.rank) {
            const unsigned int j = slab / sw.world;
            const double* src = d_chunk_depth + ((size_t)plan.slot_base + (size_t)j * plan.chunks) * (2u * n_depth) + entry;
#pragma unroll 8  // loads of eight chunks in flight; the additions stay in chunk order
            for (unsigned int c = 0; c < plan.chunks; ++c) s += src[(size_t)c * (2u * n_depth)];
        }
        d_slab_depth[((size_t)plan.atm * sw.n_slabs + slab) * (2u * n_depth) + entry] = s;
        return;
    }
    if (blockIdx.x >= n_reduce) {
        const unsigned int i = (blockIdx.x - n_reduce) * 64u + (unsigned int)lane;
        const bool is_count = i < n_counts;
        const unsigned int j = is_count ? i : i - n_counts;
        const unsigned int n = is_count ? n_counts : n_hist;
        if (j >= n) return;
        const unsigned long long* in = is_count ? d_count_copies : d_hist_copies;
        unsigned long long s = 0ull;
#pragma unroll 8  // eight independent loads in flight per lane instead of one (the copies are 8 n bytes apart)
        for (uint32_t c = 0; c < sw.n_copies; ++c) s += in[(size_t)c * n + j];
        (is_count ? d_counts : d_mu_hist)[j] = s;
        return;
    }
    const unsigned int a = blockIdx.x / sw.n_slabs;
    const unsigned int slab = blockIdx.x - a * sw.n_slabs;
    const PlanEntry plan = sw.plan[a];
    if (slab % sw.world != sw.rank) {
        if (lane < RT_NSUM) d_slab_sums[((size_t)plan.atm * sw.n_slabs + slab) * RT_NSUM + lane] = 0.0;
        return;
    }
    const unsigned int j = slab / sw.world;  // ordinal among the slabs this rank owns
    const double* src = d_chunk_sums + ((size_t)plan.slot_base + (size_t)j * plan.chunks) * RT_NSUM;
    double s[RT_NSUM];
#pragma unroll
    for (int k = 0; k < RT_NSUM; ++k) s[k] = 0.0;
    for (unsigned int c = (unsigned int)lane; c < plan.chunks; c += 64u) {
#pragma unroll
        for (int k = 0; k < RT_NSUM; ++k) s[k] += src[(size_t)c * RT_NSUM + k];
    }
#pragma unroll
    for (int k = 0; k < RT_NSUM; ++k) {
        const double v = wave_sum(s[k]);
        if (lane == 0) d_slab_sums[((size_t)plan.atm * sw.n_slabs + slab) * RT_NSUM + k] = v;
    }
}

// The wavelength index CreatePhotons would assign (MC code.py:389-394) and the direction draw (:385), per photon id:
// the emission block of the photon's stream through the same sampler the sweeps use.
__global__ __launch_bounds__(256) void emission_kernel(const Cfg cfg, const int rng_mode, const uint32_t atm,
                                                       const unsigned long long first_id, const long long n,
                                                       int32_t* __restrict__ out_index, double* __restrict__ out_mu0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long gid = first_id + (unsigned long long)i + cfg.id_base;
    // the one-block streams take the emission draw from words 0-1 of walk block 0, the direction from the emission block
    const uint32_t block = rng_mode != RT_RNG_PHILOX52 ? 0u : kEmitBlock;
    const PhiloxOut o = philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), block, atm, cfg.k0, cfg.k1);
    out_index[i] = sample_index(cfg, u52(o.w0, o.w1)) - cfg.table_base;
    if (out_mu0) {
        const PhiloxOut e = philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), kEmitBlock, atm, cfg.k0, cfg.k1);
        out_mu0[i] = cfg.rand_mu ? u52(e.w2, e.w3) : 1.0;
    }
}

__global__ __launch_bounds__(kWalkBlock) void replay_kernel(const Cfg cfg, const double tau, const long long n,
                                                            const int max_draws, const double* __restrict__ xi,
                                                            const double* __restrict__ lam,
                                                            const double* __restrict__ g,
                                                            const double* __restrict__ albedo,
                                                            const double* __restrict__ scale,
                                                            const double* __restrict__ mu0, double* d_sums,
                                                            unsigned long long* d_counts,
                                                            unsigned long long* d_mu_hist, double* d_depth,
                                                            rt_photon_trace* __restrict__ trace) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nmu = cfg.n_mu_bins + 1;
    const int nd = cfg.n_depth_bins;
    const Hist hist = wave_hist(lds_raw, wave, nmu, nd);
    stage_log_table(cfg, lds_raw);
    clear_hist(hist, lane, nmu, nd);

    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    Acc acc;
    acc.clear();
    if (i < n) {
        Photon ph = {};
        ReplaySrc src;
        src.xi = xi + (size_t)i * (size_t)max_draws;
        src.max_draws = max_draws;
        src.j = 0;
        src.n_used = 0;
        set_constants(cfg, tau, lam[i], g[i], albedo[i], scale[i], mu0[i], ph);
        start_photon(ph, acc);
        auto reemit = [&](Photon& p) {
            p.z = 0.0;
            p.mu = mu0[i];
        };
        int fate = 0;
        while (fate == 0) fate = flight<Runtime>(cfg, ph, src, acc, hist, reemit);
        retire(cfg, ph, fate, acc, hist);
        if (trace) {
            rt_photon_trace t;
            t.final_z = ph.z;
            t.final_mu = ph.mu;
            t.fate = fate;
            t.n_scatter = ph.n_scatter;
            t.n_draws = src.n_used;
            t.pad_ = 0;
            trace[i] = t;
        }
    }
#pragma unroll
    for (int k = 0; k < RT_NSUM; ++k) {
        const double v = wave_sum(acc.s[k]);
        if (lane == 0 && v != 0.0) atomicAdd(&d_sums[k], v);
    }
#pragma unroll
    for (int k = 0; k < RT_NCNT; ++k) {
        const uint32_t v = wave_sum_u32(acc.c[k]);
        if (lane == 0 && v) atomicAdd(&d_counts[k], (unsigned long long)v);
    }
    flush_hist<false>(hist, lane, nmu, nd, 0u, d_mu_hist, d_depth);
}

// The kernels' own ln() and cos(2 pi u) applied element-wise (test hook).
__global__ void selftest_math_kernel(const LogTableEntry* __restrict__ table, long long n, const double* __restrict__ u,
                                     double* __restrict__ out_log, double* __restrict__ out_cos,
                                     double* __restrict__ out_sqrt, double* __restrict__ out_div) {
    __shared__ LogTableEntry lds_table[kLogTableEntries];
    __shared__ __align__(16) double lds_coef[16];
    for (int k = threadIdx.x; k < kLogTableEntries; k += blockDim.x) lds_table[k] = table[k];
    if (threadIdx.x < kMathCoefs) lds_coef[threadIdx.x] = math_coef((int)threadIdx.x);
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        // both forms of each function (coefficients in registers / read from LDS) must agree bit for bit; a
        // difference is reported as NaN, which no caller's tolerance accepts
        const double l0 = log_unit(u[i], lds_table), l1 = log_unit(u[i], lds_table, lds_coef);
        const double c0 = cos_2pi(u[i]), c1 = cos_2pi(u[i], lds_coef);
        const bool same_l = __double_as_longlong(l0) == __double_as_longlong(l1);
        const bool same_c = __double_as_longlong(c0) == __double_as_longlong(c1);
        out_log[i] = same_l ? l0 : __builtin_nan("");
        out_cos[i] = same_c ? c0 : __builtin_nan("");
        out_sqrt[i] = sqrt_unit(u[i]);
        out_div[i] = div_unit(1.0 - u[i] * u[i], 2.0 * u[i]);
    }
}

// ---------------------------------------------------------------------------
// Grain-size integration of the Draine tables: one thread per output wavelength.
// For every grain size the three tabulated columns are interpolated linearly to the output
// wavelength exactly as np.interp does on the flipped (ascending) columns (read_BPASS.py:94-96:
// clamped outside the tabulated range, slope * (x - x_k) + y_k inside), and the size integrals of
// GetKappas (read_BPASS.py:159-162) and GetDraineAverage (MC code.py:306-312) are accumulated with the
// trapezoid rule over consecutive sizes (np.trapz).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void interp3(const double* __restrict__ xw, const double* __restrict__ ya,
                                        const double* __restrict__ ys, const double* __restrict__ yg, int rows, double x,
                                        double& qa, double& qs, double& gg) {
    // columns are stored descending in wavelength: ascending index r corresponds to stored row rows-1-r
    const double x_first = xw[rows - 1], x_last = xw[0];
    if (x <= x_first) {  // np.interp: left clamp (also the exact hit on the first abscissa)
        qa = ya[rows - 1];
        qs = ys[rows - 1];
        gg = yg[rows - 1];
        return;
    }
    if (x >= x_last) {
        qa = ya[0];
        qs = ys[0];
        gg = yg[0];
        return;
    }
    int lo = 0, hi = rows - 1;  // ascending indices with xasc[lo] <= x < xasc[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xw[rows - 1 - mid] <= x)
            lo = mid;
        else
            hi = mid;
    }
    const int k0 = rows - 1 - lo, k1 = rows - 1 - hi;
    const double x0 = xw[k0], dx = xw[k1] - x0, t = x - x0;
    qa = (ya[k1] - ya[k0]) / dx * t + ya[k0];
    qs = (ys[k1] - ys[k0]) / dx * t + ys[k0];
    gg = (yg[k1] - yg[k0]) / dx * t + yg[k0];
}

__global__ __launch_bounds__(256) void grain_integrals_kernel(const GrainSetDev gs, const int n_wl,
                                                              const double* __restrict__ wl, const int lo_k,
                                                              const int hi_k, const int lo_a, const int hi_a,
                                                              double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_wl) return;
    const double x = wl[j];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    double prev[6] = {0, 0, 0, 0, 0, 0};
    double a_prev = 0.0;
    const int first = lo_k < lo_a ? lo_k : lo_a, last = hi_k > hi_a ? hi_k : hi_a;
    for (int i = first; i < last; ++i) {
        const int off = gs.offset[i], rows = gs.offset[i + 1] - off;
        double qa, qs, gg;
        interp3(gs.wl + off, gs.q_abs + off, gs.q_sca + off, gs.g + off, rows, x, qa, qs, gg);
        const double a = gs.a[i];
        const double pi = 3.141592653589793;
        const double w15 = gs.w15[i], w35 = gs.w35[i];
        double cur[6];
        cur[0] = (qa + (1.0 - gg) * qs) * w15 * pi;  // Q_ext_RP * a**-1.5 * pi, read_BPASS.py:100,161
        cur[1] = (qa + qs) * w15 * pi;               // Q_ext_F  * a**-1.5 * pi, read_BPASS.py:99,162
        cur[2] = w15 * pi * qa;                      // a**(-1.5) * pi * Q_abs,  MC code.py:308
        cur[3] = w15 * pi * qs;
        cur[4] = w35 * gg;
        cur[5] = w35 * (qa + qs);
        const double da = a - a_prev;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int lo = k < 2 ? lo_k : lo_a, hi = k < 2 ? hi_k : hi_a;
            if (i > lo && i < hi) acc[k] += da * (cur[k] + prev[k]) / 2.0;
            // a single size: no integral; opacities keep the integrand, averages the tabulated values
            if (hi - lo == 1 && i == lo) acc[k] = k < 2 ? cur[k] : (k == 2 ? qa : k == 3 ? qs : k == 4 ? gg : qa + qs);
            prev[k] = cur[k];
        }
        a_prev = a;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) out[(size_t)k * n_wl + j] = acc[k];
}

// ------------------------------- launchers -------------------------------
hipError_t launch_grain_integrals(const GrainSetDev& gs, int n_wl, const double* d_wl, int lo_k, int hi_k, int lo_a,
                                  int hi_a, double* d_out, hipStream_t stream) {
    grain_integrals_kernel<<<(n_wl + 255) / 256, 256, 0, stream>>>(gs, n_wl, d_wl, lo_k, hi_k, lo_a, hi_a, d_out);
    return hipGetLastError();
}

hipError_t launch_selftest_math(const LogTableEntry* d_table, long long n, const double* d_u, double* d_log,
                                double* d_cos, double* d_sqrt, double* d_div, hipStream_t stream) {
    selftest_math_kernel<<<(unsigned int)((n + 255) / 256), 256, 0, stream>>>(d_table, n, d_u, d_log, d_cos, d_sqrt, d_div);
    return hipGetLastError();
}

size_t walk_lds_bytes(int n_mu_bins, int n_depth_bins) {
    return kMathLdsBytes + sizeof(double) * kWavesPerBlock * 2 * (size_t)n_depth_bins +
           sizeof(uint32_t) * kWavesPerBlock * (size_t)(n_mu_bins + 1);
}

// The instantiations, collected once from the translation units that hold them.
static const std::vector<WalkVariantInfo>& walk_variants() {
    static const std::vector<WalkVariantInfo> all = [] {
        std::vector<WalkVariantInfo> v;
        register_walk_base(v);  // variants 0 to 2: the run-time kernels
        register_walk_slab_mc(v);
        register_walk_slab_gray(v);
        register_walk_shell(v);
        register_walk_options(v);
        return v;
    }();
    return all;
}

int walk_variant_count() { return (int)walk_variants().size(); }
const WalkVariantInfo& walk_variant(int v) { return walk_variants()[(size_t)v]; }

int walk_find_variant(const WalkKey& k) {
    const std::vector<WalkVariantInfo>& all = walk_variants();
    for (size_t i = 0; i < all.size(); ++i) {
        const WalkKey& q = all[i].key;
        if (all[i].fixed && q.order == k.order && q.boundary == k.boundary && q.law == k.law && q.depth == k.depth &&
            q.abs_only == k.abs_only && q.shell == k.shell && q.emu == k.emu && q.rng == k.rng && q.plain == k.plain)
            return (int)i;
    }
    return -1;
}

int walk_max_blocks_per_cu(int v, size_t lds_bytes) {
    int nb = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, walk_variant(v).fn, kWalkBlock, lds_bytes);
    if (e != hipSuccess || nb < 1) nb = 1;
    return nb;
}

hipError_t launch_walk(const Cfg& cfg, const Sweep& sw, int v, int grid, hipStream_t stream,
                       unsigned int* d_next, double* d_chunk_sums, unsigned long long* d_counts,
                       unsigned long long* d_mu_hist, double* d_chunk_depth) {
    const WalkVariantInfo& info = walk_variant(v);
    const size_t lds = walk_lds_bytes(cfg.n_mu_bins, info.has_depth ? cfg.n_depth_bins : 0);
    info.fn<<<grid, kWalkBlock, lds, stream>>>(cfg, sw, d_next, d_chunk_sums, d_counts, d_mu_hist, d_chunk_depth);
    return hipGetLastError();
}

hipError_t launch_zero_results(unsigned int* d_next, unsigned long long* d_tallies, size_t n_tallies, hipStream_t stream) {
    size_t grid = (n_tallies + 255) / 256;
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    zero_results_kernel<<<(unsigned int)grid, 256, 0, stream>>>(d_next, d_tallies, n_tallies);
    return hipGetLastError();
}

hipError_t launch_finish(const Sweep& sw, const double* d_chunk_sums, double* d_slab_sums,
                         const unsigned long long* d_count_copies, unsigned long long* d_counts, size_t n_counts,
                         const unsigned long long* d_hist_copies, unsigned long long* d_mu_hist, size_t n_hist,
                         const double* d_chunk_depth, double* d_slab_depth, int n_depth, hipStream_t stream) {
    const size_t grid = (size_t)sw.n_atm * sw.n_slabs + (n_counts + n_hist + 63) / 64 +
                        (size_t)sw.n_atm * sw.n_slabs * ((2 * (size_t)n_depth + 63) / 64);
    if (grid == 0 || grid > 0x7fffffffull) return grid ? hipErrorInvalidValue : hipSuccess;
    finish_kernel<<<(unsigned int)grid, 64, 0, stream>>>(sw, d_chunk_sums, d_slab_sums, d_count_copies, d_counts,
                                                         (uint32_t)n_counts, d_hist_copies, d_mu_hist, (uint32_t)n_hist,
                                                         d_chunk_depth, d_slab_depth, (uint32_t)n_depth);
    return hipGetLastError();
}

hipError_t launch_emission(const Cfg& cfg, int rng_mode, uint32_t atm, unsigned long long first_id, long long n,
                           int32_t* d_index, double* d_mu0, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    emission_kernel<<<(unsigned int)((n + 255) / 256), 256, 0, stream>>>(cfg, rng_mode, atm, first_id, n, d_index, d_mu0);
    return hipGetLastError();
}

hipError_t launch_replay(const Cfg& cfg, double tau, long long n, int max_draws, const double* d_xi,
                         const double* d_lam, const double* d_g, const double* d_albedo, const double* d_scale,
                         const double* d_mu0, double* d_sums, unsigned long long* d_counts,
                         unsigned long long* d_mu_hist, double* d_depth, rt_photon_trace* d_trace,
                         hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const size_t lds = walk_lds_bytes(cfg.n_mu_bins, cfg.n_depth_bins);
    const unsigned int grid = (unsigned int)((n + kWalkBlock - 1) / kWalkBlock);
    replay_kernel<<<grid, kWalkBlock, lds, stream>>>(cfg, tau, n, max_draws, d_xi, d_lam, d_g, d_albedo, d_scale, d_mu0,
                                                     d_sums, d_counts, d_mu_hist, d_depth, d_trace);
    return hipGetLastError();
}

}  // namespace rt
