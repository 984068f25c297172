// rt_kernels.hpp -- host-visible launch interface of the walk kernels.
#pragma once

#include "rt_device.hpp"

namespace rt {

// How one atmosphere's photon-id range [0, n_photons) is cut up.  The cut depends only on n_photons and n_slabs,
// never on the number of GPUs:
//   slab s   = ids [s*slab_len, min((s+1)*slab_len, n_photons))        s = 0..n_slabs-1
//   chunk c  = ids [slab_lo + c*chunk, ...) inside a slab               one wave walks one chunk
// A rank owns the slabs s with s % world == rank; the partial sums of the chunks of its j-th slab
// (j = (s - rank) / world) live at slots  slot_base + j*chunks + c.
struct PlanEntry {
    double tau;
    unsigned long long n_photons;
    unsigned long long slab_len;
    uint32_t atm;        // index of this atmosphere in the caller's arrays
    uint32_t chunk;      // photon ids per chunk
    uint32_t chunks;     // chunks per slab: ceil(slab_len / chunk)
    uint32_t slot_base;  // first chunk-partial slot of this atmosphere
    uint32_t table;      // which of the context's tables this atmosphere uses
    uint32_t pad_;
    unsigned long long mu_base;  // where this atmosphere's photons start in Cfg::escaped_mu
};
static_assert(sizeof(PlanEntry) == 56, "plan entries are uploaded as raw bytes");

struct Sweep {
    const PlanEntry* plan;        // device, n_atm entries, heaviest atmospheres first
    const uint32_t* item_prefix;  // device, n_atm + 1 entries: first queue index of each entry
    uint32_t n_atm;
    uint32_t n_items;
    uint32_t rank, world;
    uint32_t n_slabs, owned_slabs;
    uint32_t n_copies;  // copies of the integer tally arrays the waves spread their atomics over
    // Order in which the queue hands out the items of the list: positions below n_interleaved
    // (= kQueueRows * row_len) go through the list in kQueueRows rows at once, position q -> item
    // (q % kQueueRows) * row_len + q / kQueueRows, so that thick and thin atmospheres are in flight together
    // from the first wave to (almost) the last; the remaining items, the cheapest of the list, follow in order
    // and keep the end of the launch balanced.
    uint32_t n_interleaved, row_len;
    // Or (mix_groups > 0): the first mix_h items of the list, the heavy ones, alternate with the light ones behind them,
    // mix_rh heavy then mix_rl light per group for mix_groups groups; what is left follows in list order.
    uint32_t mix_h, mix_rh, mix_rl, mix_groups;
};

constexpr uint32_t kQueueRowsLog2 = 3;  // measured: 2 to 1024 rows give the same time, 8192 rows 1 % more
constexpr uint32_t kQueueRows = 1u << kQueueRowsLog2;

constexpr int kWalkBlock = 256;

// LDS per workgroup of the walk / replay kernels: math tables + per-wave histograms of n_mu_bins + 1 and 2 * n_depth_bins
// entries.  The limit leaves room for several workgroups per compute unit.
size_t walk_lds_bytes(int n_mu_bins, int n_depth_bins);
constexpr size_t kWalkLdsLimit = 64 * 1024;
// The walk kernel's instantiations (rt_walk.hpp): which run flags are compile-time constants (see Fixed<> in
// rt_device.hpp).  Variants 0 to 2 are the kernels that read every flag at run time (one per random stream);
// the others are found by their key.
typedef void (*WalkKernel)(const Cfg, const Sweep, unsigned int*, double*, unsigned long long*, unsigned long long*, double*);
struct WalkVariantInfo {
    WalkKey key;
    bool fixed;      // false: the run-time kernel
    bool has_depth;  // the kernel contains the depth tally (and needs its LDS)
    WalkKernel fn;
    char name[112];
};
constexpr int kWalkRuntime = 0;  // + rng_mode: variants 0, 1, 2 are the run-time kernels of the three random streams
int walk_variant_count();
const WalkVariantInfo& walk_variant(int v);
// the instantiation with exactly these flags fixed, or -1 (the sweep then runs a run-time kernel)
int walk_find_variant(const WalkKey& key);
int walk_max_blocks_per_cu(int v, size_t lds_bytes);

// d_counts / d_mu_hist: sw.n_copies copies each; d_chunk_depth: 2 * n_depth_bins doubles per chunk slot (may be null
// when the variant has no depth tally or n_depth_bins == 0)
hipError_t launch_walk(const Cfg& cfg, const Sweep& sw, int variant, int grid, hipStream_t stream,
                       unsigned int* d_next, double* d_chunk_sums, unsigned long long* d_counts,
                       unsigned long long* d_mu_hist, double* d_chunk_depth);

// *d_next = 0 and the tally copies zeroed (length in 64-bit words)
hipError_t launch_zero_results(unsigned int* d_next, unsigned long long* d_tallies, size_t n_tallies, hipStream_t stream);
// after the walk: every slab of every atmosphere written (the owned ones from their chunk partials, the others zero)
// -- the ten sums to d_slab_sums [n_atm][n_slabs][RT_NSUM] and, with n_depth > 0, the depth-resolved deposition to
// d_slab_depth [n_atm][n_slabs][2][n_depth] -- and the copies of the integer tallies folded into d_counts [n_counts]
// and d_mu_hist [n_hist]
hipError_t launch_finish(const Sweep& sw, const double* d_chunk_sums, double* d_slab_sums,
                         const unsigned long long* d_count_copies, unsigned long long* d_counts, size_t n_counts,
                         const unsigned long long* d_hist_copies, unsigned long long* d_mu_hist, size_t n_hist,
                         const double* d_chunk_depth, double* d_slab_depth, int n_depth, hipStream_t stream);

// the wavelength index (and direction draw) photon ids [first_id, first_id + n) of atmosphere `atm` are emitted with
hipError_t launch_emission(const Cfg& cfg, int rng_mode, uint32_t atm, unsigned long long first_id, long long n,
                           int32_t* d_index, double* d_mu0, hipStream_t stream);

hipError_t launch_replay(const Cfg& cfg, double tau, long long n, int max_draws, const double* d_xi,
                         const double* d_lam, const double* d_g, const double* d_albedo, const double* d_scale,
                         const double* d_mu0, double* d_sums, unsigned long long* d_counts,
                         unsigned long long* d_mu_hist, double* d_depth, rt_photon_trace* d_trace,
                         hipStream_t stream);

hipError_t launch_selftest_math(const LogTableEntry* d_table, long long n, const double* d_u, double* d_log,
                                double* d_cos, double* d_sqrt, double* d_div, hipStream_t stream);

}  // namespace rt

// ---- device table builder (grain-size integration of the Draine tables) ----
namespace rt {

struct GrainSetDev {
    const double* a;        // [n_sizes] grain radii, ascending
    const double* w15;      // [n_sizes] a^-1.5 (host pow, like NumPy's)
    const double* w35;      // [n_sizes] a^-3.5
    const int32_t* offset;  // [n_sizes + 1] start of each size's rows in the concatenated columns
    const double* wl;       // concatenated per-size wavelength columns, DESCENDING as in the files
    const double* q_abs;
    const double* q_sca;
    const double* g;
    int32_t n_sizes;
};

// out[k * n_wl + j], k = 0..5: int Q_ext_RP a^-1.5 pi da, int Q_ext_F a^-1.5 pi da  (sizes in [lo_k, hi_k))
//                              int Q_abs a^-1.5 pi da, int Q_sca a^-1.5 pi da, int g a^-3.5 da,
//                              int (Q_abs+Q_sca) a^-3.5 da                          (sizes in [lo_a, hi_a))
hipError_t launch_grain_integrals(const GrainSetDev& gs, int n_wl, const double* d_wl, int lo_k, int hi_k, int lo_a,
                                  int hi_a, double* d_out, hipStream_t stream);

}  // namespace rt

// ---- shell velocity ODE (rt_shell.hip); the structs are the C ABI's with device pointers ----
namespace rt {

typedef rt_shell_model ShellModel;
typedef rt_shell_regions ShellRegions;
typedef rt_shell_control ShellControl;
typedef rt_shell_result ShellResult;
typedef rt_shell_galaxy ShellGalaxy;

// gal = nullptr: one region (or point) per lane.  With the galaxy's other regions (gal != nullptr, device pointers;
// the model's own `galaxy` field is not read on the device) one wave serves a region and its lanes share the sums
// over gal_data's rows.
hipError_t launch_shell_rhs(const ShellModel& m, const ShellRegions& rg, const ShellGalaxy* gal, int n,
                            const int32_t* d_region, const double* d_r, const double* d_v, const double* d_t,
                            double* d_dvdr, double* d_dtdr, hipStream_t stream);
hipError_t launch_shell_ode(const ShellModel& m, const ShellRegions& rg, const ShellGalaxy* gal, const ShellControl& ct,
                            const ShellResult& out, hipStream_t stream);

}  // namespace rt
