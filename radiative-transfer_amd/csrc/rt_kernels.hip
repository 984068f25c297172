// rt_kernels.hip -- gfx950 kernels of the photon random walk and their launchers.
//
//   walk_kernel<FAST, PACKED, PLAIN>
//                       persistent scatter kernel (FAST: the production configuration fixed at compile
//                       time; otherwise every flag read at run time).  Every wave pulls work items
//                       (atmosphere, chunk of photon ids) from a global queue; inside a
//                       chunk each lane carries one photon in registers and dead lanes are
//                       refilled from the chunk with a ballot / prefix count.  A chunk is
//                       walked by exactly one wave and its lanes are reduced in a fixed
//                       butterfly, so a chunk's partial sums depend only on the photon
//                       ids in it, not on which wave, CU or GPU ran it.
//   slab_reduce_kernel  sums the chunk partials of a slab in chunk order.
//   replay_kernel       one photon per thread on caller-supplied uniforms.
//
// Replaces RunPhotons (MC code.py:620-779, Numpy_random_walk.py:280-408) and the
// atmosphere loop of runMC (MC code.py:1056-1062).
#include <algorithm>
#include <type_traits>

#include "rt_kernels.hpp"

namespace rt {

constexpr int kWavesPerBlock = kWalkBlock / 64;

#ifdef RT_ABLATION
// Experiment builds only (tools/wave_trace.py): per wave of the last walk launch, 100 MHz clock at its first queue pop,
// at the start of its last chunk and at its end, and the number of chunks it walked.
__device__ unsigned long long g_wave_trace[8 * 8192];
extern "C" int rt_debug_wave_trace(unsigned long long* out, int n_waves) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace), sizeof(unsigned long long) * 8 * (size_t)n_waves);
}
#else
__device__ unsigned long long g_wave_trace[1];  // never written: Ablation::kWaveTrace is false
#endif

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr size_t kLogTableBytes = sizeof(LogTableEntry) * kLogTableEntries;
constexpr size_t kCoefBytes = sizeof(double) * 16;  // kMathCoefs polynomial coefficients, padded
constexpr size_t kMathLdsBytes = kLogTableBytes + kCoefBytes;
static_assert(kMathCoefs <= 16, "coefficient block");

// LDS layout per block: the logarithm table (2 KiB) and the polynomial coefficients of ln / cos (128 B), then
// [waves][2*nd] doubles (depth momentum, depth energy), then [waves][nmu] u32 (escape-cosine histogram).
// The histograms are private to a wave; table and coefficients are shared and read-only after
// stage_log_table().  The production kernel reads the coefficients from LDS where it uses them: held in
// registers across the loop they took 28 of its 115 vector registers, without them it fits 5 waves per SIMD
// instead of 4.
__device__ __forceinline__ Hist wave_hist(unsigned char* lds_raw, int wave, int nmu, int nd) {
    double* depth_base = reinterpret_cast<double*>(lds_raw + kMathLdsBytes) + (size_t)wave * 2 * nd;
    uint32_t* mu_base =
        reinterpret_cast<uint32_t*>(lds_raw + kMathLdsBytes + sizeof(double) * (size_t)kWavesPerBlock * 2 * nd) +
        (size_t)wave * nmu;
    Hist h;
    h.mu = mu_base;
    h.depth_mom = nd ? depth_base : nullptr;
    h.depth_en = nd ? depth_base + nd : nullptr;
    h.log_table = reinterpret_cast<const LogTableEntry*>(lds_raw);
    h.coef = reinterpret_cast<const double*>(lds_raw + kLogTableBytes);
    return h;
}

// Every thread of the block calls this once, before any divergence.
__device__ __forceinline__ void stage_log_table(const Cfg& cfg, unsigned char* lds_raw) {
    LogTableEntry* dst = reinterpret_cast<LogTableEntry*>(lds_raw);
    for (int k = threadIdx.x; k < kLogTableEntries; k += blockDim.x) dst[k] = cfg.log_table[k];
    if (threadIdx.x < kMathCoefs) reinterpret_cast<double*>(lds_raw + kLogTableBytes)[threadIdx.x] = math_coef((int)threadIdx.x);
    __syncthreads();
}

__device__ __forceinline__ void clear_hist(const Hist& h, int lane, int nmu, int nd) {
    for (int k = lane; k < nmu; k += 64) h.mu[k] = 0u;
    for (int k = lane; k < 2 * nd; k += 64) h.depth_mom[k] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Moves the wave's LDS histograms into the global tallies of atmosphere `atm` and zeroes them.
__device__ __forceinline__ void flush_hist(const Hist& h, int lane, int nmu, int nd, uint32_t atm, uint32_t n_atm,
                                           unsigned long long* d_mu_hist, double* d_depth) {
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < nmu; k += 64) {
        const uint32_t v = atomicExch(&h.mu[k], 0u);
        if (v) atomicAdd(&d_mu_hist[(size_t)atm * nmu + k], (unsigned long long)v);
    }
    for (int k = lane; k < 2 * nd; k += 64) {
        const unsigned long long bits = atomicExch(reinterpret_cast<unsigned long long*>(&h.depth_mom[k]), 0ull);
        const double v = __longlong_as_double((long long)bits);
        const int which = k >= nd;  // 0: momentum, 1: energy
        if (v != 0.0) atomicAdd(&d_depth[((size_t)which * n_atm + atm) * nd + (k - which * nd)], v);
    }
    __builtin_amdgcn_wave_barrier();
}

// A wave-uniform value moved to scalar registers, so that what is computed from it runs on the scalar unit.
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// One unit of work for a wave: a chunk of consecutive photon ids of one atmosphere.
struct WorkItem {
    unsigned long long cursor, end;  // photon ids [cursor, end) before id_base is added
    double tau;
    uint32_t atm;
    uint32_t table;
    size_t slot;  // where this chunk's partial sums go
    unsigned long long mu_base;  // first entry of the atmosphere in Cfg::escaped_mu
};

// Pops the next item of the global queue (wave-uniform).  Returns false when the queue is empty.
__device__ __forceinline__ bool pop_item(const Sweep& sw, unsigned int* d_next, int lane, WorkItem& it) {
    unsigned int q = 0;
    if (lane == 0) q = atomicAdd(d_next, 1u);
    q = __builtin_amdgcn_readfirstlane(q);
    if (q >= sw.n_items) return false;
    if (sw.mix_groups) {
        const uint32_t per = sw.mix_rh + sw.mix_rl, mixed = sw.mix_groups * per;
        if (q < mixed) {
            const uint32_t g = q / per, p = q - g * per;
            q = p < sw.mix_rh ? g * sw.mix_rh + p : sw.mix_h + g * sw.mix_rl + (p - sw.mix_rh);
        } else {
            const uint32_t t = q - mixed, rem_h = sw.mix_h - sw.mix_groups * sw.mix_rh;
            q = t < rem_h ? sw.mix_groups * sw.mix_rh + t : sw.mix_h + sw.mix_groups * sw.mix_rl + (t - rem_h);
        }
    } else if (q < sw.n_interleaved) {
        q = (q & (kQueueRows - 1u)) * sw.row_len + (q >> kQueueRowsLog2);
    }
    // Sorted position lo with item_prefix[lo] <= q < item_prefix[lo+1], searched 64 ways at a time: every lane
    // reads one entry of the candidate range, a ballot picks the sub-range -- two dependent loads for up to 4096
    // atmospheres where a binary search has twelve (the wave idles through each of them once per chunk).
    uint32_t lo = 0u, span = sw.n_atm;  // candidates lo .. lo + span - 1; item_prefix[lo] <= q holds throughout
    while (span > 1u) {
        const uint32_t step = (span + 63u) >> 6;
        const uint32_t idx = lo + (uint32_t)lane * step;
        const uint32_t v = idx < lo + span ? sw.item_prefix[idx] : 0xffffffffu;
        const uint32_t below = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(v <= q));  // >= 1: lane 0 reads item_prefix[lo]
        const uint32_t end = lo + span;
        lo += (below - 1u) * step;
        span = end - lo < step ? end - lo : step;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const PlanEntry plan = sw.plan[lo];
    const unsigned int local = q - sw.item_prefix[lo];
    const unsigned int j = local / plan.chunks;  // ordinal among the slabs this rank owns
    const unsigned int c = local - j * plan.chunks;
    const unsigned long long slab = (unsigned long long)sw.rank + (unsigned long long)j * sw.world;
    const unsigned long long slab_lo = slab * plan.slab_len;
    unsigned long long slab_hi = slab_lo + plan.slab_len;
    slab_hi = slab_hi < plan.n_photons ? slab_hi : plan.n_photons;
    const unsigned long long first = slab_lo + (unsigned long long)c * plan.chunk;
    const unsigned long long e = first + plan.chunk;
    it.cursor = uniform64(first);
    it.end = uniform64(e < slab_hi ? e : slab_hi);
    it.tau = __longlong_as_double((long long)uniform64((unsigned long long)__double_as_longlong(plan.tau)));
    it.atm = __builtin_amdgcn_readfirstlane(plan.atm);
    it.table = __builtin_amdgcn_readfirstlane(plan.table);
    it.slot = (size_t)uniform64((unsigned long long)plan.slot_base + (unsigned long long)j * plan.chunks + c);
    it.mu_base = uniform64(plan.mu_base);
    return true;
}

// Chunk epilogue: lanes reduced in a fixed butterfly, one partial per chunk; integer tallies
// and histograms go to the global arrays with atomics (order-independent).
__device__ __forceinline__ void store_chunk(const Acc& acc, const WorkItem& it, const Hist& hist, int lane, int nmu,
                                            int nd, uint32_t n_atm, double* d_chunk_sums,
                                            unsigned long long* d_counts, unsigned long long* d_mu_hist,
                                            double* d_depth) {
#pragma unroll
    for (int k = 0; k < RT_NSUM; ++k) {
        const double v = wave_sum(acc.s[k]);
        if (lane == 0) d_chunk_sums[it.slot * RT_NSUM + k] = v;
    }
#pragma unroll
    for (int k = 0; k < RT_NCNT; ++k) {
        const uint32_t v = wave_sum_u32(acc.c[k]);
        if (lane == 0 && v) atomicAdd(&d_counts[(size_t)it.atm * RT_NCNT + k], (unsigned long long)v);
    }
    flush_hist(hist, lane, nmu, nd, it.atm, n_atm, d_mu_hist, d_depth);
}

// Same for the fast kernel: the integer tallies are already wave totals (wave-uniform values).
template <bool REEMITS>
__device__ __forceinline__ void store_chunk_wave(const AccWave& acc, const unsigned long long (&wc)[RT_NCNT],
                                                 const WorkItem& it, const Hist& hist, int lane, int nmu, int nd,
                                                 uint32_t n_atm, double* d_chunk_sums, unsigned long long* d_counts,
                                                 unsigned long long* d_mu_hist, double* d_depth) {
#pragma unroll
    for (int k = 0; k < RT_NSUM; ++k) {
        const double v = wave_sum(acc.s[k]);
        if (lane == 0) d_chunk_sums[it.slot * RT_NSUM + k] = v;
    }
    const uint32_t hits = wave_sum_u32(acc.hits);
    const uint32_t reemits = REEMITS ? wave_sum_u32(acc.reemits) : 0u;
#pragma unroll
    for (int k = 0; k < RT_NCNT; ++k) {
        const unsigned long long v = k == RT_CNT_BOUNDARY_HITS ? (unsigned long long)hits
                                     : k == RT_CNT_REEMITTED   ? (unsigned long long)reemits
                                                               : wc[k];
        if (lane == 0 && v) atomicAdd(&d_counts[(size_t)it.atm * RT_NCNT + k], v);
    }
    flush_hist(hist, lane, nmu, nd, it.atm, n_atm, d_mu_hist, d_depth);
}

// The persistent walk kernel.  The loop is rotated so that the two lane populations that exist after a
// flight -- lanes whose photon ended (they need a new photon) and lanes whose photon goes on to
// scatter -- share ONE Philox call: each lane feeds its own counter (emission block of the new
// photon id, or the scattering block of the current one) and the 128 output bits are then read
// as the emission draws or as (azimuth, cosine).  The alias-table entry of the new photons is
// requested before the scattering arithmetic runs, so the first gather's latency hides under it.
//   V = Fixed<order, boundary, law, depth>  these flags fixed at compile time (abs_only off, slab geometry): the lean
//                 loop with integer tallies per wave.  FixedBase is the reference's production configuration
//                 (RT_ORDER_MC, reflecting boundary, Henyey-Greenstein, no depth bins).
//   V = Runtime   every boundary / scattering law / loop order, depth bins, abs_only, shell geometry, read from cfg.
// Both give the same photon histories as one fly() + scatter() per photon per iteration (flight()).
//   PLAIN (with FAST) also fixes the reference's defaults rand_mu = 0 and monochromatic = 0 at compile time: the
//                 selects they cost per emission, and the scalar registers their operands occupy, go away.
template <class V, bool PACKED, bool PLAIN>
__global__ __launch_bounds__(kWalkBlock) void walk_kernel(const Cfg cfg0, const Sweep sw,
                                                           unsigned int* __restrict__ d_next,
                                                           double* __restrict__ d_chunk_sums,
                                                           unsigned long long* __restrict__ d_counts,
                                                           unsigned long long* __restrict__ d_mu_hist,
                                                           double* __restrict__ d_depth) {
    // d_counts [n_copies][n_atm][RT_NCNT], d_mu_hist [n_copies][n_atm][n_mu_bins + 1]
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nmu = cfg0.n_mu_bins + 1;
    constexpr bool FAST = V::kFixed;  // lean loop: wave-level integer tallies, scalar lane bookkeeping, buffer loads
    const int nd = V::kDepth ? cfg0.n_depth_bins : 0;
    const Hist hist = wave_hist(lds_raw, wave, nmu, nd);
    stage_log_table(cfg0, lds_raw);
    clear_hist(hist, lane, nmu, nd);
    // Integer tallies and the escape histogram go to one of n_copies copies of the tally arrays (folded afterwards):
    // with one copy, all the waves that finish chunks of the same atmosphere queue up on the same few cache lines.
    {
        const uint32_t copy = __builtin_amdgcn_readfirstlane(((uint32_t)blockIdx.x * kWavesPerBlock + (uint32_t)wave) % sw.n_copies);
        d_counts += (size_t)copy * sw.n_atm * RT_NCNT;
        d_mu_hist += (size_t)copy * sw.n_atm * (size_t)nmu;
    }

    unsigned long long t_begin = 0ull, t_last = 0ull, n_done = 0ull;
    unsigned long long n_iter = 0ull, n_drain = 0ull, n_drain_few = 0ull, n_lanes = 0ull;  // experiment builds only
    if constexpr (Ablation::kWaveTrace) t_begin = t_last = wall_clock64();
    WorkItem it;
    while (pop_item(sw, d_next, lane, it)) {
        if constexpr (Ablation::kWaveTrace) {
            t_last = wall_clock64();
            ++n_done;
        }
        const double tau = it.tau;
        Cfg cfg = cfg0;  // this atmosphere's table (wave-uniform)
        cfg.table_base = (int32_t)(it.table * (uint32_t)cfg0.n_wl);
        typename std::conditional<FAST, AccWave, Acc>::type acc;
        acc.clear();
        // FAST: wave totals in scalar registers.  A chunk has at most 16384 photons, so everything that happens
        // once per photon fits 32 bits; scatterings do not (up to max_events per photon).
        uint32_t n_photons = 0u, n_escaped = 0u, n_uncounted = 0u, n_guarded = 0u, n_floor = 0u;
        unsigned long long n_scatter = 0ull;
        Photon ph = {};
        PhiloxSrc src = {};
        src.atm = it.atm;
        src.k0 = cfg.k0;
        src.k1 = cfg.k1;
        bool alive = false;  // at the top of the loop a live lane is one that must scatter

        // CheckBoundary 'reemit' (MC code.py:539-547): fresh table values and direction, same history
        auto reemit = [&](Photon& p) {
            if (V::boundary(cfg) == RT_BOUNDARY_REEMIT) {
                const double p0 = p.p0, e1 = p.e1, b = p.b;
                const uint32_t n_it = p.iter;
                const int ns = p.n_scatter;
                emit_from_table(cfg, tau, src, p);
                p.p0 = p0;
                p.e1 = e1;
                p.b = b;
                p.iter = n_it;
                p.n_scatter = ns;
            }
        };

        // FAST: the lanes without a photon as a wave-uniform mask, carried from one iteration to the next, so that
        // every count the loop needs is scalar arithmetic on masks (a ballot of a lane predicate that lives in a
        // mask register costs a select and a compare per use)
        unsigned long long dead = ~0ull;
        // position in the chunk as 32-bit scalars (a chunk has at most 16384 ids; the scalar unit compares 32 bits)
        const unsigned long long id0 = it.cursor + cfg.id_base;
        // (a queued chunk can lie wholly beyond the end of a short last slab: it.end < it.cursor, no photons)
        const uint32_t len = it.end > it.cursor ? (uint32_t)(it.end - it.cursor) : 0u;
        uint32_t cur = 0u;
        for (;;) {
            // ---- hand new photon ids to dead lanes (ballot + prefix count) ----
            bool fresh = false;
            if (!FAST) dead = __builtin_amdgcn_ballot_w64(!alive);
            const uint32_t n_dead = (uint32_t)__popcll(dead);
            uint32_t n_new = 0u;               // photons started in this iteration (wave-uniform)
            unsigned long long idle = dead;  // FAST: the lanes without a photon that get none in this iteration
            if (dead != 0ull && cur < len) {
                const unsigned int my_off =
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
                const uint32_t left = len - cur;
                n_new = left < n_dead ? left : n_dead;
                // the first n_new dead lanes, in lane order, take the chunk's next ids
                const bool take = my_off < n_new;
                // (the comparison's own lane mask: a ballot of `take` would go through a select and a second compare)
                if (FAST) idle = dead & ~__builtin_amdgcn_uicmp(my_off, n_new, 36 /* unsigned < */);
                if (!alive && take) {
                    const unsigned long long gid = (id0 + cur) + my_off;  // scalar base + lane offset
                    src.id_lo = (uint32_t)gid;
                    src.id_hi = (uint32_t)(gid >> 32);
                    src.blk = 0u;
                    src.n_emit = 1u;  // the emission block of this iteration is number 0
                    fresh = true;
                }
                cur += n_dead;  // may pass len: the chunk is used up then
            }
            // FAST: the lanes whose photon goes on to scatter wait (in place, masked out) until at least `quorum` of
            // them have gathered, while the other lanes keep starting photons: in atmospheres where only a few per cent
            // of the flights end in a scattering, the scattering code then runs once for many lanes instead of in almost
            // every iteration for one or two.  With nothing left to start they always go.  The decision depends on the
            // wave's own lane states only, so a chunk's results stay reproducible.
            const bool go = !FAST || 64u - n_dead >= (uint32_t)cfg.quorum || n_new == 0u;
            const bool active = (alive && go) || fresh;
            // FAST: 64 - n_dead lanes hold a photon, n_new start one (a wave of this kernel has 64 lanes)
            if (FAST ? n_dead - n_new == 64u : __builtin_amdgcn_ballot_w64(active) == 0ull) break;
            if (FAST) {
                n_photons += n_new;
                n_scatter += go ? 64u - n_dead : 0u;
            }
            if constexpr (Ablation::kWaveTrace && FAST) {  // iterations, those after the chunk ran out, lanes at work
                const uint32_t busy = n_new + (go ? 64u - n_dead : 0u);
                ++n_iter;
                n_lanes += busy;
                if (cur >= len) {
                    ++n_drain;
                    if (busy <= 4u) ++n_drain_few;
                }
            }
            int fate = 0;
            if (active) {
                // ---- one Philox call: emission block for fresh lanes, scattering block otherwise ----
                // (PACKED: the photon's next block serves both, and the flight below as well)
                const uint32_t block = PACKED ? src.blk : (fresh ? kEmitBlock : src.blk);
                const PhiloxOut o = philox4x32_10(src.id_lo, src.id_hi, block, src.atm, src.k0, src.k1);
                // the production kernel keeps its first draw as m = 1 + u (MFORM): both uses are exact affine forms
                constexpr bool MFORM = FAST && !PACKED;
                const double ua = MFORM ? m52(o.w0, o.w1) : ((!PACKED || fresh) ? u52(o.w0, o.w1) : u32mid(o.w0));
                double ub = PACKED ? u32mid(o.w1) : u52(o.w2, o.w3);
                // alias entry of the new photons' bucket, requested now; scattering lanes read entry 0 (broadcast)
                const double xb = MFORM ? fma(ua, (double)cfg.n_wl, -(double)cfg.n_wl) : ua * (double)cfg.n_wl;
                int bucket = (int)xb;
                bucket = bucket < cfg.n_wl ? bucket : cfg.n_wl - 1;
                bucket = fresh ? bucket : 0;
                const uint4 raw = FAST ? gather16(cfg.alias, (uint32_t)(cfg.table_base + bucket) * 16u)
                                       : *reinterpret_cast<const uint4*>(&cfg.alias[cfg.table_base + bucket]);
                int scatter_fate = 0;
                if (!fresh) {
                    ++src.blk;
                    scatter_fate = scatter<V, FAST, MFORM>(cfg, ph, src, ua, ub, 2, acc, hist);
                }
                const double prob = __hiloint2double((int)raw.y, (int)raw.x);
                int idx = (xb - (double)bucket) < prob ? bucket : (int)raw.z;
                if (!PLAIN) idx = cfg.mono_index >= 0 ? cfg.mono_index : idx;
                const int local = fresh ? idx : 0;
                idx = fresh ? cfg.table_base + idx : 0;
                const TableRec rec = FAST ? load_record_buffer(cfg, idx) : load_record(cfg, idx);
                // the iteration guard: ph.iter flights so far.  In the production kernel a photon that has flown k times
                // has used 2k blocks at this point, so the block counter serves and ph.iter need not be kept.
                const bool too_long = MFORM ? src.blk >= 2u * (uint32_t)cfg.max_events : ph.iter >= (uint32_t)cfg.max_events;
                const bool guard = scatter_fate != 0 || (!fresh && too_long);
                double u_step, u_abs;
                if (PACKED) {
                    if (fresh) ++src.blk;
                    u_step = u32mid(o.w2);
                    u_abs = u32mid(o.w3);
                    if (!PLAIN && cfg.rand_mu && fresh) {  // the direction draw of a new photon: second uniform of emission block 0
                        const PhiloxOut e = philox4x32_10(src.id_lo, src.id_hi, kEmitBlock, src.atm, src.k0, src.k1);
                        ub = u52(e.w2, e.w3);
                    }
                } else {
                    src.pair(u_step, u_abs);
                }
                if (fresh) {
                    set_constants<V>(cfg, tau, rec, local, (!PLAIN && cfg.rand_mu) ? ub : 1.0, ph);
                    start_photon(ph, acc);
                    alive = true;
                }
                if (guard) {
                    fate = RT_FATE_GUARD;
                } else {
                    int n_used = 0;
                    // FAST: polynomial coefficients from LDS (5 waves per SIMD); the generic kernel is bound by its
                    // other registers (3 waves per SIMD either way) and keeps them in registers
                    const double path = FAST ? -log_unit(u_step, hist.log_table, hist.coef) : -log_unit(u_step, hist.log_table);
                    fate = fly_path<V>(cfg, ph, path, u_abs, 2, false, acc, hist, reemit, n_used);
                }
                if (fate != 0) {
                    if (!FAST && cfg.escaped_mu) {  // the reference's escaped_mu list, photon by photon
                        const unsigned long long gid = ((unsigned long long)src.id_hi << 32) | src.id_lo;
                        const bool logged = fate == RT_FATE_ESCAPED_UNCOUNTED ||
                                            (fate == RT_FATE_ESCAPED && (V::gray(cfg) || V::shell(cfg)));
                        cfg.escaped_mu[it.mu_base + (gid - cfg.id_base)] = logged ? ph.mu : __builtin_nan("");
                    }
                    retire<PLAIN, V>(cfg, ph, fate, acc, hist);
                    alive = false;
                }
            }
            if (FAST) {
                n_guarded += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fate == RT_FATE_GUARD));
                n_escaped += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fate == RT_FATE_ESCAPED));
                n_uncounted += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fate == RT_FATE_ESCAPED_UNCOUNTED));
                if (V::boundary(cfg) == RT_BOUNDARY_ABSORB)
                    n_floor += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fate == RT_FATE_BOUNDARY));
                // lanes without a photon at the top of the next iteration: those whose photon ended now (fate is 0
                // in lanes that were not active) and the dead lanes that got no new one (only when the chunk runs out)
                dead = __builtin_amdgcn_ballot_w64(fate != 0) | idle;
            }
        }
        if constexpr (FAST) {
            unsigned long long wc[RT_NCNT] = {};
            wc[RT_CNT_PHOTONS] = n_photons;
            wc[RT_CNT_SCATTER] = n_scatter;
            wc[RT_CNT_ESCAPED] = n_escaped;
            wc[RT_CNT_ESCAPED_UNCOUNTED] = n_uncounted;
            wc[RT_CNT_GUARD] = n_guarded;
            wc[RT_CNT_BOUNDARY_ABSORBED] = n_floor;
            // one flight per photon started and per scattering, except where the iteration guard ended the photon
            // after its scattering (fly_path itself never returns RT_FATE_GUARD for an unbounded stream)
            wc[RT_CNT_FLIGHTS] = n_scatter + n_photons - n_guarded;
            // every photon started in this chunk also ended in it
            wc[RT_CNT_ABSORBED] = wc[RT_CNT_PHOTONS] - wc[RT_CNT_ESCAPED] - wc[RT_CNT_ESCAPED_UNCOUNTED] - wc[RT_CNT_GUARD] - n_floor;
            store_chunk_wave<V::kReemit>(acc, wc, it, hist, lane, nmu, nd, sw.n_atm, d_chunk_sums, d_counts, d_mu_hist, d_depth);
        } else {
            store_chunk(acc, it, hist, lane, nmu, nd, sw.n_atm, d_chunk_sums, d_counts, d_mu_hist, d_depth);
        }
    }
    if constexpr (Ablation::kWaveTrace) {
        if (lane == 0) {
            unsigned long long* t = g_wave_trace + 8 * (size_t)(blockIdx.x * kWavesPerBlock + wave);
            t[0] = t_begin;
            t[1] = t_last;  // start of this wave's last chunk
            t[2] = wall_clock64();
            t[3] = n_done;
            t[4] = n_iter;
            t[5] = n_drain;      // iterations after the chunk's last id was handed out
            t[6] = n_drain_few;  // ... with at most 4 lanes at work
            t[7] = n_lanes;      // lanes at work summed over the iterations
        }
    }
}

// Zeroes what a sweep accumulates into with atomics: the queue counter, the tally copies and the depth arrays.
__global__ void zero_results_kernel(unsigned int* __restrict__ d_next, unsigned long long* __restrict__ a, size_t na,
                                    unsigned long long* __restrict__ b, size_t nb) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) *d_next = 0u;
    for (size_t i = t; i < na; i += stride) a[i] = 0ull;
    for (size_t i = t; i < nb; i += stride) b[i] = 0ull;
}

// What follows the walk, in one launch of 64-thread workgroups.
//   Workgroups [0, n_atm * n_slabs): one wave per (atmosphere, slab); the slabs of other ranks are set to zero.  An
//   owned slab's chunk partials are added in a fixed shape -- lane l takes the slots l, l + 64, ... in order, then the
//   lanes are combined by the butterfly -- that depends on the slab's chunk count only.  Every slot of the range was
//   written by the walk kernel (a chunk beyond the end of a short last slab holds zeros).
//   The remaining workgroups fold the copies of the integer tallies (counts, then the escape histogram) into the
//   caller's arrays: out[i] = sum over the copies of in[c * n + i] (integers: any order gives the same bits).
__global__ __launch_bounds__(64) void finish_kernel(const Sweep sw, const double* __restrict__ d_chunk_sums,
                                                    double* __restrict__ d_slab_sums,
                                                    const unsigned long long* __restrict__ d_count_copies,
                                                    unsigned long long* __restrict__ d_counts, uint32_t n_counts,
                                                    const unsigned long long* __restrict__ d_hist_copies,
                                                    unsigned long long* __restrict__ d_mu_hist, uint32_t n_hist) {
    const int lane = threadIdx.x;
    const unsigned int n_reduce = sw.n_atm * sw.n_slabs;
    if (blockIdx.x >= n_reduce) {
        const unsigned int i = (blockIdx.x - n_reduce) * 64u + (unsigned int)lane;
        const bool is_count = i < n_counts;
        const unsigned int j = is_count ? i : i - n_counts;
        const unsigned int n = is_count ? n_counts : n_hist;
        if (j >= n) return;
        const unsigned long long* in = is_count ? d_count_copies : d_hist_copies;
        unsigned long long s = 0ull;
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
    flush_hist(hist, lane, nmu, nd, 0u, 1u, d_mu_hist, d_depth);
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

// The instantiations.  A sweep whose flags match none of the fixed ones runs the Runtime kernel.
typedef Fixed<RT_ORDER_MC, RT_BOUNDARY_REFLECT, RT_SCATTER_HG, true> FixedDepth;
typedef Fixed<RT_ORDER_MC, RT_BOUNDARY_ABSORB, RT_SCATTER_HG, false> FixedAbsorb;
typedef Fixed<RT_ORDER_MC, RT_BOUNDARY_REEMIT, RT_SCATTER_HG, false> FixedReemit;
typedef Fixed<RT_ORDER_MC, RT_BOUNDARY_REFLECT, RT_SCATTER_ISO, false> FixedIso;
typedef Fixed<RT_ORDER_GRAY, RT_BOUNDARY_REFLECT, RT_SCATTER_HG, false> FixedGray;

typedef void (*WalkKernel)(const Cfg, const Sweep, unsigned int*, double*, unsigned long long*, unsigned long long*, double*);

static WalkKernel walk_kernel_of(WalkVariant v) {
    switch (v) {
        case kWalkPlain: return walk_kernel<FixedBase, false, true>;
        case kWalkBase: return walk_kernel<FixedBase, false, false>;
        case kWalkBasePacked: return walk_kernel<FixedBase, true, false>;
        case kWalkPlainPacked: return walk_kernel<FixedBase, true, true>;
        case kWalkDepth: return walk_kernel<FixedDepth, false, false>;
        case kWalkAbsorb: return walk_kernel<FixedAbsorb, false, false>;
        case kWalkReemit: return walk_kernel<FixedReemit, false, false>;
        case kWalkIso: return walk_kernel<FixedIso, false, false>;
        case kWalkGray: return walk_kernel<FixedGray, false, false>;
        case kWalkRuntimePacked: return walk_kernel<Runtime, true, false>;
        default: return walk_kernel<Runtime, false, false>;
    }
}

const char* walk_variant_name(WalkVariant v) {
    switch (v) {
        case kWalkPlain: return "walk_kernel<FixedBase, false, true>";
        case kWalkBase: return "walk_kernel<FixedBase, false, false>";
        case kWalkBasePacked: return "walk_kernel<FixedBase, true, false>";
        case kWalkPlainPacked: return "walk_kernel<FixedBase, true, true>";
        case kWalkDepth: return "walk_kernel<FixedDepth, false, false>";
        case kWalkAbsorb: return "walk_kernel<FixedAbsorb, false, false>";
        case kWalkReemit: return "walk_kernel<FixedReemit, false, false>";
        case kWalkIso: return "walk_kernel<FixedIso, false, false>";
        case kWalkGray: return "walk_kernel<FixedGray, false, false>";
        case kWalkRuntimePacked: return "walk_kernel<Runtime, true, false>";
        default: return "walk_kernel<Runtime, false, false>";
    }
}

int walk_max_blocks_per_cu(WalkVariant v, size_t lds_bytes) {
    int nb = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, walk_kernel_of(v), kWalkBlock, lds_bytes);
    if (e != hipSuccess || nb < 1) nb = 1;
    return nb;
}

hipError_t launch_walk(const Cfg& cfg, const Sweep& sw, WalkVariant v, int grid, hipStream_t stream,
                       unsigned int* d_next, double* d_chunk_sums, unsigned long long* d_counts,
                       unsigned long long* d_mu_hist, double* d_depth) {
    const size_t lds = walk_lds_bytes(cfg.n_mu_bins, walk_variant_has_depth(v) ? cfg.n_depth_bins : 0);
    walk_kernel_of(v)<<<grid, kWalkBlock, lds, stream>>>(cfg, sw, d_next, d_chunk_sums, d_counts, d_mu_hist, d_depth);
    return hipGetLastError();
}

hipError_t launch_zero_results(unsigned int* d_next, unsigned long long* d_tallies, size_t n_tallies, double* d_depth,
                               size_t n_depth, hipStream_t stream) {
    size_t grid = (std::max(n_tallies, n_depth) + 255) / 256;
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    zero_results_kernel<<<(unsigned int)grid, 256, 0, stream>>>(d_next, d_tallies, n_tallies,
                                                                reinterpret_cast<unsigned long long*>(d_depth), n_depth);
    return hipGetLastError();
}

hipError_t launch_finish(const Sweep& sw, const double* d_chunk_sums, double* d_slab_sums,
                         const unsigned long long* d_count_copies, unsigned long long* d_counts, size_t n_counts,
                         const unsigned long long* d_hist_copies, unsigned long long* d_mu_hist, size_t n_hist,
                         hipStream_t stream) {
    const size_t grid = (size_t)sw.n_atm * sw.n_slabs + (n_counts + n_hist + 63) / 64;
    if (grid == 0 || grid > 0x7fffffffull) return grid ? hipErrorInvalidValue : hipSuccess;
    finish_kernel<<<(unsigned int)grid, 64, 0, stream>>>(sw, d_chunk_sums, d_slab_sums, d_count_copies, d_counts,
                                                         (uint32_t)n_counts, d_hist_copies, d_mu_hist, (uint32_t)n_hist);
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
