// rt_walk.hpp -- the persistent walk kernel (gfx950) as a template, and the registration of its instantiations.
//
//   walk_kernel<V, RNG, PLAIN>
//       Every wave pulls work items (atmosphere, chunk of photon ids) from a global queue; inside a chunk each lane
//       carries one photon in registers and dead lanes are refilled from the chunk with a ballot / prefix count.  A
//       chunk is walked by exactly one wave, its lanes are reduced in a fixed butterfly and its partial results are
//       written to the chunk's own slots, so they depend only on the photon ids in it, not on which wave, CU or GPU
//       ran it.
//
// Replaces RunPhotons (MC code.py:620-779, Numpy_random_walk.py:280-408) and the atmosphere loop of runMC
// (MC code.py:1056-1062).  The instantiations are spread over several translation units (rt_walk_*.hip), each of
// which registers its own with add_walk_variant(); rt_kernels.hip collects them.
#pragma once

#include <cstdio>
#include <type_traits>
#include <vector>

#include "rt_kernels.hpp"

namespace rt {

constexpr int kWavesPerBlock = kWalkBlock / 64;
constexpr int kHeldWaves = 5;  // waves per SIMD the lean instantiations are held to (see walk_kernel)

#ifdef RT_ABLATION
// Experiment builds only (tools/wave_trace.py): per wave of the last walk launch, 100 MHz clock at its first queue pop,
// at the start of its last chunk and at its end, and the number of chunks it walked.  One copy per translation unit
// (device symbols do not link across them); rt_debug_wave_trace reads the copy of rt_walk_base.hip, where the
// production configuration's kernels are.
[[maybe_unused]] static __device__ unsigned long long g_wave_trace[8 * 8192];
#else
[[maybe_unused]] static __device__ unsigned long long g_wave_trace[1];  // never written: Ablation::kWaveTrace is false
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
// stage_log_table().  The lean kernels read the coefficients from LDS where they use them: held in
// registers across the loop they took 28 of 115 vector registers, without them the production kernel fits 5 waves
// per SIMD instead of 4.
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

// Moves the wave's LDS histograms into global memory and zeroes them: the escape cosines into the tally copy of
// atmosphere `atm` with integer atomics (any order gives the same bits), the depth-resolved deposition into the 2*nd
// doubles at `depth_out`.  SLOT: depth_out belongs to this chunk alone and is written with plain stores (the walk
// kernel: a chunk's deposition depends on its photon ids only); otherwise it is shared and added to with atomics
// (replay kernel).
template <bool SLOT>
__device__ __forceinline__ void flush_hist(const Hist& h, int lane, int nmu, int nd, uint32_t atm,
                                           unsigned long long* d_mu_hist, double* depth_out) {
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < nmu; k += 64) {
        const uint32_t v = atomicExch(&h.mu[k], 0u);
        if (v) atomicAdd(&d_mu_hist[(size_t)atm * nmu + k], (unsigned long long)v);
    }
    for (int k = lane; k < 2 * nd; k += 64) {
        const unsigned long long bits = atomicExch(reinterpret_cast<unsigned long long*>(&h.depth_mom[k]), 0ull);
        const double v = __longlong_as_double((long long)bits);
        if (SLOT)
            depth_out[k] = v;
        else if (v != 0.0)
            atomicAdd(&depth_out[k], v);
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

// Chunk epilogue: lanes reduced in a fixed butterfly, one partial row per chunk (sums, then the depth bins); integer
// tallies and the escape histogram go to the tally copies with atomics (order-independent).
__device__ __forceinline__ void store_chunk(const Acc& acc, const WorkItem& it, const Hist& hist, int lane, int nmu,
                                            int nd, double* d_chunk_sums, unsigned long long* d_counts,
                                            unsigned long long* d_mu_hist, double* d_chunk_depth) {
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
    flush_hist<true>(hist, lane, nmu, nd, it.atm, d_mu_hist, d_chunk_depth + it.slot * (size_t)(2 * nd));
}

// Same for the lean kernels: the integer tallies are already wave totals (wave-uniform values).
template <bool REEMITS>
__device__ __forceinline__ void store_chunk_wave(const AccWave& acc, const unsigned long long (&wc)[RT_NCNT],
                                                 const WorkItem& it, const Hist& hist, int lane, int nmu, int nd,
                                                 double* d_chunk_sums, unsigned long long* d_counts,
                                                 unsigned long long* d_mu_hist, double* d_chunk_depth) {
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
    flush_hist<true>(hist, lane, nmu, nd, it.atm, d_mu_hist, d_chunk_depth + it.slot * (size_t)(2 * nd));
}

// The persistent walk kernel.  The loop is rotated so that the two lane populations that exist after a
// flight -- lanes whose photon ended (they need a new photon) and lanes whose photon goes on to
// scatter -- share ONE Philox call: each lane feeds its own counter (emission block of the new
// photon id, or the scattering block of the current one) and the 128 output bits are then read
// as the emission draws or as (azimuth, cosine).  The alias-table entry of the new photons is
// requested before the scattering arithmetic runs, so the first gather's latency hides under it.
//   V = Fixed<order, boundary, law, depth, abs_only, shell, emu>  these flags fixed at compile time: the lean
//                 loop with integer tallies per wave.  FixedBase is the reference's production configuration
//                 (RT_ORDER_MC, reflecting boundary, Henyey-Greenstein, no depth bins, slab).
//   V = Runtime   every flag read from cfg.
// Both give the same photon histories as one fly() + scatter() per photon per iteration (flight()).
//   RNG           RT_RNG_PHILOX52 (two blocks per iteration), RT_RNG_PHILOX32X4 or RT_RNG_PHILOX40 (one).
//   PLAIN (with a Fixed V) also fixes the reference's defaults rand_mu = 0 and monochromatic = 0 at compile time: the
//                 selects they cost per emission, and the scalar registers their operands occupy, go away.
//
// The lean kernels other than the production configuration's are held to 5 waves per SIMD (96 vector registers).  Left alone the compiler takes 97-117 for those
// with depth bins, the re-emitting boundary, the shell or the exact Draine law, and one wave fewer; told to, it fits the
// depth-bin kernels without a spill (+3 ... +4 %) and the others with 3 to 18 spilled dwords outside the hot blocks
// (+0 ... +5 %; the re-emitting ones gain most: the two dependent gathers of a re-emission sit in the middle of the
// flight with nothing of the wave's own to run under them).  The production configuration is left alone: its kernels
// need 87-99 registers and none of them spills.
template <class V, int RNG, bool PLAIN>
__global__ __launch_bounds__(kWalkBlock, (V::kFixed && !std::is_same<V, FixedBase>::value) ? kHeldWaves : 1) void walk_kernel(const Cfg cfg0, const Sweep sw,
                                                           unsigned int* __restrict__ d_next,
                                                           double* __restrict__ d_chunk_sums,
                                                           unsigned long long* __restrict__ d_counts,
                                                           unsigned long long* __restrict__ d_mu_hist,
                                                           double* __restrict__ d_chunk_depth) {
    // d_counts [n_copies][n_atm][RT_NCNT], d_mu_hist [n_copies][n_atm][n_mu_bins + 1], d_chunk_depth [slot][2][nd]
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr bool PACKED = RNG != RT_RNG_PHILOX52;  // the one-block streams
    typedef BlockCut<RNG> Cut;
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
        unsigned long long n_scatter = 0ull, n_scatter_failed = 0ull;
        Photon ph = {};
        PhiloxSrc src = {};
        src.atm = it.atm;
        src.k0 = cfg.k0;
        src.k1 = cfg.k1;
        bool alive = false;  // at the top of the loop a live lane is one that must scatter

        // CheckBoundary 'reemit' (MC code.py:539-547): fresh table values and direction, same history
        auto reemit = [&](Photon& p) {
            if (V::boundary(cfg) == RT_BOUNDARY_REEMIT) {
                const double p0 = p.p0, e1 = p.e1, b = p.b, dep = p.dep;
                const uint32_t n_it = p.iter;
                const int ns = p.n_scatter;
                emit_from_table<V, FAST, PLAIN>(cfg, tau, src, p);
                p.p0 = p0;
                p.e1 = e1;
                p.b = b;
                p.dep = dep;
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
            int scatter_fate = 0;
            if (active) {
                // ---- one Philox call: emission block for fresh lanes, scattering block otherwise ----
                // (PACKED: the photon's next block serves both, and the flight below as well)
                const uint32_t block = PACKED ? src.blk : (fresh ? kEmitBlock : src.blk);
                const PhiloxOut o = philox4x32_10(src.id_lo, src.id_hi, block, src.atm, src.k0, src.k1);
                // the lean kernels keep their first draw as m = 1 + u (MFORM): both uses are exact affine forms
                constexpr bool MFORM = FAST && !PACKED;
                // a photon that has flown k times has used 2k blocks -- unless its scattering law draws blocks of its own
                constexpr bool BLOCK_GUARD = MFORM && !V::kRejection;
                const double ua = MFORM ? m52(o.w0, o.w1) : ((!PACKED || fresh) ? u52(o.w0, o.w1) : Cut::azimuth(o));
                double ub = PACKED ? Cut::cosine(o) : u52(o.w2, o.w3);
                // alias entry of the new photons' bucket, requested now; scattering lanes read entry 0 (broadcast)
                const double xb = MFORM ? fma(ua, (double)cfg.n_wl, -(double)cfg.n_wl) : ua * (double)cfg.n_wl;
                int bucket = (int)xb;
                bucket = bucket < cfg.n_wl ? bucket : cfg.n_wl - 1;
                bucket = fresh ? bucket : 0;
                const uint4 raw = FAST ? gather16(cfg.alias, (uint32_t)(cfg.table_base + bucket) * 16u)
                                       : *reinterpret_cast<const uint4*>(&cfg.alias[cfg.table_base + bucket]);
                if (!fresh) {
                    ++src.blk;
                    scatter_fate = scatter<V, FAST, MFORM>(cfg, ph, src, ua, ub, 2, acc, hist);
                }
                const double prob = __hiloint2double((int)raw.y, (int)raw.x);
                int idx = (xb - (double)bucket) < prob ? bucket : (int)raw.z;
                if (!PLAIN) idx = cfg.mono_index >= 0 ? cfg.mono_index : idx;
                idx = fresh ? cfg.table_base + idx : 0;
                const TableRec rec = FAST ? load_record_buffer(cfg, idx) : load_record(cfg, idx);
                // the iteration guard: ph.iter flights so far.  In the production kernel the block counter serves and
                // ph.iter need not be kept.
                const bool too_long = BLOCK_GUARD ? src.blk >= 2u * (uint32_t)cfg.max_events : ph.iter >= (uint32_t)cfg.max_events;
                const bool guard = scatter_fate != 0 || (!fresh && too_long);
                double u_step, u_abs;
                if (PACKED) {
                    if (fresh) ++src.blk;
                    u_step = Cut::step(o);
                    u_abs = Cut::absorb(o);
                    if (!PLAIN && cfg.rand_mu && fresh) {  // the direction draw of a new photon: second uniform of emission block 0
                        const PhiloxOut e = philox4x32_10(src.id_lo, src.id_hi, kEmitBlock, src.atm, src.k0, src.k1);
                        ub = u52(e.w2, e.w3);
                    }
                } else {
                    src.pair(u_step, u_abs);
                }
                if (fresh) {
                    set_constants<V>(cfg, tau, rec, (!PLAIN && cfg.rand_mu) ? ub : 1.0, ph);
                    start_photon(ph, acc);
                    alive = true;
                }
                if (guard) {
                    fate = RT_FATE_GUARD;
                } else {
                    int n_used = 0;
                    // FAST: polynomial coefficients from LDS (5 waves per SIMD); the run-time kernel is bound by its
                    // other registers (3 waves per SIMD either way) and keeps them in registers
                    const double path = FAST ? -log_unit(u_step, hist.log_table, hist.coef) : -log_unit(u_step, hist.log_table);
                    fate = fly_path<V>(cfg, ph, path, u_abs, 2, false, acc, hist, reemit, n_used);
                }
                if (fate != 0) {
                    if (V::emu(cfg)) {  // the reference's escaped_mu list, photon by photon
                        const unsigned long long gid = ((unsigned long long)src.id_hi << 32) | src.id_lo;
                        const bool logged = fate == RT_FATE_ESCAPED_UNCOUNTED ||
                                            (fate == RT_FATE_ESCAPED && (V::gray(cfg) || V::shell(cfg)));
                        cfg.escaped_mu[it.mu_base + (gid - cfg.id_base)] = logged ? ph.mu : __builtin_nan("");
                    }
                    // (after a re-emission the photon's weight is the new wavelength's: its initial momentum is kept in ph.p0)
                    retire<PLAIN && !V::kReemit, V>(cfg, ph, fate, acc, hist);
                    alive = false;
                }
            }
            if (FAST) {
                n_guarded += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fate == RT_FATE_GUARD));
                n_escaped += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fate == RT_FATE_ESCAPED));
                if (!V::gray(cfg) && !V::shell(cfg))  // the second escape test exists in the MC order of the slab only
                    n_uncounted += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fate == RT_FATE_ESCAPED_UNCOUNTED));
                if (V::boundary(cfg) == RT_BOUNDARY_ABSORB)
                    n_floor += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fate == RT_FATE_BOUNDARY));
                if (V::kRejection)  // a rejection loop that found no cosine: the photon ends without having scattered
                    n_scatter_failed += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(scatter_fate != 0));
                // lanes without a photon at the top of the next iteration: those whose photon ended now (fate is 0
                // in lanes that were not active) and the dead lanes that got no new one (only when the chunk runs out)
                dead = __builtin_amdgcn_ballot_w64(fate != 0) | idle;
            }
        }
        if constexpr (FAST) {
            unsigned long long wc[RT_NCNT] = {};
            wc[RT_CNT_PHOTONS] = n_photons;
            wc[RT_CNT_SCATTER] = n_scatter - n_scatter_failed;
            wc[RT_CNT_ESCAPED] = n_escaped;
            wc[RT_CNT_ESCAPED_UNCOUNTED] = n_uncounted;
            wc[RT_CNT_GUARD] = n_guarded;
            wc[RT_CNT_BOUNDARY_ABSORBED] = n_floor;
            // one flight per photon started and per lane that went on from a flight, except where the photon was ended
            // before its next flight (iteration guard, failed rejection loop: fly_path itself never returns
            // RT_FATE_GUARD for an unbounded stream)
            wc[RT_CNT_FLIGHTS] = n_scatter + n_photons - n_guarded;
            // every photon started in this chunk also ended in it
            wc[RT_CNT_ABSORBED] = wc[RT_CNT_PHOTONS] - wc[RT_CNT_ESCAPED] - wc[RT_CNT_ESCAPED_UNCOUNTED] - wc[RT_CNT_GUARD] - n_floor;
            store_chunk_wave<V::kReemit>(acc, wc, it, hist, lane, nmu, nd, d_chunk_sums, d_counts, d_mu_hist, d_chunk_depth);
        } else {
            store_chunk(acc, it, hist, lane, nmu, nd, d_chunk_sums, d_counts, d_mu_hist, d_chunk_depth);
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

// ---- registration of the instantiations ----
inline void walk_variant_name(const WalkKey& k, bool fixed, char* out, size_t n) {
    static const char* const order[] = {"mc", "gray"};
    static const char* const boundary[] = {"reflect", "absorb", "reemit"};
    static const char* const law[] = {"hg", "iso", "draine", "draine_exact"};
    int at = fixed ? snprintf(out, n, "walk_kernel<%s,%s,%s", order[k.order], boundary[k.boundary], law[k.law])
                   : snprintf(out, n, "walk_kernel<runtime");
    auto add = [&](bool on, const char* s) {
        if (on && at > 0 && (size_t)at < n) at += snprintf(out + at, n - (size_t)at, ",%s", s);
    };
    if (fixed) {
        add(k.depth, "depth");
        add(k.abs_only, "abs_only");
        add(k.shell, "shell");
        add(k.emu, "escaped_mu");
    }
    add(k.rng == RT_RNG_PHILOX32X4, "philox32x4");
    add(k.rng == RT_RNG_PHILOX40, "philox40");
    add(k.plain, "plain");
    if (at > 0 && (size_t)at < n) snprintf(out + at, n - (size_t)at, ">");
}

template <class V, int RNG = RT_RNG_PHILOX52, bool PLAIN = false>
void add_walk_variant(std::vector<WalkVariantInfo>& out) {
    WalkVariantInfo v;
    v.key = V::key();
    v.key.rng = RNG;
    v.key.plain = PLAIN;
    v.fixed = V::kFixed;
    v.has_depth = V::kDepth;
    v.fn = walk_kernel<V, RNG, PLAIN>;
    walk_variant_name(v.key, v.fixed, v.name, sizeof v.name);
    out.push_back(v);
}

// the groups, one translation unit each (compiled in parallel)
void register_walk_base(std::vector<WalkVariantInfo>&);       // run-time kernels + the production configuration
void register_walk_slab_mc(std::vector<WalkVariantInfo>&);    // boundary x {hg, iso} x depth, MC loop order
void register_walk_slab_gray(std::vector<WalkVariantInfo>&);  // the same in the gray script's order
void register_walk_shell(std::vector<WalkVariantInfo>&);      // spherical shell x boundary x {hg, draine_exact} x depth
void register_walk_options(std::vector<WalkVariantInfo>&);    // draine_exact in the slab, abs_only, per-photon escaped_mu

}  // namespace rt
