// rt_api.hip -- C ABI of the library (include/rt_amd.h): context, tables, sweep, replay.
//
// Host side only; the kernels are in rt_kernels.hip.  Everything here talks to one GPU through
// the HIP runtime.  There is no CPU implementation of the walk in this library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is opened at run time (rt_comm_init), not linked

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "rt_kernels.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define RT_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct EventPair {
    hipEvent_t a, b;
};

}  // namespace

// Work memory of rt_run_device_static: callers that cannot allocate (HIP graph capture, tight serving loops) get the
// sweep's queue, plan and chunk partial sums from this array instead of hipMalloc.  Every GPU the library runs on
// holds its own copy until the process ends; host code never uses the array's name as an address, only the
// per-device pointer hipGetSymbolAddress returns (rt_create).  All calls on one GPU share it; the library orders them
// (g_static_last below), so that calls from different streams or contexts take turns instead of racing.
constexpr size_t kStaticScratchBytes = (size_t)32 << 20;
constexpr int kStaticMaxAtm = 4096;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

// The static work array of a GPU serves one sweep at a time.  Static sweeps enqueued on different streams (or from
// different contexts) of one GPU are therefore ordered by the library itself: every static sweep records an event
// behind its last kernel, and the next one on that GPU, if it goes to another stream, makes its stream wait for that
// event before its plan upload.  Neither operation allocates.  g_static_mutex also covers the enqueue itself, so that
// the order of the events is the order of the uploads.
constexpr int kMaxDevices = 64;
struct StaticLast {
    hipEvent_t done = nullptr;     // owned by the context that enqueued the sweep
    hipStream_t stream = nullptr;
};
static StaticLast g_static_last[kMaxDevices];
static std::mutex g_static_mutex;

struct rt_ctx {
    int device = 0;
    hipDeviceProp_t prop;
    hipStream_t stream = nullptr;
    // tables
    int32_t n_wl = 0;
    int32_t n_tables = 0;
    std::vector<double> wl;
    DevBuf<rt::TableRec> d_rec, d_rec_gray;  // weight = 1/lambda (MC loop order), = lambda (gray loop order)
    DevBuf<rt::AliasEntry> d_alias;
    DevBuf<rt::LogTableEntry> d_log_table;
    // sweep workspace
    DevBuf<unsigned char> d_work;  // plan + item prefix + queue counter + chunk partial sums (see WorkLayout)
    unsigned char* static_scratch = nullptr;  // this device's copy of the static work array (rt_run_device_static)
    std::vector<int> order;
    std::vector<unsigned long long> mu_base;
    DevBuf<double> d_escaped_mu;  // rt_run's device copy of rt_params.escaped_mu
    // pinned staging for the plan upload: the copy is asynchronous, `staged` says when the host side may be reused
    void* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    hipEvent_t staged = nullptr;
    bool staged_pending = false;
    hipEvent_t static_done = nullptr;  // recorded behind this context's latest static sweep (see g_static_last)
    // result staging for rt_run / rt_replay
    DevBuf<double> d_slab_sums, d_depth, d_sums;  // d_depth: [n_atm][n_slabs][2][n_depth_bins] (rt_run), [2][n_depth_bins] (rt_replay)
    DevBuf<unsigned long long> d_counts, d_mu_hist;
    std::vector<int> blocks_per_cu;  // occupancy of each walk-kernel variant for blocks_lds bytes of LDS (0: not yet asked)
    std::vector<size_t> blocks_lds;
    int last_variant = -1;  // the variant the last sweep ran (rt_last_kernel)
    // the collective (rt_comm_init): one RCCL communicator per context, rank and size as given there
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    // kernel timing
    std::vector<EventPair> events;
    size_t events_used = 0;
    double ms_total = 0.0;
    long long launches = 0;
};

namespace {

int check_common(const rt_params* p) {
    if (!p) return fail(RT_ERR_INVALID, "params is NULL");
    if (p->struct_size != sizeof(rt_params))  // before any other member is read: a shorter struct ends before them
        return fail(RT_ERR_INVALID,
                    "rt_params.struct_size is %u, but this library's rt_params has %zu bytes (ABI %d): the caller's binding "
                    "was typed from another version of include/rt_amd.h", p->struct_size, sizeof(rt_params), RT_ABI_VERSION);
    if (p->n_atm < 1) return fail(RT_ERR_INVALID, "n_atm must be >= 1 (got %d)", p->n_atm);
    if (p->flags & ~(uint32_t)RT_FLAG_RUNTIME_KERNEL) return fail(RT_ERR_INVALID, "unknown bits in flags (%#x)", p->flags);
    if (!p->tau_atm || !p->n_photons) return fail(RT_ERR_INVALID, "tau_atm / n_photons is NULL");
    if (p->boundary < 0 || p->boundary > 2)
        return fail(RT_ERR_INVALID, "No valid boundary condition specified (boundary=%d)", p->boundary);
    if (p->scatter < 0 || p->scatter > 3)
        return fail(RT_ERR_INVALID, "scatter=%d is not hg/iso/draine/draine_exact", p->scatter);
    if (p->loop_order < 0 || p->loop_order > 1) return fail(RT_ERR_INVALID, "loop_order=%d", p->loop_order);
    if (p->n_mu_bins < 0 || p->n_mu_bins > 4096) return fail(RT_ERR_LIMIT, "n_mu_bins must be in [0, 4096]");
    if (p->n_depth_bins < 0 || p->n_depth_bins > 1024) return fail(RT_ERR_LIMIT, "n_depth_bins must be in [0, 1024]");
    {   // both histograms are private to each of a workgroup's waves and live in LDS next to the math tables
        const size_t lds = rt::walk_lds_bytes(p->n_mu_bins, p->n_depth_bins);
        if (lds > rt::kWalkLdsLimit)
            return fail(RT_ERR_LIMIT, "n_mu_bins=%d with n_depth_bins=%d needs %zu bytes of LDS per workgroup (limit %zu: "
                        "16*(n_mu_bins+1) + 64*n_depth_bins <= %zu)", p->n_mu_bins, p->n_depth_bins, lds, rt::kWalkLdsLimit,
                        rt::kWalkLdsLimit - rt::walk_lds_bytes(-1, 0));
    }
    if (p->scatter == RT_SCATTER_DRAINE && p->loop_order != RT_ORDER_GRAY)
        return fail(RT_ERR_INVALID,
                    "scatter='draine' is only available with the gray loop order (it is disabled in the "
                    "wavelength-dependent reference script)");
    if (p->rng_mode != RT_RNG_PHILOX52 && p->rng_mode != RT_RNG_PHILOX32X4 && p->rng_mode != RT_RNG_PHILOX40)
        return fail(RT_ERR_INVALID, "rng_mode=%d is not RT_RNG_PHILOX52 / RT_RNG_PHILOX32X4 / RT_RNG_PHILOX40", p->rng_mode);
    if (p->rng_mode != RT_RNG_PHILOX52 && (p->scatter == RT_SCATTER_DRAINE || p->scatter == RT_SCATTER_DRAINE_EXACT))
        return fail(RT_ERR_INVALID, "RT_RNG_PHILOX32X4 / RT_RNG_PHILOX40 have one block per loop iteration; the Draine rejection loop needs more");
    if (p->scatter == RT_SCATTER_DRAINE_EXACT && !(p->alpha >= 0.0 && p->alpha <= 1.0))
        return fail(RT_ERR_INVALID, "scatter='draine_exact' needs 0 <= alpha <= 1 (got %g)", p->alpha);
    if (p->geometry != RT_GEOM_SLAB && p->geometry != RT_GEOM_SHELL)
        return fail(RT_ERR_INVALID, "geometry=%d is not RT_GEOM_SLAB / RT_GEOM_SHELL", p->geometry);
    if (p->geometry == RT_GEOM_SHELL && !(p->shell_inner > 0.0 && p->shell_inner < 1e300))
        return fail(RT_ERR_INVALID, "RT_GEOM_SHELL needs a finite shell_inner > 0 (got %g)", p->shell_inner);
    return RT_OK;
}

// Scattering lanes of the lean kernels wait for this many of their kind (measured: tools/quorum_sweep.py).
constexpr int kScatterQuorum = 8;

rt::Cfg make_cfg(const rt_ctx* ctx, const rt_params* p) {
    rt::Cfg c;
    memset(&c, 0, sizeof c);
    c.boundary = p->boundary;
    c.scatter = p->scatter;
    c.order = p->loop_order;
    c.rand_mu = p->rand_mu != 0;
    c.abs_only = p->abs_only != 0;
    c.n_mu_bins = p->n_mu_bins;
    c.n_depth_bins = p->n_depth_bins;
    c.max_events = p->max_events_per_photon > 0 ? p->max_events_per_photon : (1 << 20);
    if (c.max_events > (1 << 30)) c.max_events = 1 << 30;  // walk blocks stay below the emission blocks (2^31)
    c.alpha = p->alpha;
    c.h = p->h;
    c.geometry = p->geometry;
    c.shell_inner = p->shell_inner;
    c.k0 = (uint32_t)p->seed;
    c.k1 = (uint32_t)(p->seed >> 32);
    c.id_base = p->photon_id_base;
    c.rec = p->loop_order == RT_ORDER_GRAY ? ctx->d_rec_gray.p : ctx->d_rec.p;
    c.alias = ctx->d_alias.p;
    c.n_wl = ctx->n_wl;
    c.mono_index = -1;
    c.log_table = ctx->d_log_table.p;
    c.quorum = rt::Ablation::scatter_quorum(kScatterQuorum);
    c.escaped_mu = p->escaped_mu;  // a device address here (rt_run substitutes its own buffer for the caller's host array)
    return c;
}

// CreatePhotons rounds the wavelength to 4 decimals and joins on exact equality (MC code.py:394,403-404).
int find_monochromatic(const rt_ctx* ctx, double wl, int32_t* index) {
    const double want = std::round(wl * 1e4) / 1e4;
    for (int i = 0; i < ctx->n_wl; ++i)
        if (ctx->wl[i] == want || ctx->wl[i] == wl) {
            *index = i;
            return RT_OK;
        }
    return fail(RT_ERR_INVALID,
                "monochromatic wavelength %.6g has no exact match in the table (the reference would loop forever on NaN "
                "constants)", wl);
}

// Chunk length as a function of the atmosphere's slab length and optical depth only, never of the number of
// ranks, so that the chunk partial sums -- and with them every result bit -- are the same for any number of GPUs.
struct ChunkRule {
    uint32_t thin = 8192u, thick = 8192u;
    ChunkRule() { (void)rt::Ablation::chunk_rule(thin, thick); }  // product builds: no-op
};
const ChunkRule g_chunk_rule;

uint32_t chunk_for(unsigned long long slab_len, double tau) {
    // short slabs: shorter chunks, so that a small run still fills the GPU
    const uint32_t base = slab_len >= 1000000ull ? 16384u : slab_len >= 250000ull ? 8192u : slab_len >= 31250ull ? 4096u : 1024u;
    const uint32_t cap = tau >= 0.5 ? g_chunk_rule.thick : g_chunk_rule.thin;
    return base < cap ? base : cap;
}

void drain_events(rt_ctx* ctx);

hipError_t next_events(rt_ctx* ctx, EventPair** out) {
    if (ctx->events_used >= 256) drain_events(ctx);  // bound the number of unresolved timings
    if (ctx->events_used == ctx->events.size()) {
        EventPair e;
        hipError_t r = hipEventCreate(&e.a);
        if (r != hipSuccess) return r;
        r = hipEventCreate(&e.b);
        if (r != hipSuccess) return r;
        ctx->events.push_back(e);
    }
    *out = &ctx->events[ctx->events_used++];
    return hipSuccess;
}

void drain_events(rt_ctx* ctx) {
    for (size_t i = 0; i < ctx->events_used; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(ctx->events[i].b) == hipSuccess &&
            hipEventElapsedTime(&ms, ctx->events[i].a, ctx->events[i].b) == hipSuccess) {
            ctx->ms_total += ms;
            ctx->launches += 1;
        }
    }
    ctx->events_used = 0;
}


// Which instantiation of the walk kernel serves a sweep: the one with exactly the sweep's flags fixed at compile time
// where there is one, else the kernel that reads every flag at run time.
int pick_variant(const rt_params* p, const rt::Cfg& cfg) {
    const int runtime = rt::kWalkRuntime + p->rng_mode;  // one run-time kernel per random stream
    if (p->flags & RT_FLAG_RUNTIME_KERNEL) return runtime;
    rt::WalkKey k;
    k.order = p->loop_order;
    k.boundary = p->boundary;
    k.law = p->scatter;
    k.depth = p->n_depth_bins > 0;
    k.abs_only = p->abs_only != 0;
    k.shell = p->geometry == RT_GEOM_SHELL;
    k.emu = p->escaped_mu != nullptr;
    k.rng = p->rng_mode;
    k.plain = false;
    if (!p->rand_mu && cfg.mono_index < 0) {  // the reference's defaults throughout: a kernel without their selects
        k.plain = true;
        const int v = rt::walk_find_variant(k);
        if (v >= 0) return v;
        k.plain = false;
    }
    const int v = rt::walk_find_variant(k);
    return v >= 0 ? v : runtime;
}

// Device work memory of one sweep, one block of bytes:
//   [plan: n_atm PlanEntry][item prefix: n_atm + 1 u32]           uploaded with one copy
//   [queue counter, 256 bytes]                                    zeroed per sweep
//   [tally copies: n_copies x n_atm x (RT_NCNT + n_mu_bins + 1) u64]  zeroed per sweep, folded into the caller's arrays
//   [chunk partial sums: slots x RT_NSUM doubles]                 every slot is written by the wave that walks its chunk
//   [chunk depth bins: slots x 2 n_depth_bins doubles]            likewise
struct WorkLayout {
    size_t prefix_off, upload_bytes, next_off, tally_off, tally_words, sums_off, depth_off, total;
    uint32_t n_copies;
};

// Copies of the integer tally arrays: as many as fit kTallyCopyBytes, at most 64.
constexpr size_t kTallyCopyBytes = (size_t)8 << 20;
uint32_t tally_copies(size_t words_per_copy) {
    const size_t fit = kTallyCopyBytes / (8 * (words_per_copy ? words_per_copy : 1));
    return (uint32_t)(fit < 1 ? 1 : (fit > 64 ? 64 : fit));
}

WorkLayout work_layout(size_t n_atm, size_t slots, size_t tally_words_per_copy = 0, size_t n_depth = 0) {
    WorkLayout w;
    w.prefix_off = sizeof(rt::PlanEntry) * n_atm;
    w.upload_bytes = w.prefix_off + sizeof(uint32_t) * (n_atm + 1);
    w.next_off = (w.upload_bytes + 255) & ~(size_t)255;
    w.tally_off = w.next_off + 256;
    w.n_copies = tally_copies(tally_words_per_copy);
    w.tally_words = tally_words_per_copy * w.n_copies;
    w.sums_off = (w.tally_off + 8 * w.tally_words + 255) & ~(size_t)255;
    w.depth_off = (w.sums_off + sizeof(double) * slots * RT_NSUM + 255) & ~(size_t)255;
    w.total = w.depth_off + sizeof(double) * slots * 2 * n_depth;
    return w;
}

// Largest work memory a sweep may ask for (the chunk depth bins dominate: 16 n_depth_bins bytes per chunk).
constexpr size_t kMaxWorkBytes = (size_t)64 << 30;

// The sweep behind rt_run_device (work memory from the context's growable buffer, kernel timed with events) and
// rt_run_device_static (work memory = the static device array, nothing allocated, no timing).  Everything that can
// be refused is checked before the first operation is enqueued.
int enqueue_sweep(rt_ctx* ctx, const rt_params* p, void* d_slab_sums, void* d_counts, void* d_mu_hist, void* d_depth,
                  void* stream_v, bool use_static) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    int rc = check_common(p);
    if (rc) return rc;
    if (ctx->n_wl < 1) return fail(RT_ERR_STATE, "rt_set_tables must be called before rt_run");
    if (p->table_of_atm)
        for (int a = 0; a < p->n_atm; ++a)
            if (p->table_of_atm[a] < 0 || p->table_of_atm[a] >= ctx->n_tables)
                return fail(RT_ERR_INVALID, "atmosphere %d asks for table %d of %d", a, p->table_of_atm[a], ctx->n_tables);
    if (!d_slab_sums || !d_counts || !d_mu_hist) return fail(RT_ERR_INVALID, "device result buffer is NULL");
    if (p->n_depth_bins > 0 && !d_depth) return fail(RT_ERR_INVALID, "d_slab_depth is NULL but n_depth_bins > 0");
    const int n_slabs = p->n_slabs > 0 ? p->n_slabs : 8;
    const int world = p->world > 0 ? p->world : 1;
    if (p->rank < 0 || p->rank >= world) return fail(RT_ERR_INVALID, "rank %d not in [0,%d)", p->rank, world);
    if (n_slabs % world != 0) return fail(RT_ERR_INVALID, "n_slabs (%d) must be a multiple of world (%d)", n_slabs, world);
    if (use_static && ctx->device >= kMaxDevices)
        return fail(RT_ERR_LIMIT, "rt_run_device_static serves device ids below %d (this context is on %d)", kMaxDevices, ctx->device);
    if (use_static && p->n_atm > kStaticMaxAtm)
        return fail(RT_ERR_LIMIT, "rt_run_device_static takes at most %d atmospheres per call (got %d)", kStaticMaxAtm, p->n_atm);
    RT_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_v;  // NULL is the (legacy) default stream

    rt::Cfg cfg = make_cfg(ctx, p);
    if (p->monochromatic_wl != 0.0) {
        rc = find_monochromatic(ctx, p->monochromatic_wl, &cfg.mono_index);
        if (rc) return rc;
    }

    // ---- plan: heaviest atmospheres first, chunked per slab ----
    const size_t na = (size_t)p->n_atm;
    const uint32_t owned = (uint32_t)(n_slabs / world);
    if (ctx->order.size() < na) {  // (never in the static path: n_atm <= kStaticMaxAtm)
        ctx->order.resize(na);
        ctx->mu_base.resize(na);
    }
    int* order = ctx->order.data();
    std::iota(order, order + na, 0);
    std::sort(order, order + na, [&](int a, int b) {  // ties keep the caller's order (no scratch memory needed)
        const double ca = std::min(p->tau_atm[a], 4.0) * (double)p->n_photons[a];
        const double cb = std::min(p->tau_atm[b], 4.0) * (double)p->n_photons[b];
        return ca != cb ? ca > cb : a < b;
    });
    const WorkLayout up = work_layout(na, 0);
    if (ctx->staged_pending) {  // the previous call's upload must have left the staging buffer
        RT_HIP(hipEventSynchronize(ctx->staged));
        ctx->staged_pending = false;
    }
    if (ctx->h_stage_bytes < up.upload_bytes) {  // (never in the static path: the staging buffer holds kStaticMaxAtm)
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr;
        ctx->h_stage_bytes = 0;
        RT_HIP(hipHostMalloc(&ctx->h_stage, up.upload_bytes, hipHostMallocDefault));
        ctx->h_stage_bytes = up.upload_bytes;
    }
    rt::PlanEntry* plan = static_cast<rt::PlanEntry*>(ctx->h_stage);
    uint32_t* prefix = reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->h_stage) + up.prefix_off);
    unsigned long long items = 0;
    if (p->escaped_mu) {  // atmosphere a's photons start at the sum of the photon counts before it, in the caller's order
        unsigned long long at = 0;
        for (int a = 0; a < p->n_atm; ++a) {
            ctx->mu_base[(size_t)a] = at;
            at += p->n_photons[a];
        }
    }
    for (int k = 0; k < p->n_atm; ++k) {
        const int a = order[k];
        rt::PlanEntry& pl = plan[k];
        pl.mu_base = p->escaped_mu ? ctx->mu_base[(size_t)a] : 0ull;
        pl.tau = p->tau_atm[a];
        pl.n_photons = p->n_photons[a];
        pl.slab_len = (pl.n_photons + (unsigned long long)n_slabs - 1) / (unsigned long long)n_slabs;
        pl.atm = (uint32_t)a;
        pl.table = p->table_of_atm ? (uint32_t)p->table_of_atm[a] : 0u;
        pl.pad_ = 0u;
        pl.chunk = chunk_for(pl.slab_len, pl.tau);
        pl.chunks = (uint32_t)((pl.slab_len + pl.chunk - 1) / pl.chunk);
        if (pl.chunks == 0) pl.chunks = 1;  // an empty atmosphere still has one (empty) chunk per slab
        pl.slot_base = (uint32_t)items;
        prefix[k] = (uint32_t)items;
        items += (unsigned long long)owned * pl.chunks;  // one slot per queued chunk
        if (items > 0x7fffffffull) return fail(RT_ERR_LIMIT, "too many work items (%llu)", items);
    }
    prefix[p->n_atm] = (uint32_t)items;
    const size_t nmu = (size_t)p->n_mu_bins + 1;
    const WorkLayout lay = work_layout(na, (size_t)items, na * (RT_NCNT + nmu), (size_t)p->n_depth_bins);
    if (lay.total > kMaxWorkBytes)
        return fail(RT_ERR_LIMIT, "this sweep needs %zu bytes of work memory (%llu chunks x %d depth bins); the limit is %zu: "
                    "fewer depth bins or fewer photons per call", lay.total, items, p->n_depth_bins, kMaxWorkBytes);

    unsigned char* work = nullptr;
    if (use_static) {
        if (lay.total > kStaticScratchBytes)
            return fail(RT_ERR_LIMIT,
                        "rt_run_device_static: this sweep needs %zu bytes of work memory (%llu chunks), the static array "
                        "holds %zu; use rt_run_device", lay.total, items, kStaticScratchBytes);
        work = ctx->static_scratch;
    } else {
        RT_HIP(ctx->d_work.reserve(lay.total));
        work = ctx->d_work.p;
    }
    unsigned int* d_next = reinterpret_cast<unsigned int*>(work + lay.next_off);
    double* d_chunk_sums = reinterpret_cast<double*>(work + lay.sums_off);
    double* d_chunk_depth = reinterpret_cast<double*>(work + lay.depth_off);
    unsigned long long* d_tally = reinterpret_cast<unsigned long long*>(work + lay.tally_off);
    unsigned long long* d_count_copies = d_tally;                                           // [n_copies][n_atm][RT_NCNT]
    unsigned long long* d_hist_copies = d_tally + (size_t)lay.n_copies * na * RT_NCNT;     // [n_copies][n_atm][nmu]

    rt::Sweep sw;
    sw.plan = reinterpret_cast<const rt::PlanEntry*>(work);
    sw.item_prefix = reinterpret_cast<const uint32_t*>(work + lay.prefix_off);
    sw.n_atm = (uint32_t)p->n_atm;
    sw.n_items = (uint32_t)items;
    sw.n_interleaved = 0u;
    sw.row_len = 0u;
    sw.mix_h = sw.mix_rh = sw.mix_rl = sw.mix_groups = 0u;
    sw.rank = (uint32_t)p->rank;
    sw.world = (uint32_t)world;
    sw.n_slabs = (uint32_t)n_slabs;
    sw.owned_slabs = owned;
    sw.n_copies = lay.n_copies;

    const int variant = pick_variant(p, cfg);
    const size_t lds = rt::walk_lds_bytes(cfg.n_mu_bins, rt::walk_variant(variant).has_depth ? cfg.n_depth_bins : 0);  // within the limit: check_common
    int& cached = ctx->blocks_per_cu[(size_t)variant];  // asked once per variant and histogram size (no allocation)
    if (cached <= 0 || ctx->blocks_lds[(size_t)variant] != lds) {
        cached = rt::walk_max_blocks_per_cu(variant, lds);
        ctx->blocks_lds[(size_t)variant] = lds;
    }
    ctx->last_variant = variant;
    const int per_cu = cached;
    long long grid = (long long)ctx->prop.multiProcessorCount * per_cu;
    const long long need = ((long long)items + 3) / 4;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;

    {   // Order of the queue (results do not depend on it: a chunk's partial sums depend on its photon ids only).
        // Long launches, from about 8 chunks per resident wave on: the list is walked in interleaved rows, so that
        // thick atmospheres (bound by instruction issue) and thin ones (all emission, bound by cache-line traffic) are
        // resident together and each hides under the other; the cheapest 2 chunks per wave stay in order for the end
        // of the launch (measured: no ordered tail costs 2 %, 1 to 8 per wave are equal; rows at 5 chunks per wave
        // -4 %, at 10 +1.5 %, at 24 +3.5 %).
        // Short launches (the share of one of 8 ranks): rows would put thick chunks at the very end, where their
        // length decides the launch's; instead the chunks of the atmospheres with tau >= 0.1, heaviest first,
        // alternate with the thin ones behind them in the ratio of their numbers, so that the thin chunks still
        // have thick ones beside them and the launch ends on thin, short chunks (measured at 6 chunks per wave:
        // -2.8 % against heaviest-first order, which leaves the thin third to itself at the end; +4 % at 48).
        const unsigned long long per_wave = rt::Ablation::interleave_min(8);
        unsigned tail_per_wave = 2u, tail_cap_div = 8u;
        rt::Ablation::tail_rule(tail_per_wave, tail_cap_div);  // product builds: no-op
        const unsigned long long tail = std::min<unsigned long long>(items / tail_cap_div, (unsigned long long)grid * 4 * tail_per_wave);
        const unsigned long long rows = rt::kQueueRows;
        double tau_split = 0.1, beta = 1.0;
        const bool forced = rt::Ablation::mix_rule(tau_split, beta);  // product builds: false
        if (!forced && items >= (unsigned long long)grid * 4 * per_wave && items - tail >= rows * 8) {
            sw.row_len = (uint32_t)((items - tail) / rows);
            sw.n_interleaved = (uint32_t)(sw.row_len * rows);
        } else {
            unsigned long long heavy = 0;
            for (int k = 0; k < p->n_atm; ++k)
                if (plan[k].tau >= tau_split) heavy = prefix[k + 1];  // (the list is sorted by cost: a prefix when the photon counts are equal)
            const unsigned long long light = items - heavy, mixed_light = (unsigned long long)(beta * (double)light);
            if (heavy > 0 && mixed_light > 0) {
                sw.mix_h = (uint32_t)heavy;
                if (heavy >= mixed_light) {
                    sw.mix_rl = 1u;
                    sw.mix_rh = (uint32_t)((heavy + mixed_light / 2) / mixed_light);
                } else {
                    sw.mix_rh = 1u;
                    sw.mix_rl = (uint32_t)((mixed_light + heavy / 2) / heavy);
                }
                sw.mix_groups = (uint32_t)std::min(heavy / sw.mix_rh, mixed_light / sw.mix_rl);
            }
        }
    }

    // ---- device work: one upload, one kernel that zeroes what the walk accumulates into, the walk, slab sums and tally fold
    std::unique_lock<std::mutex> static_lock;
    if (use_static) {  // the GPU's static array is free once the previous static sweep's last kernel has run
        static_lock = std::unique_lock<std::mutex>(g_static_mutex);
        const StaticLast& last = g_static_last[ctx->device];
        if (last.done && last.stream != stream) RT_HIP(hipStreamWaitEvent(stream, last.done, 0));
    }
    RT_HIP(hipMemcpyAsync(work, ctx->h_stage, lay.upload_bytes, hipMemcpyHostToDevice, stream));
    RT_HIP(hipEventRecord(ctx->staged, stream));
    ctx->staged_pending = true;
    RT_HIP(rt::launch_zero_results(d_next, d_tally, lay.tally_words, stream));
    EventPair* ev = nullptr;
    if (!use_static) {  // timing events are created on demand, which allocates: the static path is not timed
        RT_HIP(next_events(ctx, &ev));
        RT_HIP(hipEventRecord(ev->a, stream));
    }
    RT_HIP(rt::launch_walk(cfg, sw, variant, (int)grid, stream, d_next, d_chunk_sums, d_count_copies,
                           d_hist_copies, d_chunk_depth));
    if (ev) RT_HIP(hipEventRecord(ev->b, stream));
    RT_HIP(rt::launch_finish(sw, d_chunk_sums, (double*)d_slab_sums, d_count_copies, (unsigned long long*)d_counts,
                             na * RT_NCNT, d_hist_copies, (unsigned long long*)d_mu_hist, na * nmu, d_chunk_depth,
                             (double*)d_depth, p->n_depth_bins, stream));
    if (use_static) {
        RT_HIP(hipEventRecord(ctx->static_done, stream));
        g_static_last[ctx->device].done = ctx->static_done;
        g_static_last[ctx->device].stream = stream;
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION | rt::Ablation::kAbiTag; }  // experiment builds are tagged

const char* rt_last_error(void) { return g_last_error.c_str(); }

int rt_create(rt_ctx** out, int device_id) {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RT_ERR_DEVICE, "no HIP device visible (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(RT_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, n);
    RT_HIP(hipSetDevice(device_id));
    rt_ctx* ctx = new rt_ctx();
    ctx->device = device_id;
    e = hipGetDeviceProperties(&ctx->prop, device_id);
    if (e != hipSuccess) {
        delete ctx;
        return fail(RT_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    }
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = ctx->prop.gcnArchName;
        delete ctx;
        return fail(RT_ERR_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id,
                    arch.c_str());
    }
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->staged, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->static_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (ctx->staged) (void)hipEventDestroy(ctx->staged);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(RT_ERR_DEVICE, "hipStreamCreate / hipEventCreate: %s", hipGetErrorString(e));
    }
    // this device's address of the static work array, and host staging for the largest plan the static path accepts
    {
        void* sp = nullptr;
        e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
        if (e == hipSuccess) {
            ctx->static_scratch = static_cast<unsigned char*>(sp);
            ctx->h_stage_bytes = work_layout((size_t)kStaticMaxAtm, 0).upload_bytes;
            e = hipHostMalloc(&ctx->h_stage, ctx->h_stage_bytes, hipHostMallocDefault);
            if (e != hipSuccess) ctx->h_stage_bytes = 0;
        }
        if (e != hipSuccess) {
            rt_destroy(ctx);
            return fail(RT_ERR_DEVICE, "static work array / staging buffer: %s", hipGetErrorString(e));
        }
        ctx->order.resize(kStaticMaxAtm);
        ctx->mu_base.resize(kStaticMaxAtm);
        // Load the library's device code now (the runtime does that, and allocates for it, at the first launch from a
        // code object): rt_run_device_static must find nothing left to set up.  The launch writes nothing.
        e = rt::launch_zero_results(nullptr, nullptr, 0, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            rt_destroy(ctx);
            return fail(RT_ERR_DEVICE, "first kernel launch: %s", hipGetErrorString(e));
        }
        // occupancy of every walk-kernel variant with the default histogram (other sizes are asked when they occur)
        ctx->blocks_per_cu.assign((size_t)rt::walk_variant_count(), 0);
        ctx->blocks_lds.assign((size_t)rt::walk_variant_count(), 0);
        for (int v = 0; v < rt::walk_variant_count(); ++v) {
            ctx->blocks_lds[(size_t)v] = rt::walk_lds_bytes(10, 0);  // the default histogram; other sizes are queried when they occur
            ctx->blocks_per_cu[(size_t)v] = rt::walk_max_blocks_per_cu(v, ctx->blocks_lds[(size_t)v]);
        }
    }
    // logarithm table for the kernels' ln(): c_j = 1 + (j + 1/2)/128, inv_c = fl(1/c_j), l = -ln(inv_c)
    std::vector<rt::LogTableEntry> tab(rt::kLogTableEntries);
    for (int j = 0; j < rt::kLogTableEntries; ++j) {
        const double c = 1.0 + ((double)j + 0.5) / (double)rt::kLogTableEntries;
        tab[j].inv_c = 1.0 / c;
        tab[j].l = (double)(-logl((long double)tab[j].inv_c));
    }
    e = ctx->d_log_table.reserve(tab.size());
    if (e == hipSuccess)
        e = hipMemcpy(ctx->d_log_table.p, tab.data(), sizeof(rt::LogTableEntry) * tab.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        rt_destroy(ctx);
        return fail(RT_ERR_DEVICE, "log table upload: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return RT_OK;
}

int rt_destroy(rt_ctx* ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->static_done) (void)hipEventSynchronize(ctx->static_done);  // its latest static sweep, on whatever stream
    (void)rt_comm_destroy(ctx);
    for (auto& e : ctx->events) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    ctx->d_rec.release();
    ctx->d_alias.release();
    ctx->d_rec_gray.release();
    ctx->d_log_table.release();
    ctx->d_work.release();
    ctx->d_escaped_mu.release();
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->staged) (void)hipEventDestroy(ctx->staged);
    if (ctx->static_done) {
        std::lock_guard<std::mutex> lock(g_static_mutex);
        if (ctx->device < kMaxDevices && g_static_last[ctx->device].done == ctx->static_done) g_static_last[ctx->device] = StaticLast();
        (void)hipEventDestroy(ctx->static_done);
    }
    ctx->d_slab_sums.release();
    ctx->d_depth.release();
    ctx->d_sums.release();
    ctx->d_counts.release();
    ctx->d_mu_hist.release();
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

int rt_device_info(rt_ctx* ctx, char* name, int name_len, int* n_cu, int* wave_size, int* clock_khz) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (wave_size) *wave_size = ctx->prop.warpSize;
    if (clock_khz) *clock_khz = ctx->prop.clockRate;
    return RT_OK;
}

int rt_set_tables_multi(rt_ctx* ctx, int32_t n_tables, int32_t n_wl, const double* wl, const double* g,
                        const double* albedo, const double* scale, const double* pdf) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    if (n_tables < 1 || n_wl < 1 || !wl || !g || !albedo || !scale) return fail(RT_ERR_INVALID, "empty or NULL table");
    if (n_wl > (1 << 24)) return fail(RT_ERR_LIMIT, "n_wl %d exceeds 2^24", n_wl);
    // the production kernel addresses the records with 32-bit byte offsets (32 bytes per record)
    if ((long long)n_tables * n_wl > (1ll << 26)) return fail(RT_ERR_LIMIT, "n_tables * n_wl exceeds 2^26");
    RT_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)n_wl, total_n = n * (size_t)n_tables;
    std::vector<rt::TableRec> rec(total_n);
    std::vector<rt::AliasEntry> alias(total_n);
    std::vector<double> w(n), q(n);
    std::vector<int32_t> small, large;
    small.reserve(n);
    large.reserve(n);
    for (int t = 0; t < n_tables; ++t) {
        const size_t o = (size_t)t * n;
        double total = 0.0;
        for (int i = 0; i < n_wl; ++i) {
            rec[o + i] = {g[o + i], albedo[o + i], scale[o + i], 1.0 / wl[i]};
            double p = pdf ? pdf[o + i] : 1.0;
            if (!(p > 0.0)) p = 0.0;  // also maps NaN to 0
            if (p > 1.0) p = 1.0;     // AcceptReject accepts with probability min(PDF, 1)
            w[i] = p;
            total += p;
        }
        if (!(total > 0.0)) return fail(RT_ERR_INVALID, "emission law of table %d has no positive weight", t);
        // Alias table (Vose).  The construction order is part of the definition of the photon streams:
        // q_i = (w_i * n) / total; wavelengths with q < 1 ("small") and q >= 1 ("large") are kept on two
        // stacks filled in ascending index order; pop one of each, the small one keeps q and points to the
        // large one, whose excess (q_l + q_s) - 1 decides on which stack it goes back.
        small.clear();
        large.clear();
        for (int i = 0; i < n_wl; ++i) {
            q[i] = (w[i] * (double)n_wl) / total;
            (q[i] < 1.0 ? small : large).push_back(i);
        }
        rt::AliasEntry* al = alias.data() + o;
        while (!small.empty() && !large.empty()) {
            const int32_t s = small.back(), l = large.back();
            small.pop_back();
            large.pop_back();
            al[s] = {q[s], l, 0};
            q[l] = (q[l] + q[s]) - 1.0;
            (q[l] < 1.0 ? small : large).push_back(l);
        }
        for (int32_t i : large) al[i] = {1.0, i, 0};
        for (int32_t i : small) al[i] = {1.0, i, 0};  // only reachable through rounding; q is 1 to within an ulp
    }
    RT_HIP(ctx->d_rec.reserve(total_n));
    RT_HIP(ctx->d_alias.reserve(total_n));
    // a sweep enqueued on a caller's stream (rt_run_device) may still be reading the old tables
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipMemcpy(ctx->d_rec.p, rec.data(), sizeof(rt::TableRec) * total_n, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->d_alias.p, alias.data(), sizeof(rt::AliasEntry) * total_n, hipMemcpyHostToDevice));
    for (size_t k = 0; k < total_n; ++k) rec[k].weight = wl[k % n];  // the gray loop order's records
    RT_HIP(ctx->d_rec_gray.reserve(total_n));
    RT_HIP(hipMemcpy(ctx->d_rec_gray.p, rec.data(), sizeof(rt::TableRec) * total_n, hipMemcpyHostToDevice));
    ctx->n_wl = n_wl;
    ctx->n_tables = n_tables;
    ctx->wl.assign(wl, wl + n_wl);
    return RT_OK;
}

int rt_set_tables(rt_ctx* ctx, int32_t n_wl, const double* wl, const double* g, const double* albedo,
                  const double* scale, const double* pdf) {
    return rt_set_tables_multi(ctx, 1, n_wl, wl, g, albedo, scale, pdf);
}

int rt_run_device(rt_ctx* ctx, const rt_params* p, void* d_slab_sums, void* d_counts, void* d_mu_hist, void* d_depth,
                  void* stream_v) {
    return enqueue_sweep(ctx, p, d_slab_sums, d_counts, d_mu_hist, d_depth, stream_v, false);
}

int rt_run_device_static(rt_ctx* ctx, const rt_params* p, void* d_slab_sums, void* d_counts, void* d_mu_hist,
                         void* d_depth, void* stream_v) {
    return enqueue_sweep(ctx, p, d_slab_sums, d_counts, d_mu_hist, d_depth, stream_v, true);
}

int rt_run(rt_ctx* ctx, const rt_params* p, rt_tallies* out) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    int rc = check_common(p);
    if (rc) return rc;
    const int n_slabs = p->n_slabs > 0 ? p->n_slabs : 8;
    const size_t na = (size_t)p->n_atm, nmu = (size_t)p->n_mu_bins + 1, nd = (size_t)p->n_depth_bins;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(ctx->d_slab_sums.reserve(na * n_slabs * RT_NSUM));
    RT_HIP(ctx->d_counts.reserve(na * RT_NCNT));
    RT_HIP(ctx->d_mu_hist.reserve(na * nmu));
    if (nd) RT_HIP(ctx->d_depth.reserve(na * n_slabs * 2 * nd));
    rt_params run = *p;
    size_t n_mu = 0;
    if (p->escaped_mu) {  // the kernel writes a device array; the caller's host array is filled from it below
        for (size_t a = 0; a < na; ++a) n_mu += (size_t)p->n_photons[a];
        const hipError_t e = ctx->d_escaped_mu.reserve(n_mu ? n_mu : 1);
        if (e != hipSuccess)
            return fail(RT_ERR_DEVICE, "escaped_mu: %zu bytes of device memory for %zu photons: %s", 8 * n_mu, n_mu, hipGetErrorString(e));
        // entries of slabs this rank does not walk are reported as NaN ("did not leave through the logged test here")
        RT_HIP(hipMemsetAsync(ctx->d_escaped_mu.p, 0xff, sizeof(double) * n_mu, ctx->stream));
        run.escaped_mu = ctx->d_escaped_mu.p;
    }
    rc = rt_run_device(ctx, &run, ctx->d_slab_sums.p, ctx->d_counts.p, ctx->d_mu_hist.p, nd ? ctx->d_depth.p : nullptr,
                       ctx->stream);
    if (rc) return rc;
    if (p->escaped_mu && n_mu)
        RT_HIP(hipMemcpyAsync(p->escaped_mu, ctx->d_escaped_mu.p, sizeof(double) * n_mu, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<double> slab(na * n_slabs * RT_NSUM);
    RT_HIP(hipMemcpyAsync(slab.data(), ctx->d_slab_sums.p, sizeof(double) * slab.size(), hipMemcpyDeviceToHost,
                          ctx->stream));
    if (out->counts)
        RT_HIP(hipMemcpyAsync(out->counts, ctx->d_counts.p, sizeof(uint64_t) * na * RT_NCNT, hipMemcpyDeviceToHost,
                              ctx->stream));
    if (out->mu_hist)
        RT_HIP(hipMemcpyAsync(out->mu_hist, ctx->d_mu_hist.p, sizeof(uint64_t) * na * nmu, hipMemcpyDeviceToHost,
                              ctx->stream));
    std::vector<double> slab_depth(nd && (out->depth_momentum || out->depth_energy || out->slab_depth) ? na * n_slabs * 2 * nd : 0);
    if (!slab_depth.empty())
        RT_HIP(hipMemcpyAsync(slab_depth.data(), ctx->d_depth.p, sizeof(double) * slab_depth.size(), hipMemcpyDeviceToHost,
                              ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "walk kernel failed: %s", hipGetErrorString(e));
    if (out->slab_sums) memcpy(out->slab_sums, slab.data(), sizeof(double) * slab.size());
    if (out->sums) {
        for (size_t a = 0; a < na; ++a)
            for (int k = 0; k < RT_NSUM; ++k) {
                double s = 0.0;
                for (int sl = 0; sl < n_slabs; ++sl) s += slab[(a * n_slabs + sl) * RT_NSUM + k];
                out->sums[a * RT_NSUM + k] = s;
            }
    }
    if (!slab_depth.empty() && out->slab_depth) memcpy(out->slab_depth, slab_depth.data(), sizeof(double) * slab_depth.size());
    if (!slab_depth.empty()) {  // the slab axis summed in index order, like the sums: the same bits on any number of GPUs
        for (size_t a = 0; a < na; ++a)
            for (size_t k = 0; k < 2 * nd; ++k) {
                double s = 0.0;
                for (int sl = 0; sl < n_slabs; ++sl) s += slab_depth[(a * n_slabs + sl) * 2 * nd + k];
                double* dst = k < nd ? out->depth_momentum : out->depth_energy;
                if (dst) dst[a * nd + (k < nd ? k : k - nd)] = s;
            }
    }
    return RT_OK;
}

int rt_replay(rt_ctx* ctx, const rt_params* p, int64_t n, int32_t max_draws, const double* xi, const double* lambda,
              const double* g, const double* albedo, const double* scale, const double* mu0, rt_tallies* out,
              rt_photon_trace* per_photon) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    int rc = check_common(p);
    if (rc) return rc;
    if (n < 0 || max_draws < 0) return fail(RT_ERR_INVALID, "negative size");
    if (p->geometry != RT_GEOM_SLAB) return fail(RT_ERR_INVALID, "rt_replay walks the slab geometry only");
    if (n > 0 && (!xi || !lambda || !g || !albedo || !scale || !mu0)) return fail(RT_ERR_INVALID, "NULL input array");
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    if ((double)n * (double)max_draws > 4e9) return fail(RT_ERR_LIMIT, "xi matrix larger than 4e9 entries");
    RT_HIP(hipSetDevice(ctx->device));
    rt::Cfg cfg = make_cfg(ctx, p);
    const size_t nn = (size_t)n, nmu = (size_t)p->n_mu_bins + 1, nd = (size_t)p->n_depth_bins;
    DevBuf<double> d_xi, d_in;
    DevBuf<rt_photon_trace> d_tr;
    auto cleanup = [&]() {
        d_xi.release();
        d_in.release();
        d_tr.release();
    };
    hipError_t e = hipSuccess;
    if ((e = d_xi.reserve(std::max<size_t>(nn * (size_t)max_draws, 1))) != hipSuccess ||
        (e = d_in.reserve(std::max<size_t>(5 * nn, 1))) != hipSuccess ||
        (e = d_tr.reserve(std::max<size_t>(nn, 1))) != hipSuccess ||
        (e = ctx->d_sums.reserve(RT_NSUM)) != hipSuccess || (e = ctx->d_counts.reserve(RT_NCNT)) != hipSuccess ||
        (e = ctx->d_mu_hist.reserve(nmu)) != hipSuccess || (e = ctx->d_depth.reserve(std::max<size_t>(2 * nd, 1))) != hipSuccess) {
        cleanup();
        return fail(RT_ERR_DEVICE, "device allocation failed: %s", hipGetErrorString(e));
    }
    hipStream_t s = ctx->stream;
    const double* cols[5] = {lambda, g, albedo, scale, mu0};
    if (nn) {
        e = hipMemcpyAsync(d_xi.p, xi, sizeof(double) * nn * (size_t)max_draws, hipMemcpyHostToDevice, s);
        for (int c = 0; c < 5 && e == hipSuccess; ++c)
            e = hipMemcpyAsync(d_in.p + c * nn, cols[c], sizeof(double) * nn, hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_sums.p, 0, sizeof(double) * RT_NSUM, s);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_counts.p, 0, sizeof(uint64_t) * RT_NCNT, s);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_mu_hist.p, 0, sizeof(uint64_t) * nmu, s);
    if (e == hipSuccess && nd) e = hipMemsetAsync(ctx->d_depth.p, 0, sizeof(double) * 2 * nd, s);
    if (e == hipSuccess)
        e = rt::launch_replay(cfg, p->tau_atm[0], (long long)n, max_draws, d_xi.p, d_in.p, d_in.p + nn, d_in.p + 2 * nn,
                              d_in.p + 3 * nn, d_in.p + 4 * nn, ctx->d_sums.p, ctx->d_counts.p, ctx->d_mu_hist.p,
                              ctx->d_depth.p, d_tr.p, s);
    if (e == hipSuccess && out->sums) e = hipMemcpyAsync(out->sums, ctx->d_sums.p, sizeof(double) * RT_NSUM, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && out->counts)
        e = hipMemcpyAsync(out->counts, ctx->d_counts.p, sizeof(uint64_t) * RT_NCNT, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && out->mu_hist)
        e = hipMemcpyAsync(out->mu_hist, ctx->d_mu_hist.p, sizeof(uint64_t) * nmu, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && nd && out->depth_momentum)
        e = hipMemcpyAsync(out->depth_momentum, ctx->d_depth.p, sizeof(double) * nd, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && nd && out->depth_energy)
        e = hipMemcpyAsync(out->depth_energy, ctx->d_depth.p + nd, sizeof(double) * nd, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && per_photon && nn)
        e = hipMemcpyAsync(per_photon, d_tr.p, sizeof(rt_photon_trace) * nn, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "replay failed: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_build_grain_tables(rt_ctx* ctx, const rt_grain_set* gs, double density, double grain_min, double grain_max,
                          int32_t average_all_sizes, int32_t n_wl, const double* wl, const double* sed,
                          rt_grain_tables* out) {
    if (!ctx || !gs || !out || !wl) return fail(RT_ERR_INVALID, "NULL argument");
    if (gs->n_sizes < 1 || n_wl < 1 || !gs->a || !gs->offset || !gs->wl || !gs->q_abs || !gs->q_sca || !gs->g)
        return fail(RT_ERR_INVALID, "empty grain set");
    const int ns = gs->n_sizes;
    for (int i = 0; i < ns; ++i) {
        if (i && !(gs->a[i] > gs->a[i - 1])) return fail(RT_ERR_INVALID, "grain sizes must be strictly ascending");
        if (gs->offset[i + 1] - gs->offset[i] < 2) return fail(RT_ERR_INVALID, "grain size %d has fewer than 2 rows", i);
    }
    // sizes inside [grain_min, grain_max] (read_BPASS.py:115) form a contiguous range of the ascending list
    int lo_k = 0, hi_k = ns;
    while (lo_k < ns && gs->a[lo_k] < grain_min) ++lo_k;
    while (hi_k > lo_k && gs->a[hi_k - 1] > grain_max) --hi_k;
    if (hi_k <= lo_k) return fail(RT_ERR_INVALID, "no grain size inside [grain_min, grain_max]");
    const int lo_a = average_all_sizes ? 0 : lo_k, hi_a = average_all_sizes ? ns : hi_k;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t rows = (size_t)gs->offset[ns];
    std::vector<double> w15((size_t)ns), w35((size_t)ns);
    for (int i = 0; i < ns; ++i) {
        w15[i] = std::pow(gs->a[i], -1.5);
        w35[i] = std::pow(gs->a[i], -3.5);
    }
    DevBuf<double> d_cols, d_small, d_wl, d_out;
    DevBuf<int32_t> d_off;
    auto cleanup = [&]() {
        d_cols.release();
        d_small.release();
        d_wl.release();
        d_out.release();
        d_off.release();
    };
    hipError_t e = hipSuccess;
    hipStream_t s = ctx->stream;
    if ((e = d_cols.reserve(4 * rows)) != hipSuccess || (e = d_small.reserve(3 * (size_t)ns)) != hipSuccess ||
        (e = d_wl.reserve((size_t)n_wl)) != hipSuccess || (e = d_out.reserve(6 * (size_t)n_wl)) != hipSuccess ||
        (e = d_off.reserve((size_t)ns + 1)) != hipSuccess) {
        cleanup();
        return fail(RT_ERR_DEVICE, "device allocation failed: %s", hipGetErrorString(e));
    }
    const double* cols[4] = {gs->wl, gs->q_abs, gs->q_sca, gs->g};
    for (int c = 0; c < 4 && e == hipSuccess; ++c)
        e = hipMemcpyAsync(d_cols.p + c * rows, cols[c], sizeof(double) * rows, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_small.p, gs->a, sizeof(double) * ns, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_small.p + ns, w15.data(), sizeof(double) * ns, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_small.p + 2 * ns, w35.data(), sizeof(double) * ns, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off.p, gs->offset, sizeof(int32_t) * (ns + 1), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_wl.p, wl, sizeof(double) * (size_t)n_wl, hipMemcpyHostToDevice, s);
    rt::GrainSetDev dev;
    dev.a = d_small.p;
    dev.w15 = d_small.p + ns;
    dev.w35 = d_small.p + 2 * ns;
    dev.offset = d_off.p;
    dev.wl = d_cols.p;
    dev.q_abs = d_cols.p + rows;
    dev.q_sca = d_cols.p + 2 * rows;
    dev.g = d_cols.p + 3 * rows;
    dev.n_sizes = ns;
    if (e == hipSuccess) e = rt::launch_grain_integrals(dev, n_wl, d_wl.p, lo_k, hi_k, lo_a, hi_a, d_out.p, s);
    std::vector<double> host(6 * (size_t)n_wl);
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), d_out.p, sizeof(double) * host.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "table build failed: %s", hipGetErrorString(e));
    // size-only normalisations (np.trapz over the same size ranges) and the wavelength integrals, on the host
    const double pi = 3.141592653589793;
    auto trapz_sizes = [&](int lo, int hi, auto f) {
        if (hi - lo == 1) return f(lo);
        double acc = 0.0;
        for (int i = lo + 1; i < hi; ++i) acc += (gs->a[i] - gs->a[i - 1]) * (f(i) + f(i - 1)) / 2.0;
        return acc;
    };
    const double mass = trapz_sizes(lo_k, hi_k, [&](int i) { return 4 * pi / 3 * std::pow(gs->a[i], -0.5) * density; });
    const double norm = hi_a - lo_a == 1 ? 1.0 : trapz_sizes(lo_a, hi_a, [&](int i) { return w35[i]; });
    const size_t n = (size_t)n_wl;
    double lum = 0.0;
    for (size_t j = 0; j < n; ++j) lum += sed ? sed[j] : 1.0;
    double int_rp = 0.0, int_f = 0.0;
    for (size_t j = 1; j < n; ++j) {
        const double l0 = sed ? sed[j - 1] : 1.0, l1 = sed ? sed[j] : 1.0, d = wl[j] - wl[j - 1];
        int_rp += d * (host[j] * l1 + host[j - 1] * l0) / 2.0;
        int_f += d * (host[n + j] * l1 + host[n + j - 1] * l0) / 2.0;
    }
    out->kappa_av_rp = int_rp * 1e4 / mass / lum;
    out->kappa_av_f = int_f * 1e4 / mass / lum;
    out->lum = lum;
    const bool one_avg = hi_a - lo_a == 1;
    for (size_t j = 0; j < n; ++j) {
        if (out->kappa_rp) out->kappa_rp[j] = host[j] / mass;
        if (out->kappa_f) out->kappa_f[j] = host[n + j] / mass;
        // one size: pi a^2 Q and the tabulated g, Q_ext
        const double area = one_avg ? pi * (gs->a[lo_a] * gs->a[lo_a]) : 0.0;
        if (out->sigma_abs) out->sigma_abs[j] = one_avg ? area * host[2 * n + j] : host[2 * n + j] / norm;
        if (out->sigma_sca) out->sigma_sca[j] = one_avg ? area * host[3 * n + j] : host[3 * n + j] / norm;
        if (out->g_av) out->g_av[j] = host[4 * n + j] / norm;
        if (out->q_ext) out->q_ext[j] = host[5 * n + j] / norm;
    }
    return RT_OK;
}

// ---- shell velocity ODE ----
namespace {

// Host arrays of the model and the regions packed into one device buffer; fills device-pointer copies.
struct ShellUpload {
    DevBuf<double> buf;
    DevBuf<int32_t> ibuf;  // view_of_region, self_row
    std::vector<int32_t> ihost;
    std::vector<double> host;
    std::vector<size_t> off;
    size_t add(const double* p, size_t n) {
        const size_t o = host.size();
        if (p && n) host.insert(host.end(), p, p + n);
        off.push_back(o);
        return o;
    }
};

int check_shell_model(const rt_shell_model* m) {
    if (!m) return fail(RT_ERR_INVALID, "model is NULL");
    if (m->struct_size != sizeof(rt_shell_model))
        return fail(RT_ERR_INVALID, "rt_shell_model.struct_size is %u, but this library's rt_shell_model has %zu bytes (ABI %d)",
                    m->struct_size, sizeof(rt_shell_model), RT_ABI_VERSION);
    if (m->n_tau < 2 || m->n_age < 1 || !m->mc_tau || !m->mc_age || !m->mc_val)
        return fail(RT_ERR_INVALID, "the curve needs at least 2 optical depths and 1 age");
    for (int i = 1; i < m->n_tau; ++i)
        if (!(m->mc_tau[i] > m->mc_tau[i - 1])) return fail(RT_ERR_INVALID, "mc_tau must be strictly ascending");
    for (int i = 1; i < m->n_age; ++i)
        if (!(m->mc_age[i] > m->mc_age[i - 1])) return fail(RT_ERR_INVALID, "mc_age must be strictly ascending");
    if (m->aging && (m->n_lage < 2 || !m->l_age || !m->l_val || !(m->l_norm > 0.0)))
        return fail(RT_ERR_INVALID, "aging needs the luminosity per age (l_age, l_val, l_norm)");
    if (m->tau_cutoff > 0.0 && (m->n_lage < 1 || !m->l_age || !m->ratio_val))
        return fail(RT_ERR_INVALID, "tau_cutoff > 0 needs ratio_val per age");
    if (m->aging && m->n_age < 2) return fail(RT_ERR_INVALID, "aging needs a curve with at least 2 ages");
    return RT_OK;
}

// The rows of gal_data and the views onto them (model->galaxy); the per-region arrays must name existing views / rows.
int check_shell_galaxy(const rt_shell_galaxy* g, int n_regions) {
    if (!g) return RT_OK;
    if (g->n_gal < 1 || g->n_views < 1) return fail(RT_ERR_INVALID, "galaxy: n_gal and n_views must be >= 1");
    if ((long long)g->n_gal * g->n_views > (1ll << 28)) return fail(RT_ERR_LIMIT, "galaxy: n_views * n_gal beyond 2^28");
    if (!g->lha_ratio_f || !g->dist_to_ref || !g->mass_tot) return fail(RT_ERR_INVALID, "galaxy: a column is NULL");
    if (!(g->cm_pc > 0.0)) return fail(RT_ERR_INVALID, "galaxy: cm_pc must be positive");
    if (n_regions > 0 && (!g->view_of_region || !g->self_row)) return fail(RT_ERR_INVALID, "galaxy: view_of_region / self_row is NULL");
    for (int i = 0; i < n_regions; ++i) {
        if (g->view_of_region[i] < 0 || g->view_of_region[i] >= g->n_views)
            return fail(RT_ERR_INVALID, "galaxy: region %d names view %d of %d", i, g->view_of_region[i], g->n_views);
        if (g->self_row[i] < -1 || g->self_row[i] >= g->n_gal)
            return fail(RT_ERR_INVALID, "galaxy: region %d names row %d of %d", i, g->self_row[i], g->n_gal);
    }
    const size_t cells = (size_t)g->n_gal * (size_t)g->n_views;
    for (size_t k = 0; k < cells; ++k)
        if (!(g->dist_to_ref[k] >= 0.0)) return fail(RT_ERR_INVALID, "galaxy: dist_to_ref[%zu] is negative or not a number", k);
    return RT_OK;
}

int check_shell_regions(const rt_shell_regions* rg) {
    if (!rg) return fail(RT_ERR_INVALID, "regions is NULL");
    if (rg->n < 0) return fail(RT_ERR_INVALID, "negative region count");
    if (rg->n > 0 && (!rg->m_g || !rg->m_new || !rg->m_old || !rg->m_co_hi || !rg->h_old || !rg->h_g_i || !rg->tau_i ||
                      !rg->l_bol || !rg->sigma_g || !rg->r_end))
        return fail(RT_ERR_INVALID, "a region array is NULL");
    for (int i = 0; i < rg->n; ++i) {
        if (!(rg->h_g_i[i] > 0.0) || !(rg->m_g[i] > 0.0)) return fail(RT_ERR_INVALID, "region %d: h_g_i and m_g must be positive", i);
        if (!(rg->r_end[i] >= rg->h_g_i[i])) return fail(RT_ERR_INVALID, "region %d: r_end below the starting radius", i);
    }
    return RT_OK;
}

// Packs model + regions (+ the galaxy's rows), uploads, and returns device-pointer views; *dgal_out is &dg or nullptr.
hipError_t upload_shell(rt_ctx* ctx, const rt_shell_model* m, const rt_shell_regions* rg, ShellUpload& up,
                        rt::ShellModel& dm, rt::ShellRegions& dr, rt::ShellGalaxy& dg, const rt::ShellGalaxy** dgal_out) {
    const size_t n = (size_t)rg->n;
    const rt_shell_galaxy* gal = m->galaxy;
    const size_t cells = gal ? (size_t)gal->n_gal * (size_t)gal->n_views : 0;
    const size_t o_lha = up.add(gal ? gal->lha_ratio_f : nullptr, cells), o_dist = up.add(gal ? gal->dist_to_ref : nullptr, cells);
    const size_t o_mass = up.add(gal ? gal->mass_tot : nullptr, gal ? (size_t)gal->n_gal : 0);
    const size_t o_tau = up.add(m->mc_tau, (size_t)m->n_tau), o_age = up.add(m->mc_age, (size_t)m->n_age);
    const size_t o_val = up.add(m->mc_val, (size_t)m->n_tau * (size_t)m->n_age);
    const size_t nl = (m->aging || m->tau_cutoff > 0.0) ? (size_t)m->n_lage : 0;
    const size_t o_la = up.add(m->l_age, nl), o_lv = up.add(m->aging ? m->l_val : nullptr, m->aging ? nl : 0);
    const size_t o_rv = up.add(m->tau_cutoff > 0.0 ? m->ratio_val : nullptr, m->tau_cutoff > 0.0 ? nl : 0);
    const double* cols[10] = {rg->m_g, rg->m_new, rg->m_old, rg->m_co_hi, rg->h_old, rg->h_g_i, rg->tau_i, rg->l_bol,
                              rg->sigma_g, rg->r_end};
    size_t o_col[10];
    for (int k = 0; k < 10; ++k) o_col[k] = up.add(cols[k], n);
    if (up.host.empty()) up.host.push_back(0.0);
    hipError_t e = up.buf.reserve(up.host.size());
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(up.buf.p, up.host.data(), sizeof(double) * up.host.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return e;
    *dgal_out = nullptr;
    if (gal) {
        up.ihost.assign(gal->view_of_region, gal->view_of_region + n);
        up.ihost.insert(up.ihost.end(), gal->self_row, gal->self_row + n);
        e = up.ibuf.reserve(2 * n);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(up.ibuf.p, up.ihost.data(), sizeof(int32_t) * 2 * n, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return e;
        dg = *gal;
        dg.lha_ratio_f = up.buf.p + o_lha;
        dg.dist_to_ref = up.buf.p + o_dist;
        dg.mass_tot = up.buf.p + o_mass;
        dg.view_of_region = up.ibuf.p;
        dg.self_row = up.ibuf.p + n;
        *dgal_out = &dg;
    }
    dm = *m;
    dm.galaxy = nullptr;  // a host address: the kernels get the galaxy as an argument of its own
    dm.mc_tau = up.buf.p + o_tau;
    dm.mc_age = up.buf.p + o_age;
    dm.mc_val = up.buf.p + o_val;
    dm.l_age = up.buf.p + o_la;
    dm.l_val = up.buf.p + o_lv;
    dm.ratio_val = up.buf.p + o_rv;
    dr = *rg;
    const double** dcols[10] = {&dr.m_g, &dr.m_new, &dr.m_old, &dr.m_co_hi, &dr.h_old, &dr.h_g_i, &dr.tau_i, &dr.l_bol,
                                &dr.sigma_g, &dr.r_end};
    for (int k = 0; k < 10; ++k) *dcols[k] = up.buf.p + o_col[k];
    return hipSuccess;
}

}  // namespace

int rt_shell_rhs(rt_ctx* ctx, const rt_shell_model* model, const rt_shell_regions* regions, int32_t n,
                 const int32_t* region, const double* r, const double* v, const double* t, double* dvdr,
                 double* dtdr) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    int rc = check_shell_model(model);
    if (rc == RT_OK) rc = check_shell_regions(regions);
    if (rc == RT_OK) rc = check_shell_galaxy(model->galaxy, regions->n);
    if (rc != RT_OK) return rc;
    if (n < 0 || (n > 0 && (!region || !r || !v || !t || !dvdr || !dtdr))) return fail(RT_ERR_INVALID, "bad argument");
    if (n == 0) return RT_OK;
    for (int i = 0; i < n; ++i)
        if (region[i] < 0 || region[i] >= regions->n) return fail(RT_ERR_INVALID, "point %d names region %d of %d", i, region[i], regions->n);
    RT_HIP(hipSetDevice(ctx->device));
    ShellUpload up;
    rt::ShellModel dm;
    rt::ShellRegions dr;
    rt::ShellGalaxy dg;
    const rt::ShellGalaxy* dgal = nullptr;
    DevBuf<double> pts;
    DevBuf<int32_t> idx;
    hipError_t e = upload_shell(ctx, model, regions, up, dm, dr, dg, &dgal);
    if (e == hipSuccess) e = pts.reserve(5 * (size_t)n);
    if (e == hipSuccess) e = idx.reserve((size_t)n);
    const double* in[3] = {r, v, t};
    for (int k = 0; k < 3 && e == hipSuccess; ++k)
        e = hipMemcpyAsync(pts.p + (size_t)k * n, in[k], sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(idx.p, region, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = rt::launch_shell_rhs(dm, dr, dgal, n, idx.p, pts.p, pts.p + n, pts.p + 2 * (size_t)n, pts.p + 3 * (size_t)n,
                                 pts.p + 4 * (size_t)n, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dvdr, pts.p + 3 * (size_t)n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dtdr, pts.p + 4 * (size_t)n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    up.buf.release();
    up.ibuf.release();
    pts.release();
    idx.release();
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "shell right-hand side failed: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_shell_velocity(rt_ctx* ctx, const rt_shell_model* model, const rt_shell_regions* regions,
                      const rt_shell_control* control, rt_shell_result* out) {
    if (!ctx || !control || !out) return fail(RT_ERR_INVALID, "NULL argument");
    int rc = check_shell_model(model);
    if (rc == RT_OK) rc = check_shell_regions(regions);
    if (rc == RT_OK) rc = check_shell_galaxy(model->galaxy, regions->n);
    if (rc != RT_OK) return rc;
    if (!(control->v0 > 0.0) || !(control->t0 > 0.0) || !(control->rtol > 0.0) || !(control->max_step > 0.0) ||
        control->atol_v < 0.0 || control->atol_t < 0.0 || control->n_out < 0 || control->max_steps < 1)
        return fail(RT_ERR_INVALID, "bad integration control");
    const size_t n = (size_t)regions->n, no = (size_t)control->n_out;
    if (n == 0) return RT_OK;
    if (!out->v_max || !out->r_vmax || !out->t_vmax || !out->r_last || !out->v_last || !out->t_last || !out->status ||
        !out->n_steps || (no > 0 && (!out->v_out || !out->t_out)))
        return fail(RT_ERR_INVALID, "a result array is NULL");
    RT_HIP(hipSetDevice(ctx->device));
    ShellUpload up;
    rt::ShellModel dm;
    rt::ShellRegions dr;
    rt::ShellGalaxy dg;
    const rt::ShellGalaxy* dgal = nullptr;
    DevBuf<double> res;
    DevBuf<int32_t> ires;
    hipError_t e = upload_shell(ctx, model, regions, up, dm, dr, dg, &dgal);
    if (e == hipSuccess) e = res.reserve(6 * n + 2 * n * no + 1);
    if (e == hipSuccess) e = ires.reserve(2 * n);
    rt::ShellResult dout;
    dout.v_max = res.p;
    dout.r_vmax = res.p + n;
    dout.t_vmax = res.p + 2 * n;
    dout.r_last = res.p + 3 * n;
    dout.v_last = res.p + 4 * n;
    dout.t_last = res.p + 5 * n;
    dout.v_out = res.p + 6 * n;
    dout.t_out = res.p + 6 * n + n * no;
    dout.status = ires.p;
    dout.n_steps = ires.p + n;
    if (e == hipSuccess) e = rt::launch_shell_ode(dm, dr, dgal, *control, dout, ctx->stream);
    double* hd[6] = {out->v_max, out->r_vmax, out->t_vmax, out->r_last, out->v_last, out->t_last};
    for (int k = 0; k < 6 && e == hipSuccess; ++k)
        e = hipMemcpyAsync(hd[k], res.p + (size_t)k * n, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && no) e = hipMemcpyAsync(out->v_out, dout.v_out, sizeof(double) * n * no, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && no) e = hipMemcpyAsync(out->t_out, dout.t_out, sizeof(double) * n * no, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out->status, ires.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out->n_steps, ires.p + n, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    up.buf.release();
    up.ibuf.release();
    res.release();
    ires.release();
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "shell integration failed: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_selftest_math(rt_ctx* ctx, int64_t n, const double* u, double* out_log, double* out_cos, double* out_sqrt,
                     double* out_div) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    if (n < 0 || (n > 0 && (!u || !out_log || !out_cos || !out_sqrt || !out_div))) return fail(RT_ERR_INVALID, "bad argument");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    DevBuf<double> d;
    hipError_t e = d.reserve(5 * (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(d.p, u, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt::launch_selftest_math(ctx->d_log_table.p, (long long)n, d.p, d.p + n, d.p + 2 * n, d.p + 3 * n, d.p + 4 * n,
                                                      ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_log, d.p + n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_cos, d.p + 2 * n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_sqrt, d.p + 3 * n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_div, d.p + 4 * n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d.release();
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "selftest failed: %s", hipGetErrorString(e));
    return RT_OK;
}

// ---- the collective: RCCL, opened at run time ----
namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

// librccl.so.1 is opened on first use, so that a process that never uses the collective does not load RCCL, and a
// process that already has an RCCL of that name loaded (PyTorch ships one) shares it instead of loading a second.
struct RcclLoad {
    Rccl lib;
    std::string error;  // why the library is not usable (kept: dlerror() reports a failure only once)
    RcclLoad() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib.handle) break;
            const char* msg = dlerror();
            error = msg ? msg : "dlopen failed";
        }
        if (!lib.handle) return;
        lib.get_unique_id = reinterpret_cast<decltype(lib.get_unique_id)>(dlsym(lib.handle, "ncclGetUniqueId"));
        lib.comm_init_rank = reinterpret_cast<decltype(lib.comm_init_rank)>(dlsym(lib.handle, "ncclCommInitRank"));
        lib.comm_destroy = reinterpret_cast<decltype(lib.comm_destroy)>(dlsym(lib.handle, "ncclCommDestroy"));
        lib.all_reduce = reinterpret_cast<decltype(lib.all_reduce)>(dlsym(lib.handle, "ncclAllReduce"));
        lib.error_string = reinterpret_cast<decltype(lib.error_string)>(dlsym(lib.handle, "ncclGetErrorString"));
        if (!lib.get_unique_id || !lib.comm_init_rank || !lib.comm_destroy || !lib.all_reduce || !lib.error_string) {
            error = "the RCCL library that was opened lacks an entry point (ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / "
                    "ncclAllReduce / ncclGetErrorString)";
            dlclose(lib.handle);
            lib.handle = nullptr;
        }
    }
};

// Opened once, by whichever thread asks first (a function-local static: initialised exactly once, thread-safe).
const RcclLoad* rccl_load() {
    static const RcclLoad load;
    return &load;
}

const Rccl* rccl() { return rccl_load()->lib.handle ? &rccl_load()->lib : nullptr; }

#define RT_RCCL(lib, call)                                                                          \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) return fail(RT_ERR_DEVICE, "%s failed: %s", #call, (lib)->error_string(r_)); \
    } while (0)

}  // namespace

static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "rt_comm_unique_id hands out an ncclUniqueId");

int rt_comm_unique_id(void* id) {
    if (!id) return fail(RT_ERR_INVALID, "id is NULL");
    const Rccl* lib = rccl();
    if (!lib) return fail(RT_ERR_DEVICE, "RCCL is not usable: %s", rccl_load()->error.c_str());
    ncclUniqueId u;
    RT_RCCL(lib, lib->get_unique_id(&u));
    memcpy(id, &u, sizeof u);
    return RT_OK;
}

int rt_comm_init(rt_ctx* ctx, int rank, int world, const void* id) {
    if (!ctx || !id) return fail(RT_ERR_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "rank %d not in [0,%d)", rank, world);
    if (ctx->comm) return fail(RT_ERR_STATE, "the context already has a communicator");
    const Rccl* lib = rccl();
    if (!lib) return fail(RT_ERR_DEVICE, "RCCL is not usable: %s", rccl_load()->error.c_str());
    RT_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    RT_RCCL(lib, lib->comm_init_rank(&ctx->comm, world, u, rank));
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return RT_OK;
}

int rt_comm_destroy(rt_ctx* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    if (!ctx->comm) return RT_OK;
    const Rccl* lib = rccl();
    (void)hipSetDevice(ctx->device);
    const ncclResult_t r = lib ? lib->comm_destroy(ctx->comm) : ncclSuccess;
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    if (r != ncclSuccess) return fail(RT_ERR_DEVICE, "ncclCommDestroy failed: %s", lib->error_string(r));
    return RT_OK;
}

int rt_run_allreduce(rt_ctx* ctx, const rt_params* p, void* d_tallies, void* stream_v) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    if (!ctx->comm) return fail(RT_ERR_STATE, "rt_comm_init must be called before rt_run_allreduce");
    if (!d_tallies) return fail(RT_ERR_INVALID, "d_tallies is NULL");
    int rc = check_common(p);  // before p's sizes are used
    if (rc) return rc;
    const int n_slabs = p->n_slabs > 0 ? p->n_slabs : 8;
    const size_t na = (size_t)p->n_atm, nmu = (size_t)p->n_mu_bins + 1, nd = (size_t)p->n_depth_bins;
    const size_t n_slab = na * n_slabs * RT_NSUM, n_cnt = na * RT_NCNT, n_hist = na * nmu, n_depth = na * n_slabs * 2 * nd;
    unsigned long long* words = static_cast<unsigned long long*>(d_tallies);
    rt_params run = *p;  // the communicator says which slabs this process walks
    run.rank = ctx->comm_rank;
    run.world = ctx->comm_world;
    // A failure here is this process's alone (a refused argument, an allocation, a launch): the peers are then already
    // waiting in ncclAllReduce, so the caller must tear the job down (see include/rt_amd.h).
    rc = enqueue_sweep(ctx, &run, words, words + n_slab, words + n_slab + n_cnt, nd ? words + n_slab + n_cnt + n_hist : nullptr,
                       stream_v, false);
    if (rc) return rc;
    const Rccl* lib = rccl();
    hipStream_t stream = (hipStream_t)stream_v;
    // One collective: every slab word (sums and depth bins) is non-zero on exactly one rank and all-zero bits on the
    // others, so the integer sum of the bit patterns returns that rank's double unchanged; counts and histogram add
    // as integers.
    RT_RCCL(lib, lib->all_reduce(words, words, n_slab + n_cnt + n_hist + n_depth, ncclInt64, ncclSum, ctx->comm, stream));
    return RT_OK;
}

// The wavelength index (into the table's wl[]) and the starting cosine the sweeps give photon ids
// [first_id, first_id + n) of atmosphere `atm` under p's seed, id base, stream and tables.
int rt_emission_indices(rt_ctx* ctx, const rt_params* p, int32_t atm, uint64_t first_id, int64_t n, int32_t* out_index,
                        double* out_mu0) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    int rc = check_common(p);
    if (rc) return rc;
    if (ctx->n_wl < 1) return fail(RT_ERR_STATE, "rt_set_tables must be called before rt_emission_indices");
    if (atm < 0 || atm >= p->n_atm) return fail(RT_ERR_INVALID, "atmosphere %d not in [0,%d)", atm, p->n_atm);
    if (n < 0 || (n > 0 && !out_index)) return fail(RT_ERR_INVALID, "bad output array");
    if (n > (1ll << 31)) return fail(RT_ERR_LIMIT, "at most 2^31 photon ids per call");
    if (n == 0) return RT_OK;
    const int table = p->table_of_atm ? p->table_of_atm[atm] : 0;
    if (table < 0 || table >= ctx->n_tables) return fail(RT_ERR_INVALID, "atmosphere %d asks for table %d of %d", atm, table, ctx->n_tables);
    RT_HIP(hipSetDevice(ctx->device));
    rt::Cfg cfg = make_cfg(ctx, p);
    cfg.table_base = table * ctx->n_wl;
    if (p->monochromatic_wl != 0.0) {
        rc = find_monochromatic(ctx, p->monochromatic_wl, &cfg.mono_index);
        if (rc) return rc;
    }
    DevBuf<int32_t> d_idx;
    DevBuf<double> d_mu;
    hipError_t e = d_idx.reserve((size_t)n);
    if (e == hipSuccess && out_mu0) e = d_mu.reserve((size_t)n);
    if (e == hipSuccess) e = rt::launch_emission(cfg, p->rng_mode, (uint32_t)atm, first_id, (long long)n, d_idx.p, out_mu0 ? d_mu.p : nullptr, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_index, d_idx.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && out_mu0) e = hipMemcpyAsync(out_mu0, d_mu.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d_idx.release();
    d_mu.release();
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "emission kernel failed: %s", hipGetErrorString(e));
    return RT_OK;
}

const char* rt_last_kernel(rt_ctx* ctx) {
    if (!ctx || ctx->last_variant < 0) return "";
    return rt::walk_variant(ctx->last_variant).name;
}

int rt_kernel_time(rt_ctx* ctx, double* avg_ms, int64_t* launches, int reset) {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    (void)hipSetDevice(ctx->device);
    drain_events(ctx);
    if (avg_ms) *avg_ms = ctx->launches ? ctx->ms_total / (double)ctx->launches : 0.0;
    if (launches) *launches = ctx->launches;
    if (reset) {
        ctx->ms_total = 0.0;
        ctx->launches = 0;
    }
    return RT_OK;
}

}  // extern "C"
