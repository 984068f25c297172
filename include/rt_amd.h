/*
 * rt_amd.h -- C ABI of the MI355X (gfx950) Monte Carlo dust photon random walk.
 *
 * This is the drop-in boundary for ONE hot path of ianblackstone/Radiative-Transfer:
 * the photon random walk `RunPhotons` and the atmosphere sweep that calls it.
 * The reference has no FFI layer of its own; its boundary is two Python call
 * signatures, and every entry point below names the reference interface it
 * replaces (file:line into the reference checkout).  The thin Python host in
 * radiative-transfer_amd/ binds these with ctypes and re-creates the Python
 * signatures on top.
 *
 * Conventions
 *   - plain C types only; all array arguments are caller-owned HOST buffers
 *     unless a name starts with d_ (device pointer of the context's GPU);
 *   - every function returns 0 on success or a negative rt_status; the text of
 *     the last failure on the calling thread is returned by rt_last_error();
 *   - a context is bound to one GPU and is not thread-safe; use one context
 *     per host thread / per process rank;
 *   - there is no CPU fallback: if no gfx950 device is usable rt_create fails.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 8

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1, /* bad argument */
    RT_ERR_DEVICE = -2,  /* HIP runtime / no device */
    RT_ERR_STATE = -3,   /* call order (e.g. run before tables) */
    RT_ERR_LIMIT = -4    /* size beyond a documented maximum */
} rt_status;

/* boundary condition of the lower face, MC code.py:522-574 */
enum { RT_BOUNDARY_REFLECT = 0, RT_BOUNDARY_ABSORB = 1, RT_BOUNDARY_REEMIT = 2 };
/* scattering law, MC code.py:732-760 and Numpy_random_walk.py:369-396 */
enum { RT_SCATTER_HG = 0, RT_SCATTER_ISO = 1, RT_SCATTER_DRAINE = 2, RT_SCATTER_DRAINE_EXACT = 3 };
/* RT_SCATTER_DRAINE restates the reference's rejection sampler (Numpy_random_walk.py:106-130) photon by photon --
 * a deliberate difference in pairing: the reference draws candidates for all scattering photons in batches and hands
 * the accepted cosines to the photons in acceptance order (and may return more cosines than photons), here every
 * photon runs the same accept / reject test on pairs (y, candidate) of its own stream until one is accepted.  Each
 * accepted cosine is an independent draw from the same law either way; that law is pinned to the reference's own
 * output (histograms of 1e6 cosines, tests/test_draine_reference_law.py).  It keeps the reference's defect: candidates are accepted where pdf(mu) > y with y uniform on [0,1), so the density is
 * clipped wherever pdf > 1 and the sampler is slow where pdf << 1.  Gray loop order only, as in the reference.
 * RT_SCATTER_DRAINE_EXACT (no counterpart in the reference, which marks its sampler "not currently functioning",
 * MC code.py:1006) draws from the same phase function
 *     p(mu) ~ (1 + alpha mu^2) / (1 + g^2 - 2 g mu)^(3/2),      0 <= alpha <= 1,
 * exactly: a Henyey-Greenstein candidate by the reference's own inverse CDF (MC code.py:439), accepted with
 * probability (1 + alpha mu^2) / (1 + alpha); one uniform pair (y, xi) of the photon's stream per try, at most
 * 1 + alpha tries on average.  Any loop order; g per photon (per wavelength) as for RT_SCATTER_HG.
 * alpha = 0 accepts every candidate and gives Henyey-Greenstein; g = 0, alpha = 1 is Rayleigh scattering. */
/* geometry of the dust layer.
 *   RT_GEOM_SLAB   the reference's plane-parallel slab: z is the optical depth above the emitting face and mu
 *                  the cosine to the slab normal (TakeStep, MC code.py:493-508).
 *   RT_GEOM_SHELL  (no counterpart in the reference) a spherical shell of uniform density between the radii
 *                  r_in and r_in + thickness around the source: z = r - r_in in optical-depth units of the
 *                  photon's wavelength, 0 <= z <= tau*Scale, and mu the cosine to the local radial direction.
 *                  r_in = shell_inner * tau*Scale (shell_inner = inner radius / thickness > 0).  A flight of
 *                  optical length s moves the photon along a straight line:
 *                      r' = sqrt(r^2 + s^2 + 2 r s mu),   mu' = (r mu + s) / r'.
 *                  The cavity r < r_in is empty: a photon that reaches it crosses it and re-enters the shell
 *                  on the far side (RT_BOUNDARY_REFLECT, counted in RT_CNT_BOUNDARY_HITS, no momentum
 *                  exchanged), is removed there (RT_BOUNDARY_ABSORB) or is replaced by a new photon
 *                  (RT_BOUNDARY_REEMIT).  One escape test per flight, r' > r_in + thickness, in either loop
 *                  order; every escaping photon is tallied (RT_SUM_ESCAPED = radial momentum at the outer
 *                  face) and enters mu_hist.  Momentum is not conserved along a straight line in radial
 *                  coordinates, so the deposited radial momentum is RT_SUM_DUST_MOMENTUM (summed at the
 *                  collisions) and t_i in RT_SUM_TRANSFER_* is photon i's share of it.  depth bins are radial
 *                  bins.  shell_inner -> infinity recovers the slab with a reflecting lower face.
 *                  Walk only (rt_run / rt_run_device), not rt_replay. */
enum { RT_GEOM_SLAB = 0, RT_GEOM_SHELL = 1 };
/* order of the tests inside one loop iteration:
 *   RT_ORDER_MC    step, escape(tallied), boundary, escape(logged), absorb, scatter  (MC code.py:657-767)
 *   RT_ORDER_GRAY  step, boundary, escape(tallied+logged), absorb, scatter           (Numpy_random_walk.py:313-396)
 * RT_ORDER_GRAY also weights momentum by lambda instead of 1/lambda (Numpy_random_walk.py:55). */
enum { RT_ORDER_MC = 0, RT_ORDER_GRAY = 1 };

/* per-photon fates reported by rt_replay */
enum {
    RT_FATE_ESCAPED = 1,           /* left through the top, momentum tallied */
    RT_FATE_ESCAPED_UNCOUNTED = 2, /* removed by the second escape test of RT_ORDER_MC (MC code.py:690) */
    RT_FATE_ABSORBED = 3,          /* absorbed by dust (MC code.py:479) */
    RT_FATE_BOUNDARY = 4,          /* absorbed by the lower boundary (MC code.py:551-556) */
    RT_FATE_GUARD = 5              /* max_events_per_photon reached; the reference would loop forever */
};

/* Floating-point tallies per atmosphere: sums[atm * RT_NSUM + k].
 * p = mu * w * h is a photon's z momentum, w = 1/lambda (RT_ORDER_MC) or lambda (RT_ORDER_GRAY),
 * as in GetZMomentum (MC code.py:360-369). */
enum {
    RT_SUM_INITIAL = 0,           /* initial_momentum,  MC code.py:642 */
    RT_SUM_ESCAPED = 1,           /* escaped_momentum,  MC code.py:670 (first test only) */
    RT_SUM_BOUNDARY = 2,          /* boundary_momentum, MC code.py:541,553,561 (reference signs) */
    RT_SUM_ESCAPED_UNCOUNTED = 3, /* momentum the reference drops at MC code.py:690 */
    RT_SUM_TRANSFER_SQ = 4,       /* sum_i t_i^2,  t_i = p0_i - escaped_i + boundary_i of photon i */
    RT_SUM_TRANSFER_P0 = 5,       /* sum_i t_i * p0_i */
    RT_SUM_P0_SQ = 6,             /* sum_i p0_i^2 */
    RT_SUM_ABSORBED_ENERGY = 7,   /* sum of w*h over dust-absorbed photons */
    RT_SUM_DUST_MOMENTUM = 8,     /* sum over collisions of (p_before - p_after); the reference's
                                     commented-out second estimator, MC code.py:648-650,728-729,773-776 */
    RT_SUM_REEMITTED = 9,         /* momentum of photons re-emitted at the boundary (RT_BOUNDARY_REEMIT) */
    RT_NSUM = 10
};

/* Integer tallies per atmosphere: counts[atm * RT_NCNT + k]. */
enum {
    RT_CNT_SCATTER = 0,           /* scatter_count, MC code.py:767 -- the throughput metric's event */
    RT_CNT_ESCAPED = 1,
    RT_CNT_ESCAPED_UNCOUNTED = 2,
    RT_CNT_ABSORBED = 3,
    RT_CNT_BOUNDARY_ABSORBED = 4,
    RT_CNT_REEMITTED = 5,
    RT_CNT_GUARD = 6,
    RT_CNT_PHOTONS = 7,           /* photon histories started */
    RT_CNT_FLIGHTS = 8,           /* loop iterations summed over photons (one TakeStep each) */
    RT_CNT_BOUNDARY_HITS = 9,     /* z <= 0 events */
    RT_NCNT = 10
};

/* One sweep over n_atm atmospheres (the loop of runMC, MC code.py:1056-1062). */
/* How uniforms are cut from the Philox4x32-10 stream of a photon (counter = id, block, atmosphere).
 * RT_RNG_PHILOX52 (default): every uniform has 52 random bits (two words); loop iteration k uses block 2k
 *   for (step, absorb) and block 2k+1 for (azimuth, cosine); emission draws come from blocks >= 2^31.
 * RT_RNG_PHILOX32X4 (opt-in, faster): ONE block per loop iteration, one 32-bit word per uniform,
 *   u = (w + 1/2) / 2^32: block k gives (azimuth, cosine) of the scattering BEFORE flight k in words 0-1 and
 *   (step, absorb) of flight k in words 2-3; block 0's words 0-1 (no scattering precedes the first flight)
 *   are the 52-bit emission draw.  Steps are then quantised to 2^-32 in xi (longest step 22.9), angles to
 *   2^-32 of a turn.
 * RT_RNG_PHILOX40 (opt-in, as fast): the same one block per iteration with its 128 bits spent where the resolution
 *   matters: azimuth 24 bits (top of word 0), cosine 40 bits (low 8 bits of word 0, word 1), step 40 bits (word 2,
 *   top 8 bits of word 3), absorption 24 bits (low 24 bits of word 3), each uniform at the midpoint of its cell,
 *   u = (k + 1/2) / 2^bits.  Steps are quantised to 2^-40 in xi (longest step 27.7 mean free paths, reached once in
 *   1e12 flights), albedos effectively to 2^-24.  Emission as for RT_RNG_PHILOX32X4.
 * Neither one-block stream is available with RT_SCATTER_DRAINE / RT_SCATTER_DRAINE_EXACT (their rejection loops need
 *   more than one block per iteration). */
enum rt_rng_mode { RT_RNG_PHILOX52 = 0, RT_RNG_PHILOX32X4 = 1, RT_RNG_PHILOX40 = 2 };

/* rt_params.flags */
enum {
    RT_FLAG_RUNTIME_KERNEL = 1u /* walk with the kernel that reads every option at run time even where an instantiation
                                   with the sweep's options fixed at compile time exists (both give the same bits;
                                   for testing one against the other) */
};

typedef struct rt_params {
    uint32_t struct_size;          /* = sizeof(rt_params) of the header the caller was built against.  Every entry point
                                      that takes an rt_params refuses any other value with RT_ERR_INVALID before it reads
                                      another member: a binding typed from an older header (a shorter struct) cannot make
                                      the library read past the caller's object. */
    int32_t n_atm;
    int32_t boundary;              /* RT_BOUNDARY_* ; reference flag `boundary`, MC code.py:78 */
    int32_t scatter;               /* RT_SCATTER_*  ; reference flag `scatter`,  MC code.py:81 */
    int32_t loop_order;            /* RT_ORDER_* */
    int32_t rand_mu;               /* reference global `rand_mu`,  MC code.py:84, :384-385 */
    int32_t abs_only;              /* reference global `abs_only`, MC code.py:85, :478-481 */
    int32_t n_mu_bins;             /* escape-cosine histogram bins (0..1), plus one overflow bin for mu >= 1; <= 4096 */
    int32_t n_depth_bins;          /* depth-resolved deposition bins (extension; 0 = off); <= 1024, and together
                                      16 * (n_mu_bins + 1) + 64 * n_depth_bins <= 63360 (both histograms are kept per wave
                                      in 64 KiB of LDS); RT_ERR_LIMIT beyond that, before anything is enqueued */
    int32_t max_events_per_photon; /* iteration guard per photon; <= 0 selects 1<<20; capped at 1<<30 */
    int32_t n_slabs;               /* fixed partition of every atmosphere's photon-id range (default 8) */
    int32_t rank;                  /* this context walks slabs s with s % world == rank */
    int32_t world;
    uint32_t flags;                /* RT_FLAG_* bits, 0 by default */
    uint64_t seed;                 /* Philox key */
    uint64_t photon_id_base;       /* added to every photon id (independent repeat runs) */
    double monochromatic_wl;       /* reference global `monochromatic`, MC code.py:88, :387-391; 0 = sample the table */
    double alpha;                  /* Draine phase-function parameter, Numpy_random_walk.py:16 */
    double h;                      /* Planck constant used in GetZMomentum (constants.py:10; 1 in the gray script) */
    const double* tau_atm;         /* [n_atm] reference optical depth of each atmosphere (tau_list_calc) */
    const uint64_t* n_photons;     /* [n_atm] photons per atmosphere (photon_reduction already applied) */
    int32_t rng_mode;              /* RT_RNG_* */
    int32_t geometry;              /* RT_GEOM_* */
    const int32_t* table_of_atm;   /* [n_atm] which table of rt_set_tables_multi an atmosphere uses; NULL = table 0 */
    double shell_inner;            /* RT_GEOM_SHELL: inner radius in units of the shell's thickness (> 0) */
    double* escaped_mu;            /* opt-in, NULL = off: one double per photon, atmosphere after atmosphere in the
                                      caller's order (atmosphere a starts at sum of n_photons[0..a-1]); photon i of an
                                      atmosphere gets the cosine it left with if the reference would have put it into
                                      RunPhotons' escaped_mu list (MC code.py:690-693, :779: the second escape test;
                                      Numpy_random_walk.py:336-343: every escape), NaN otherwise.  A HOST array for
                                      rt_run, a DEVICE array for rt_run_device / rt_run_device_static; only the
                                      entries of this rank's slabs are written.  The sweep then runs the kernel that
                                      reads its flags at run time.  Walk only, not rt_replay (whose per-photon trace
                                      has the same information). */
} rt_params;

/* Host-side result arrays of one sweep; any pointer may be NULL to skip it. */
typedef struct rt_tallies {
    double* sums;          /* [n_atm][RT_NSUM] */
    uint64_t* counts;      /* [n_atm][RT_NCNT] */
    uint64_t* mu_hist;     /* [n_atm][n_mu_bins + 1] escape cosines: second-test population for
                              RT_ORDER_MC (MC code.py:693), the single test for RT_ORDER_GRAY */
    double* depth_momentum; /* [n_atm][n_depth_bins] dust momentum deposition by depth z / (tau*Scale) */
    double* depth_energy;   /* [n_atm][n_depth_bins] absorbed w*h by depth.  Both are added per chunk of photon ids by
                               the wave that walks it, then per slab in chunk order, then over the slabs in index
                               order: bit for bit reproducible and independent of the number of GPUs, like `sums` */
    double* slab_sums;     /* [n_atm][n_slabs][RT_NSUM] per-slab partial sums (zero for slabs of other ranks);
                              summing the slab axis in index order gives `sums` bit for bit on any GPU count */
    double* slab_depth;    /* [n_atm][n_slabs][2][n_depth_bins] the same for depth_momentum ([..][0][..]) and
                              depth_energy ([..][1][..]) */
} rt_tallies;

/* Per-photon record written by rt_replay. */
typedef struct rt_photon_trace {
    double final_z;
    double final_mu;
    int32_t fate;      /* RT_FATE_* */
    int32_t n_scatter; /* scatterings of this photon */
    int32_t n_draws;   /* uniforms consumed from its stream */
    int32_t pad_;
} rt_photon_trace;

typedef struct rt_ctx rt_ctx;

/* Library / device management (no reference counterpart: the reference is single-process NumPy). */
int rt_abi_version(void);
const char* rt_last_error(void);
int rt_create(rt_ctx** out, int device_id);
int rt_destroy(rt_ctx* ctx);
/* name[<=256], compute units, wavefront size, clock (kHz) of the context's device */
int rt_device_info(rt_ctx* ctx, char* name, int name_len, int* n_cu, int* wave_size, int* clock_khz);

/* Optical constants per wavelength and the emission law.
 * Replaces the table side of CreatePhotons (MC code.py:371-412): wl/g/albedo/scale are the
 * WL, g, Albedo, Scale columns of Grain_data after the join on WL (MC code.py:403-408);
 * pdf[i] is the acceptance probability of wavelength i in AcceptReject (MC code.py:239-268),
 * i.e. P(lambda = wl[i]) is proportional to min(pdf[i], 1).  pdf == NULL means a uniform law.
 * n_wl == 1 gives gray / monochromatic dust.
 * Limits (RT_ERR_LIMIT beyond them): n_wl <= 2^24, n_tables * n_wl <= 2^26. */
int rt_set_tables(rt_ctx* ctx, int32_t n_wl, const double* wl, const double* g, const double* albedo,
                  const double* scale, const double* pdf);

/* Several tables at once on one wavelength grid, e.g. one per grain species or size window, so that a sweep
 * over (species x radius x optical depth) is ONE rt_run: g, albedo, scale, pdf are [n_tables][n_wl] (pdf may be
 * NULL), wl is [n_wl].  Atmospheres pick their table with rt_params.table_of_atm.  rt_set_tables is the
 * n_tables = 1 case. */
int rt_set_tables_multi(rt_ctx* ctx, int32_t n_tables, int32_t n_wl, const double* wl, const double* g,
                        const double* albedo, const double* scale, const double* pdf);

/* The sweep: for every atmosphere a, walk n_photons[a] photons through a slab of optical depth
 * tau_atm[a] and return the tallies.  Replaces the loop body of runMC (MC code.py:1056-1062)
 * and RunPhotons (MC code.py:620-779; gray: Numpy_random_walk.py:280-408).
 * Randomness: Philox4x32-10, key = seed, counter = (photon id lo, hi, block, atmosphere);
 * results do not depend on rank/world except through which slabs are walked. */
int rt_run(rt_ctx* ctx, const rt_params* p, rt_tallies* out);

/* Same sweep, results left on the device for a collective:
 * d_slab_sums  [n_atm][n_slabs][RT_NSUM] doubles, d_counts [n_atm][RT_NCNT] u64,
 * d_mu_hist [n_atm][n_mu_bins+1] u64, d_slab_depth [n_atm][n_slabs][2][n_depth_bins] doubles (momentum bins, then
 * energy bins, of every slab; may be NULL when n_depth_bins == 0).  Buffers are overwritten (every entry: this rank's
 * share, zeros for the slabs of other ranks); summing the slab axis in index order gives rt_run's sums / depth_momentum /
 * depth_energy, with the same bits on any number of GPUs.  The work
 * is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and NOT synchronised.
 * A context owns one block of device work memory (plan, work queue, tally copies, and per chunk of photon ids the ten
 * partial sums and 2 * n_depth_bins depth bins; it grows on demand, so a call may allocate; RT_ERR_LIMIT beyond
 * 64 GiB): successive calls on one
 * context must go to the same stream, or the caller orders them itself (event or synchronise) -- two
 * sweeps of one context in flight on different streams race.  Use one context per concurrent stream. */
int rt_run_device(rt_ctx* ctx, const rt_params* p, void* d_slab_sums, void* d_counts, void* d_mu_hist,
                  void* d_slab_depth, void* stream);

/* The same sweep for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc, code that
 * must not disturb the allocator (e.g. while another stream is being captured into a HIP graph).  No counterpart in
 * the reference.  Same arguments and results as rt_run_device, bit for bit, but the work memory (work queue, plan,
 * tally copies, chunk partial sums) is a static device array of 32 MiB that the library holds on every GPU it runs on
 * from load to process exit, and the call itself allocates nothing on the device or the host and frees nothing; its
 * kernel is not timed (rt_kernel_time does not see it).  (The call is not itself capturable: it records an event
 * after the plan upload and waits for the previous call's before it reuses the staging buffer.)
 * Limits (RT_ERR_LIMIT beyond them, before anything is enqueued): n_atm <= 4096, and the work memory within the
 * array: 60 * n_atm bytes of plan, up to 8 MiB of tally copies (8 * n_atm * (RT_NCNT + n_mu_bins + 1) bytes each, as
 * many as fit, at most 64) and 80 + 16 * n_depth_bins bytes per chunk of this rank's slabs (a chunk is 8192 photon ids
 * for slabs of at least 250 000 ids, shorter below) <= 33554432 bytes in all -- the headline sweep (100 atmospheres x 2e7 photons on
 * one GPU: 244 800 chunks) needs 21 MB of it.  rt_set_tables must have been called before (that call allocates).
 * All contexts of one GPU share the array, and it keeps the last call's contents: the call clears what must start at
 * zero, and the library orders the static sweeps of a GPU among themselves -- a call that goes to another stream than
 * the GPU's previous static sweep makes its stream wait (hipStreamWaitEvent, on the device, no allocation) for that
 * sweep's last kernel, so static sweeps from different streams or contexts of one process take turns on the array
 * instead of racing, in the order of the calls.  What the library cannot see still races: the array is per process
 * and per GPU, but a caller that enqueues static sweeps of ONE context from two host threads at once, or frees a
 * result buffer while a sweep is in flight, is on its own (contexts are not thread-safe). */
int rt_run_device_static(rt_ctx* ctx, const rt_params* p, void* d_slab_sums, void* d_counts, void* d_mu_hist,
                         void* d_slab_depth, void* stream);

/* ---- the collective, for callers without a Python runtime (one process per GPU) ----
 * No counterpart in the reference (single process).  RCCL is opened at run time by these calls (librccl.so.1; a
 * process that already has it loaded -- PyTorch ships one -- shares that copy) and is not a link dependency.
 *   rt_comm_unique_id  on one process: fills id[RT_COMM_ID_BYTES] (an ncclUniqueId); the caller hands the bytes to
 *                      the other processes by whatever it launches them with (file, environment, MPI, a socket);
 *   rt_comm_init       on every process: joins the communicator as `rank` of `world` on the context's GPU
 *                      (collective: returns when all ranks have called it);
 *   rt_run_allreduce   rt_run_device for the slabs of this process's rank (rank / world of the communicator replace
 *                      p->rank / p->world; n_slabs must be a multiple of world) followed by ONE all-reduce on the
 *                      same stream, not synchronised.  d_tallies is one device buffer of 64-bit words,
 *                          [n_atm][n_slabs][RT_NSUM] doubles | [n_atm][RT_NCNT] u64 | [n_atm][n_mu_bins + 1] u64 |
 *                          [n_atm][n_slabs][2][n_depth_bins] doubles,
 *                      summed as 64-bit integers: every slab entry is non-zero on exactly one rank and zero bits on
 *                      the others, so the doubles come back unchanged, and summing the slab axis in index order
 *                      gives the same bits on any number of GPUs (depth bins included).
 *                      A non-zero return on one process (a refused argument, a failed allocation or launch) is that
 *                      process's alone: its peers are already waiting in the collective, so the caller must end the
 *                      job on every rank; the communicator is not usable afterwards.
 *   rt_comm_destroy    leaves the communicator (rt_destroy does it too). */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(void* id);
int rt_comm_init(rt_ctx* ctx, int rank, int world, const void* id);
int rt_comm_destroy(rt_ctx* ctx);
int rt_run_allreduce(rt_ctx* ctx, const rt_params* p, void* d_tallies, void* stream);

/* The wavelengths the sweeps emit, photon by photon: for photon ids [first_id, first_id + n) of atmosphere `atm`
 * (its index in p->tau_atm) under p's seed, photon_id_base, rng_mode, monochromatic_wl and table_of_atm, out_index[i]
 * is the index into wl[] that CreatePhotons' wavelength draw (AcceptReject, MC code.py:239-268, :389-394) is replaced
 * by, and out_mu0[i] (may be NULL) the starting cosine (rand_mu, MC code.py:384-385).  Host arrays; n <= 2^31.
 * For comparing the emission law with the reference's and for rebuilding what a sweep walked. */
int rt_emission_indices(rt_ctx* ctx, const rt_params* p, int32_t atm, uint64_t first_id, int64_t n, int32_t* out_index,
                        double* out_mu0);

/* Replay: photon i consumes xi[i*max_draws + j], j = 0,1,... instead of Philox, with its constants
 * given directly (what CreatePhotons would have put in its row).  Used to compare trajectories with
 * the reference run on identical pre-drawn numbers.  Only n_atm == 1 (tau_atm[0]) is used; a
 * re-emitted photon (RT_BOUNDARY_REEMIT) restarts with its own constants and mu0.  A photon that
 * needs more than max_draws uniforms ends with RT_FATE_GUARD. */
int rt_replay(rt_ctx* ctx, const rt_params* p, int64_t n, int32_t max_draws, const double* xi,
              const double* lambda, const double* g, const double* albedo, const double* scale,
              const double* mu0, rt_tallies* out, rt_photon_trace* per_photon);

/* ---- table builder on the device (the step before the walk) ----
 * One grain species as stored in a 'Draine data XXX' folder: per grain radius a table of
 * w(micron) Q_abs Q_sca g with wavelength DESCENDING.  Columns of all sizes are concatenated;
 * offset[i]..offset[i+1] are the rows of size i; sizes ascending in a[]. */
typedef struct rt_grain_set {
    int32_t n_sizes;
    const double* a;       /* [n_sizes] */
    const int32_t* offset; /* [n_sizes + 1] */
    const double* wl;      /* [offset[n_sizes]] */
    const double* q_abs;
    const double* q_sca;
    const double* g;
} rt_grain_set;

/* Host result arrays, each [n_wl] (any may be NULL), and three scalars. */
typedef struct rt_grain_tables {
    double* kappa_rp;  /* GetKappas: kappa_RP(lambda), read_BPASS.py:161 */
    double* kappa_f;   /* GetKappas: kappa_F(lambda),  read_BPASS.py:162 */
    double* sigma_abs; /* GetDraineAverage: Sigma_abs_av, MC code.py:308 */
    double* sigma_sca; /* Sigma_scatt_av, MC code.py:309 */
    double* g_av;      /* MC code.py:310 */
    double* q_ext;     /* MC code.py:312 */
    double kappa_av_rp; /* luminosity-weighted means, read_BPASS.py:159-160 */
    double kappa_av_f;
    double lum;         /* sum of the spectrum column, read_BPASS.py:136 */
} rt_grain_tables;

/* Interpolates every grain size to wl[n_wl] (np.interp semantics) and integrates over grain size
 * (np.trapz) on the GPU.  Replaces GetKappas (read_BPASS.py:56-170; sizes in [grain_min, grain_max],
 * `density` in g cm^-2 um^-1) and the commented GetDraineAverage recipe (MC code.py:270-324; all sizes,
 * or the same size range if average_all_sizes == 0).  sed[n_wl] is the spectrum column (NULL = flat). */
int rt_build_grain_tables(rt_ctx* ctx, const rt_grain_set* gs, double density, double grain_min, double grain_max,
                          int32_t average_all_sizes, int32_t n_wl, const double* wl, const double* sed,
                          rt_grain_tables* out);

/* ---- the consumer of the curve: shell velocities of many H II regions (the step after the walk) ---- */
enum rt_shell_status {
    RT_SHELL_REACHED_END = 0, /* integrated to r_end */
    RT_SHELL_STALLED = 1,     /* v -> 0 before r_end; r_last is the stall radius */
    RT_SHELL_ESCAPED = 2,     /* passed r_escape (the reference sets dv/dr = -inf there, MC code.py:951-953) */
    RT_SHELL_STEP_LIMIT = 3   /* max_steps used up */
};

/* gal_data as ODEs reads it when the reference's includeWholeGalaxy flag is set (MC code.py:919-920, 940-941): the
 * light and the gravity of every region of the galaxy as seen from the region being integrated.  The reference
 * prepares the two per-view columns for one region_ID at a time (commented set-up, MC code.py:1682-1683); here
 * n_views such views are handed over together and integrated region i uses view view_of_region[i].  The formulas
 * are taken as written, units included: sin(arctan(r / dist_to_ref / cm_pc)) weights both sums, the gravity sum
 * divides by (dist_to_ref / cm_pc)^2 + r^2 and leaves out row self_row[i] (gal_data.ID != region_ID), the light
 * sum runs over all rows (the region's own row has dist_to_ref = 0: weight 1).  All pointers are host arrays. */
typedef struct rt_shell_galaxy {
    int32_t n_gal;                 /* rows of gal_data (>= 1) */
    int32_t n_views;               /* >= 1 */
    double cm_pc;                  /* constants.py:2 */
    double ratio_rp_lambda_rp;     /* GetRatios(...)[2] at time_slice (MC code.py:1678) */
    const double* lha_ratio_f;     /* [n_views][n_gal] gal_data.LHa_ratio_f */
    const double* dist_to_ref;     /* [n_views][n_gal] gal_data.dist_to_ref, pc, >= 0 */
    const double* mass_tot;        /* [n_gal] gal_data.Mass_tot, g */
    const int32_t* view_of_region; /* [regions->n] in [0, n_views) */
    const int32_t* self_row;       /* [regions->n] in [-1, n_gal): the region's own row of gal_data, -1 = none */
} rt_shell_galaxy;

/* What ODEs reads besides the region (MC code.py:905-955): the curve MC_data (GetMC, :843-863), the
 * luminosity of the population per age (L_ratio, :917) and the physical constants.  All pointers are host
 * arrays; grids ascending. */
typedef struct rt_shell_model {
    int32_t aging;           /* stellar_aging */
    int32_t static_momentum; /* MC code.py:119 */
    int32_t n_tau, n_age, n_lage;
    uint32_t struct_size;    /* = sizeof(rt_shell_model); any other value is refused with RT_ERR_INVALID */
    double G, c, sigma_sb, year_s; /* cgs; year_s = 31556952 (MC code.py:943) */
    double h_hi_cm;                /* h_Hi * cm_pc */
    double r_escape;               /* 10000 * cm_pc */
    double time_slice;             /* log10 age used when aging == 0 */
    double tau_cutoff;             /* GetMC's tau_cutoff; 0 = off */
    double l_norm;                 /* summed luminosity at time_slice */
    const double* mc_tau;          /* [n_tau]  MC_data.tau */
    const double* mc_age;          /* [n_age]  the age columns of MC_data as numbers */
    const double* mc_val;          /* [n_tau][n_age] */
    const double* l_age;           /* [n_lage] ages of l_val / ratio_val */
    const double* l_val;           /* [n_lage] summed luminosity per age (needed if aging) */
    const double* ratio_val;       /* [n_lage] GetRatios' ratio_rp_lambda_rp per age (needed if tau_cutoff > 0) */
    const rt_shell_galaxy* galaxy; /* includeWholeGalaxy (MC code.py:118): NULL = off, the reference's setting */
} rt_shell_model;

/* regionData of GetRegionVelocity (MC code.py:1438-1457), one entry per region, cgs units. */
typedef struct rt_shell_regions {
    int32_t n, pad_;
    const double *m_g, *m_new, *m_old, *m_co_hi, *h_old, *h_g_i, *tau_i, *l_bol, *sigma_g;
    const double* r_end; /* integrate from h_g_i to here (the reference: 100 * H_old, MC code.py:1459) */
} rt_shell_regions;

typedef struct rt_shell_control {
    double v0, t0;               /* v_0 (cm/s), t_0 (yr): MC code.py:102-103 */
    double rtol, atol_v, atol_t; /* local error control of the embedded pair */
    double max_step;             /* cm; the reference uses cm_pc */
    int32_t n_out;               /* samples per region at r0 + (k+1)/n_out (r_end - r0); 0 = none */
    int32_t max_steps;
} rt_shell_control;

/* Host result arrays, [n] each; v_out / t_out are [n][n_out] (NaN beyond the last radius reached). */
typedef struct rt_shell_result {
    double *v_max, *r_vmax, *t_vmax; /* largest velocity and where / when */
    double *r_last, *v_last, *t_last;
    int32_t *status, *n_steps;
    double *v_out, *t_out;
} rt_shell_result;

/* Replaces the per-region loop over GetRegionVelocity (MC code.py:1433-1472, spherical case). */
int rt_shell_velocity(rt_ctx* ctx, const rt_shell_model* model, const rt_shell_regions* regions,
                      const rt_shell_control* control, rt_shell_result* out);

/* dv/dr and dt/dr of ODEs (MC code.py:905-955) at n points (region[i], r[i], v[i], t[i]). */
int rt_shell_rhs(rt_ctx* ctx, const rt_shell_model* model, const rt_shell_regions* regions, int32_t n,
                 const int32_t* region, const double* r, const double* v, const double* t, double* dvdr,
                 double* dtdr);

/* Test hook (no reference counterpart): applies the kernels' own fp64 ln(u), cos(2*pi*u), sqrt(u) and
 * division ((1 - u*u) / (2*u)) to n host values u in [0,1) so that they can be checked against a host
 * math library. */
int rt_selftest_math(rt_ctx* ctx, int64_t n, const double* u, double* out_log, double* out_cos, double* out_sqrt,
                     double* out_div);

/* Average kernel time (ms) and launch count of the walk kernel since the last call with reset != 0,
 * measured with HIP events on the stream the kernel ran on. */
int rt_kernel_time(rt_ctx* ctx, double* avg_ms, int64_t* launches, int reset);

/* Name of the walk-kernel instantiation that served the context's last sweep ("" before the first): the sweep's
 * flags decide whether a kernel with those flags fixed at compile time or the one that reads them at run time runs. */
const char* rt_last_kernel(rt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
