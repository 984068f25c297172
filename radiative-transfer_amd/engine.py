"""Thin ctypes host over the C ABI (include/rt_amd.h): one ``Engine`` per GPU.

All numerical work happens in the HIP library; this module only marshals
NumPy arrays and mirrors the argument meaning of the reference's
``RunPhotons`` (MC code.py:620-779, Numpy_random_walk.py:280-408).
"""
import ctypes as C

import numpy as np

from . import _abi as abi

H_PLANCK = 6.6261 * (10 ** (-27))  # constants.py:10
DEFAULT_SEED = 0x5EED2026


class RtError(RuntimeError):
    pass


def _dp(a):
    return a.ctypes.data_as(abi.c_double_p)


def _u64p(a):
    return a.ctypes.data_as(abi.c_u64_p)


class SweepResult:
    """Tallies of one sweep, one row per atmosphere (see the RT_SUM_* / RT_CNT_* enums)."""

    def __init__(self, n_atm, n_mu_bins, n_depth_bins, n_slabs):
        self.sums = np.zeros((n_atm, abi.NSUM))
        self.counts = np.zeros((n_atm, abi.NCNT), dtype=np.uint64)
        self.mu_hist = np.zeros((n_atm, n_mu_bins + 1), dtype=np.uint64)
        self.depth_momentum = np.zeros((n_atm, max(n_depth_bins, 1)))
        self.depth_energy = np.zeros((n_atm, max(n_depth_bins, 1)))
        self.slab_sums = np.zeros((n_atm, n_slabs, abi.NSUM))
        self.slab_depth = np.zeros((n_atm, n_slabs, 2, max(n_depth_bins, 1)))
        self.n_depth_bins = n_depth_bins
        self._c = abi.RtTallies(_dp(self.sums), _u64p(self.counts), _u64p(self.mu_hist), _dp(self.depth_momentum),
                                _dp(self.depth_energy), _dp(self.slab_sums), _dp(self.slab_depth) if n_depth_bins else None)

    # the reference's names (RunPhotons return values, MC code.py:779)
    @property
    def initial_momentum(self):
        return self.sums[:, abi.SUM_INITIAL]

    @property
    def escaped_momentum(self):
        return self.sums[:, abi.SUM_ESCAPED]

    @property
    def boundary_momentum(self):
        return self.sums[:, abi.SUM_BOUNDARY]

    @property
    def scatter_count(self):
        return self.counts[:, abi.CNT_SCATTER]

    def finalize_from_slabs(self):
        """sums = slab partials added in slab order (the same order on any GPU count)."""
        self.sums[...] = 0.0
        for s in range(self.slab_sums.shape[1]):
            self.sums += self.slab_sums[:, s, :]
        if self.n_depth_bins:
            tot = np.zeros_like(self.slab_depth[:, 0])
            for s in range(self.slab_depth.shape[1]):
                tot += self.slab_depth[:, s]
            self.depth_momentum[...], self.depth_energy[...] = tot[:, 0], tot[:, 1]

    def momentum_fraction(self):
        """(initial - escaped + boundary) / initial per atmosphere, runMC, MC code.py:1062."""
        s = self.sums
        return (s[:, abi.SUM_INITIAL] - s[:, abi.SUM_ESCAPED] + s[:, abi.SUM_BOUNDARY]) / s[:, abi.SUM_INITIAL]

    def deposited_fraction(self):
        """Momentum given to the dust at the collisions / initial momentum (RT_SUM_DUST_MOMENTUM): the reference's
        commented-out second estimator (MC code.py:773-776).  In the slab with a reflecting face it equals
        (initial - escaped - escaped_uncounted + 2 boundary) / initial, one boundary term more than
        momentum_fraction().  In a shell (RT_GEOM_SHELL) it is THE result: radial momentum is not conserved along a
        straight flight, so a balance of initial, escaped and boundary momentum does not apply."""
        return self.sums[:, abi.SUM_DUST_MOMENTUM] / self.sums[:, abi.SUM_INITIAL]

    def momentum_fraction_sigma(self, shell=False):
        """One-sigma statistical error of momentum_fraction (ratio estimator over photons); with `shell`, of
        deposited_fraction() of a shell sweep (whose per-photon terms are the deposited momenta)."""
        s = self.sums
        n = self.counts[:, abi.CNT_PHOTONS].astype(float)
        m = self.deposited_fraction() if shell else self.momentum_fraction()
        resid = s[:, abi.SUM_TRANSFER_SQ] - 2 * m * s[:, abi.SUM_TRANSFER_P0] + m * m * s[:, abi.SUM_P0_SQ]
        with np.errstate(invalid="ignore", divide="ignore"):
            var = np.maximum(resid, 0.0) * n / np.maximum(n - 1, 1) / s[:, abi.SUM_INITIAL] ** 2
        return np.sqrt(var)


def make_params(tau, n_photons, *, boundary="reflect", scatter="hg", order="mc", rand_mu=0, abs_only=0,
                n_mu_bins=10, n_depth_bins=0, max_events=0, n_slabs=8, rank=0, world=1, seed=DEFAULT_SEED,
                photon_id_base=0, monochromatic=0.0, alpha=0.3, h=H_PLANCK, rng="philox52", table_of_atm=None,
                geometry="slab", shell_inner=0.0, escaped_mu=False, runtime_kernel=False):
    """rng: "philox52" (default; 52-bit uniforms, two Philox blocks per loop iteration), "philox32x4" (one
    block per iteration, 32-bit uniforms; faster) or "philox40" (one block per iteration cut as 24 | 40 | 40 | 24 bits:
    as fast, finer steps and cosines); see rt_rng_mode in include/rt_amd.h.
    geometry: "slab" (the reference's) or "shell", a uniform spherical shell whose inner radius is `shell_inner`
    thicknesses (see RT_GEOM_SHELL in include/rt_amd.h; use SweepResult.deposited_fraction() for its result).
    escaped_mu: True asks Engine.run for the escape cosine of every photon (SweepResult.escaped_mu); an integer is taken
    as the address of a caller-owned DEVICE array for Engine.run_device (rt_params.escaped_mu).
    runtime_kernel: walk with the kernel that reads every option at run time even where one with these options fixed at
    compile time exists (RT_FLAG_RUNTIME_KERNEL; same results, for testing one against the other)."""
    if geometry not in abi.GEOMETRY:
        raise ValueError(f"geometry must be one of {sorted(abi.GEOMETRY)}")
    if rng not in abi.RNG:
        raise ValueError(f"rng must be one of {sorted(abi.RNG)}")
    if boundary not in abi.BOUNDARY:
        # the reference only prints this and carries on (MC code.py:571); here it is an error
        raise ValueError("No valid boundary condition specified, enter a boundary condition.")
    if scatter not in abi.SCATTER:
        raise ValueError(f"scatter must be one of {sorted(abi.SCATTER)}")
    tau = np.ascontiguousarray(np.atleast_1d(tau), dtype=np.float64)
    npn = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(n_photons), tau.shape), dtype=np.uint64)
    p = abi.RtParams(struct_size=C.sizeof(abi.RtParams), n_atm=len(tau), boundary=abi.BOUNDARY[boundary],
                     scatter=abi.SCATTER[scatter], loop_order=abi.ORDER[order], rand_mu=int(rand_mu), abs_only=int(abs_only),
                     n_mu_bins=int(n_mu_bins), n_depth_bins=int(n_depth_bins), max_events_per_photon=int(max_events),
                     n_slabs=int(n_slabs), rank=int(rank), world=int(world), seed=int(seed),
                     photon_id_base=int(photon_id_base), monochromatic_wl=float(monochromatic), alpha=float(alpha),
                     h=float(h), tau_atm=_dp(tau), n_photons=_u64p(npn), rng_mode=abi.RNG[rng],
                     geometry=abi.GEOMETRY[geometry], shell_inner=float(shell_inner),
                     flags=abi.FLAG_RUNTIME_KERNEL if runtime_kernel else 0)
    p._keep = (tau, npn)
    p._want_escaped_mu = escaped_mu is True
    if escaped_mu is not True and escaped_mu:
        p.escaped_mu = C.cast(C.c_void_p(int(escaped_mu)), abi.c_double_p)
    if table_of_atm is not None:  # which of the tables given to Engine.set_tables (2-D columns) each atmosphere uses
        tab = np.ascontiguousarray(np.broadcast_to(np.atleast_1d(table_of_atm), tau.shape), dtype=np.int32)
        p.table_of_atm = tab.ctypes.data_as(C.POINTER(C.c_int32))
        p._keep = (tau, npn, tab)
    return p


class Engine:
    """One GPU context.  ``Engine(device)`` fails if the HIP library or an MI355X is missing."""

    def __init__(self, device=0):
        self._lib = abi.load_library()
        self._ctx = C.c_void_p()
        self._check(self._lib.rt_create(C.byref(self._ctx), int(device)))
        self.device = int(device)

    def _check(self, rc):
        if rc != 0:
            raise RtError(f"rt_amd error {rc}: {self._lib.rt_last_error().decode()}")

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, ws, clk = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.rt_device_info(self._ctx, name, 256, C.byref(cu), C.byref(ws), C.byref(clk)))
        return dict(name=name.value.decode(), compute_units=cu.value, wave_size=ws.value, clock_khz=clk.value)

    def set_tables(self, wl, g, albedo, scale, pdf=None):
        """Columns of Grain_data on the wavelength grid `wl`.  2-D columns [n_tables][n_wl] load several tables
        (species, size windows ...) at once; make_params(table_of_atm=...) says which atmosphere uses which."""
        wl_a = np.ascontiguousarray(wl, dtype=np.float64)
        cols = [np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64) for a in (g, albedo, scale)]
        n = len(wl_a)
        n_tables = cols[0].shape[0]
        if any(c.shape != (n_tables, n) for c in cols):
            raise ValueError("table columns differ in length")
        pdf_a = None
        if pdf is not None:
            pdf_a = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(pdf), (n_tables, n)), dtype=np.float64)
        self._check(self._lib.rt_set_tables_multi(self._ctx, n_tables, n, _dp(wl_a), *[_dp(c) for c in cols],
                                                  None if pdf_a is None else _dp(pdf_a)))
        self.n_wl = n
        self.n_tables = n_tables

    def run(self, params):
        n_slabs = params.n_slabs if params.n_slabs > 0 else 8
        res = SweepResult(params.n_atm, params.n_mu_bins, params.n_depth_bins, n_slabs)
        if getattr(params, "_want_escaped_mu", False):
            counts = np.ctypeslib.as_array(params.n_photons, shape=(params.n_atm,)).astype(np.int64)
            flat = np.full(int(counts.sum()), np.nan)
            params.escaped_mu = _dp(flat)
            self._check(self._lib.rt_run(self._ctx, C.byref(params), C.byref(res._c)))
            params.escaped_mu = None
            # per atmosphere: the cosines of the photons that left through the logged escape test, in photon order
            # (RunPhotons' escaped_mu, MC code.py:693,779)
            parts = np.split(flat, np.cumsum(counts)[:-1])
            res.escaped_mu = [q[~np.isnan(q)] for q in parts]
            return res
        self._check(self._lib.rt_run(self._ctx, C.byref(params), C.byref(res._c)))
        return res

    def run_device(self, params, slab_sums_ptr, counts_ptr, mu_hist_ptr, depth_ptr=None, stream=None, static=False):
        """Enqueue a sweep whose results stay in caller-owned device buffers (raw addresses; depth_ptr:
        [n_atm][n_slabs][2][n_depth_bins] doubles).  `static`: the entry point that allocates nothing
        (rt_run_device_static: work memory in the library's static device array)."""
        fn = self._lib.rt_run_device_static if static else self._lib.rt_run_device
        self._check(fn(self._ctx, C.byref(params), C.c_void_p(slab_sums_ptr),
                                            C.c_void_p(counts_ptr), C.c_void_p(mu_hist_ptr),
                                            C.c_void_p(depth_ptr) if depth_ptr else None,
                                            C.c_void_p(stream) if stream else None))

    # ---- the library's own collective (RCCL opened by the library; no torch.distributed involved) ----
    def comm_unique_id(self):
        """128 bytes identifying a new communicator; create on one process, hand to the others, give to comm_init."""
        buf = C.create_string_buffer(128)
        self._check(self._lib.rt_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        self._check(self._lib.rt_comm_init(self._ctx, int(rank), int(world), C.c_char_p(bytes(unique_id))))

    def comm_destroy(self):
        self._check(self._lib.rt_comm_destroy(self._ctx))

    def run_allreduce(self, params, tallies_ptr, stream=None):
        """Sweep of this rank's slabs + one all-reduce of the packed tally buffer (layout: distributed.TallyBuffer)."""
        self._check(self._lib.rt_run_allreduce(self._ctx, C.byref(params), C.c_void_p(tallies_ptr),
                                               C.c_void_p(stream) if stream else None))

    def emission_indices(self, params, atm, n, first_id=0, with_mu0=False):
        """Index into the table's wavelengths (and, with_mu0, the starting cosine) that the sweeps give photon ids
        first_id .. first_id + n - 1 of atmosphere `atm`: the stand-in for CreatePhotons' wavelength draw
        (MC code.py:389-394)."""
        idx = np.zeros(int(n), dtype=np.int32)
        mu0 = np.zeros(int(n)) if with_mu0 else None
        self._check(self._lib.rt_emission_indices(self._ctx, C.byref(params), int(atm), int(first_id), int(n),
                                                  idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                                  None if mu0 is None else _dp(mu0)))
        return (idx, mu0) if with_mu0 else idx

    def replay(self, params, xi, lam, g, albedo, scale, mu0):
        xi = np.ascontiguousarray(xi, dtype=np.float64)
        n, max_draws = xi.shape
        cols = [np.ascontiguousarray(a, dtype=np.float64) for a in (lam, g, albedo, scale, mu0)]
        if any(len(c) != n for c in cols):
            raise ValueError("per-photon arrays differ in length")
        res = SweepResult(1, params.n_mu_bins, params.n_depth_bins, 1)
        trace = np.zeros(n, dtype=np.dtype([("final_z", "f8"), ("final_mu", "f8"), ("fate", "i4"),
                                            ("n_scatter", "i4"), ("n_draws", "i4"), ("pad_", "i4")]))
        self._check(self._lib.rt_replay(self._ctx, C.byref(params), n, max_draws, _dp(xi), *[_dp(c) for c in cols],
                                        C.byref(res._c), trace.ctypes.data_as(C.POINTER(abi.RtPhotonTrace))))
        return res, trace

    def build_grain_tables(self, a, columns, density, grain_min, grain_max, wl, sed=None, average_all_sizes=True):
        """Grain-size integration on the GPU.  `a` ascending grain radii; `columns` = (wl, q_abs, q_sca, g), each a
        list with one array per size (stored order, wavelength descending).  Returns a dict with the columns of
        GetKappas / GetDraineAverage on `wl`."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        rows = np.array([len(c) for c in columns[0]], dtype=np.int64)
        offset = np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)
        cols = [np.ascontiguousarray(np.concatenate(c), dtype=np.float64) for c in columns]
        wl = np.ascontiguousarray(wl, dtype=np.float64)
        sed_a = None if sed is None else np.ascontiguousarray(sed, dtype=np.float64)
        gs = abi.RtGrainSet(len(a), _dp(a), offset.ctypes.data_as(C.POINTER(C.c_int32)), *[_dp(c) for c in cols])
        out = {k: np.empty(len(wl)) for k in ("Kappa_RP", "Kappa_F", "Sigma-Abs", "Sigma-Scatt", "g", "Q_ext")}
        res = abi.RtGrainTables(*[_dp(out[k]) for k in ("Kappa_RP", "Kappa_F", "Sigma-Abs", "Sigma-Scatt", "g", "Q_ext")],
                                0.0, 0.0, 0.0)
        self._check(self._lib.rt_build_grain_tables(self._ctx, C.byref(gs), float(density), float(grain_min),
                                                    float(grain_max), int(bool(average_all_sizes)), len(wl), _dp(wl),
                                                    None if sed_a is None else _dp(sed_a), C.byref(res)))
        out.update(kappa_av_RP=res.kappa_av_rp, kappa_av_F=res.kappa_av_f, L=res.lum, WL=wl)
        with np.errstate(invalid="ignore", divide="ignore"):
            out["Albedo"] = out["Sigma-Scatt"] / (out["Sigma-Scatt"] + out["Sigma-Abs"])  # MC code.py:320
        return out

    def selftest_math(self, u):
        """The device's ln(u), cos(2 pi u), sqrt(u) and (1 - u*u) / (2*u) for u in [0,1) (test hook)."""
        u = np.ascontiguousarray(u, dtype=np.float64)
        outs = [np.empty_like(u) for _ in range(4)]
        self._check(self._lib.rt_selftest_math(self._ctx, len(u), _dp(u), *[_dp(o) for o in outs]))
        return tuple(outs)

    def last_kernel(self):
        """Name of the walk-kernel instantiation that served the last sweep."""
        return self._lib.rt_last_kernel(self._ctx).decode()

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.rt_kernel_time(self._ctx, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value
