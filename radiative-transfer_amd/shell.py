"""The consumer of the Monte Carlo curve: shell velocities of many H II regions on the GPU.

Mirrors, for all regions of a galaxy table at once,

    get_tau_IR          MC code.py:781-807
    GetRatios           MC code.py:1090-1157   (ratio_rp_lambda_rp, used by GetMC's tau_cutoff)
    GetMC               MC code.py:843-863
    ODEs                MC code.py:905-955     (spherical shell; includeWholeGalaxy through ShellGalaxy)
    GetRegionVelocity   MC code.py:1433-1472

The integration itself runs in librt_amd.so (rt_shell_velocity); this module prepares its inputs from the
reference's tables (MC_data, L_Edd_DF, BPASS luminosities, gal_data columns) and keeps the reference's
names for the knobs (h_g, h_Hi, v_0, t_0, time_slice, tau_cutoff, stellar_aging, static_momentum).
"""
import ctypes as C

import numpy as np

from . import _abi as abi
from .engine import _dp

# constants.py of the reference (cgs)
CM_PC = 3.086 * 10 ** 18
C_LIGHT = 2.99792458 * (10 ** 10)
G_NEWTON = 6.67259 * (10 ** -8)
SIGMA_SB = 5.67037441918442945397 * 10 ** -5
YEAR_S = 31556952.0  # MC code.py:943

STATUS_NAMES = {abi.SHELL_REACHED_END: "reached r_end", abi.SHELL_STALLED: "stalled", abi.SHELL_ESCAPED: "escaped",
                abi.SHELL_STEP_LIMIT: "step limit"}


def get_tau_ir(r, l_bol, sigma_g):
    """Infrared optical depth of the shell (get_tau_IR, MC code.py:790-803), vectorised."""
    r, l_bol, sigma_g = np.broadcast_arrays(np.asarray(r, float), np.asarray(l_bol, float), np.asarray(sigma_g, float))
    t_eff = (l_bol / (4 * np.pi * r ** 2 * SIGMA_SB)) ** 0.25
    tau_r = 2.4 * 10 ** -4 * t_eff ** 2 * sigma_g
    temp = (3 / 4 * (tau_r + 2 / 3) * t_eff ** 4) ** 0.25
    return np.where(tau_r > 1, 2.4 * 10 ** -4 * temp ** 2 * sigma_g, tau_r)


def ratio_rp_lambda_rp(l_edd, kappa_rp_ref, grain_type="mix", grain_mix=(0.5, 0.45, 0.05)):
    """GetRatios' third return value for every age row of the L_Edd table (MC code.py:1106-1125): the
    luminosity-weighted <kappa_RP> over kappa_RP at the reference wavelength.  `l_edd` maps column names
    (kappa_av_RP_Sil, ...) to arrays."""
    if grain_type == "mix":
        k = sum(f * np.asarray(l_edd[f"kappa_av_RP_{s}"], float) for f, s in zip(grain_mix, ("Sil", "Gra", "SiC")))
    else:
        k = np.asarray(l_edd[f"kappa_av_RP_{grain_type}"], float)
    return k / float(kappa_rp_ref)


class ShellGalaxy:
    """gal_data as ODEs reads it with includeWholeGalaxy (MC code.py:919-920, 940-941): LHa_ratio_f and dist_to_ref
    as [n_views, n_gal] (one view = the columns prepared for one region_ID), Mass_tot [n_gal], and per integrated
    region its view and its own row of gal_data (-1: none, nothing is left out of the gravity sum)."""

    def __init__(self, lha_ratio_f, dist_to_ref, mass_tot, view_of_region, self_row, ratio_rp_lambda_rp):
        self.lha_ratio_f = np.ascontiguousarray(np.atleast_2d(lha_ratio_f), dtype=np.float64)
        self.dist_to_ref = np.ascontiguousarray(np.atleast_2d(dist_to_ref), dtype=np.float64)
        self.mass_tot = np.ascontiguousarray(mass_tot, dtype=np.float64)
        self.view_of_region = np.ascontiguousarray(np.atleast_1d(view_of_region), dtype=np.int32)
        self.self_row = np.ascontiguousarray(np.atleast_1d(self_row), dtype=np.int32)
        self.ratio_rp_lambda_rp = float(ratio_rp_lambda_rp)
        if self.lha_ratio_f.shape != self.dist_to_ref.shape or self.lha_ratio_f.shape[1] != len(self.mass_tot):
            raise ValueError("lha_ratio_f and dist_to_ref must be [n_views, len(mass_tot)]")
        if self.view_of_region.shape != self.self_row.shape:
            raise ValueError("view_of_region and self_row must have one entry per integrated region")

    @classmethod
    def from_gal_data(cls, gal, rows, ratio_rp_lambda_rp, distance_pc):
        """The set-up the reference keeps as comments (MC code.py:1682-1683), for every row in `rows` taken as
        region_ID in turn: LHa_ratio_f = LHa exp(-tau ratio) / LHa[ref], dist_to_ref = Distance sqrt(tan^2(dRA) +
        tan^2(dDec)) with RA, Dec in degrees.  `gal` maps the gal_data columns LHa, tau, RA, Dec, Mass_tot to
        arrays; region k of the integration is gal_data row rows[k]."""
        rows = np.atleast_1d(np.asarray(rows, dtype=np.int64))
        lha, tau, ra, dec = (np.asarray(gal[k], float) for k in ("LHa", "tau", "RA", "Dec"))
        lha_f = lha[None, :] * np.exp(-tau * ratio_rp_lambda_rp)[None, :] / lha[rows, None]
        dist = distance_pc * np.sqrt(np.tan(np.radians(ra[None, :] - ra[rows, None])) ** 2
                                     + np.tan(np.radians(dec[None, :] - dec[rows, None])) ** 2)
        return cls(lha_f, dist, gal["Mass_tot"], np.arange(len(rows)), rows, ratio_rp_lambda_rp)

    def c_struct(self, n_regions):
        if len(self.view_of_region) != n_regions:
            raise ValueError(f"the galaxy views are for {len(self.view_of_region)} regions, {n_regions} are integrated")
        i32p = C.POINTER(C.c_int32)
        g = abi.RtShellGalaxy(self.lha_ratio_f.shape[1], self.lha_ratio_f.shape[0], CM_PC, self.ratio_rp_lambda_rp,
                              _dp(self.lha_ratio_f), _dp(self.dist_to_ref), _dp(self.mass_tot),
                              self.view_of_region.ctypes.data_as(i32p), self.self_row.ctypes.data_as(i32p))
        g._keep = self
        return g


class ShellModel:
    """Everything ODEs reads that is not a property of one region.  galaxy: a ShellGalaxy switches the reference's
    includeWholeGalaxy terms on (MC code.py:118 has them off)."""

    def __init__(self, mc_tau, mc_ages, mc_values, *, time_slice=6.0, stellar_aging=False, static_momentum=False,
                 tau_cutoff=0.0, l_ages=None, l_values=None, ratio_values=None, h_Hi=100.0, galaxy=None):
        self.galaxy = galaxy
        self.mc_tau = np.ascontiguousarray(mc_tau, dtype=np.float64)
        self.mc_ages = np.ascontiguousarray(mc_ages, dtype=np.float64)
        self.mc_values = np.ascontiguousarray(mc_values, dtype=np.float64)
        if self.mc_values.shape != (len(self.mc_tau), len(self.mc_ages)):
            raise ValueError("mc_values must be [len(mc_tau), len(mc_ages)]")
        self.time_slice = float(time_slice)
        self.stellar_aging = bool(stellar_aging)
        self.static_momentum = bool(static_momentum)
        self.tau_cutoff = float(tau_cutoff)
        self.l_ages = None if l_ages is None else np.ascontiguousarray(l_ages, dtype=np.float64)
        self.l_values = None if l_values is None else np.ascontiguousarray(l_values, dtype=np.float64)
        self.ratio_values = None if ratio_values is None else np.ascontiguousarray(ratio_values, dtype=np.float64)
        self.h_Hi = float(h_Hi)
        if self.stellar_aging and (self.l_ages is None or self.l_values is None):
            raise ValueError("stellar_aging needs l_ages / l_values (summed luminosity per age)")
        if self.tau_cutoff > 0 and (self.l_ages is None or self.ratio_values is None):
            raise ValueError("tau_cutoff > 0 needs l_ages / ratio_values")

    @classmethod
    def from_mc_frame(cls, mc_data, **kw):
        """mc_data: the frame runMC's CSVs make -- a `tau` column and one column per log age
        (MC code.py:70, :1069-1071)."""
        cols = [c for c in mc_data.columns if c != "tau"]
        ages = np.array([float(c) for c in cols])
        order = np.argsort(ages)
        vals = np.stack([np.asarray(mc_data[cols[i]], float) for i in order], axis=1)
        return cls(np.asarray(mc_data["tau"], float), ages[order], vals, **kw)

    def _l_norm(self):
        if self.l_values is None:
            return 1.0
        hit = np.where(self.l_ages == self.time_slice)[0]
        # BPASS_data[str(time_slice)].sum(): the tabulated value at time_slice (MC code.py:917)
        return float(self.l_values[hit[0]]) if len(hit) else float(np.interp(self.time_slice, self.l_ages, self.l_values))

    def c_struct(self, n_regions=0):
        n_l = 0 if self.l_ages is None else len(self.l_ages)
        gal = None if self.galaxy is None else self.galaxy.c_struct(n_regions)
        m = abi.RtShellModel(int(self.stellar_aging), int(self.static_momentum), len(self.mc_tau), len(self.mc_ages), n_l,
                             C.sizeof(abi.RtShellModel),
                             G_NEWTON, C_LIGHT, SIGMA_SB, YEAR_S, self.h_Hi * CM_PC, 10000 * CM_PC, self.time_slice,
                             self.tau_cutoff, self._l_norm(), _dp(self.mc_tau), _dp(self.mc_ages), _dp(self.mc_values),
                             None if self.l_ages is None else _dp(self.l_ages),
                             None if self.l_values is None else _dp(self.l_values),
                             None if self.ratio_values is None else _dp(self.ratio_values),
                             None if gal is None else C.pointer(gal))
        m._keep = (self, gal)
        return m


class ShellRegions:
    """regionData of GetRegionVelocity for many regions (MC code.py:1438-1457); all cgs.
    r_end defaults to 100 * H_old (MC code.py:1459)."""

    def __init__(self, *, M_g, M_new, M_old, H_old, tau_i, L_Bol, Sigma_g, h_g=5.0, M_CO_Hi=0.0, r_end=None):
        cols = np.broadcast_arrays(*[np.atleast_1d(np.asarray(x, float)) for x in
                                     (M_g, M_new, M_old, M_CO_Hi, H_old, np.asarray(h_g, float) * CM_PC, tau_i, L_Bol, Sigma_g)])
        self.cols = [np.ascontiguousarray(c, dtype=np.float64) for c in cols]
        self.n = len(self.cols[0])
        r_end = 100 * self.cols[4] if r_end is None else np.broadcast_to(np.asarray(r_end, float), (self.n,))
        self.cols.append(np.ascontiguousarray(r_end, dtype=np.float64))

    @classmethod
    def from_gal_data(cls, gal, h_g=5.0, r_end=None):
        """gal: mapping with the gal_data columns Mass_g, Mass_new, Mass_old, H_old, tau, L_Bol, Sigma_g and,
        if present, Mass_CO_Hi."""
        co = gal["Mass_CO_Hi"] if "Mass_CO_Hi" in gal else 0.0
        return cls(M_g=gal["Mass_g"], M_new=gal["Mass_new"], M_old=gal["Mass_old"], H_old=gal["H_old"], tau_i=gal["tau"],
                   L_Bol=gal["L_Bol"], Sigma_g=gal["Sigma_g"], h_g=h_g, M_CO_Hi=co, r_end=r_end)

    @property
    def r_start(self):
        return self.cols[5]

    @property
    def r_end(self):
        return self.cols[9]

    def c_struct(self):
        s = abi.RtShellRegions(self.n, 0, *[_dp(c) for c in self.cols])
        s._keep = self
        return s


class ShellSolution:
    def __init__(self, n, n_out):
        self.v_max, self.r_vmax, self.t_vmax = np.empty(n), np.empty(n), np.empty(n)
        self.r_last, self.v_last, self.t_last = np.empty(n), np.empty(n), np.empty(n)
        self.status = np.empty(n, dtype=np.int32)
        self.n_steps = np.empty(n, dtype=np.int32)
        self.v_out = np.empty((n, n_out))
        self.t_out = np.empty((n, n_out))
        self.r_out = None


def shell_velocity(engine, model, regions, *, v_0=100.0, t_0=None, rtol=1e-8, atol_v=1e-6, atol_t=1e-6,
                   max_step=CM_PC, n_out=0, max_steps=2000000):
    """Integrates dv/dr, dt/dr of every region from h_g to r_end (GetRegionVelocity, MC code.py:1459-1469:
    v(h_g) = v_0 cm/s, t(h_g) = t_0 = 10**time_slice years, steps of at most max_step).  Returns a
    ShellSolution: per region the largest velocity (cm/s) with its radius and time, the last state reached,
    a status (see STATUS_NAMES) and, if n_out > 0, v and t at r_out[:, k]."""
    t_0 = 10.0 ** model.time_slice if t_0 is None else float(t_0)
    sol = ShellSolution(regions.n, n_out)
    ctl = abi.RtShellControl(float(v_0), t_0, float(rtol), float(atol_v), float(atol_t), float(max_step), int(n_out),
                             int(max_steps))
    res = abi.RtShellResult(_dp(sol.v_max), _dp(sol.r_vmax), _dp(sol.t_vmax), _dp(sol.r_last), _dp(sol.v_last),
                            _dp(sol.t_last), sol.status.ctypes.data_as(C.POINTER(C.c_int32)),
                            sol.n_steps.ctypes.data_as(C.POINTER(C.c_int32)), _dp(sol.v_out), _dp(sol.t_out))
    m, r = model.c_struct(regions.n), regions.c_struct()
    engine._check(engine._lib.rt_shell_velocity(engine._ctx, C.byref(m), C.byref(r), C.byref(ctl), C.byref(res)))
    if n_out:
        frac = (np.arange(n_out) + 1.0) / n_out
        sol.r_out = regions.r_start[:, None] + frac[None, :] * (regions.r_end - regions.r_start)[:, None]
    return sol


def shell_rhs(engine, model, regions, region_index, r, v, t):
    """dv/dr, dt/dr of ODEs (MC code.py:905-955) at the given points."""
    idx = np.ascontiguousarray(region_index, dtype=np.int32)
    r, v, t = (np.ascontiguousarray(np.broadcast_to(np.asarray(x, float), idx.shape), dtype=np.float64) for x in (r, v, t))
    dvdr, dtdr = np.empty(len(idx)), np.empty(len(idx))
    m, rg = model.c_struct(regions.n), regions.c_struct()
    engine._check(engine._lib.rt_shell_rhs(engine._ctx, C.byref(m), C.byref(rg), len(idx),
                                           idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(r), _dp(v), _dp(t), _dp(dvdr),
                                           _dp(dtdr)))
    return dvdr, dtdr


def get_mc(model, tau, new_time):
    """GetMC (MC code.py:843-863) on the host, for single lookups such as gal_data['Momentum']
    (MC code.py:1385): np.interp along tau at a tabulated age, bilinear in (tau, age) otherwise."""
    tau = np.asarray(tau, float)
    vals = model.mc_values
    if model.tau_cutoff > 0:
        ratio = float(np.interp(new_time, model.l_ages, model.ratio_values))
        thin = model.mc_tau * ratio < model.tau_cutoff
        vals = np.where(thin[:, None], (1 - np.exp(-model.mc_tau * ratio))[:, None], vals)
    hit = np.where(model.mc_ages == new_time)[0]
    if len(hit):
        return np.interp(tau, model.mc_tau, vals[:, hit[0]])
    k = int(np.clip(np.searchsorted(model.mc_ages, new_time, side="right") - 1, 0, len(model.mc_ages) - 2))
    y1 = (np.clip(new_time, model.mc_ages[0], model.mc_ages[-1]) - model.mc_ages[k]) / (model.mc_ages[k + 1] - model.mc_ages[k])
    return (1 - y1) * np.interp(tau, model.mc_tau, vals[:, k]) + y1 * np.interp(tau, model.mc_tau, vals[:, k + 1])
