"""The reference's script-level entry points, running on the GPU engine.

``RunPhotons`` and ``runMC`` keep the names, argument order and meaning of the functions in
``MC code.py`` (:620-779 and :992-1073); ``RunPhotonsGray`` is the gray-dust ``RunPhotons`` of
``Numpy_random_walk.py`` (:280-408).  The flags the reference reads as module globals
(``rand_mu``, ``abs_only``, ``monochromatic``, ``h``) are keyword arguments with the same names
and defaults.  Differences a caller can see:

* random numbers come from Philox4x32-10 keyed by ``seed`` and the photon index, not from NumPy's
  global Mersenne Twister, so runs are reproducible and independent of the number of GPUs;
* ``escaped_mu`` is returned as a histogram (``MuHistogram``) by default -- the reference's only
  use of the list is to histogram it (Numpy_random_walk.py:436); ``return_escaped_mu=True`` returns
  the reference's float array, one cosine per escaped photon (in photon order, not escape order);
* a wavelength without a table entry, an unknown ``boundary`` and similar input errors raise
  instead of printing a message or looping forever (MC code.py:571, :403-408).
"""
import csv
import os

import numpy as np

from . import _abi as abi
from . import tables
from .engine import DEFAULT_SEED, H_PLANCK, Engine, make_params

_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


class MuHistogram:
    """Escape cosines binned as floor(n_bins * mu) (Numpy_random_walk.py:436); the last entry of
    ``counts`` holds mu >= 1 (photons that left without ever scattering when mu starts at 1)."""

    def __init__(self, counts):
        self.counts = np.asarray(counts, dtype=np.int64)
        self.n_bins = len(self.counts) - 1

    def __len__(self):
        return int(self.counts.sum())

    def midpoints(self):
        return (np.arange(self.n_bins) + 0.5) / self.n_bins

    def intensity(self, num_photons):
        """counts * n_bins / (2 N cos(theta)) per bin, the normalisation of Numpy_random_walk.py:446."""
        c = self.counts[:-1].astype(float).copy()
        c[-1] += self.counts[-1]
        return c / (2.0 * num_photons * self.midpoints()) * self.n_bins

    def as_samples(self):
        """One bin-midpoint value per escaped photon (for code that expects the reference's array)."""
        mids = np.append(self.midpoints(), 1.0)
        return np.repeat(mids, self.counts)


def _column(grain, name):
    col = grain[name]
    return col.to_numpy() if hasattr(col, "to_numpy") else np.asarray(col)


def _tables_for(PDF, Range, Grain_data, monochromatic):
    """(wl, g, albedo, scale, pdf) as rt_set_tables wants them: Grain_data joined on the rounded
    wavelength for every candidate wavelength, exactly like CreatePhotons' left-join (MC code.py:394-408)."""
    grain = {k: _column(Grain_data, k) for k in ("WL", "g", "Albedo", "Scale")}
    if monochromatic:
        wl, g, alb, sc = tables.join_on_wavelength([monochromatic], grain)
        if np.isnan(g[0]):
            raise ValueError(f"monochromatic wavelength {monochromatic} has no entry in Grain_data.WL")
        return wl, g, alb, sc, None
    if Range is None or PDF is None:
        raise ValueError("PDF and Range are required unless monochromatic is set")
    wl, g, alb, sc = tables.join_on_wavelength(Range, grain)
    pdf = np.asarray(PDF, dtype=float).copy()
    missing = np.isnan(g) & (pdf > 0)
    if missing.any():
        raise ValueError(f"{int(missing.sum())} wavelengths with non-zero emission probability have no entry in "
                         "Grain_data.WL (the reference would loop forever on their NaN constants)")
    pdf[np.isnan(g)] = 0.0
    return wl, np.nan_to_num(g), np.nan_to_num(alb), np.nan_to_num(sc), pdf


def run_sweep(tau_list, num_photons, boundary, scatter, PDF, Range, Grain_data, *, rand_mu=False, abs_only=False,
              monochromatic=0, h=H_PLANCK, order="mc", seed=DEFAULT_SEED, photon_id_base=0, n_mu_bins=10,
              n_depth_bins=0, max_events=0, alpha=0.3, geometry="slab", shell_inner=0.0, engine=None, escaped_mu=False):
    """All atmospheres of tau_list in one kernel launch; returns the engine's SweepResult.
    scatter may also be 'draine_exact' and geometry 'shell' (extensions, see include/rt_amd.h)."""
    eng = engine or default_engine()
    wl, g, alb, sc, pdf = _tables_for(PDF, Range, Grain_data, monochromatic)
    eng.set_tables(wl, g, alb, sc, pdf)
    prm = make_params(tau_list, num_photons, boundary=boundary, scatter=scatter, order=order, rand_mu=int(bool(rand_mu)),
                      abs_only=int(bool(abs_only)), n_mu_bins=n_mu_bins, n_depth_bins=n_depth_bins,
                      max_events=max_events, seed=seed, photon_id_base=photon_id_base,
                      monochromatic=float(monochromatic) if monochromatic else 0.0, alpha=alpha, h=h,
                      geometry=geometry, shell_inner=shell_inner, escaped_mu=bool(escaped_mu))
    return eng.run(prm)


def RunPhotons(tau_atm, num_photons, boundary, scatter, PDF, Range, Grain_data, *, rand_mu=False, abs_only=False,
               monochromatic=0, h=H_PLANCK, seed=DEFAULT_SEED, n_mu_bins=10, engine=None, return_escaped_mu=False):
    """MC code.py:620.  Returns (escaped_mu, boundary_momentum, escaped_momentum, initial_momentum, scatter_count).
    escaped_mu is a MuHistogram, or with return_escaped_mu=True the reference's own form: a float array with the
    cosine of every photon that left through the logged escape test (MC code.py:690-693)."""
    res = run_sweep([tau_atm], num_photons, boundary, scatter, PDF, Range, Grain_data, rand_mu=rand_mu,
                    abs_only=abs_only, monochromatic=monochromatic, h=h, seed=seed, n_mu_bins=n_mu_bins, engine=engine,
                    escaped_mu=return_escaped_mu)
    emu = res.escaped_mu[0] if return_escaped_mu else MuHistogram(res.mu_hist[0])
    return (emu, float(res.boundary_momentum[0]), float(res.escaped_momentum[0]),
            float(res.initial_momentum[0]), int(res.scatter_count[0]))


def RunPhotonsGray(tau_atm, num_photons, boundary, albedo, scatter, g, *, rand_mu=0, alpha=0.3, seed=DEFAULT_SEED,
                   n_mu_bins=10, engine=None, return_escaped_mu=False):
    """Numpy_random_walk.py:280 (scalar albedo and g, lambda = 1, h = 1, boundary tested before escape).
    Returns (escaped_mu, boundary_momentum, escaped_momentum, initial_momentum)."""
    grain = {"WL": np.array([1.0]), "g": np.array([float(g)]), "Albedo": np.array([float(albedo)]),
             "Scale": np.array([1.0])}
    res = run_sweep([tau_atm], num_photons, boundary, scatter, None, None, grain, rand_mu=rand_mu, monochromatic=1.0,
                    h=1.0, order="gray", seed=seed, n_mu_bins=n_mu_bins, alpha=alpha, engine=engine,
                    escaped_mu=return_escaped_mu)
    emu = res.escaped_mu[0] if return_escaped_mu else MuHistogram(res.mu_hist[0])
    return (emu, float(res.boundary_momentum[0]), float(res.escaped_momentum[0]), float(res.initial_momentum[0]))


# ----------------------------------------------------------------------------------------------
# runMC
# ----------------------------------------------------------------------------------------------
BPASS_LOG_AGES = tuple(f"{6.0 + 0.1 * i:.1f}" for i in range(51))  # the 51 ages of every BPASS v2 output table


def read_bpass_spectra(path):
    """A BPASS 'spectra-*.dat' file as {column name: array}: 'WL' (Angstrom) and the 51 log-age columns '6.0' ...
    '11.0' that the reference indexes as BPASS_data[str(time_slice)] (MC code.py:1013, :917).  The file is a
    whitespace-separated table of Fortran E-format numbers, one row per wavelength, like the BPASS tables of the
    same model that ship with the reference (ionizing-*.dat, colours-*.dat: one row per age, the same 51 ages).
    Anything else -- another column count, a wavelength column that does not ascend, a non-number -- raises
    ValueError instead of being read as something it is not."""
    data = np.loadtxt(path, ndmin=2)
    if data.shape[1] != 1 + len(BPASS_LOG_AGES):
        raise ValueError(f"{path}: {data.shape[1]} columns; a BPASS spectra file has 52 (wavelength + 51 ages)")
    wl = data[:, 0]
    if not (np.all(np.isfinite(data)) and np.all(wl > 0) and np.all(np.diff(wl) > 0)):
        raise ValueError(f"{path}: the wavelength column must be positive and ascending, and every entry finite")
    cols = {"WL": wl}
    for i, name in enumerate(BPASS_LOG_AGES):
        cols[name] = data[:, i + 1]
    return cols


def photon_reduction_counts(num_photons, photon_reduction, n_atm):
    """np.linspace(N, N / photon_reduction, n_atm).astype(int), MC code.py:1053-1054."""
    if not photon_reduction:
        return np.full(n_atm, int(num_photons), dtype=np.int64)
    return np.linspace(num_photons, num_photons / photon_reduction, n_atm).astype(int).astype(np.int64)


def extrapolate(tau_list, tau_list_calc, momentum_method_1, min_tau, max_tau):
    """Thin-limit scaling below min_tau and plateau above max_tau, MC code.py:1064-1067."""
    tau_list = np.asarray(tau_list, dtype=float)
    out = np.zeros(len(tau_list))
    lo = tau_list < min_tau
    out[lo] = tau_list[lo] * np.mean(momentum_method_1[0:10] / (1 - np.exp(-tau_list_calc[0:10])))
    out[(tau_list >= min_tau) & (tau_list <= max_tau)] = momentum_method_1
    out[tau_list > max_tau] = np.mean(momentum_method_1[-5:])
    return out


def mc_csv_name(grain_type, BPASS_file):
    return "detailed MC {} {}csv".format(grain_type, BPASS_file.replace(".z", " z").replace("dat", ""))  # MC code.py:70


def save_mc_column(path, tau_list, time_slice, momentum_transfer):
    """Adds (or replaces) the column str(time_slice) of the MC table, keeping the other ages: what
    ``MC_data[str(time_slice)] = ...; MC_data.to_csv(..., index=False)`` does (MC code.py:1069-1071)."""
    header, rows = ["tau"], [[repr(float(t))] for t in tau_list]
    if os.path.exists(path):
        with open(path, newline="") as fh:
            rd = list(csv.reader(fh))
        if rd and len(rd) - 1 == len(tau_list):
            header, rows = rd[0], rd[1:]
        elif rd:
            raise ValueError(f"{path} has {len(rd) - 1} rows but tau_list has {len(tau_list)}")
    name = str(time_slice)
    vals = [repr(float(v)) for v in momentum_transfer]
    if name in header:
        j = header.index(name)
        for r, v in zip(rows, vals):
            r[j] = v
    else:
        header = header + [name]
        rows = [r + [v] for r, v in zip(rows, vals)]
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh, lineterminator="\n")  # pandas' to_csv ends lines with os.linesep's "\n", csv's default is "\r\n"
        w.writerow(header)
        w.writerows(rows)


def write_deposition_csv(path, tau_list, result, h_g=None):
    """One row per atmosphere: escape fraction, momentum fraction and its 1-sigma error, counters and the
    depth-resolved momentum / energy deposition (columns dep_p_k, dep_e_k, in units of the initial momentum
    / emitted energy).  Extension of the reference's single-column output."""
    s, c = result.sums, result.counts
    n = c[:, abi.CNT_PHOTONS].astype(float)
    nd = result.n_depth_bins
    emitted_energy = s[:, abi.SUM_INITIAL]  # with mu0 = 1 the initial momentum is sum(w*h), the emitted energy / c
    header = ["tau", "photons", "escape_fraction", "momentum_fraction", "momentum_sigma", "scatter_count",
              "absorbed_fraction", "uncounted_escape_fraction"]
    if h_g is not None:
        header.insert(1, "radius_over_h_g")
    header += [f"dep_p_{k}" for k in range(nd)] + [f"dep_e_{k}" for k in range(nd)]
    mf, sg = result.momentum_fraction(), result.momentum_fraction_sigma()
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(header)
        for a, tau in enumerate(tau_list):
            row = [repr(float(tau)), int(n[a]), (c[a, abi.CNT_ESCAPED] + c[a, abi.CNT_ESCAPED_UNCOUNTED]) / n[a],
                   mf[a], sg[a], int(c[a, abi.CNT_SCATTER]), c[a, abi.CNT_ABSORBED] / n[a],
                   c[a, abi.CNT_ESCAPED_UNCOUNTED] / n[a]]
            if h_g is not None:
                row.insert(1, float(np.sqrt(tau_list[0] / tau)))  # tau ~ (h_g / r)^2, MC code.py:924
            if nd:
                row += list(result.depth_momentum[a] / s[a, abi.SUM_INITIAL]) + list(result.depth_energy[a] / emitted_energy[a])
            w.writerow(row)


def runMC(BPASS_file, time_slice, num_photons, Range, Grain_data, momentum_method_1, photon_reduction=0,
          boundary="reflect", scatter="hg", save_MC=False, *, sed=None, tau_list=None, min_tau=10 ** -3, max_tau=100,
          wl_min=0.001, wl_max=10, grain_type="mix", rand_mu=False, abs_only=False, monochromatic=0, h=H_PLANCK,
          seed=DEFAULT_SEED, n_depth_bins=0, out_dir=".", engine=None, return_result=False, alpha=0.3,
          geometry="slab", shell_inner=0.0):
    """MC code.py:992.  Runs every atmosphere of tau_list inside [min_tau, max_tau], fills the caller's
    ``momentum_method_1`` in place and returns ``momentum_transfer`` over the whole tau_list.

    The spectrum is ``sed`` (luminosity per unit wavelength on ``Range``, in microns) or, if ``sed`` is
    None, column str(time_slice) of the BPASS text file ``BPASS_file`` (wavelengths in Angstrom, cut to
    [wl_min, wl_max] microns as at MC code.py:1017-1020; ``Range`` is then taken from the file).
    With ``save_MC`` the column str(time_slice) of 'detailed MC {grain_type} {BPASS name}csv' in
    ``out_dir`` is written (created if absent).

    Extensions: scatter='draine_exact' (with ``alpha``) and geometry='shell' with ``shell_inner`` = inner radius /
    thickness; for a shell the curve is the momentum deposited at the collisions over the initial momentum
    (SweepResult.deposited_fraction) instead of the slab's (initial - escaped + boundary) / initial."""
    tau_list = np.logspace(-3, 2, 100, base=10) if tau_list is None else np.asarray(tau_list, dtype=float)
    tau_list_calc = tau_list[(tau_list >= min_tau) & (tau_list <= max_tau)]   # MC code.py:74
    if sed is None:
        spec = read_bpass_spectra(BPASS_file)
        wl = spec["WL"] * 10 ** -4
        keep = (wl >= wl_min) & (wl <= wl_max)
        Range, sed = wl[keep], spec[str(time_slice)][keep]
    Range = np.asarray(Range, dtype=float)
    PDF, _ = tables.photon_pdf(Range, sed)
    counts = photon_reduction_counts(num_photons, photon_reduction, len(tau_list_calc))
    res = run_sweep(tau_list_calc, counts, boundary, scatter, PDF, Range, Grain_data, rand_mu=rand_mu,
                    abs_only=abs_only, monochromatic=monochromatic, h=h, seed=seed, n_depth_bins=n_depth_bins,
                    alpha=alpha, geometry=geometry, shell_inner=shell_inner, engine=engine)
    momentum_method_1[:len(tau_list_calc)] = (res.deposited_fraction() if geometry == "shell"
                                              else res.momentum_fraction())      # MC code.py:1062
    momentum_transfer = extrapolate(tau_list, tau_list_calc, np.asarray(momentum_method_1[:len(tau_list_calc)]),
                                    min_tau, max_tau)
    if save_MC:
        save_mc_column(os.path.join(out_dir, mc_csv_name(grain_type, os.path.basename(str(BPASS_file)))), tau_list,
                       time_slice, momentum_transfer)
    return (momentum_transfer, res) if return_result else momentum_transfer


def runMC_ages(time_slices, seds, num_photons, Range, Grain_data, photon_reduction=0, boundary="reflect", scatter="hg",
               save_MC=False, *, BPASS_file="", tau_list=None, min_tau=10 ** -3, max_tau=100, grain_type="mix",
               rand_mu=False, abs_only=False, h=H_PLANCK, seed=DEFAULT_SEED, out_dir=".", engine=None,
               return_result=False):
    """runMC (MC code.py:992-1073) for several stellar ages in ONE launch.

    The reference builds its 'detailed MC ...csv' table one age at a time: one runMC call -- one Python loop over the
    optical depths -- per column (MC code.py:1069-1071).  The ages share the grain tables and differ only in the
    spectrum the wavelengths are drawn from, so here every (age, optical depth) pair is one atmosphere of a single
    sweep: the emission law of age k is table k, its g / Albedo / Scale rows are the same rows again.  A sweep of
    100 optical depths x 1e5 photons is too short to fill the GPU; thirteen or fifty-one of them in one launch are not.

    time_slices: the column names (log ages); seds[k]: luminosity per unit wavelength of age k on `Range` (microns).
    Returns {str(time_slice): momentum_transfer over tau_list}; with save_MC the columns are written to the
    reference's CSV (created or extended, other columns kept).  Unlike successive runMC calls, which would give every
    age the same random streams, every (age, optical depth) pair has streams of its own."""
    eng = engine or default_engine()
    tau_list = np.logspace(-3, 2, 100, base=10) if tau_list is None else np.asarray(tau_list, dtype=float)
    tau_list_calc = tau_list[(tau_list >= min_tau) & (tau_list <= max_tau)]
    n_ages, n_calc = len(time_slices), len(tau_list_calc)
    if n_ages < 1 or len(seds) != n_ages:
        raise ValueError("one spectrum per time slice is needed")
    Range = np.asarray(Range, dtype=float)
    pdfs = np.stack([tables.photon_pdf(Range, sed)[0] for sed in seds])
    grain = {k: _column(Grain_data, k) for k in ("WL", "g", "Albedo", "Scale")}
    wl, g, alb, sc = tables.join_on_wavelength(Range, grain)
    missing = np.isnan(g)[None, :] & (pdfs > 0)
    if missing.any():
        raise ValueError(f"{int(missing.any(axis=0).sum())} wavelengths with non-zero emission probability have no entry in "
                         "Grain_data.WL (the reference would loop forever on their NaN constants)")
    pdfs[:, np.isnan(g)] = 0.0
    rows = lambda a: np.broadcast_to(np.nan_to_num(a), (n_ages, len(wl)))
    eng.set_tables(wl, rows(g), rows(alb), rows(sc), pdfs)
    counts = photon_reduction_counts(num_photons, photon_reduction, n_calc)
    prm = make_params(np.tile(tau_list_calc, n_ages), np.tile(counts, n_ages), boundary=boundary, scatter=scatter,
                      rand_mu=int(bool(rand_mu)), abs_only=int(bool(abs_only)), seed=seed, h=h,
                      table_of_atm=np.repeat(np.arange(n_ages), n_calc))
    res = eng.run(prm)
    method_1 = res.momentum_fraction().reshape(n_ages, n_calc)        # MC code.py:1062, one row per age
    out = {}
    for k, time_slice in enumerate(time_slices):
        out[str(time_slice)] = extrapolate(tau_list, tau_list_calc, method_1[k], min_tau, max_tau)
        if save_MC:
            save_mc_column(os.path.join(out_dir, mc_csv_name(grain_type, os.path.basename(str(BPASS_file)))), tau_list,
                           time_slice, out[str(time_slice)])
    return (out, res) if return_result else out
