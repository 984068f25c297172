"""Generates tests/golden/runmc_chain.npz: the reference's OWN chain runMC -> RunPhotons -> CreatePhotons ->
AcceptReject (MC code.py:992-1073, :620-779, :371-412, :239-268), executed unmodified.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (it needs the reference checkout at
/root/reference); tests read the committed .npz, never the reference.

The reference's ``def``s are loaded by oracle/ref_loader.py.  What runMC reads besides its arguments is supplied from
outside, nothing inside it is touched:

* ``load.model_output(BPASS_file)`` (the absent third-party BPASS reader, MC code.py:1013) is a data provider that
  returns a synthetic spectra frame: wavelengths in Angstrom, some outside [wl_min, wl_max] so that the reference's
  own cut (:1020) leaves a frame whose index does not start at 0, one luminosity column per age;
* the module globals ``tau_list``, ``tau_list_calc``, ``MC_data``, ``grain_type`` (:70-75, :41), set as the script's
  top level would; ``min_tau``, ``max_tau``, ``wl_min``, ``wl_max``, ``h``, ``c`` are the reference's own values;
* ``np.random.seed`` before each run (the reference never seeds).

``RunPhotons`` and ``CreatePhotons`` are wrapped only to LOOK at their arguments and results (the PDF runMC built,
the photon count of each atmosphere, the tallies, the wavelengths emitted).  ``Grain_data`` lacks one wavelength of
the range, placed where the emission law is zero, and holds rows in shuffled order plus rows outside the range, so
that the pandas left-join (:403-408) is exercised as the reference performs it; a separate direct call of
``CreatePhotons`` with a law that does emit the missing wavelength records the NaN rows the join then produces.

Usage:  python oracle/make_golden_runmc.py
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
BPASS_FILE = "spectra-bin-imf135_300.z020.dat"


def synthetic_spectra():
    """A frame shaped like hoki's model_output of a spectra file: WL in Angstrom + log-age columns."""
    wl_a = np.concatenate([[2.0, 5.0], 500.0 + 100.0 * np.arange(96), [120000.0, 150000.0]])
    wl_cm = wl_a * 1e-8
    cols = {"WL": wl_a}
    for age, temp in (("6.0", 3.0e4), ("6.3", 2.0e4)):
        x = 6.6261e-27 * 2.99792458e10 / (wl_cm * 1.380649e-16 * temp)
        with np.errstate(over="ignore"):
            lum = 1.0 / wl_cm ** 5 / np.expm1(x) * (1.0 + 0.3 * np.sin(np.arange(len(wl_a)) * 0.7))
        lum[wl_a < 912.0] = 0.0
        cols[age] = lum / lum.max() * 1e30
    return pd.DataFrame(cols)


def grain_frame(range_um, missing, rng):
    """g / Albedo / Scale per wavelength with `missing` left out, rows outside the range added, order shuffled."""
    wl = np.round(range_um, 4)
    x = (wl - wl.min()) / (wl.max() - wl.min())
    frame = pd.DataFrame({"WL": wl, "g": 0.75 * np.exp(-3.0 * x) * np.where(np.arange(len(wl)) % 11 == 5, -1.0, 1.0) + 1e-6,
                          "Albedo": 0.35 + 0.6 * x * (1.0 - 0.5 * x), "Scale": 2.5 / (1.0 + 4.0 * x) + 0.1,
                          "Kappa_RP": 1e4 * (1 + x)})
    frame = frame[frame.WL != np.round(missing, 4)]
    extra = pd.DataFrame({"WL": [0.0123, 3.3333, 9.9999], "g": [0.1, 0.2, 0.3], "Albedo": [0.5, 0.5, 0.5],
                          "Scale": [1.0, 1.0, 1.0], "Kappa_RP": [1.0, 1.0, 1.0]})
    frame = pd.concat([frame, extra], ignore_index=True)
    return frame.iloc[rng.permutation(len(frame))].reset_index(drop=True)


class _Provider:
    """Stands in for ``hoki.load``: runMC only calls model_output(file) on it."""

    def __init__(self, frame):
        self.frame, self.calls = frame, []

    def model_output(self, name):
        self.calls.append(name)
        return self.frame.copy()


def run_chain(case, spectra, out):
    mod = ref_loader.load_mc_code()
    rng = np.random.default_rng(case["seed"])
    wl_um_all = spectra.WL.to_numpy() * 10 ** -4
    keep = (wl_um_all >= mod.wl_min) & (wl_um_all <= mod.wl_max)
    range_um = np.round(wl_um_all[keep], 4)          # `Range` as the reference's callers build it (BPASS_data_r.WL)
    missing = range_um[2]                            # 0.07 micron: below the Lyman limit, never emitted
    grain = grain_frame(range_um, missing, rng)
    tau_list = np.logspace(-4, 3, 22)
    mod.tau_list = tau_list
    mod.tau_list_calc = tau_list[(tau_list >= mod.min_tau) & (tau_list <= mod.max_tau)]
    mod.grain_type = "mix"
    mod.MC_data = pd.DataFrame({"tau": tau_list, "6.0": np.linspace(0.0, 1.5, len(tau_list))})
    mod.load = _Provider(spectra)
    mod.rand_mu = case["rand_mu"]
    seen = dict(pdf=None, n=[], tallies=[], mu_hist=[], emitted=np.zeros(len(range_um), dtype=np.int64), nan_rows=0,
                mu0_sum=0.0, mu0_n=0)
    run0, create0 = mod.RunPhotons, mod.CreatePhotons

    def create(num, init, rand_mu, PDF, Range, Grain_data):
        ps = create0(num, init, rand_mu, PDF, Range, Grain_data)
        idx = np.searchsorted(range_um, ps[:, 5])
        assert np.array_equal(range_um[idx], ps[:, 5])
        seen["emitted"] += np.bincount(idx, minlength=len(range_um))
        seen["nan_rows"] += int(np.isnan(ps[:, 6]).sum())
        seen["mu0_sum"] += float(ps[:, 4].sum())
        seen["mu0_n"] += len(ps)
        return ps

    def run(tau_atm, num_photons, boundary, scatter, PDF, Range, Grain_data):
        seen["pdf"] = np.array(PDF, dtype=float)
        seen["n"].append(int(num_photons))
        emu, B, E, I, count = run0(tau_atm, num_photons, boundary, scatter, PDF, Range, Grain_data)
        seen["tallies"].append([float(I), float(E), float(B), float(count), float(len(emu))])
        k = np.clip(np.floor(10 * np.asarray(emu, dtype=float)).astype(int), 0, 10)
        seen["mu_hist"].append(np.bincount(k, minlength=11))
        return emu, B, E, I, count

    mod.RunPhotons, mod.CreatePhotons = run, create
    m1 = np.zeros(len(mod.tau_list_calc))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            np.random.seed(case["seed"])
            with contextlib.redirect_stdout(io.StringIO()):
                mt = mod.runMC(BPASS_FILE, case["time_slice"], case["num_photons"], range_um, grain, m1,
                               photon_reduction=case["photon_reduction"], boundary=case["boundary"],
                               scatter=case["scatter"], save_MC=case["save_MC"])
            files = sorted(os.listdir(tmp))
            csv_name = files[0] if files else ""
            csv_bytes = open(os.path.join(tmp, csv_name), "rb").read() if files else b""
        finally:
            os.chdir(cwd)
    assert len(files) == (1 if case["save_MC"] else 0), files
    p = case["name"] + "/"
    out[p + "spectra_wl_angstrom"] = spectra.WL.to_numpy()
    out[p + "spectra_column"] = spectra[str(case["time_slice"])].to_numpy()
    for k in ("time_slice", "num_photons", "photon_reduction", "seed", "rand_mu"):
        out[p + k] = np.asarray(case[k])
    for k in ("boundary", "scatter"):
        out[p + k] = np.array(case[k])
    out[p + "bpass_file"] = np.array(BPASS_FILE)
    out[p + "range_um"] = range_um
    out[p + "missing_wl"] = np.asarray(missing)
    for col in ("WL", "g", "Albedo", "Scale"):
        out[p + "grain_" + col] = grain[col].to_numpy()
    out[p + "tau_list"] = tau_list
    out[p + "tau_list_calc"] = np.asarray(mod.tau_list_calc)
    out[p + "min_tau"], out[p + "max_tau"] = np.asarray(mod.min_tau), np.asarray(mod.max_tau)
    out[p + "wl_min"], out[p + "wl_max"] = np.asarray(mod.wl_min), np.asarray(mod.wl_max)
    out[p + "h"] = np.asarray(mod.h)
    out[p + "ref_pdf"] = seen["pdf"]
    out[p + "ref_photons_per_atm"] = np.asarray(seen["n"], dtype=np.int64)
    out[p + "ref_tallies"] = np.asarray(seen["tallies"])        # [atm][initial, escaped, boundary, scatter_count, len(escaped_mu)]
    out[p + "ref_escaped_mu_hist"] = np.asarray(seen["mu_hist"], dtype=np.int64)
    out[p + "ref_momentum_method_1"] = m1
    out[p + "ref_momentum_transfer"] = np.asarray(mt, dtype=float)
    out[p + "ref_emitted_hist"] = seen["emitted"]
    out[p + "ref_nan_rows"] = np.asarray(seen["nan_rows"])
    out[p + "ref_mu0_mean"] = np.asarray(seen["mu0_sum"] / seen["mu0_n"])
    out[p + "ref_csv_name"] = np.array(csv_name)
    out[p + "ref_csv_bytes"] = np.frombuffer(csv_bytes, dtype=np.uint8)
    out[p + "mc_data_before_tau"] = tau_list
    out[p + "mc_data_before_6.0"] = np.linspace(0.0, 1.5, len(tau_list))
    print(case["name"], "atmospheres", len(m1), "photons", seen["n"][0], "->", seen["n"][-1], "emitted", int(seen["emitted"].sum()),
          "nan rows", seen["nan_rows"], "csv", repr(csv_name), len(csv_bytes), "bytes")
    return mod, range_um, grain, missing


def join_case(out, spectra):
    """CreatePhotons called directly with a law that emits the wavelength Grain_data lacks: the left-join's NaN rows."""
    mod = ref_loader.load_mc_code()
    wl_um_all = spectra.WL.to_numpy() * 10 ** -4
    range_um = np.round(wl_um_all[(wl_um_all >= mod.wl_min) & (wl_um_all <= mod.wl_max)], 4)
    rng = np.random.default_rng(77)
    missing = range_um[40]
    grain = grain_frame(range_um, missing, rng)
    pdf = np.linspace(0.2, 1.0, len(range_um)) / 8.0
    pdf[7] = 0.0
    pdf[13] = 1.7          # a weight above 1: accepted with probability 1 (the comparison at MC code.py:264)
    pdf[20] = -0.5         # a negative weight: never accepted
    np.random.seed(4242)
    n = 400000
    ps = mod.CreatePhotons(n, mod.initial_state, 1, pdf, range_um, grain)
    idx = np.searchsorted(range_um, ps[:, 5])
    assert np.array_equal(range_um[idx], ps[:, 5])
    nan = np.isnan(ps[:, 6])
    assert np.array_equal(nan, np.isnan(ps[:, 7])) and np.array_equal(nan, np.isnan(ps[:, 8]))
    # the value each emitted wavelength carries (one value per wavelength: the join is a function of WL)
    cols = np.full((3, len(range_um)), np.nan)
    for k, c in enumerate((6, 7, 8)):
        for i in np.unique(idx):
            v = np.unique(ps[idx == i, c])
            assert len(v) == 1 or np.all(np.isnan(v)), (i, v)
            cols[k, i] = v[0]
    p = "join/"
    out[p + "range_um"] = range_um
    out[p + "pdf"] = pdf
    out[p + "missing_wl"] = np.asarray(missing)
    for col in ("WL", "g", "Albedo", "Scale"):
        out[p + "grain_" + col] = grain[col].to_numpy()
    out[p + "n"] = np.asarray(n)
    out[p + "ref_emitted_hist"] = np.bincount(idx, minlength=len(range_um))
    out[p + "ref_nan_hist"] = np.bincount(idx[nan], minlength=len(range_um))
    out[p + "ref_joined_g"], out[p + "ref_joined_albedo"], out[p + "ref_joined_scale"] = cols
    out[p + "ref_mu0_mean"] = np.asarray(ps[:, 4].mean())
    out[p + "ref_mu0_hist"] = np.bincount(np.floor(ps[:, 4] * 20).astype(int), minlength=20)
    print("join: emitted", n, "NaN rows", int(nan.sum()), "at", range_um[np.unique(idx[nan])])


CASES = [
    dict(name="reflect_hg_red4", time_slice=6.3, num_photons=1000000, photon_reduction=4, boundary="reflect", scatter="hg",
         save_MC=True, seed=20260, rand_mu=False),
    dict(name="reemit_hg_randmu", time_slice=6.0, num_photons=60000, photon_reduction=0, boundary="reemit", scatter="hg",
         save_MC=False, seed=20261, rand_mu=True),
    dict(name="absorb_iso", time_slice=6.3, num_photons=60000, photon_reduction=0, boundary="absorb", scatter="iso",
         save_MC=True, seed=20262, rand_mu=False),
]


def main():
    spectra = synthetic_spectra()
    out = {"names": np.array([c["name"] for c in CASES])}
    for case in CASES:
        run_chain(case, spectra, out)
    join_case(out, spectra)
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, "runmc_chain.npz"), **out)
    print("wrote", os.path.join(GOLDEN, "runmc_chain.npz"), os.path.getsize(os.path.join(GOLDEN, "runmc_chain.npz")), "bytes")


if __name__ == "__main__":
    main()
