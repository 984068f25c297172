"""GPU tests of the reference-shaped host API (RunPhotons / runMC), of statistical parity with the
NumPy restatement of the reference loop (different generator, so agreement is statistical: the north
star's bar is 3 sigma per bin, reported here as chi-square over atmospheres and a KS distance on the
escape cosines), and of the distributed wrapper on one rank."""
import csv
import importlib
import os

import numpy as np
import pytest

from oracle import numpy_walk as nw

pytestmark = pytest.mark.gpu

mc = importlib.import_module("radiative-transfer_amd.mc")
tables = importlib.import_module("radiative-transfer_amd.tables")


def grain_and_sed(n=400):
    wl = np.round(0.09 + 1e-3 * np.arange(n), 4)
    grain = {"WL": wl, "g": 0.2 + 0.6 * np.linspace(0, 1, n) ** 2, "Albedo": 0.45 + 0.3 * np.sin(np.linspace(0, 3, n)) ** 2,
             "Scale": np.linspace(2.5, 0.4, n)}
    sed = tables.planck_sed(wl, 2.0e4)
    return wl, grain, sed


def numpy_reference(tau, n, wl, grain, pdf, rs, **kw):
    w = np.clip(pdf, 0, 1)
    cdf = np.cumsum(w)
    idx = np.minimum(np.searchsorted(cdf, rs.rand(n) * cdf[-1], side="right"), len(cdf) - 1)
    return nw.run_photons(tau, wl[idx], grain["g"][idx], grain["Albedo"][idx], grain["Scale"][idx], np.ones(n),
                          stream=nw.GlobalStream(rs), **kw)


def test_statistical_parity_with_numpy_reference(rt, engine):
    wl, grain, sed = grain_and_sed()
    pdf, _ = tables.photon_pdf(wl, sed)
    taus = np.logspace(-2, 1.5, 10)
    n_gpu, n_np = 400000, 40000
    res = mc.run_sweep(taus, n_gpu, "reflect", "hg", pdf, wl, grain, engine=engine, n_mu_bins=20)
    m_gpu, s_gpu = res.momentum_fraction(), res.momentum_fraction_sigma()
    rs = np.random.RandomState(2026)
    chi2, mu_np = 0.0, []
    for a, tau in enumerate(taus):
        r = numpy_reference(tau, n_np, wl, grain, pdf, rs)
        m_np = nw.momentum_fraction(r)
        s_np = s_gpu[a] * np.sqrt(n_gpu / n_np)          # same per-photon variance, fewer photons
        zscore = (m_gpu[a] - m_np) / np.hypot(s_gpu[a], s_np)
        assert abs(zscore) < 4.0, (tau, m_gpu[a], m_np, zscore)
        chi2 += zscore ** 2
        ev_gpu = res.scatter_count[a] / n_gpu
        assert abs(ev_gpu - r["scatter_count"] / n_np) < 5 * np.sqrt(max(ev_gpu, 1e-3) * 4 / n_np) + 1e-3
        mu_np.append(r["escaped_mu"])
    assert chi2 < 29.6  # chi-square(10 dof) 99.9th percentile
    # KS distance between the escape-cosine populations (second escape test), thickest atmospheres pooled
    a_sel = [a for a in range(len(taus)) if len(mu_np[a]) > 30]
    if a_sel:
        h_gpu = res.mu_hist[a_sel].sum(0).astype(float)
        h_gpu[-2] += h_gpu[-1]
        h_np = np.bincount(np.minimum(np.floor(20 * np.concatenate([mu_np[a] for a in a_sel])).astype(int), 19), minlength=20)
        n1, n2 = h_gpu[:-1].sum(), h_np.sum()
        d = np.max(np.abs(np.cumsum(h_gpu[:-1]) / n1 - np.cumsum(h_np) / n2))
        assert d < 1.95 * np.sqrt((n1 + n2) / (n1 * n2))  # alpha = 0.001


def test_run_photons_signature_and_gray(rt, engine, c_oracle):
    wl, grain, sed = grain_and_sed(50)
    pdf, _ = tables.photon_pdf(wl, sed)
    emu, B, E, I, count = mc.RunPhotons(2.0, 50000, "reflect", "hg", pdf, wl, grain, engine=engine)
    assert isinstance(count, int) and count > 0 and I > 0 and 0 < E < I
    assert len(emu) == int(emu.counts.sum()) and emu.n_bins == 10
    assert 0 < (I - E + B) / I < 2
    # monochromatic: joins on the rounded wavelength like CreatePhotons
    _, B1, E1, I1, c1 = mc.RunPhotons(2.0, 20000, "reflect", "hg", None, None, grain, monochromatic=0.1230000001, engine=engine)
    np.testing.assert_allclose(I1, 20000 * rt.H_PLANCK / 0.123, rtol=1e-12)
    with pytest.raises(ValueError):
        mc.RunPhotons(2.0, 100, "reflect", "hg", None, None, grain, monochromatic=5.0, engine=engine)
    # gray script signature: exact agreement with the oracle run on the same streams
    emu, B, E, I = mc.RunPhotonsGray(3.0, 30000, "reflect", 0.9, "hg", 0.5, engine=engine)
    ref = c_oracle.run(c_oracle.make_params(np.array([3.0]), 30000, order="gray", h=1.0, monochromatic=1.0),
                       [1.0], [0.5], [0.9], [1.0])
    assert I == 30000.0 and np.array_equal(emu.counts, ref.mu_hist[0].astype(np.int64))
    np.testing.assert_allclose([B, E], [ref.sums[0, 2], ref.sums[0, 1]], rtol=1e-12)


def test_runmc_estimator_extrapolation_and_csv(rt, engine, tmp_path):
    wl, grain, sed = grain_and_sed(200)
    tau_list = np.logspace(-4, 2.5, 40)
    calc = tau_list[(tau_list >= 1e-3) & (tau_list <= 100)]
    buf = np.zeros(len(calc))
    mt, res = mc.runMC("spectra-bin-imf135_300.z020.dat", 6.0, 20000, wl, grain, buf, photon_reduction=4, sed=sed,
                       tau_list=tau_list, save_MC=True, grain_type="mix", out_dir=str(tmp_path), engine=engine,
                       return_result=True, n_depth_bins=8)
    assert np.array_equal(buf, res.momentum_fraction())                      # caller's buffer filled in place
    assert list(res.counts[:, rt._abi.CNT_PHOTONS]) == list(np.linspace(20000, 5000, len(calc)).astype(int))
    mid = (tau_list >= 1e-3) & (tau_list <= 100)
    assert np.array_equal(mt[mid], buf)
    lo = tau_list < 1e-3
    np.testing.assert_allclose(mt[lo], tau_list[lo] * np.mean(buf[:10] / (1 - np.exp(-calc[:10]))))
    assert np.all(mt[tau_list > 100] == np.mean(buf[-5:]))
    # thin limit ~ tau * <Scale-weighted absorption>, thick plateau of order one
    assert 0.3 < mt[-1] < 2.0 and mt[0] < 1e-3
    path = tmp_path / "detailed MC mix spectra-bin-imf135_300 z020.csv"
    rows = list(csv.reader(open(path)))
    assert rows[0] == ["tau", "6.0"] and len(rows) == 41
    np.testing.assert_allclose([float(r[1]) for r in rows[1:]], mt)
    # a second age adds a column and keeps the first
    mc.runMC("spectra-bin-imf135_300.z020.dat", 6.5, 5000, wl, grain, buf, sed=sed, tau_list=tau_list, save_MC=True,
             out_dir=str(tmp_path), engine=engine)
    rows2 = list(csv.reader(open(path)))
    assert rows2[0] == ["tau", "6.0", "6.5"] and [r[1] for r in rows2[1:]] == [r[1] for r in rows[1:]]
    dep = tmp_path / "deposition.csv"
    mc.write_deposition_csv(str(dep), calc, res, h_g=10)
    rows3 = list(csv.reader(open(dep)))
    assert rows3[0][:3] == ["tau", "radius_over_h_g", "photons"] and len(rows3) == len(calc) + 1
    A = rt._abi
    np.testing.assert_allclose(res.depth_momentum.sum(1), res.sums[:, A.SUM_DUST_MOMENTUM], rtol=1e-9)


def test_bpass_text_reader(tmp_path):
    wl = np.arange(1, 2001, dtype=float)
    cols = np.column_stack([wl] + [wl * (i + 1) for i in range(51)])
    p = tmp_path / "spectra-test.dat"
    np.savetxt(p, cols)
    spec = mc.read_bpass_spectra(str(p))
    assert len(spec) == 52 and "6.0" in spec and "11.0" in spec and np.array_equal(spec["6.3"], wl * 4)


def test_distributed_wrapper_single_rank(rt, engine):
    D = importlib.import_module("radiative-transfer_amd.distributed")
    wl, grain, sed = grain_and_sed(64)
    pdf, _ = tables.photon_pdf(wl, sed)
    engine.set_tables(wl, grain["g"], grain["Albedo"], grain["Scale"], pdf)
    tau = np.array([0.5, 6.0])
    ds = D.DistributedSweep(engine)
    got = ds.run(tau, 30000, n_mu_bins=10)
    want = engine.run(rt.make_params(tau, 30000, n_mu_bins=10))
    assert np.array_equal(got.sums, want.sums) and np.array_equal(got.counts, want.counts)
    assert np.array_equal(got.mu_hist, want.mu_hist)
