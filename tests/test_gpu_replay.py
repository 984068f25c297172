"""GPU parity against the reference's own outputs: the HIP walk on pre-drawn uniforms (replay mode), and the
PRODUCTION sweeps wherever the reference was run on exactly the uniforms and photon rows a sweep produces
(cases gray_*, mcu_*, mct_* and the file walk_reemit_table.npz; see the tests below).

tests/golden/walk_replay.npz holds what the UNMODIFIED reference functions returned
(RunPhotons, MC code.py:620-779 and Numpy_random_walk.py:280-408) when every photon consumed
the stream xi(id, j) rebuilt here with oracle.philox; generator: oracle/make_golden.py.
Bar: discrete results (fates, draws consumed, scatter_count, escape histogram) exact; tallies
within 1e-12 relative (the tolerance the north star states for replay mode); per-photon final
(z, mu) within 1e-12 of their natural scale (mu is a cosine, scale 1; z scale max(1, |z|)): the
lab-frame rotation is ill-conditioned near |mu| = 1, so a last-bit difference between the device's
log/cos and NumPy's is amplified over successive scatterings and a purely relative bound on a
cosine that happens to be ~1e-3 is not meaningful.
"""
import os

import numpy as np
import pytest

from oracle import philox

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RTOL = 1e-12

G = np.load(os.path.join(GOLDEN, "walk_replay.npz"))
NAMES = [str(n) for n in G["names"]]


def case(name):
    return {k.split("/", 1)[1]: G[k] for k in G.files if k.startswith(name + "/")}


@pytest.mark.parametrize("name", NAMES)
def test_replay_matches_reference(rt, engine, name):
    c = case(name)
    n = int(c["n"])
    order = str(c["order"])
    max_draws = int(c["ref_draws"].max())
    xi = philox.walk_matrix(n, max_draws, 0, int(c["seed"]))
    prm = rt.make_params(float(c["tau"]), n, boundary=str(c["boundary"]), scatter=str(c["scatter"]), order=order,
                         abs_only=int(c["abs_only"]), h=(1.0 if order == "gray" else rt.H_PLANCK), n_mu_bins=10)
    res, tr = engine.replay(prm, xi, c["lam"], c["g"], c["albedo"], c["scale"], c["mu0"])
    A = rt._abi
    assert np.array_equal(tr["fate"], c["ref_fate"])
    assert np.array_equal(tr["n_draws"], c["ref_draws"])
    np.testing.assert_allclose(tr["final_mu"], c["ref_final_mu"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(tr["final_z"], c["ref_final_z"], rtol=RTOL, atol=1e-12)
    # and the bulk agrees far better than the bound
    assert np.median(np.abs(tr["final_mu"] - c["ref_final_mu"])) < 1e-15
    s = res.sums[0]
    np.testing.assert_allclose(s[A.SUM_INITIAL], float(c["ref_initial_momentum"]), rtol=RTOL)
    np.testing.assert_allclose(s[A.SUM_ESCAPED], float(c["ref_escaped_momentum"]), rtol=RTOL)
    np.testing.assert_allclose(s[A.SUM_BOUNDARY], float(c["ref_boundary_momentum"]), rtol=RTOL,
                               atol=1e-12 * abs(float(c["ref_initial_momentum"])))
    if int(c["ref_scatter_count"]) >= 0:
        assert int(res.counts[0, A.CNT_SCATTER]) == int(c["ref_scatter_count"])
    # escape-cosine population the reference returns as escaped_mu
    ref_hist = np.bincount(np.minimum(np.floor(10 * c["ref_escaped_mu"]).astype(int), 10), minlength=11)
    assert np.array_equal(res.mu_hist[0], ref_hist)
    assert int(res.counts[0, A.CNT_PHOTONS]) == n
    assert int(res.counts[0, A.CNT_GUARD]) == 0


def test_replay_guard_on_short_stream(rt, engine):
    """A photon whose stream runs out is reported as guard-killed, not looped on."""
    n = 256
    xi = philox.walk_matrix(n, 3, 0, 5)
    prm = rt.make_params(50.0, n, h=1.0)
    res, tr = engine.replay(prm, xi, np.ones(n), np.full(n, 0.5), np.ones(n), np.ones(n), np.ones(n))
    assert int(res.counts[0, rt._abi.CNT_GUARD]) == int((tr["fate"] == rt._abi.FATE_GUARD).sum()) > 0


UNIFORM = [n for n in NAMES if n.startswith("gray_") or n.startswith("mcu_")]


@pytest.mark.parametrize("name", UNIFORM)
def test_production_sweep_matches_reference_where_the_streams_coincide(rt, engine, name):
    """The cases of the golden file whose photons all carry the same constants -- the gray script's (scalar albedo and
    g, lambda = 1; Numpy_random_walk.RunPhotons) and the wavelength-dependent script's with one wavelength (MC
    code.py RunPhotons) -- were produced by the unmodified reference on the uniforms xi(id, j) of oracle.philox for
    atmosphere 0 and the case's seed: exactly what the PRODUCTION sweep draws for photon ids 0..n-1 of atmosphere 0
    under that seed on a one-wavelength table.  So the production kernels (not the replay kernel) must return the
    reference's own numbers, with no restatement in between -- the kernel with the case's flags fixed at compile
    time where there is one (with and without the per-photon cosines), and the run-time-flag kernel."""
    c = case(name)
    n, seed, order = int(c["n"]), int(c["seed"]), str(c["order"])
    gray = order == "gray"
    engine.set_tables([float(c["lam"][0])], [float(c["g"][0])], [float(c["albedo"][0])], [float(c["scale"][0])])
    A = rt._abi
    kw = dict(boundary=str(c["boundary"]), scatter=str(c["scatter"]), order=order, h=(1.0 if gray else rt.H_PLANCK), seed=seed,
              abs_only=int(c["abs_only"]), n_mu_bins=10)
    fates = np.bincount(c["ref_fate"], minlength=6)
    kernels = set()
    for per_photon, runtime in ((False, False), (True, False), (True, True)):
        res = engine.run(rt.make_params(float(c["tau"]), n, escaped_mu=per_photon, runtime_kernel=runtime, **kw))
        kernels.add(engine.last_kernel())
        s, cnt = res.sums[0], res.counts[0].astype(np.int64)
        np.testing.assert_allclose(s[A.SUM_INITIAL], float(c["ref_initial_momentum"]), rtol=RTOL)
        np.testing.assert_allclose(s[A.SUM_ESCAPED], float(c["ref_escaped_momentum"]), rtol=RTOL)
        np.testing.assert_allclose(s[A.SUM_BOUNDARY], float(c["ref_boundary_momentum"]), rtol=RTOL,
                                   atol=1e-12 * abs(float(c["ref_initial_momentum"])))
        assert cnt[A.CNT_ESCAPED] == fates[A.FATE_ESCAPED] and cnt[A.CNT_ESCAPED_UNCOUNTED] == fates[A.FATE_ESCAPED_UNCOUNTED]
        assert cnt[A.CNT_ABSORBED] == fates[A.FATE_ABSORBED] and cnt[A.CNT_BOUNDARY_ABSORBED] == fates[A.FATE_BOUNDARY]
        assert cnt[A.CNT_GUARD] == 0 and cnt[A.CNT_PHOTONS] == n
        if int(c["ref_scatter_count"]) >= 0:
            assert cnt[A.CNT_SCATTER] == int(c["ref_scatter_count"])
        ref_hist = np.bincount(np.minimum(np.floor(10 * c["ref_escaped_mu"]).astype(int), 10), minlength=11)
        assert np.array_equal(res.mu_hist[0], ref_hist)
        if per_photon:   # the reference's escaped_mu list itself (it appends in escape order: compare as sorted lists)
            np.testing.assert_allclose(np.sort(res.escaped_mu[0]), np.sort(c["ref_escaped_mu"]), rtol=0, atol=1e-12)
    assert "walk_kernel<runtime>" in kernels, kernels
    if name.startswith("mcu_"):   # these flag sets all have a kernel of their own
        assert len(kernels) >= 2, kernels


def test_the_headline_kernel_is_among_them(rt, engine):
    """mcu_reflect_hg_* run the instantiation that serves the headline sweep."""
    c = case("mcu_reflect_hg_tau3")
    engine.set_tables([float(c["lam"][0])], [float(c["g"][0])], [float(c["albedo"][0])], [float(c["scale"][0])])
    engine.run(rt.make_params(float(c["tau"]), int(c["n"]), seed=int(c["seed"]), n_mu_bins=10))
    assert engine.last_kernel() == "walk_kernel<mc,reflect,hg,plain>"
    assert len(UNIFORM) == 12


TABLE = [n for n in NAMES if n.startswith("mct_")]


@pytest.mark.parametrize("name", TABLE)
def test_production_sweep_on_a_table_matches_reference(rt, engine, name):
    """The same with a table of 64 wavelengths: the golden's photons carry the rows the production sweep emits for
    their ids under the case's seed (stored as emitted_index), and the unmodified reference walked them on the sweep's
    own uniforms.  The production sweep on that table -- emission through the alias table, record gather, walk --
    must return the reference's numbers; with rand_mu also the direction draw of the emission block."""
    c = case(name)
    n, seed = int(c["n"]), int(c["seed"])
    engine.set_tables(c["table_wl"], c["table_g"], c["table_albedo"], c["table_scale"], c["table_pdf"])
    A = rt._abi
    kw = dict(boundary=str(c["boundary"]), scatter=str(c["scatter"]), seed=seed, rand_mu=int(c["rand_mu"]), n_mu_bins=10)
    fates = np.bincount(c["ref_fate"], minlength=6)
    for per_photon in (False, True):
        res = engine.run(rt.make_params(float(c["tau"]), n, escaped_mu=per_photon, **kw))
        s, cnt = res.sums[0], res.counts[0].astype(np.int64)
        np.testing.assert_allclose(s[A.SUM_INITIAL], float(c["ref_initial_momentum"]), rtol=RTOL)
        np.testing.assert_allclose(s[A.SUM_ESCAPED], float(c["ref_escaped_momentum"]), rtol=RTOL)
        np.testing.assert_allclose(s[A.SUM_BOUNDARY], float(c["ref_boundary_momentum"]), rtol=RTOL,
                                   atol=1e-12 * abs(float(c["ref_initial_momentum"])))
        assert cnt[A.CNT_SCATTER] == int(c["ref_scatter_count"])
        assert cnt[A.CNT_ESCAPED] == fates[A.FATE_ESCAPED] and cnt[A.CNT_ESCAPED_UNCOUNTED] == fates[A.FATE_ESCAPED_UNCOUNTED]
        assert cnt[A.CNT_ABSORBED] == fates[A.FATE_ABSORBED] and cnt[A.CNT_BOUNDARY_ABSORBED] == fates[A.FATE_BOUNDARY]
        assert cnt[A.CNT_GUARD] == 0 and cnt[A.CNT_PHOTONS] == n
        ref_hist = np.bincount(np.minimum(np.floor(10 * c["ref_escaped_mu"]).astype(int), 10), minlength=11)
        assert np.array_equal(res.mu_hist[0], ref_hist)
        if per_photon:
            np.testing.assert_allclose(np.sort(res.escaped_mu[0]), np.sort(c["ref_escaped_mu"]), rtol=0, atol=1e-12)
    assert len(TABLE) == 5 and len(np.unique(c["emitted_index"])) > 40


GR = np.load(os.path.join(GOLDEN, "walk_reemit_table.npz"))


@pytest.mark.parametrize("name", [str(n) for n in GR["names"]])
def test_production_sweep_with_reemission_on_a_table_matches_reference(rt, engine, name):
    """boundary='reemit' on a table of 64 wavelengths: a photon that reaches the floor is replaced by a fresh
    CreatePhotons row (MC code.py:539-547) -- a new wavelength and, with rand_mu, a new direction, which the sweep
    draws from the photon's next emission block.  The golden holds the unmodified reference handed exactly those rows
    (oracle/make_golden.py), so the sweep's second emission inside a flight is pinned to the reference as well."""
    c = {k.split("/", 1)[1]: GR[k] for k in GR.files if k.startswith(name + "/")}
    n = int(c["n"])
    engine.set_tables(c["table_wl"], c["table_g"], c["table_albedo"], c["table_scale"], c["table_pdf"])
    A = rt._abi
    fates = np.bincount(c["ref_fate"], minlength=6)
    for per_photon in (False, True):
        res = engine.run(rt.make_params(float(c["tau"]), n, boundary="reemit", scatter=str(c["scatter"]), seed=int(c["seed"]),
                                        rand_mu=int(c["rand_mu"]), n_mu_bins=10, escaped_mu=per_photon))
        s, cnt = res.sums[0], res.counts[0].astype(np.int64)
        np.testing.assert_allclose(s[A.SUM_INITIAL], float(c["ref_initial_momentum"]), rtol=RTOL)
        np.testing.assert_allclose(s[A.SUM_ESCAPED], float(c["ref_escaped_momentum"]), rtol=RTOL)
        np.testing.assert_allclose(s[A.SUM_BOUNDARY], float(c["ref_boundary_momentum"]), rtol=RTOL,
                                   atol=1e-12 * abs(float(c["ref_initial_momentum"])))
        assert cnt[A.CNT_SCATTER] == int(c["ref_scatter_count"])
        assert cnt[A.CNT_ESCAPED] == fates[A.FATE_ESCAPED] and cnt[A.CNT_ESCAPED_UNCOUNTED] == fates[A.FATE_ESCAPED_UNCOUNTED]
        assert cnt[A.CNT_ABSORBED] == fates[A.FATE_ABSORBED] and cnt[A.CNT_GUARD] == 0 and cnt[A.CNT_PHOTONS] == n
        assert cnt[A.CNT_REEMITTED] > 50
        ref_hist = np.bincount(np.minimum(np.floor(10 * c["ref_escaped_mu"]).astype(int), 10), minlength=11)
        assert np.array_equal(res.mu_hist[0], ref_hist)
