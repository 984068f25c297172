"""The two modes that have no counterpart in the reference -- the exact Draine sampler and the spherical
shell geometry -- checked in the CPU restatement (oracle/rt_oracle.c) against closed-form answers.  The GPU
tests then compare the device with this restatement photon population by photon population.

Parity unpinned by construction: the reference has no working Draine sampler (MC code.py:1006) and no
spherical geometry, so the anchors here are the phase function's own density and moments, the slab limit
of the shell, and chord-length integrals.
"""
import numpy as np
import pytest
from scipy import integrate, stats


@pytest.fixture(scope="module")
def ora(c_oracle):
    return c_oracle


def first_scatter_cosines(ora, g, alpha, scatter, n=200000):
    """mu' of the first scattering of n photons: they start at mu = 1 (where the rotation is the identity),
    cannot escape (tau = 1e6), are never absorbed (albedo 1) and stop after one iteration."""
    prm = ora.make_params([1e6], n, order="gray", h=1.0, scatter=scatter, alpha=alpha, max_events=1, n_slabs=1)
    t, tr = ora.run(prm, [1.0], [g], [1.0], [1.0], trace=True)
    assert np.all(tr["n_scatter"] == 1)
    return tr["final_mu"]


def draine_density(mu, g, alpha):
    return (1 + alpha * mu * mu) / (1 + g * g - 2 * g * mu) ** 1.5


@pytest.mark.parametrize("g,alpha", [(0.6, 0.3), (-0.4, 1.0), (0.0, 1.0), (0.85, 0.3), (0.5, 0.0)])
def test_draine_exact_follows_the_phase_function(ora, g, alpha):
    mu = first_scatter_cosines(ora, g, alpha, "draine_exact")
    assert mu.min() >= -1 and mu.max() <= 1
    edges = np.linspace(-1, 1, 41)
    norm = integrate.quad(draine_density, -1, 1, args=(g, alpha))[0]
    expect = np.array([integrate.quad(draine_density, a, b, args=(g, alpha))[0] for a, b in zip(edges[:-1], edges[1:])])
    expect *= len(mu) / norm
    got = np.histogram(mu, edges)[0]
    chi2 = ((got - expect) ** 2 / expect).sum()
    assert chi2 < stats.chi2.ppf(0.9999, 39), chi2
    # first moment in closed form: <mu> = g (1 + alpha (3 + 2 g^2) / 5) / (1 + alpha (1 + 2 g^2) / 3)
    mean = g * (1 + alpha * (3 + 2 * g * g) / 5) / (1 + alpha * (1 + 2 * g * g) / 3)
    assert abs(mu.mean() - mean) < 4.5 * mu.std() / np.sqrt(len(mu))


def test_reference_sampler_is_clipped_where_exact_is_not(ora):
    """g = 0.85: the density exceeds 1 near mu = 1, where the reference's acceptance test pdf > y saturates
    (Numpy_random_walk.py:124-127), so its forward peak is too low; the exact sampler has no such limit."""
    g, alpha = 0.85, 0.3
    ref = first_scatter_cosines(ora, g, alpha, "draine", n=100000)
    exact = first_scatter_cosines(ora, g, alpha, "draine_exact", n=100000)
    norm = integrate.quad(draine_density, -1, 1, args=(g, alpha))[0]
    true_peak = integrate.quad(draine_density, 0.95, 1, args=(g, alpha))[0] / norm
    assert abs((exact > 0.95).mean() - true_peak) < 0.006
    assert (ref > 0.95).mean() < true_peak - 0.05


def test_shell_pure_absorption_chords(ora, rt):
    """abs_only + rand_mu: a photon leaves the inner surface at cosine mu ~ U(0,1) and is absorbed at its first
    interaction, so it escapes with probability exp(-L(mu)), L = the chord from r_in to r_out:
    L = -r_in mu + sqrt(r_in^2 mu^2 + r_out^2 - r_in^2)."""
    A = rt._abi
    n = 400000
    for x, tau in ((0.05, 1.0), (1.0, 0.7), (20.0, 2.0)):
        prm = ora.make_params([tau], n, geometry="shell", shell_inner=x, abs_only=1, rand_mu=1, n_slabs=1, order="gray", h=1.0)
        t = ora.run(prm, [1.0], [0.5], [0.5], [1.0])
        rin, rout = x * tau, (x + 1) * tau
        chord = lambda m: -rin * m + np.sqrt(rin * rin * m * m + rout * rout - rin * rin)
        p_esc = integrate.quad(lambda m: np.exp(-chord(m)), 0, 1)[0]
        got = t.counts[0, A.CNT_ESCAPED] / n
        assert abs(got - p_esc) < 4.5 * np.sqrt(p_esc * (1 - p_esc) / n), (x, tau, got, p_esc)
        assert t.counts[0, A.CNT_ESCAPED] + t.counts[0, A.CNT_ABSORBED] == n
        assert t.counts[0, A.CNT_BOUNDARY_HITS] == 0      # outward photons never meet the cavity


def test_shell_radial_photons_deposit_like_the_slab(ora, rt):
    """abs_only without rand_mu: every photon flies radially (mu stays 1), so the deposited fraction is the
    absorbed fraction 1 - exp(-tau), and each absorbed photon deposits exactly its own momentum."""
    A = rt._abi
    n, tau = 200000, 1.3
    prm = ora.make_params([tau], n, geometry="shell", shell_inner=0.5, abs_only=1, n_slabs=1)
    t = ora.run(prm, [0.7], [0.5], [0.5], [1.0])
    s, c = t.sums[0], t.counts[0]
    np.testing.assert_allclose(s[A.SUM_DUST_MOMENTUM] / s[A.SUM_INITIAL], c[A.CNT_ABSORBED] / n, rtol=1e-12)
    p = 1 - np.exp(-tau)
    assert abs(c[A.CNT_ABSORBED] / n - p) < 4.5 * np.sqrt(p * (1 - p) / n)


def test_shell_with_huge_radius_is_the_reflecting_slab(ora, rt):
    """r_in / thickness -> infinity: the same photons (same Philox streams) follow the slab's paths, the cavity
    crossing becomes the reflection at z = 0, and the deposited momentum becomes the slab's collision tally
    initial - escaped + 2 boundary (a reflection turns -|mu| into +|mu|; the reference's production estimator
    books one boundary term, its commented-out second estimator is this tally, MC code.py:773-776)."""
    A = rt._abi
    tau = np.array([0.3, 3.0, 30.0])
    n = 40000
    kw = dict(order="gray", h=1.0, n_slabs=1, n_mu_bins=10)
    flat = ora.run(ora.make_params(tau, n, **kw), [1.0], [0.5], [0.8], [1.0])
    ball = ora.run(ora.make_params(tau, n, geometry="shell", shell_inner=1e9, **kw), [1.0], [0.5], [0.8], [1.0])
    f = flat.sums[:, A.SUM_DUST_MOMENTUM] / flat.sums[:, A.SUM_INITIAL]
    b = ball.sums[:, A.SUM_DUST_MOMENTUM] / ball.sums[:, A.SUM_INITIAL]
    np.testing.assert_allclose(b, f, rtol=0, atol=2e-4)     # a few histories in 40000 may branch differently
    two_b = (flat.sums[:, A.SUM_INITIAL] - flat.sums[:, A.SUM_ESCAPED] + 2 * flat.sums[:, A.SUM_BOUNDARY]) / flat.sums[:, A.SUM_INITIAL]
    np.testing.assert_allclose(f, two_b, rtol=1e-9)
    assert np.all(np.abs(ball.counts[:, A.CNT_ESCAPED].astype(float) - flat.counts[:, A.CNT_ESCAPED]) <= 5)
    assert np.all(np.abs(ball.counts[:, A.CNT_BOUNDARY_HITS].astype(float) - flat.counts[:, A.CNT_BOUNDARY_HITS]) <= 5)


def test_shell_conservative_scattering_loses_no_photon(ora, rt):
    """albedo 1 with the cavity crossing: every photon leaves through the outer face."""
    A = rt._abi
    n, tau = 30000, np.array([2.0])
    kw = dict(scatter="iso", order="gray", h=1.0, n_slabs=1)
    t = ora.run(ora.make_params(tau, n, geometry="shell", shell_inner=0.1, **kw), [1.0], [0.5], [1.0], [1.0])
    assert t.counts[0, A.CNT_ESCAPED] == n and t.counts[0, A.CNT_GUARD] == 0 and t.counts[0, A.CNT_ABSORBED] == 0
    assert np.isfinite(t.sums).all()
    hist_total = t.mu_hist.sum()
    assert hist_total == n                      # every escape of a shell is logged
    assert t.mu_hist[0, :5].sum() < t.mu_hist[0, 5:].sum()   # radially beamed at the outer face


def test_shell_absorbing_and_reemitting_cavity(ora, rt):
    A = rt._abi
    n, tau = 50000, np.array([1.5])
    for boundary in ("absorb", "reemit"):
        prm = ora.make_params(tau, n, geometry="shell", shell_inner=2.0, boundary=boundary, scatter="iso", n_slabs=1,
                              rand_mu=(boundary == "reemit"))
        t = ora.run(prm, [1.0], [0.5], [0.9], [1.0])
        c = t.counts[0]
        if boundary == "absorb":
            assert c[A.CNT_BOUNDARY_ABSORBED] == c[A.CNT_BOUNDARY_HITS] > 0
            assert c[A.CNT_ESCAPED] + c[A.CNT_ABSORBED] + c[A.CNT_BOUNDARY_ABSORBED] == n
            assert t.sums[0, A.SUM_BOUNDARY] < 0          # photons enter the cavity moving inwards
        else:
            assert c[A.CNT_REEMITTED] == c[A.CNT_BOUNDARY_HITS] > 0
            assert c[A.CNT_ESCAPED] + c[A.CNT_ABSORBED] == n
