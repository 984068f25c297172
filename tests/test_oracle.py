"""CPU tests: the oracle against the reference's golden vectors (no GPU, no reference checkout).

walk_replay.npz / walk_seeded.npz were produced by the unmodified reference functions
(oracle/make_golden.py).  These tests pin the two restatements the GPU tests and the CPU baseline
rely on: oracle/rt_oracle.c (scalar C) and oracle/numpy_walk.py (vectorised NumPy).
"""
import os

import numpy as np
import pytest

from oracle import numpy_walk as nw
from oracle import philox

from conftest import GOLDEN

REPLAY = np.load(os.path.join(GOLDEN, "walk_replay.npz"))
SEEDED = np.load(os.path.join(GOLDEN, "walk_seeded.npz"))


def case(z, name):
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def test_philox_known_answers(c_oracle):
    # Random123 known-answer vectors for philox4x32-10
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert c_oracle.philox(ctr, key) == want
        got = philox.philox4x32_10(*ctr, *key)
        assert [int(x) for x in got] == want


def test_uniform_is_52_bit_and_below_one(c_oracle):
    u = philox.walk_matrix(1000, 8, 3, 12345)
    assert u.min() >= 0.0 and u.max() < 1.0
    assert np.all(u * 2.0 ** 52 == np.floor(u * 2.0 ** 52))
    assert np.any(u * 2.0 ** 51 != np.floor(u * 2.0 ** 51))  # the last bit is used
    # extremes of the word pair
    assert philox.u52(0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -52 and philox.u52(0, 0xFFF) == 0.0
    assert philox.u52(0x80000000, 0) == 0.5 and philox.u52(0, 0x1000) == 2.0 ** -52


@pytest.mark.parametrize("name", [str(n) for n in REPLAY["names"]])
def test_c_oracle_replay_matches_reference(c_oracle, name):
    c = case(REPLAY, name)
    n, order = int(c["n"]), str(c["order"])
    xi = philox.walk_matrix(n, int(c["ref_draws"].max()), 0, int(c["seed"]))
    prm = c_oracle.make_params(float(c["tau"]), n, boundary=str(c["boundary"]), scatter=str(c["scatter"]),
                               order=order, abs_only=int(c["abs_only"]), h=(1.0 if order == "gray" else 6.6261e-27))
    t, tr = c_oracle.replay(prm, xi, c["lam"], c["g"], c["albedo"], c["scale"], c["mu0"])
    A = c_oracle.abi
    assert np.array_equal(tr["fate"], c["ref_fate"])
    assert np.array_equal(tr["n_draws"], c["ref_draws"])
    np.testing.assert_allclose(tr["final_mu"], c["ref_final_mu"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(tr["final_z"], c["ref_final_z"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(t.sums[0, A.SUM_INITIAL], float(c["ref_initial_momentum"]), rtol=1e-12)
    np.testing.assert_allclose(t.sums[0, A.SUM_ESCAPED], float(c["ref_escaped_momentum"]), rtol=1e-12)
    np.testing.assert_allclose(t.sums[0, A.SUM_BOUNDARY], float(c["ref_boundary_momentum"]), rtol=1e-12,
                               atol=1e-12 * abs(float(c["ref_initial_momentum"])))
    if int(c["ref_scatter_count"]) >= 0:
        assert int(t.counts[0, A.CNT_SCATTER]) == int(c["ref_scatter_count"])
    ref_hist = np.bincount(np.minimum(np.floor(10 * c["ref_escaped_mu"]).astype(int), 10), minlength=11)
    assert np.array_equal(t.mu_hist[0], ref_hist)


@pytest.mark.parametrize("name", [str(n) for n in REPLAY["names"]])
def test_numpy_walk_replay_matches_reference(name):
    c = case(REPLAY, name)
    n, order = int(c["n"]), str(c["order"])
    seed = int(c["seed"])
    stream = nw.PerPhotonStream(n, lambda ids, j: philox.walk_uniform(ids, j, 0, seed))
    r = nw.run_photons(float(c["tau"]), c["lam"], c["g"], c["albedo"], c["scale"], c["mu0"],
                       boundary=str(c["boundary"]), scatter=str(c["scatter"]), order=order,
                       stream=stream, h=(1 if order == "gray" else nw.H_PLANCK), abs_only=bool(c["abs_only"]), trace=True)
    # same NumPy operations on the same numbers: bitwise
    assert np.array_equal(r["fate"], c["ref_fate"])
    assert np.array_equal(stream.ctr, c["ref_draws"])
    assert np.array_equal(r["final_mu"], c["ref_final_mu"])
    assert np.array_equal(r["escaped_mu"], c["ref_escaped_mu"])
    assert r["boundary_momentum"] == float(c["ref_boundary_momentum"])
    assert r["escaped_momentum"] == float(c["ref_escaped_momentum"])
    assert r["initial_momentum"] == float(c["ref_initial_momentum"])
    if int(c["ref_scatter_count"]) >= 0:
        assert r["scatter_count"] == int(c["ref_scatter_count"])


@pytest.mark.parametrize("name", [str(n) for n in SEEDED["names"]])
def test_numpy_walk_seeded_matches_reference(name):
    """Same MT19937 seed, same draw order -> bitwise the reference's return values."""
    c = case(SEEDED, name)
    n, order = int(c["n"]), str(c["order"])
    rs = np.random.RandomState(int(c["seed"]))
    full = lambda v: np.full(n, float(v))
    r = nw.run_photons(float(c["tau"]), full(c["lam"]), full(c["g"]), full(c["albedo"]), full(c["scale"]), np.ones(n),
                       boundary=str(c["boundary"]), scatter=str(c["scatter"]), order=order,
                       stream=nw.GlobalStream(rs), h=(1 if order == "gray" else nw.H_PLANCK), alpha=float(c["alpha"]))
    assert np.array_equal(r["escaped_mu"], c["ref_escaped_mu"])
    assert r["boundary_momentum"] == float(c["ref_boundary_momentum"])
    assert r["escaped_momentum"] == float(c["ref_escaped_momentum"])
    assert r["initial_momentum"] == float(c["ref_initial_momentum"])
    if int(c["ref_scatter_count"]) >= 0:
        assert r["scatter_count"] == int(c["ref_scatter_count"])


def test_c_oracle_production_statistics(c_oracle):
    """Philox-driven C oracle vs MT-driven NumPy restatement: same physics within 4 sigma."""
    n = 40000
    tau = np.array([0.5, 3.0])
    t = c_oracle.run(c_oracle.make_params(tau, n, monochromatic=0.5), [0.5], [0.6], [0.7], [1.3])
    A = c_oracle.abi
    for a, ta in enumerate(tau):
        r = nw.run_photons(ta, np.full(n, 0.5), np.full(n, 0.6), np.full(n, 0.7), np.full(n, 1.3), np.ones(n),
                           stream=nw.GlobalStream(np.random.RandomState(3 + a)))
        m_np = nw.momentum_fraction(r)
        s = t.sums[a]
        m_c = (s[A.SUM_INITIAL] - s[A.SUM_ESCAPED] + s[A.SUM_BOUNDARY]) / s[A.SUM_INITIAL]
        sigma = np.sqrt(2.0 / n)  # per-photon transfer is O(1); generous bound
        assert abs(m_np - m_c) < 4 * sigma
        ev_c, ev_np = float(t.counts[a, A.CNT_SCATTER]), float(r["scatter_count"])
        assert abs(ev_c - ev_np) < 6 * np.sqrt(ev_np * 3)


def test_emission_index_law(c_oracle):
    """P(index) proportional to min(max(pdf, 0), 1): the law of AcceptReject (MC code.py:239-268)."""
    pdf = np.array([0.0, 0.2, 0.0, np.nan, 0.5, 2.0, -1.0, 1e-9])  # 2.0 is clipped to 1 (acceptance probability)
    w = np.array([0, .2, 0, 0, .5, 1., 0, 1e-9])
    w = w / w.sum()
    us = (np.arange(400000) + 0.5) / 400000
    idx = c_oracle.sample_indices(pdf, us)
    freq = np.bincount(idx, minlength=len(pdf)) / len(us)
    np.testing.assert_allclose(freq, w, atol=1e-5)
    assert freq[0] == freq[2] == freq[3] == freq[6] == 0          # zero weight is never emitted
    assert c_oracle.sample_index(pdf, np.nextafter(1.0, 0.0)) in (1, 4, 5, 7)
    # a broad, very uneven law (4 decades) is reproduced to the resolution of the stratified sample
    rng = np.random.default_rng(3)
    pdf = 10.0 ** rng.uniform(-9, -5, 5000)
    idx = c_oracle.sample_indices(pdf, us)
    freq = np.bincount(idx, minlength=5000) / len(us)
    assert np.abs(freq - pdf / pdf.sum()).max() < 3.0 / len(us)


def test_known_answer_two_stream(c_oracle):
    """Isotropic conservative slab, photons lost below z=0: transmitted fraction ~ 1/(1+tau/2)
    (Old code/1D test.py:105-106; the Monte Carlo sits a few per cent below at tau=10)."""
    n = 60000
    tau = np.array([1.0, 10.0])
    t = c_oracle.run(c_oracle.make_params(tau, n, order="gray", h=1.0, boundary="absorb", scatter="iso",
                                          monochromatic=1.0, rand_mu=0), [1.0], [0.0], [1.0], [1.0])
    frac = t.counts[:, c_oracle.abi.CNT_ESCAPED] / n
    # photons start travelling straight up (mu=1), which transmits more than the diffuse two-stream value
    assert 0.5 < frac[0] < 0.9
    assert abs(frac[1] - 1 / (1 + 10 / 2)) < 0.06


def test_known_answer_emergent_intensity(c_oracle):
    """tau=10 isotropic conservative slab with the re-emitting boundary: emergent intensity vs angle
    (reference table intensity.dat / Numpy_random_walk.py:448-449).  The table is itself a Monte Carlo result: column 2
    is the fraction of photons per cosine bin, column 3 its error, column 4 the intensity; compared within the combined
    errors (2e5 photons here; tests/test_gpu_known_answers.py does it at 2e7)."""
    import os
    from conftest import GOLDEN
    theta, frac, err, want = np.loadtxt(os.path.join(GOLDEN, "intensity.dat")).T
    assert np.allclose(want, [1.224, 1.145, 1.065, 0.9885, 0.9117, 0.8375, 0.7508, 0.6769, 0.5786, 0.5135])  # Numpy_random_walk.py:449
    n = 200000
    t = c_oracle.run(c_oracle.make_params(np.array([10.0]), n, order="gray", h=1.0, boundary="reemit", scatter="iso",
                                          monochromatic=1.0, rand_mu=1, n_mu_bins=10), [1.0], [0.0], [1.0], [1.0])
    counts = t.mu_hist[0, :10].astype(float)
    counts[9] += t.mu_hist[0, 10]
    mids = (np.arange(10) + 0.5) / 10
    order = np.argsort(-mids)  # table runs from small theta (mu near 1) to large
    assert np.all(np.abs(np.degrees(np.arccos(mids[order])) - theta) < 0.25)
    got_frac = counts[order] / n
    intensity = got_frac * 10 / (2 * mids[order])  # Numpy_random_walk.py:446
    z = (got_frac - frac) / np.hypot(err, np.sqrt(got_frac / n))
    assert np.max(np.abs(z)) < 4.0 and np.sum(z * z) < 29.6, z
    np.testing.assert_allclose(intensity[:6], want[:6], rtol=0.03)


@pytest.mark.parametrize("rng", ["philox32x4", "philox40"])
def test_packed_stream_of_the_oracle(c_oracle, rng):
    """The one-block streams in the oracle: same physics as the default stream (statistically) on a small sweep."""
    wl = np.array([0.5]); g = np.array([0.6]); alb = np.array([0.7]); sc = np.array([1.0])
    tau = np.array([0.1, 1.0, 5.0])
    a = c_oracle.run(c_oracle.make_params(tau, 300000), wl, g, alb, sc, None)
    b = c_oracle.run(c_oracle.make_params(tau, 300000, rng=rng), wl, g, alb, sc, None)
    assert not np.array_equal(a.counts, b.counts)
    fa = (a.sums[:, 0] - a.sums[:, 1] + a.sums[:, 2]) / a.sums[:, 0]
    fb = (b.sums[:, 0] - b.sums[:, 1] + b.sums[:, 2]) / b.sums[:, 0]
    np.testing.assert_allclose(fa, fb, rtol=0.03)


ONE_BLOCK = {"philox32x4": np.load(os.path.join(os.path.dirname(__file__), "golden", "walk_replay_packed.npz")),
             "philox40": np.load(os.path.join(os.path.dirname(__file__), "golden", "walk_replay_philox40.npz"))}


@pytest.mark.parametrize("rng,name", [(r, str(n)) for r, f in ONE_BLOCK.items() for n in f["names"]])
def test_c_oracle_packed_production_matches_reference(c_oracle, rng, name):
    """The one-block streams (rng='philox32x4', 'philox40') of the C restatement in PRODUCTION mode against the
    unmodified reference run on those streams' uniforms (oracle/philox.walk_uniform_packed / walk_uniform_p40): the
    golden's photons carry the constants the sweep gives them, so the sweep must return the reference's own numbers."""
    PACKED = ONE_BLOCK[rng]
    c = {k.split("/", 1)[1]: PACKED[k] for k in PACKED.files if k.startswith(name + "/")}
    n, order = int(c["n"]), str(c["order"])
    if "table_wl" in c:
        tab = (c["table_wl"], c["table_g"], c["table_albedo"], c["table_scale"], c["table_pdf"])
    else:
        tab = ([float(c["lam"][0])], [float(c["g"][0])], [float(c["albedo"][0])], [float(c["scale"][0])])
    prm = c_oracle.make_params(float(c["tau"]), n, boundary=str(c["boundary"]), scatter=str(c["scatter"]), order=order,
                               h=(1.0 if order == "gray" else 6.6261e-27), seed=int(c["seed"]), rng=rng, n_mu_bins=10)
    res = c_oracle.run(prm, *tab)
    fates = np.bincount(c["ref_fate"], minlength=6)
    cnt = res.counts[0].astype(np.int64)
    assert cnt[1] == fates[1] and cnt[2] == fates[2] and cnt[3] == fates[3] and cnt[4] == fates[4] and cnt[6] == 0
    if int(c["ref_scatter_count"]) >= 0:
        assert cnt[0] == int(c["ref_scatter_count"])
    np.testing.assert_allclose(res.sums[0, 0], float(c["ref_initial_momentum"]), rtol=1e-12)
    np.testing.assert_allclose(res.sums[0, 1], float(c["ref_escaped_momentum"]), rtol=1e-12)
    np.testing.assert_allclose(res.sums[0, 2], float(c["ref_boundary_momentum"]), rtol=1e-12,
                               atol=1e-12 * abs(float(c["ref_initial_momentum"])))


REEMIT_TABLE = np.load(os.path.join(os.path.dirname(__file__), "golden", "walk_reemit_table.npz"))


@pytest.mark.parametrize("name", [str(n) for n in REEMIT_TABLE["names"]])
def test_c_oracle_reemission_on_a_table_matches_reference(c_oracle, name):
    """boundary='reemit' with a table: the re-emitted photon's new row comes from its next emission block; the golden is
    the unmodified reference handed those rows (oracle/make_golden.py)."""
    c = {k.split("/", 1)[1]: REEMIT_TABLE[k] for k in REEMIT_TABLE.files if k.startswith(name + "/")}
    prm = c_oracle.make_params(float(c["tau"]), int(c["n"]), boundary="reemit", scatter=str(c["scatter"]), seed=int(c["seed"]),
                               rand_mu=int(c["rand_mu"]), n_mu_bins=10)
    res = c_oracle.run(prm, c["table_wl"], c["table_g"], c["table_albedo"], c["table_scale"], c["table_pdf"])
    fates = np.bincount(c["ref_fate"], minlength=6)
    cnt = res.counts[0].astype(np.int64)
    assert cnt[0] == int(c["ref_scatter_count"]) and cnt[1] == fates[1] and cnt[2] == fates[2] and cnt[3] == fates[3]
    assert cnt[5] > 50 and cnt[6] == 0
    np.testing.assert_allclose(res.sums[0, 0], float(c["ref_initial_momentum"]), rtol=1e-12)
    np.testing.assert_allclose(res.sums[0, 1], float(c["ref_escaped_momentum"]), rtol=1e-12)
    np.testing.assert_allclose(res.sums[0, 2], float(c["ref_boundary_momentum"]), rtol=1e-12,
                               atol=1e-12 * abs(float(c["ref_initial_momentum"])))
