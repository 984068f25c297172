"""GPU edge cases, each against the CPU oracle on the same Philox streams: empty and ragged photon counts,
photon ids beyond 2^32, zero and huge optical depths, the iteration guard (NaN tables, conservative
scattering), many atmospheres in one launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def table(n=48):
    wl = np.round(np.linspace(0.1, 1.0, n), 4)
    return wl, np.linspace(0.3, 0.8, n), np.linspace(0.4, 0.95, n), np.linspace(2.0, 0.5, n), np.linspace(1.0, 0.1, n) / n


def same(res, ref):
    assert np.array_equal(res.counts, ref.counts)
    assert np.array_equal(res.mu_hist, ref.mu_hist)
    np.testing.assert_allclose(res.sums, ref.sums, rtol=1e-12, atol=0)


def test_empty_and_ragged_photon_counts(rt, engine, c_oracle):
    t = table()
    engine.set_tables(*t)
    tau = np.array([0.5, 2.0, 5.0, 1.0, 3.0])
    n = np.array([0, 1, 63, 1025, 7], dtype=np.uint64)   # photon_reduction makes counts ragged (MC code.py:1053-1059)
    res = engine.run(rt.make_params(tau, n))
    ref = c_oracle.run(c_oracle.make_params(tau, n), *t)
    same(res, ref)
    assert list(res.counts[:, rt._abi.CNT_PHOTONS]) == [0, 1, 63, 1025, 7]
    assert np.all(res.sums[0] == 0)


def test_photon_ids_beyond_32_bits(rt, engine, c_oracle):
    t = table()
    engine.set_tables(*t)
    base = (1 << 32) - 1000   # ids straddle the 32-bit boundary: the high counter word matters
    kw = dict(photon_id_base=base, seed=0xDEADBEEFCAFEF00D)
    res = engine.run(rt.make_params(np.array([2.0]), 5000, **kw))
    ref = c_oracle.run(c_oracle.make_params(np.array([2.0]), 5000, **kw), *t)
    same(res, ref)
    other = engine.run(rt.make_params(np.array([2.0]), 5000, seed=0xDEADBEEFCAFEF00D))
    assert not np.array_equal(other.sums, res.sums)


def test_zero_and_huge_optical_depth(rt, engine, c_oracle):
    t = table()
    engine.set_tables(*t)
    tau = np.array([0.0, 1e-12, 300.0])
    res = engine.run(rt.make_params(tau, 3000))
    ref = c_oracle.run(c_oracle.make_params(tau, 3000), *t)
    same(res, ref)
    A = rt._abi
    assert res.counts[0, A.CNT_ESCAPED] == 3000 and res.counts[0, A.CNT_SCATTER] == 0
    assert res.counts[2, A.CNT_ESCAPED] == 0


def test_iteration_guard(rt, engine, c_oracle):
    A = rt._abi
    # conservative scattering in a very thick reflecting slab: histories would need ~tau^2 steps
    engine.set_tables([0.5], [0.3], [1.0], [1.0])
    kw = dict(max_events=50, monochromatic=0.5)
    res = engine.run(rt.make_params(np.array([1e4]), 2000, **kw))
    ref = c_oracle.run(c_oracle.make_params(np.array([1e4]), 2000, **kw), [0.5], [0.3], [1.0], [1.0])
    same(res, ref)
    assert res.counts[0, A.CNT_GUARD] > 1900
    # a guarded photon flew exactly 50 times, every other one fewer
    assert res.counts[0, A.CNT_GUARD] * 50 <= res.counts[0, A.CNT_FLIGHTS] <= 2000 * 50
    # NaN optical constants (a wavelength without table entry in the reference's left join): never escapes,
    # never absorbed -> the reference loops forever, here the guard retires the photon
    engine.set_tables([0.5], [np.nan], [np.nan], [np.nan])
    res = engine.run(rt.make_params(np.array([1.0]), 500, max_events=20, monochromatic=0.5))
    assert res.counts[0, A.CNT_GUARD] == 500 and res.counts[0, A.CNT_FLIGHTS] == 500 * 20


def test_g_zero_divides_like_the_reference(rt, engine, c_oracle):
    """g == 0 in the wavelength-dependent script divides by zero in HGScatter (MC code.py:439): the cosine
    becomes NaN and the photon can only end by absorption.  Same arithmetic here."""
    engine.set_tables([0.5], [0.0], [0.6], [1.0])
    kw = dict(monochromatic=0.5, max_events=200)
    res = engine.run(rt.make_params(np.array([2.0]), 2000, **kw))
    ref = c_oracle.run(c_oracle.make_params(np.array([2.0]), 2000, **kw), [0.5], [0.0], [0.6], [1.0])
    assert np.array_equal(res.counts, ref.counts)
    A = rt._abi
    assert res.counts[0, A.CNT_ABSORBED] + res.counts[0, A.CNT_ESCAPED] + res.counts[0, A.CNT_GUARD] + \
        res.counts[0, A.CNT_ESCAPED_UNCOUNTED] == 2000


def test_many_atmospheres_one_launch(rt, engine, c_oracle):
    t = table()
    engine.set_tables(*t)
    tau = np.logspace(-3, 2, 700)     # the reference's 'detailed MC' tables have up to 3500 rows
    res = engine.run(rt.make_params(tau, 300))
    ref = c_oracle.run(c_oracle.make_params(tau, 300), *t)
    same(res, ref)
    m = res.momentum_fraction()
    assert m[0] < 0.01 and 0.5 < m[-1] < 2.0


def test_all_flag_combinations_smoke(rt, engine, c_oracle):
    t = table(16)
    engine.set_tables(*t)
    tau = np.array([0.7, 4.0])
    for order in ("mc", "gray"):
        for boundary in ("reflect", "absorb", "reemit"):
            for scatter in ("hg", "iso"):
                for abs_only in (0, 1):
                    kw = dict(order=order, boundary=boundary, scatter=scatter, abs_only=abs_only, rand_mu=abs_only,
                              h=(1.0 if order == "gray" else rt.H_PLANCK), n_depth_bins=(4 if abs_only else 0))
                    res = engine.run(rt.make_params(tau, 1500, **kw))
                    ref = c_oracle.run(c_oracle.make_params(tau, 1500, **kw), *t)
                    same(res, ref)


def test_slabs_beyond_the_last_photon(rt, engine, c_oracle):
    """Fewer photons than the slab partition has room for: the chunks of the last slabs start beyond the last
    photon id (their end lies below their start) and must contribute nothing."""
    t = table()
    engine.set_tables(*t)
    tau = np.array([1.0, 3.0, 0.2])
    for n in ([9, 1, 2], [17, 5, 8]):       # n_slabs = 8: slab length 2 / 1 / 1, slabs 5..7 of the first are empty
        n = np.array(n, dtype=np.uint64)
        res = engine.run(rt.make_params(tau, n))
        ref = c_oracle.run(c_oracle.make_params(tau, n), *t)
        same(res, ref)
        assert list(res.counts[:, rt._abi.CNT_PHOTONS]) == list(n)


def test_results_do_not_depend_on_the_work_order(rt, engine):
    """A launch with many chunks per resident wave (here 52 000 over 5120 waves) hands them out in interleaved
    rows, a short one alternates the chunks of thick and thin atmospheres.  A chunk's tallies depend on its photon ids only, so the first 100 of 500
    atmospheres in one launch give, bit for bit, what a launch of those 100 alone gives (same atmosphere
    indices, hence the same Philox counters)."""
    t = table()
    engine.set_tables(*t)
    tau = np.logspace(-3, 2, 500)
    n = 100000
    whole = engine.run(rt.make_params(tau, n, n_mu_bins=10))
    assert np.all(whole.counts[:, rt._abi.CNT_PHOTONS] == n)
    first = engine.run(rt.make_params(tau[:100], n, n_mu_bins=10))
    assert np.array_equal(first.counts, whole.counts[:100])
    assert np.array_equal(first.mu_hist, whole.mu_hist[:100])
    assert np.array_equal(first.sums, whole.sums[:100])
    # a short launch with thick and thin atmospheres (5200 chunks over 5120 waves) alternates thick and thin chunks;
    # every other atmosphere of it alone, at its own index, is a different list in a different order
    tau = np.logspace(-3, 2, 50)
    short = engine.run(rt.make_params(tau, n, n_mu_bins=10))
    half = engine.run(rt.make_params(tau, np.where(np.arange(50) % 2 == 0, n, 0), n_mu_bins=10))
    pick = np.arange(50) % 2 == 0
    assert np.array_equal(half.counts[pick], short.counts[pick]) and np.all(half.counts[~pick] == 0)
    assert np.array_equal(half.mu_hist[pick], short.mu_hist[pick])
    assert np.array_equal(half.sums[pick], short.sums[pick])


def test_histogram_limit_is_reported_before_any_device_work(rt, engine):
    """Both histograms live in LDS per wave; a combination that does not fit is refused during validation, and the
    caller's result arrays are not touched."""
    engine.set_tables([0.5], [0.3], [0.5], [1.0])
    with pytest.raises(rt.RtError, match="LDS"):
        engine.run(rt.make_params(np.array([1.0]), 100, n_mu_bins=4096, monochromatic=0.5))
    with pytest.raises(rt.RtError, match="LDS"):
        engine.run(rt.make_params(np.array([1.0]), 100, n_mu_bins=2000, n_depth_bins=512, monochromatic=0.5))
    res = engine.run(rt.make_params(np.array([1.0]), 100, n_mu_bins=3900, monochromatic=0.5))   # just inside
    assert res.counts[0, rt._abi.CNT_PHOTONS] == 100
