"""GPU parity, production mode: Philox-driven sweep vs the CPU oracle on the same streams.

The oracle (oracle/rt_oracle.c, pinned to the reference by tests/golden/walk_replay.npz) uses the
same counter layout, so the comparison is photon population by photon population: integer tallies
exact, floating-point sums to 1e-12 relative.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def small_table(n=64):
    wl = np.round(np.linspace(0.1, 1.0, n), 4)
    g = np.linspace(0.3, 0.8, n) * np.where(np.arange(n) % 7 == 0, -1, 1)
    alb = np.linspace(0.4, 0.95, n)
    scale = np.linspace(2.0, 0.5, n)
    pdf = (np.linspace(1.0, 0.1, n) ** 2) / n
    pdf[5:9] = 0.0
    return wl, g, alb, scale, pdf


def close_sums(got, want, A, rtol, floor):
    """|got - want| <= rtol |want| + floor * (the tally's natural size: the initial momentum for the sums of
    momenta, the sum of squared initial momenta for the three second-moment sums)."""
    size = np.repeat(np.abs(want[..., [A.SUM_INITIAL]]), A.NSUM, axis=-1)
    for k in (A.SUM_TRANSFER_SQ, A.SUM_TRANSFER_P0, A.SUM_P0_SQ):
        size[..., k] = np.abs(want[..., A.SUM_P0_SQ])
    err = np.abs(got - want)
    assert np.all(err <= rtol * np.abs(want) + floor * size), (err / np.maximum(np.abs(want), 1e-300)).max()


def compare(res, ref, A, nd=0, rtol=RTOL, floor=0.0):
    assert np.array_equal(res.counts, ref.counts)
    assert np.array_equal(res.mu_hist, ref.mu_hist)
    scale = np.abs(ref.sums[:, [A.SUM_INITIAL]])
    close_sums(res.sums, ref.sums, A, rtol, floor)
    close_sums(res.slab_sums, ref.slab_sums, A, rtol, floor)
    if nd:
        for got, want in ((res.depth_momentum, ref.depth_momentum), (res.depth_energy, ref.depth_energy)):
            assert np.all(np.abs(got - want) <= 1e-11 * np.abs(want) + 1e-13 * scale), np.abs(got - want).max()


@pytest.mark.parametrize("boundary", ["reflect", "absorb", "reemit"])
@pytest.mark.parametrize("scatter", ["hg", "iso"])
def test_sweep_matches_oracle(rt, engine, c_oracle, boundary, scatter):
    wl, g, alb, scale, pdf = small_table()
    engine.set_tables(wl, g, alb, scale, pdf)
    tau = np.array([1e-3, 0.3, 2.0, 20.0])
    kw = dict(boundary=boundary, scatter=scatter, n_mu_bins=10, rand_mu=(boundary == "reemit"))
    res = engine.run(rt.make_params(tau, 6000, **kw))
    ref = c_oracle.run(c_oracle.make_params(tau, 6000, **kw), wl, g, alb, scale, pdf)
    compare(res, ref, rt._abi)


def test_depth_bins_and_identities(rt, engine, c_oracle):
    A = rt._abi
    wl, g, alb, scale, pdf = small_table()
    engine.set_tables(wl, g, alb, scale, pdf)
    tau = np.array([0.5, 5.0])
    kw = dict(n_mu_bins=20, n_depth_bins=16)
    res = engine.run(rt.make_params(tau, 20000, **kw))
    ref = c_oracle.run(c_oracle.make_params(tau, 20000, **kw), wl, g, alb, scale, pdf)
    compare(res, ref, A, nd=16)
    s = res.sums
    # depth bins add up to the dust momentum; energy bins to the absorbed energy
    np.testing.assert_allclose(res.depth_momentum.sum(1), s[:, A.SUM_DUST_MOMENTUM], rtol=1e-10)
    np.testing.assert_allclose(res.depth_energy.sum(1), s[:, A.SUM_ABSORBED_ENERGY], rtol=1e-10)
    # two estimators of the transferred momentum agree (MC code.py:648-650, :728-729, :773-776):
    # initial = escaped(all) + dust + boundary deposit, and a reflection deposits -2*boundary_momentum
    lhs = s[:, A.SUM_INITIAL]
    rhs = s[:, A.SUM_ESCAPED] + s[:, A.SUM_ESCAPED_UNCOUNTED] + s[:, A.SUM_DUST_MOMENTUM] - 2 * s[:, A.SUM_BOUNDARY]
    np.testing.assert_allclose(lhs, rhs, rtol=1e-10)


def test_gray_order_and_monochromatic(rt, engine, c_oracle):
    engine.set_tables([1.0], [0.5], [0.9], [1.0])
    tau = np.logspace(-2, 2, 10)
    kw = dict(order="gray", h=1.0, monochromatic=1.0, n_mu_bins=10)
    res = engine.run(rt.make_params(tau, 10000, **kw))
    ref = c_oracle.run(c_oracle.make_params(tau, 10000, **kw), [1.0], [0.5], [0.9], [1.0])
    compare(res, ref, rt._abi)


def test_draine_phase_function(rt, engine, c_oracle):
    engine.set_tables([1.0], [0.5], [0.9], [1.0])
    tau = np.array([1.0, 3.0])
    kw = dict(order="gray", h=1.0, scatter="draine", alpha=0.3, n_mu_bins=10)
    res = engine.run(rt.make_params(tau, 8000, **kw))
    ref = c_oracle.run(c_oracle.make_params(tau, 8000, **kw), [1.0], [0.5], [0.9], [1.0])
    compare(res, ref, rt._abi)


def test_draine_with_g_and_alpha_zero_is_isotropic(rt, engine, c_oracle):
    """scatter='draine' with g == 0 and alpha == 0 takes the gray script's IsoScatter branch (Numpy_random_walk.py:388-391):
    no rejection loop, the lab-frame cosine assigned directly -- the same photon histories as scatter='iso'."""
    engine.set_tables([1.0], [0.0], [0.9], [1.0])
    tau = np.array([0.5, 2.0])
    kw = dict(order="gray", h=1.0, n_mu_bins=10)
    res = engine.run(rt.make_params(tau, 8000, scatter="draine", alpha=0.0, **kw))
    ref = c_oracle.run(c_oracle.make_params(tau, 8000, scatter="draine", alpha=0.0, **kw), [1.0], [0.0], [0.9], [1.0])
    compare(res, ref, rt._abi)
    iso = engine.run(rt.make_params(tau, 8000, scatter="iso", **kw))
    assert np.array_equal(res.counts, iso.counts) and np.array_equal(res.mu_hist, iso.mu_hist)
    np.testing.assert_allclose(res.sums, iso.sums, rtol=1e-12)
    # one block per scattering, not a rejection loop: flights + scatterings blocks in all
    assert res.counts[:, rt._abi.CNT_GUARD].sum() == 0


def test_result_independent_of_gpu_count(rt, engine):
    """Walking the slabs as 1, 2, 4 or 8 ranks gives bitwise the same tallies after the slab sum -- the
    depth-resolved deposition included (per chunk, per slab in chunk order, slabs in index order)."""
    wl, g, alb, scale, pdf = small_table()
    engine.set_tables(wl, g, alb, scale, pdf)
    tau = np.array([0.2, 3.0])
    for kw in (dict(n_depth_bins=16), dict(n_depth_bins=40, boundary="reemit", runtime_kernel=True)):
        base = engine.run(rt.make_params(tau, 50000, **kw))
        assert np.any(base.depth_momentum != 0) and np.any(base.depth_energy != 0)
        for world in (2, 4, 8):
            slab = np.zeros_like(base.slab_sums)
            depth = np.zeros_like(base.slab_depth)
            counts = np.zeros_like(base.counts)
            hist = np.zeros_like(base.mu_hist)
            for rank in range(world):
                r = engine.run(rt.make_params(tau, 50000, rank=rank, world=world, **kw))
                slab += r.slab_sums  # what an all-reduce(sum) of disjoint non-zero entries does
                depth += r.slab_depth
                counts += r.counts
                hist += r.mu_hist
            assert np.array_equal(slab, base.slab_sums) and np.array_equal(depth, base.slab_depth)
            assert np.array_equal(counts, base.counts)
            assert np.array_equal(hist, base.mu_hist)
            whole = rt.SweepResult(len(tau), 10, kw["n_depth_bins"], 8)
            whole.slab_sums[...], whole.slab_depth[...] = slab, depth
            whole.finalize_from_slabs()
            assert np.array_equal(whole.sums, base.sums)
            assert np.array_equal(whole.depth_momentum, base.depth_momentum) and np.array_equal(whole.depth_energy, base.depth_energy)


def test_rerun_is_bitwise_reproducible(rt, engine):
    wl, g, alb, scale, pdf = small_table()
    engine.set_tables(wl, g, alb, scale, pdf)
    prm = rt.make_params(np.array([1.0, 8.0]), 300000, n_depth_bins=32)
    a = engine.run(prm)
    for _ in range(3):
        b = engine.run(prm)
        assert np.array_equal(a.sums, b.sums) and np.array_equal(a.counts, b.counts) and np.array_equal(a.mu_hist, b.mu_hist)
        assert np.array_equal(a.depth_momentum, b.depth_momentum) and np.array_equal(a.depth_energy, b.depth_energy)
        assert np.array_equal(a.slab_depth, b.slab_depth)


def test_errors(rt, engine):
    engine.set_tables([0.5], [0.5], [0.5], [1.0])
    with pytest.raises(rt.RtError):
        engine.run(rt.make_params(1.0, 100, monochromatic=0.7))
    with pytest.raises(ValueError):
        rt.make_params(1.0, 100, boundary="bounce")
    with pytest.raises(rt.RtError):
        engine.run(rt.make_params(1.0, 100, scatter="draine"))
    with pytest.raises(rt.RtError):
        engine.run(rt.make_params(1.0, 100, world=3))


def test_device_math_accuracy(engine):
    """The kernels' own ln(u), cos(2 pi u) and sqrt(u) against NumPy (extended precision for ln and cos)."""
    rng = np.random.default_rng(7)
    u = np.concatenate([rng.random(200000), 2.0 ** -rng.integers(1, 53, 2000) * rng.random(2000),
                        1 - 2.0 ** -rng.integers(1, 53, 2000), [2.0 ** -53, 0.25, 0.5, 0.75, 1 - 2.0 ** -53, 0.125]])
    u = np.floor(u * 2.0 ** 53) / 2.0 ** 53
    u = u[u > 0]
    got_log, got_cos, got_sqrt, got_div = engine.selftest_math(u)
    # the hook evaluates ln and cos in both forms the kernels use (coefficients in registers / read from LDS)
    # and returns NaN where they differ in any bit
    assert not np.isnan(got_log).any() and not np.isnan(got_cos).any()
    ul = u.astype(np.longdouble)
    want_log = np.log(ul)
    err = np.abs(got_log.astype(np.longdouble) - want_log)
    assert float(np.max(err / np.maximum(np.abs(want_log), 1.0))) < 4e-16  # absolute below |ln| = 1, relative above
    pi_ld = np.longdouble("3.14159265358979323846264338327950288")
    want_cos = np.cos(2 * pi_ld * ul)  # exact angle, not the rounded product 2*pi*u
    assert float(np.max(np.abs(got_cos.astype(np.longdouble) - want_cos))) < 3.5e-16
    # and against what the reference computes: np.cos(2*np.pi*u) with the product rounded to double
    assert float(np.max(np.abs(got_cos - np.cos(2 * np.pi * u)))) < 1.2e-15
    # the square root is the correctly rounded one on its whole domain, products of two (1 - mu^2) included
    assert np.array_equal(got_sqrt, np.sqrt(u))
    tiny = np.concatenate([2.0 ** -rng.integers(53, 107, 5000) * (1 + rng.random(5000)), [2.0 ** -106, 1.0, 4.0]])
    assert np.array_equal(engine.selftest_math(tiny)[2], np.sqrt(tiny))
    # ... and so is the division of HGScatter's operand ranges
    assert np.array_equal(got_div, (1.0 - u * u) / (2.0 * u))
    # ... for either sign of the denominator, and for the reciprocal 1 / (2 g) itself (1 - g*g rounds to 1 below 7.4e-9)
    g = np.concatenate([-u[:100000], np.logspace(-300, -8.2, 20000), -np.logspace(-300, -8.2, 20000)])
    got = engine.selftest_math(g)[3]
    assert np.array_equal(got, (1.0 - g * g) / (2.0 * g))
    assert np.array_equal(got[100000:], 1.0 / (2.0 * g[100000:]))
    lz, _, sz, _ = engine.selftest_math(np.array([0.0]))
    assert np.isneginf(lz[0]) and sz[0] == 0.0


def test_which_kernel_serves_which_flags(rt, engine):
    """Every slab combination of loop order x boundary x {hg, iso} x depth bins, the shell geometry and the exact
    Draine law with every boundary and depth bins, abs_only and the per-photon escape cosines have a kernel with those
    flags fixed at compile time (the lean loop of the production kernel); what is left -- the reference's own Draine
    sampler, options on the one-block stream, pairs of the rarer options -- runs the kernel that reads every flag at
    run time, and RT_FLAG_RUNTIME_KERNEL asks for that kernel outright."""
    engine.set_tables(*small_table())
    tau = np.array([0.5, 5.0])
    want = [(dict(), "<mc,reflect,hg,plain>"), (dict(rand_mu=1), "<mc,reflect,hg>"),
            (dict(rng="philox32x4"), "<mc,reflect,hg,philox32x4,plain>"), (dict(rng="philox32x4", rand_mu=1), "<mc,reflect,hg,philox32x4>"),
            (dict(n_depth_bins=8), "<mc,reflect,hg,depth,plain>"), (dict(n_depth_bins=8, rand_mu=1), "<mc,reflect,hg,depth>"),
            (dict(boundary="absorb"), "<mc,absorb,hg,plain>"), (dict(boundary="reemit"), "<mc,reemit,hg,plain>"),
            (dict(boundary="absorb", monochromatic=1.0), "<mc,absorb,hg>"), (dict(boundary="reemit", rand_mu=1), "<mc,reemit,hg>"),
            (dict(boundary="reemit", n_depth_bins=8), "<mc,reemit,hg,depth,plain>"),
            (dict(scatter="iso"), "<mc,reflect,iso>"), (dict(order="gray", h=1.0), "<gray,reflect,hg>"),
            (dict(boundary="absorb", n_depth_bins=8), "<mc,absorb,hg,depth,plain>"),
            (dict(order="gray", h=1.0, boundary="reemit", scatter="iso", n_depth_bins=4), "<gray,reemit,iso,depth>"),
            (dict(abs_only=1), "<mc,reflect,hg,abs_only>"), (dict(abs_only=1, n_depth_bins=8), "<mc,reflect,hg,depth,abs_only>"),
            (dict(geometry="shell", shell_inner=1.0), "<mc,reflect,hg,shell>"),
            (dict(geometry="shell", shell_inner=1.0, boundary="reemit", scatter="draine_exact", n_depth_bins=8),
             "<mc,reemit,draine_exact,depth,shell>"),
            (dict(scatter="draine_exact", boundary="absorb"), "<mc,absorb,draine_exact>"),
            (dict(escaped_mu=True), "<mc,reflect,hg,escaped_mu>"), (dict(order="gray", h=1.0, escaped_mu=True), "<gray,reflect,hg,escaped_mu>"),
            (dict(order="gray", h=1.0, scatter="draine"), "<runtime>"), (dict(abs_only=1, boundary="absorb"), "<runtime>"),
            (dict(geometry="shell", shell_inner=1.0, order="gray", h=1.0), "<runtime>"),
            (dict(escaped_mu=True, n_depth_bins=4), "<runtime>"),
            (dict(scatter="iso", rng="philox32x4"), "<runtime,philox32x4>"),
            (dict(rng="philox40"), "<mc,reflect,hg,philox40,plain>"), (dict(rng="philox40", monochromatic=1.0), "<mc,reflect,hg,philox40>"),
            (dict(rng="philox40", boundary="absorb"), "<runtime,philox40>"),
            (dict(runtime_kernel=True), "<runtime>"), (dict(runtime_kernel=True, rng="philox32x4"), "<runtime,philox32x4>")]
    for kw, name in want:
        engine.run(rt.make_params(tau, 2000, **kw))
        assert engine.last_kernel() == "walk_kernel" + name, (kw, engine.last_kernel())


@pytest.mark.parametrize("order,boundary,scatter", [("mc", "reflect", "hg"), ("mc", "absorb", "hg"), ("mc", "reflect", "iso"),
                                                    ("gray", "reflect", "hg"), ("gray", "absorb", "iso")])
def test_production_sweep_equals_the_reference_restatement_on_its_own_streams(rt, engine, order, boundary, scatter):
    """The production (Philox) sweep against oracle/numpy_walk.py -- the vectorised restatement that reproduces the
    reference's seeded runs bit for bit -- fed, photon by photon, the uniforms the device draws (oracle/philox.py):
    the same histories, so the same counts and, to 1e-12, the same three momenta of runMC's estimator.  One
    wavelength (monochromatic), so that the emission draw plays no part."""
    from oracle import numpy_walk as nw
    from oracle import philox
    A = rt._abi
    wl0, g0, alb0, sc0 = 0.55, 0.63, 0.82, 1.3
    engine.set_tables([wl0], [g0], [alb0], [sc0])
    h = 1.0 if order == "gray" else rt.H_PLANCK
    tau, n, seed, base = np.array([0.3, 3.0]), 20000, 0x1234ABCD5678, 777
    res = engine.run(rt.make_params(tau, n, order=order, boundary=boundary, scatter=scatter, monochromatic=wl0, h=h, seed=seed,
                                    photon_id_base=base, n_mu_bins=10))
    for a in range(len(tau)):
        stream = nw.PerPhotonStream(n, lambda ids, j, a=a: philox.walk_uniform(np.asarray(ids, dtype=np.uint64) + np.uint64(base), j, a, seed))
        ref = nw.run_photons(tau[a], np.full(n, wl0), np.full(n, g0), np.full(n, alb0), np.full(n, sc0), np.ones(n),
                             boundary=boundary, scatter=scatter, order=order, stream=stream, h=h)
        c = res.counts[a].astype(np.int64)
        assert c[A.CNT_SCATTER] == ref["scatter_count"] and c[A.CNT_ESCAPED] == ref["n_escaped"]
        assert c[A.CNT_ESCAPED_UNCOUNTED] == ref["n_escaped_uncounted"] and c[A.CNT_ABSORBED] == ref["n_absorbed"]
        assert c[A.CNT_BOUNDARY_ABSORBED] == ref["n_boundary_absorbed"] and c[A.CNT_GUARD] == 0
        for k, name in ((A.SUM_INITIAL, "initial_momentum"), (A.SUM_ESCAPED, "escaped_momentum"), (A.SUM_BOUNDARY, "boundary_momentum")):
            assert abs(res.sums[a, k] - ref[name]) <= 1e-12 * abs(ref[name]) + 1e-14 * abs(ref["initial_momentum"]), (name, res.sums[a, k], ref[name])
        emu = np.sort(ref["escaped_mu"])
        bins = np.minimum(np.floor(10 * emu).astype(int), 10)
        assert np.array_equal(np.bincount(bins, minlength=11), res.mu_hist[a].astype(np.int64))
