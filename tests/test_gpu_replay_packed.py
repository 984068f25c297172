"""The opt-in streams with one Philox block per loop iteration -- 32-bit uniforms (rng="philox32x4") and the 24 | 40 |
40 | 24-bit cut (rng="philox40") -- against the reference itself.  tests/golden/walk_replay_packed.npz and
walk_replay_philox40.npz hold what the UNMODIFIED reference RunPhotons returned on the uniforms those streams define
(oracle.philox.walk_uniform_packed / walk_uniform_p40; generator: oracle/make_golden.py): replayed through
the replay kernel photon by photon, and -- the photons of every case carrying the constants the production sweep gives
them -- through the PRODUCTION kernels of that stream, which must return the reference's own numbers."""
import os

import numpy as np
import pytest

from oracle import philox

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RTOL = 1e-12

FILES = {"philox32x4": np.load(os.path.join(GOLDEN, "walk_replay_packed.npz")),
         "philox40": np.load(os.path.join(GOLDEN, "walk_replay_philox40.npz"))}
MATRIX = {"philox32x4": philox.walk_matrix_packed, "philox40": philox.walk_matrix_p40}
NAMES = [(rng, str(n)) for rng, G in FILES.items() for n in G["names"]]


def case(name, rng="philox32x4"):
    G = FILES[rng]
    return {k.split("/", 1)[1]: G[k] for k in G.files if k.startswith(name + "/")}


def check_tallies(rt, res, c, n):
    A = rt._abi
    s, cnt = res.sums[0], res.counts[0].astype(np.int64)
    fates = np.bincount(c["ref_fate"], minlength=6)
    np.testing.assert_allclose(s[A.SUM_INITIAL], float(c["ref_initial_momentum"]), rtol=RTOL)
    np.testing.assert_allclose(s[A.SUM_ESCAPED], float(c["ref_escaped_momentum"]), rtol=RTOL)
    np.testing.assert_allclose(s[A.SUM_BOUNDARY], float(c["ref_boundary_momentum"]), rtol=RTOL,
                               atol=1e-12 * abs(float(c["ref_initial_momentum"])))
    if int(c["ref_scatter_count"]) >= 0:
        assert cnt[A.CNT_SCATTER] == int(c["ref_scatter_count"])
    assert cnt[A.CNT_ESCAPED] == fates[A.FATE_ESCAPED] and cnt[A.CNT_ESCAPED_UNCOUNTED] == fates[A.FATE_ESCAPED_UNCOUNTED]
    assert cnt[A.CNT_ABSORBED] == fates[A.FATE_ABSORBED] and cnt[A.CNT_BOUNDARY_ABSORBED] == fates[A.FATE_BOUNDARY]
    assert cnt[A.CNT_GUARD] == 0 and cnt[A.CNT_PHOTONS] == n
    ref_hist = np.bincount(np.minimum(np.floor(10 * c["ref_escaped_mu"]).astype(int), 10), minlength=11)
    assert np.array_equal(res.mu_hist[0], ref_hist)


@pytest.mark.parametrize("rng,name", NAMES)
def test_replay_on_the_packed_uniforms_matches_reference(rt, engine, rng, name):
    c = case(name, rng)
    n, order = int(c["n"]), str(c["order"])
    xi = MATRIX[rng](n, int(c["ref_draws"].max()), 0, int(c["seed"]))
    prm = rt.make_params(float(c["tau"]), n, boundary=str(c["boundary"]), scatter=str(c["scatter"]), order=order,
                         h=(1.0 if order == "gray" else rt.H_PLANCK), n_mu_bins=10)
    res, tr = engine.replay(prm, xi, c["lam"], c["g"], c["albedo"], c["scale"], c["mu0"])
    assert np.array_equal(tr["fate"], c["ref_fate"]) and np.array_equal(tr["n_draws"], c["ref_draws"])
    np.testing.assert_allclose(tr["final_mu"], c["ref_final_mu"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(tr["final_z"], c["ref_final_z"], rtol=RTOL, atol=1e-12)
    check_tallies(rt, res, c, n)


@pytest.mark.parametrize("rng,name", NAMES)
def test_production_sweep_of_the_packed_stream_matches_reference(rt, engine, rng, name):
    c = case(name, rng)
    n, order = int(c["n"]), str(c["order"])
    if "table_wl" in c:
        engine.set_tables(c["table_wl"], c["table_g"], c["table_albedo"], c["table_scale"], c["table_pdf"])
    else:
        engine.set_tables([float(c["lam"][0])], [float(c["g"][0])], [float(c["albedo"][0])], [float(c["scale"][0])])
    res = engine.run(rt.make_params(float(c["tau"]), n, boundary=str(c["boundary"]), scatter=str(c["scatter"]), order=order,
                                    h=(1.0 if order == "gray" else rt.H_PLANCK), seed=int(c["seed"]), rng=rng, n_mu_bins=10))
    assert rng in engine.last_kernel()  # a kernel of that stream
    check_tallies(rt, res, c, n)


@pytest.mark.parametrize("rng,prefix", [("philox32x4", "mcp"), ("philox40", "mcq")])
def test_both_packed_instantiations_are_covered(rt, engine, rng, prefix):
    seen = set()
    for name in (prefix + "_reflect_hg_tau3", prefix + "_absorb_hg_tau2"):
        c = case(name, rng)
        engine.set_tables([float(c["lam"][0])], [float(c["g"][0])], [float(c["albedo"][0])], [float(c["scale"][0])])
        engine.run(rt.make_params(float(c["tau"]), 100, boundary=str(c["boundary"]), rng=rng))
        seen.add(engine.last_kernel())
    assert seen == {f"walk_kernel<mc,reflect,hg,{rng},plain>", f"walk_kernel<runtime,{rng}>"}, seen
