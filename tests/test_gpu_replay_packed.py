"""The opt-in stream with one Philox block per loop iteration and 32-bit uniforms (rng="philox32x4") against the
reference itself.  tests/golden/walk_replay_packed.npz holds what the UNMODIFIED reference RunPhotons returned on the
uniforms that stream defines (oracle.philox.walk_uniform_packed; generator: oracle/make_golden.py): replayed through
the replay kernel photon by photon, and -- the photons of every case carrying the constants the production sweep gives
them -- through the PRODUCTION kernels of that stream, which must return the reference's own numbers."""
import os

import numpy as np
import pytest

from oracle import philox

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RTOL = 1e-12

G = np.load(os.path.join(GOLDEN, "walk_replay_packed.npz"))
NAMES = [str(n) for n in G["names"]]


def case(name):
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


@pytest.mark.parametrize("name", NAMES)
def test_replay_on_the_packed_uniforms_matches_reference(rt, engine, name):
    c = case(name)
    n, order = int(c["n"]), str(c["order"])
    xi = philox.walk_matrix_packed(n, int(c["ref_draws"].max()), 0, int(c["seed"]))
    prm = rt.make_params(float(c["tau"]), n, boundary=str(c["boundary"]), scatter=str(c["scatter"]), order=order,
                         h=(1.0 if order == "gray" else rt.H_PLANCK), n_mu_bins=10)
    res, tr = engine.replay(prm, xi, c["lam"], c["g"], c["albedo"], c["scale"], c["mu0"])
    assert np.array_equal(tr["fate"], c["ref_fate"]) and np.array_equal(tr["n_draws"], c["ref_draws"])
    np.testing.assert_allclose(tr["final_mu"], c["ref_final_mu"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(tr["final_z"], c["ref_final_z"], rtol=RTOL, atol=1e-12)
    check_tallies(rt, res, c, n)


@pytest.mark.parametrize("name", NAMES)
def test_production_sweep_of_the_packed_stream_matches_reference(rt, engine, name):
    c = case(name)
    n, order = int(c["n"]), str(c["order"])
    if "table_wl" in c:
        engine.set_tables(c["table_wl"], c["table_g"], c["table_albedo"], c["table_scale"], c["table_pdf"])
    else:
        engine.set_tables([float(c["lam"][0])], [float(c["g"][0])], [float(c["albedo"][0])], [float(c["scale"][0])])
    res = engine.run(rt.make_params(float(c["tau"]), n, boundary=str(c["boundary"]), scatter=str(c["scatter"]), order=order,
                                    h=(1.0 if order == "gray" else rt.H_PLANCK), seed=int(c["seed"]), rng="philox32x4", n_mu_bins=10))
    assert "philox32x4" in engine.last_kernel()  # a kernel of the packed stream
    check_tallies(rt, res, c, n)


def test_both_packed_instantiations_are_covered(rt, engine):
    seen = set()
    for name in ("mcp_reflect_hg_tau3", "mcp_absorb_hg_tau2"):
        c = case(name)
        engine.set_tables([float(c["lam"][0])], [float(c["g"][0])], [float(c["albedo"][0])], [float(c["scale"][0])])
        engine.run(rt.make_params(float(c["tau"]), 100, boundary=str(c["boundary"]), rng="philox32x4"))
        seen.add(engine.last_kernel())
    assert seen == {"walk_kernel<mc,reflect,hg,philox32x4,plain>", "walk_kernel<runtime,philox32x4>"}, seen
