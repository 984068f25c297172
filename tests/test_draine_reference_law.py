"""RT_SCATTER_DRAINE against the reference's own sampler output.  The reference's DraineScatter
(Numpy_random_walk.py:106-130) draws candidates in batches and pairs accepted cosines to photons by acceptance order;
the device and oracle/rt_oracle.c draw per photon from that photon's own stream.  Same law, different pairing: this
file pins the LAW -- the distribution of the scattering cosine -- to histograms of 1e6 cosines the unmodified
reference function returned (tests/golden/draine_reference_hist.npz, oracle/make_golden_draine.py).

Set-up that isolates one scattering: replay mode, gray order, albedo 1, a slab too thick to leave, photons starting
along the normal (mu = 1, so the frame rotation returns the scattering cosine itself, Numpy_random_walk.py:232-238)
and an iteration guard of one flight: the photon flies, scatters once and is retired with its new cosine."""
import os

import numpy as np
import pytest
from scipy import stats

from conftest import GOLDEN

GOLD = np.load(os.path.join(GOLDEN, "draine_reference_hist.npz"))
N, MAX_DRAWS = 100000, 4 + 2 * 150   # 150 tries: 0.92**150 = 4e-6 of the g = 0.85 photons run out and are dropped


def inputs(g):
    rs = np.random.RandomState(777 + int(g * 100))
    xi = rs.random_sample((N, MAX_DRAWS))
    one = np.ones(N)
    return xi, one, np.full(N, g), one, one, one   # xi, lambda, g, albedo, scale, mu0


def chi2_against_reference(cosines, k):
    got = np.histogram(cosines, bins=GOLD["edges"])[0].astype(float)
    ref = GOLD["hist"][k].astype(float)
    ng, nr = got.sum(), ref.sum()
    chi2 = np.sum((got / ng - ref / nr) ** 2 / (got / ng ** 2 + ref / nr ** 2))
    return chi2, len(got) - 1


def check(trace, k, A):
    done = (trace["fate"] == A.FATE_GUARD) & (trace["n_scatter"] == 1)
    assert done.mean() > 0.999
    chi2, dof = chi2_against_reference(trace["final_mu"][done], k)
    assert chi2 < stats.chi2.ppf(0.999, dof), (chi2, dof)
    # the test has teeth: Henyey-Greenstein cosines of the same g are rejected outright
    g = float(GOLD["g"][k])
    s = 2 * np.random.RandomState(1).random_sample(N) - 1
    hg = 1 / (2 * g) * (1 + g ** 2 - ((1 - g ** 2) / (1 + g * s)) ** 2)
    assert chi2_against_reference(hg, k)[0] > 5 * stats.chi2.ppf(0.999, dof)


@pytest.mark.parametrize("k", [0, 1])
def test_oracle_sampler_draws_the_reference_law(c_oracle, k):
    g = float(GOLD["g"][k])
    prm = c_oracle.make_params(np.array([1e12]), N, order="gray", scatter="draine", alpha=float(GOLD["alpha"]), h=1.0,
                               max_events=1)
    _, trace = c_oracle.replay(prm, *inputs(g))
    check(trace, k, c_oracle.abi)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1])
def test_device_sampler_draws_the_reference_law(rt, engine, c_oracle, k):
    g = float(GOLD["g"][k])
    kw = dict(order="gray", scatter="draine", alpha=float(GOLD["alpha"]), h=1.0, max_events=1)
    args = inputs(g)
    _, trace = engine.replay(rt.make_params(np.array([1e12]), N, **kw), *args)
    check(trace, k, rt._abi)
    _, ref = c_oracle.replay(c_oracle.make_params(np.array([1e12]), N, **kw), *args)
    assert np.array_equal(trace["fate"], ref["fate"]) and np.array_equal(trace["n_draws"], ref["n_draws"])
    np.testing.assert_allclose(trace["final_mu"], ref["final_mu"], rtol=0, atol=1e-12)
