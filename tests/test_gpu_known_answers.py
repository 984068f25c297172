"""The reference's known answers on the device, at 2e7 photons.

* intensity.dat (reference checkout, also the curve hard-coded at Numpy_random_walk.py:448-449): emergent intensity
  of the tau = 10 isotropic conservative slab with the re-emitting floor, rand_mu = 1, 10 cosine bins.  Its columns are
  theta (degrees, at the bin's mean cosine), the fraction of the photons leaving in the bin, that fraction's statistical
  error, and I = fraction * 10 / (2 cos theta) (Numpy_random_walk.py:446): a Monte Carlo table of about 1e6 photons
  with its own error bars.  2e7 photons here, so the comparison is limited by the TABLE's errors.
* moments.dat (reference checkout, read by none of its scripts, unlabelled): 11 levels from the floor (first row) to
  the surface (last row), six columns that identify as J+, J-, H+, H-, K+, K- of the same problem per emitted photon
  (J = sum 1/|mu|, H = sum mu/|mu|, K = sum mu^2/|mu| over the photons crossing the level upwards / downwards):
  H+ + H- = 1.000 at every level (flux conservation), J+ / H+ = 1.734 = sqrt 3 at the surface (Eddington), and at the
  surface J- = H- = K- = 0.  With that reading, K+ at the surface is the escaped momentum per photon, J+ the mean of
  1/mu over the escapes, and the momentum deposited between two levels is the difference of K+ + K- -- which radiative
  equilibrium makes H * dtau = 1 per bin of 10: the depth-binned deposition of the sweep against both.
  (The floor's own bin is left out: the reference scatters a re-emitted photon at z = 0 in the iteration that re-emitted
  it, MC code.py:539-547 / Numpy_random_walk.py:324-333, which puts the momentum of every re-emission into bin 0.)
* transmission of the conservative isotropic slab with photons lost at the floor (Old code/1D test.py:105-106 plots it
  against the two-stream curve 1 / (1 + tau / 2)), at 10 optical depths; at tau = 10 also against the asymptotic
  diffusion value mu0 H(mu0) / sqrt 3 / (tau + 2 q), H(1) = 2.9078, q = 0.710446 (Chandrasekhar, Radiative Transfer).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

N = 20000000
GRAY = dict(order="gray", h=1.0, monochromatic=1.0)
ONE = ([1.0], [0.0], [1.0], [1.0])   # one wavelength: lambda = 1, g = 0, albedo = 1, scale = 1


def test_emergent_intensity_table(rt, engine):
    tab = np.loadtxt(os.path.join(GOLDEN, "intensity.dat"))
    theta, frac, err, inten = tab.T
    engine.set_tables(*ONE)
    res = engine.run(rt.make_params(np.array([10.0]), N, boundary="reemit", scatter="iso", rand_mu=1, n_mu_bins=10, **GRAY))
    assert res.counts[0, rt._abi.CNT_ESCAPED] == N                      # conservative: every photon leaves through the top
    counts = res.mu_hist[0, :10].astype(float)
    counts[9] += res.mu_hist[0, 10]
    mids = (np.arange(10) + 0.5) / 10
    order = np.argsort(-mids)                                           # the table runs from mu = 0.95 down to 0.05
    assert np.all(np.abs(np.degrees(np.arccos(mids[order])) - theta) < 0.25)   # (75.5 in the file is 75.3 in the scripts: acos 0.25 = 75.52)
    got_frac = counts[order] / N
    got_int = got_frac * 10 / (2 * mids[order])                         # Numpy_random_walk.py:446
    z = (got_frac - frac) / np.hypot(err, np.sqrt(got_frac / N))
    rel = got_int / inten - 1
    print("intensity: relative deviation per bin", np.round(rel, 4), "z against the table's own errors", np.round(z, 2))
    assert np.max(np.abs(z)) < 4.0 and np.sum(z * z) < 29.6             # chi-square(10) 99.9 %
    assert np.all(np.abs(rel) < 4 * err / frac + 2e-3)                  # 0.8 % (mu = 0.95) ... 5.6 % (mu = 0.05): the table's errors


def test_depth_moments_table(rt, engine):
    A = rt._abi
    m = np.loadtxt(os.path.join(GOLDEN, "moments.dat"))
    jp, jm, hp, hm, kp, km = m.T
    # the identification itself, on the table alone
    np.testing.assert_allclose(hp + hm, 1.0, atol=1.5e-3)
    assert jm[-1] == 0 and hm[-1] == 0 and km[-1] == 0 and abs(jp[-1] / hp[-1] - np.sqrt(3)) < 3e-3
    engine.set_tables(*ONE)
    res = engine.run(rt.make_params(np.array([10.0]), N, boundary="reemit", scatter="iso", rand_mu=1, n_mu_bins=1000,
                                    n_depth_bins=10, **GRAY))
    k_top = res.sums[0, A.SUM_ESCAPED] / N                              # sum of mu over the escapes = K+ at the surface
    h = res.mu_hist[0].astype(float)
    mu = np.append((np.arange(1000) + 0.5) / 1000, 1.0)
    j_top = np.sum(h / mu) / N
    dep = res.depth_momentum[0] / N
    table_dep = (kp + km)[:-1] - (kp + km)[1:]
    print("K+(0) %.5f (table %.4f)  J+(0) %.4f (table %.3f)  deposition per bin" % (k_top, kp[-1], j_top, jp[-1]), np.round(dep, 4),
          "table", np.round(table_dep, 4))
    assert abs(k_top - kp[-1]) < 1.5e-3 and abs(j_top - jp[-1]) < 1e-2
    assert np.all(np.abs(dep[1:] - 1.0) < 4e-3)                         # radiative equilibrium: H dtau = 1 per bin
    assert np.all(np.abs(dep[1:] - table_dep[1:]) < 0.03)               # the table's own noise is 2 %
    np.testing.assert_allclose(res.depth_momentum[0].sum(), res.sums[0, A.SUM_DUST_MOMENTUM], rtol=1e-11)


def test_two_stream_transmission(rt, engine):
    tau = np.logspace(-2, 1, 10)
    engine.set_tables(*ONE)
    res = engine.run(rt.make_params(tau, N, boundary="absorb", scatter="iso", **GRAY))
    c = res.counts.astype(np.int64)
    A = rt._abi
    assert np.array_equal(c[:, A.CNT_ESCAPED] + c[:, A.CNT_BOUNDARY_ABSORBED], c[:, A.CNT_PHOTONS])
    t = c[:, A.CNT_ESCAPED] / N
    two_stream = 1 / (1 + tau / 2)
    dev = t / two_stream - 1
    print("transmission", np.round(t, 5), "vs two-stream", np.round(dev, 4))
    assert np.all(np.diff(t) < 0)
    assert np.all(np.abs(dev[tau <= 1.0]) < 0.015) and np.all(dev <= 1e-3) and np.all(dev > -0.13)
    diffusion = 2.9078 / np.sqrt(3) / (10 + 2 * 0.710446)
    assert abs(t[-1] / diffusion - 1) < 5e-3, (t[-1], diffusion)
