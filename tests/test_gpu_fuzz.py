"""Seeded random combinations of every option of a sweep, GPU against the CPU oracle on the same streams.

The named tests cover each option on its own and the pairs the reference uses; this one draws whole
configurations -- boundary x scattering law x loop order x geometry x stream x bins x slabs x rank x tables x
photon counts x id base x seed x iteration cap -- so that options meet in combinations nobody listed, in the
kernels with flags fixed at compile time and (three cases in ten are sent there whatever their flags) in the run-time one alike.  Integer tallies exact, sums to 1e-12.
"""
import os

import numpy as np
import pytest

from test_gpu_multi_table import tables3
from test_gpu_walk import compare, small_table

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("RT_FUZZ_CASES", "256"))  # a larger number for an occasional long run


def draw_case(k):
    rs = np.random.RandomState(9000 + k)
    pick = lambda *xs: xs[rs.randint(len(xs))]
    kw = dict(boundary=pick("reflect", "reflect", "absorb", "reemit"), order=pick("mc", "mc", "gray"),
              rand_mu=int(rs.rand() < 0.3), abs_only=int(rs.rand() < 0.15), n_mu_bins=pick(0, 1, 7, 10, 33),
              n_depth_bins=pick(0, 0, 5, 16), n_slabs=pick(8, 8, 16, 24), seed=int(rs.randint(1, 2 ** 31)) * 2654435761 % 2 ** 63,
              photon_id_base=pick(0, 0, 12345, 2 ** 32 - 1000, 2 ** 40 + 7), rng=pick("philox52", "philox52", "philox32x4"),
              max_events=pick(0, 0, 0, 3, 40))
    kw["scatter"] = pick("hg", "hg", "iso", "draine_exact", "draine") if kw["order"] == "gray" else pick("hg", "hg", "iso", "draine_exact")
    if kw["scatter"] in ("draine", "draine_exact"):
        kw["rng"] = "philox52"  # the packed stream has one block per iteration; the rejection loop needs more
        kw["alpha"] = pick(0.0, 0.3, 1.0)
    if rs.rand() < 0.25:
        kw.update(geometry="shell", shell_inner=pick(0.05, 0.7, 30.0))
    world = pick(1, 1, 2, 4, 8)
    if kw["n_slabs"] % world:
        world = 1
    kw.update(world=world, rank=int(rs.randint(world)))
    if kw["order"] == "gray":
        kw["h"] = 1.0
    multi = rs.rand() < 0.35
    n_atm = int(rs.randint(1, 6))
    tau = 10 ** rs.uniform(-3, 1.7, n_atm)
    top = 9000 if rs.rand() < 0.75 else 300000  # the larger ones span several chunks per slab
    n_photons = rs.randint(0, top, n_atm)
    if rs.rand() < 0.5:
        n_photons[:] = int(rs.randint(1, top))
    if multi:
        kw["table_of_atm"] = rs.randint(0, 3, n_atm)
    mono = rs.rand() < 0.2
    runtime = rs.rand() < 0.3  # (drawn last: the cases keep the options they had before this one existed)
    if rs.rand() < 0.5 and kw["rng"] == "philox32x4":
        kw["rng"] = "philox40"  # the other one-block stream
    return kw, tau, n_photons, multi, mono, runtime


@pytest.mark.parametrize("k", range(N_CASES))
def test_random_configuration(rt, engine, c_oracle, k):
    kw, tau, n_photons, multi, mono, runtime = draw_case(k)
    wl, g, alb, scale, pdf = tables3() if multi else small_table()
    if mono:
        kw["monochromatic"] = float(wl[17])
    engine.set_tables(wl, g, alb, scale, pdf)
    res = engine.run(rt.make_params(tau, n_photons, runtime_kernel=runtime, **kw))
    assert ("<runtime" in engine.last_kernel()) or not runtime
    ref = c_oracle.run(c_oracle.make_params(tau, n_photons, **kw), wl, g, alb, scale, pdf)
    # shell geometry: sums of signed terms can cancel, same bound as the tests of that mode
    compare(res, ref, rt._abi, nd=kw["n_depth_bins"], floor=1e-14 if kw.get("geometry") == "shell" else 0.0)
    A = rt._abi
    c = res.counts.astype(np.int64)
    ended = c[:, A.CNT_ESCAPED] + c[:, A.CNT_ESCAPED_UNCOUNTED] + c[:, A.CNT_ABSORBED] + c[:, A.CNT_GUARD] + c[:, A.CNT_BOUNDARY_ABSORBED]
    assert np.array_equal(ended, c[:, A.CNT_PHOTONS])  # every history ends exactly once
