"""Several grain tables in one sweep (rt_set_tables_multi + rt_params.table_of_atm): the batched
species x radius x optical-depth configuration.  Each atmosphere must give exactly what it gives when its
table is loaded alone, and the whole batch must agree with the CPU oracle."""
import numpy as np
import pytest

from test_gpu_walk import compare, small_table

pytestmark = pytest.mark.gpu


def tables3():
    wl, g, alb, scale, pdf = small_table()
    g3 = np.stack([g, 0.5 * g, np.full_like(g, 0.9)])
    a3 = np.stack([alb, alb[::-1], np.full_like(alb, 0.5)])
    s3 = np.stack([scale, scale ** 2, np.ones_like(scale)])
    p3 = np.stack([pdf, pdf[::-1], np.full_like(pdf, 0.01)])
    return wl, g3, a3, s3, p3


def test_batch_matches_oracle_and_single_table_runs(rt, engine, c_oracle):
    wl, g3, a3, s3, p3 = tables3()
    tau = np.array([0.05, 0.5, 3.0, 30.0, 0.5, 3.0])
    which = np.array([0, 1, 2, 0, 2, 1])
    engine.set_tables(wl, g3, a3, s3, p3)
    assert engine.n_tables == 3
    kw = dict(n_mu_bins=10)
    res = engine.run(rt.make_params(tau, 8000, table_of_atm=which, **kw))
    ref = c_oracle.run(c_oracle.make_params(tau, 8000, table_of_atm=which, **kw), wl, g3, a3, s3, p3)
    compare(res, ref, rt._abi)
    # the same atmospheres with their table loaded alone: the atmosphere index is part of the stream, so they are
    # run at their own positions
    for t in range(3):
        engine.set_tables(wl, g3[t], a3[t], s3[t], p3[t])
        n = np.where(which == t, 8000, 0)
        one = engine.run(rt.make_params(tau, n, **kw))
        pick = which == t
        assert np.array_equal(one.counts[pick], res.counts[pick])
        assert np.array_equal(one.sums[pick], res.sums[pick])
        assert np.array_equal(one.mu_hist[pick], res.mu_hist[pick])


def test_generic_kernel_and_packed_stream_with_tables(rt, engine, c_oracle):
    wl, g3, a3, s3, p3 = tables3()
    engine.set_tables(wl, g3, a3, s3, p3)
    tau = np.array([0.3, 4.0, 4.0])
    which = np.array([2, 0, 1])
    for kw in (dict(boundary="reemit", rand_mu=1), dict(rng="philox32x4"), dict(order="gray", h=1.0, scatter="iso")):
        res = engine.run(rt.make_params(tau, 5000, table_of_atm=which, **kw))
        ref = c_oracle.run(c_oracle.make_params(tau, 5000, table_of_atm=which, **kw), wl, g3, a3, s3, p3)
        compare(res, ref, rt._abi)


def test_bad_table_index(rt, engine):
    wl, g3, a3, s3, p3 = tables3()
    engine.set_tables(wl, g3, a3, s3, p3)
    with pytest.raises(rt.RtError):
        engine.run(rt.make_params([1.0, 2.0], 100, table_of_atm=[0, 3]))
    with pytest.raises(ValueError):
        engine.set_tables(wl, g3, a3[:2], s3, p3)
