"""The integration guide's raw ctypes snippet, run end to end on the GPU as a maintainer would paste it: same numbers as
the package's own host layer, and a struct typed from an older header is refused instead of being read past its end."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

from conftest import ROOT
from test_integration_doc import call_block, struct_block

pytestmark = pytest.mark.gpu

tables = importlib.import_module("radiative-transfer_amd.tables")


def inputs():
    wl = np.round(np.linspace(0.1, 1.0, 64), 4)
    grain = {"WL": wl, "g": np.linspace(0.3, 0.8, 64), "Albedo": np.linspace(0.4, 0.9, 64), "Scale": np.linspace(2.0, 0.5, 64)}
    pdf, _ = tables.photon_pdf(wl, tables.planck_sed(wl, 2.0e4))
    return grain, pdf, np.array([0.05, 0.7, 9.0]), 40000


def test_the_documented_binding_runs_and_agrees_with_the_package(rt, engine):
    grain, pdf, tau, n = inputs()
    ns = dict(Grain_data=grain, PDF=pdf, tau_list_calc=tau, num_photons=n)
    cwd = os.getcwd()
    os.chdir(ROOT)                       # the snippet loads the library by its path from the repository root
    try:
        exec(struct_block(), ns)
        exec(call_block(), ns)
    finally:
        os.chdir(cwd)
    engine.set_tables(grain["WL"], grain["g"], grain["Albedo"], grain["Scale"], pdf)
    res = engine.run(rt.make_params(tau, n, n_mu_bins=10))
    assert np.array_equal(ns["sums"], res.sums) and np.array_equal(ns["counts"], res.counts) and np.array_equal(ns["hist"], res.mu_hist)
    assert np.array_equal(ns["momentum_method_1"], res.momentum_fraction())


def test_a_binding_from_an_older_header_is_refused(rt, engine):
    """ABI 7's rt_params had no size word and no escaped_mu: passed as it is, the library must refuse it on its first
    member rather than read a pointer from behind the caller's object."""
    dp, u64p = C.POINTER(C.c_double), C.POINTER(C.c_uint64)

    class OldParams(C.Structure):
        _fields_ = [(n, C.c_int32) for n in ("n_atm", "boundary", "scatter", "loop_order", "rand_mu", "abs_only", "n_mu_bins",
                                              "n_depth_bins", "max_events_per_photon", "n_slabs", "rank", "world")] \
                 + [("seed", C.c_uint64), ("photon_id_base", C.c_uint64), ("monochromatic_wl", C.c_double),
                    ("alpha", C.c_double), ("h", C.c_double), ("tau_atm", dp), ("n_photons", u64p),
                    ("rng_mode", C.c_int32), ("geometry", C.c_int32), ("table_of_atm", C.POINTER(C.c_int32)),
                    ("shell_inner", C.c_double)]

    grain, pdf, tau, n = inputs()
    engine.set_tables(grain["WL"], grain["g"], grain["Albedo"], grain["Scale"], pdf)
    npn = np.full(len(tau), n, dtype=np.uint64)
    old = OldParams(len(tau), 0, 0, 0, 0, 0, 10, 0, 0, 8, 0, 1, 0x5EED2026, 0, 0.0, 0.3, 6.6261e-27,
                    tau.ctypes.data_as(dp), npn.ctypes.data_as(u64p), 0, 0, None, 0.0)
    res = rt.SweepResult(len(tau), 10, 0, 8)
    lib = rt._abi.load_library()
    rc = lib.rt_run(engine._ctx, C.cast(C.byref(old), C.POINTER(rt._abi.RtParams)), C.byref(res._c))
    assert rc == -1 and b"struct_size" in lib.rt_last_error()
    assert not res.counts.any()
