"""The NumPy restatement of the shell-velocity right-hand side (oracle/shell_ode.py) against the reference's
own ODEs / GetMC / get_tau_IR outputs stored in tests/golden/shell.npz (oracle/make_golden_shell.py)."""
import os

import numpy as np
import pytest

from oracle import shell_ode

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "shell.npz"))
CASES = [str(c) for c in Z["rhs_case_names"]]


def same(got, want, rtol):
    got, want = np.asarray(got, float), np.asarray(want, float)
    fin = np.isfinite(want)
    assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)])      # -inf beyond the escape radius
    assert np.all(np.abs(got[fin] - want[fin]) <= rtol * np.abs(want[fin])), np.max(np.abs(got[fin] / want[fin] - 1))
    return int(fin.sum())


@pytest.mark.parametrize("case", CASES)
def test_rhs_matches_reference(case):
    aging, cutoff, static = Z["rhs_case_flags"][CASES.index(case)]
    m = shell_ode.Model(Z, aging=bool(aging), tau_cutoff=cutoff, static_momentum=bool(static))
    want = Z["rhs_" + case]
    got = np.array([shell_ode.rhs(Z["rhs_r"][j], (Z["rhs_v"][j], Z["rhs_t"][j]), m, shell_ode.Region(Z, int(Z["rhs_region"][j])))
                    for j in range(len(want))])
    n = same(got[:, 0], want[:, 0], 1e-12)
    same(got[:, 1], want[:, 1], 1e-14)
    assert n > 0.7 * len(want)  # the rest: -inf (escaped) or no reference value (its interpn raised)


def test_cutoff_changes_the_thin_end():
    assert not np.array_equal(Z["rhs_plain"][:, 0], Z["rhs_cutoff"][:, 0])


def test_trajectories_match_reference():
    """The restatement integrated by the same scipy method reproduces the reference right-hand side's trajectories.
    The curve is piecewise linear in tau, so the integrators' error estimates are optimistic: the stored
    trajectories (rtol 1e-9) are within 2e-7 of the converged solution, which is what is integrated here."""
    n_out = int(Z["n_out"])
    for k in range(len(Z["traj_region"])):
        i, aging = int(Z["traj_region"][k]), bool(Z["traj_aging"][k])
        m, q = shell_ode.Model(Z, aging=aging), shell_ode.Region(Z, i)
        v, t, r_stop, _, _ = shell_ode.integrate_region(m, q, q.h_g_i + float(Z["span_pc"]) * shell_ode.CM_PC, float(Z["v_0"]),
                                                        float(Z["t_0"]), n_out, rtol=1e-12)
        want_v, want_t = Z["traj_v"][k], Z["traj_t"][k]
        ok = np.isfinite(want_v) & np.isfinite(v)
        assert ok.sum() >= n_out // 2
        np.testing.assert_allclose(v[ok], want_v[ok], rtol=1e-6, atol=5e-6 * np.nanmax(want_v))  # a stalling shell: v -> 0
        np.testing.assert_allclose(t[ok], want_t[ok], rtol=5e-6)
        assert abs(r_stop - Z["traj_rstop"][k]) <= 1e-3 * Z["traj_rstop"][k]
