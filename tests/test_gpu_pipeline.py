"""Tables -> curve -> shell velocities on one GPU without files (examples/pipeline.py)."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_pipeline_runs_and_is_consistent():
    spec = importlib.util.spec_from_file_location("pipeline_example", os.path.join(ROOT, "examples", "pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ages, curve, sol = mod.main(20000)
    assert list(ages) == [6.0, 7.0, 8.0] and curve.shape == (60, 3)
    assert np.all(np.isfinite(curve)) and np.all(curve[0] < 0.05) and np.all(curve[-1] > 0.5)   # thin ~ tau, thick ~ 1
    assert np.all(np.diff(curve[::6, 0]) > -0.02)                                                # grows with tau up to the noise of 2e4 photons
    assert np.all(sol.v_max >= 100.0) and np.all(sol.r_last >= 5 * 3.086e18) and np.all(sol.n_steps > 0)
