"""The gray script's top-level sweep (Numpy_random_walk.py:410-427: five g values x 100 optical depths) as one launch
(examples/gray_sweep.py), against the CPU oracle on the same streams."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT
from test_gpu_walk import compare

pytestmark = pytest.mark.gpu


def _example():
    spec = importlib.util.spec_from_file_location("gray_sweep_example", os.path.join(ROOT, "examples", "gray_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("scatter", ["hg", "iso"])
def test_gray_sweep_equals_the_oracle(rt, engine, c_oracle, scatter):
    mod = _example()
    n = 3000
    hg, tau_list, momentum, res = mod.gray_sweep(engine, n, scatter=scatter)
    assert momentum.shape == (5, 100) and np.all(np.isfinite(momentum))
    assert engine.last_kernel() == f"walk_kernel<gray,reflect,{scatter}>"
    n_g = len(hg)
    kw = dict(boundary="reflect", scatter=scatter, order="gray", h=1.0, monochromatic=1.0, alpha=0.3,
              table_of_atm=np.repeat(np.arange(n_g), len(tau_list)))
    ref = c_oracle.run(c_oracle.make_params(np.tile(tau_list, n_g), n, **kw), np.array([1.0]), hg[:, None],
                       np.full((n_g, 1), 0.9), np.ones((n_g, 1)), None)
    compare(res, ref, rt._abi)
    # the script's estimator, thin and thick ends: ~ (1 - albedo (1 + g) / 2 ...) tau for tau << 1, order one beyond
    assert np.all(momentum[:, 0] < 0.02) and np.all(momentum[:, -1] > 0.3)
    if scatter == "hg":   # forward scattering carries momentum out: less is transferred at large g, at every depth
        assert np.all(momentum[0, 30:] > momentum[-1, 30:])
