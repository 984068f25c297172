#!/usr/bin/env python3
"""The sweep at the bottom of the reference's gray-dust script (Numpy_random_walk.py:410-427) as ONE launch:
for every Henyey-Greenstein parameter g of `hg` and every optical depth of `tau_list`, the momentum transferred
(initial - escaped + boundary) / initial.  The reference calls RunPhotons 5 x 100 times; here the five (albedo, g)
pairs are five one-row tables and every (g, tau) pair is one atmosphere of the sweep.

    python examples/gray_sweep.py [num_photons] [scatter]

Defaults are the script's own (Numpy_random_walk.py:10-23): 10 000 photons, 100 atmospheres, albedo 0.9,
g = linspace(-0.9, 0.9, 5), boundary 'reflect'; scatter defaults to 'hg' ('draine' is the script's setting and
runs too, through the reference's own rejection sampler, which is slow by construction).
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")


def gray_sweep(eng, num_photons=10000, num_atm=100, albedo=0.9, hg=None, boundary="reflect", scatter="hg", alpha=0.3,
               seed=rt.DEFAULT_SEED):
    hg = np.linspace(-0.9, 0.9, num=5) if hg is None else np.asarray(hg, dtype=float)
    tau_list = np.logspace(-2, 2, num_atm, base=10)                      # Numpy_random_walk.py:18
    n_g = len(hg)
    # one table per g: a single wavelength (lambda = 1), scale 1, the script's scalar albedo
    eng.set_tables(np.array([1.0]), hg[:, None], np.full((n_g, 1), float(albedo)), np.ones((n_g, 1)))
    prm = rt.make_params(np.tile(tau_list, n_g), num_photons, boundary=boundary, scatter=scatter, order="gray", h=1.0,
                         monochromatic=1.0, alpha=alpha, table_of_atm=np.repeat(np.arange(n_g), num_atm), seed=seed)
    res = eng.run(prm)
    return hg, tau_list, res.momentum_fraction().reshape(n_g, num_atm), res   # momentum_method_1 per g


def main(num_photons=10000, scatter="hg"):
    eng = rt.Engine(0)
    hg, tau_list, momentum, res = gray_sweep(eng, num_photons, scatter=scatter)
    print("log10 tau:      " + " ".join(f"{np.log10(t):7.2f}" for t in tau_list[::11]))
    for g, row in zip(hg, momentum):
        print(f"g = {g:5.2f}:      " + " ".join(f"{v:7.4f}" for v in row[::11]))
    ms, _ = eng.kernel_time()
    print(f"{len(hg) * len(tau_list)} atmospheres x {num_photons} photons, {int(res.scatter_count.sum())} scatterings, "
          f"walk kernel {ms:.3f} ms ({eng.last_kernel()})")
    eng.close()
    return hg, tau_list, momentum


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10000, sys.argv[2] if len(sys.argv) > 2 else "hg")
