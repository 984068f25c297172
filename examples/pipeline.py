#!/usr/bin/env python3
"""End to end on one GPU, everything in memory: grain tables -> Monte Carlo curve for several stellar ages ->
shell velocities of a set of H II regions.  The reference does the same through CSV files and one scipy
integration per region (MC code.py:1165-1476).

    python examples/pipeline.py [photons_per_atmosphere]

Synthetic inputs: Planck spectra of decreasing temperature stand in for the BPASS ages (the spectra files
are not part of the reference checkout), and three made-up regions.
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")
tables = importlib.import_module("radiative-transfer_amd.tables")
mc = importlib.import_module("radiative-transfer_amd.mc")
shell = importlib.import_module("radiative-transfer_amd.shell")


def build_curve(eng, grain, wl, spectra, tau_list, n_photons):
    """One column of the curve per age: the estimator of runMC (MC code.py:1062) on the GPU."""
    columns = []
    for age in sorted(spectra, key=float):
        pdf, _ = tables.photon_pdf(wl, spectra[age])
        res = mc.run_sweep(tau_list, n_photons, "reflect", "hg", pdf, wl, grain, engine=eng)
        columns.append(res.momentum_fraction())
    return np.array([float(a) for a in sorted(spectra, key=float)]), np.stack(columns, axis=1)


def main(n_photons=200000):
    eng = rt.Engine(0)
    wl = tables.reference_wavelength_grid()
    grain = tables.grain_data("mix", 0.001, 1, wl, engine=eng)            # tables on the device
    spectra = {"6.0": tables.planck_sed(wl, 4.0e4), "7.0": tables.planck_sed(wl, 2.0e4), "8.0": tables.planck_sed(wl, 8.0e3)}
    tau_list = np.logspace(-3, 2, 60)
    ages, curve = build_curve(eng, grain, wl, spectra, tau_list, n_photons)
    lum = {a: float(np.sum(s)) for a, s in spectra.items()}
    model = shell.ShellModel(tau_list, ages, curve, time_slice=6.0, stellar_aging=True, l_ages=ages,
                             l_values=np.array([lum[a] for a in sorted(lum, key=float)]))
    regions = shell.ShellRegions(M_g=[2e36, 5e36, 1e37], M_new=[8e35, 3e36, 1e36], M_old=[1e35, 4e35, 2e36],
                                 H_old=[3e21, 1e21, 5e20], tau_i=[0.3, 2.5, 40.0], L_Bol=[3e40, 2e41, 5e41],
                                 Sigma_g=[8e-4, 6e-3, 9e-2], h_g=5.0)
    sol = shell.shell_velocity(eng, model, regions, rtol=1e-6)
    for i in range(regions.n):
        print(f"region {i}: v_max {sol.v_max[i] / 1e5:8.2f} km/s at {sol.r_vmax[i] / shell.CM_PC:8.1f} pc, "
              f"ends at {sol.r_last[i] / shell.CM_PC:9.1f} pc after {sol.t_last[i]:.3e} yr ({shell.STATUS_NAMES[int(sol.status[i])]})")
    eng.close()
    return ages, curve, sol


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200000)
