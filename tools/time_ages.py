"""The reference's table of momentum transfer over (optical depth, stellar age): one runMC call per age (its own way,
MC code.py:1069-1071) against mc.runMC_ages, all ages in one launch.  Wall time of the whole call (tables, upload,
sweep, estimator) and the walk kernel's own time.  python tools/time_ages.py [n_ages] [num_photons]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")
tables = importlib.import_module("radiative-transfer_amd.tables")
mc = importlib.import_module("radiative-transfer_amd.mc")
n_ages = int(sys.argv[1]) if len(sys.argv) > 1 else 13
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
eng = rt.Engine(0)
wl = tables.reference_wavelength_grid()
grain = tables.grain_data("mix", 0.001, 1, wl, engine=eng)
ages = [round(6.0 + 0.1 * k, 1) for k in range(n_ages)]
seds = [tables.planck_sed(wl, 4.0e4 / (1.0 + 0.25 * k)) for k in range(n_ages)]  # stand-ins for the BPASS columns
for rep in range(2):
    eng.kernel_time(reset=True)
    t0 = time.perf_counter()
    for a, s in zip(ages, seds):
        mc.runMC("x.dat", a, n, wl, grain, np.zeros(100), sed=s, engine=eng)
    t1 = time.perf_counter()
    ms_seq, k_seq = eng.kernel_time(reset=True)
    out = mc.runMC_ages(ages, seds, n, wl, grain, engine=eng)
    t2 = time.perf_counter()
    ms_one, k_one = eng.kernel_time(reset=True)
    print(f"{n_ages} ages x 100 optical depths x {n} photons: one runMC per age {1e3 * (t1 - t0):.1f} ms wall "
          f"({k_seq} launches, {ms_seq * k_seq:.2f} ms of walk kernels); runMC_ages {1e3 * (t2 - t1):.1f} ms wall "
          f"({k_one} launch, {ms_one * k_one:.2f} ms)", flush=True)
eng.close()
