"""Times rt_shell_velocity on a galaxy-sized batch: the 16 regions of tests/golden/shell.npz tiled to 2080
(the reference's M51 table has 2075), integrated from h_g to 100 H_old like GetRegionVelocity."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")
shell = importlib.import_module("radiative-transfer_amd.shell")
Z = np.load(os.path.join(ROOT, "tests", "golden", "shell.npz"))
reps = 130
gal = {k[4:]: np.tile(Z[k], reps) * (1 + 1e-3 * np.random.RandomState(1).rand(16 * reps)) for k in Z.files if k.startswith("reg_")}
eng = rt.Engine(0)
for aging in (False, True):
    model = shell.ShellModel(Z["mc_tau"], Z["mc_ages"], Z["mc_values"], time_slice=6.0, stellar_aging=aging,
                             l_ages=Z["l_ages"], l_values=Z["l_values"])
    regions = shell.ShellRegions.from_gal_data(gal, h_g=5.0)
    for rtol in (1e-3, 1e-8):
        shell.shell_velocity(eng, model, regions, rtol=rtol)
        t0 = time.perf_counter()
        sol = shell.shell_velocity(eng, model, regions, rtol=rtol)
        dt = time.perf_counter() - t0
        print(f"aging={aging} rtol={rtol:g}: {regions.n} regions in {dt * 1e3:.1f} ms; steps per region median "
              f"{int(np.median(sol.n_steps))} max {sol.n_steps.max()}; status counts {np.bincount(sol.status, minlength=4)}")

# with the light and gravity of all other regions (the reference's includeWholeGalaxy): the 48 rows of
# tests/golden/shell_galaxy.npz tiled to 2064 regions scattered over the same patch of sky, every region seen from
# its own place (2064 views of 2064 rows)
ZG = np.load(os.path.join(ROOT, "tests", "golden", "shell_galaxy.npz"))
reps, rs = 43, np.random.RandomState(2)
big = {k[4:]: np.tile(ZG[k], reps).astype(float) for k in ZG.files if k.startswith("gal_")}
n = len(big["tau"])
big["RA"] = big["RA"] + 0.05 * rs.rand(n)
big["Dec"] = big["Dec"] + 0.05 * rs.rand(n)
for n_reg in (48, n):
    rows = np.arange(n_reg)
    t0 = time.perf_counter()
    galaxy = shell.ShellGalaxy.from_gal_data(big, rows, float(ZG["ratio_rp_lambda_rp"]), float(ZG["distance_pc"]))
    t_views = time.perf_counter() - t0
    regions = shell.ShellRegions.from_gal_data({k: v[rows] for k, v in big.items()}, h_g=5.0)
    for aging in (False, True):
        model = shell.ShellModel(Z["mc_tau"], Z["mc_ages"], Z["mc_values"], time_slice=6.0, stellar_aging=aging,
                                 l_ages=Z["l_ages"], l_values=Z["l_values"], galaxy=galaxy)
        shell.shell_velocity(eng, model, regions)
        t0 = time.perf_counter()
        sol = shell.shell_velocity(eng, model, regions)
        dt = time.perf_counter() - t0
        print(f"whole galaxy, aging={aging}: {regions.n} regions x {n} rows in {dt * 1e3:.1f} ms (views built on the host in "
              f"{t_views * 1e3:.0f} ms); steps per region median {int(np.median(sol.n_steps))} max {sol.n_steps.max()}; "
              f"status counts {np.bincount(sol.status, minlength=4)}", flush=True)
eng.close()
