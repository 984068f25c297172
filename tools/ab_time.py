"""A/B kernel timing of builds of the same ABI: python tools/ab_time.py lib1.so lib2.so ...  (headline sweep,
walk-kernel time from the library's own HIP events; one subprocess per library, interleaved twice)."""
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
    rt = importlib.import_module("radiative-transfer_amd")
    workloads = importlib.import_module("radiative-transfer_amd.workloads")
    cfg = workloads.WORKLOADS["mix_2e7"]()
    eng = rt.Engine(0)
    if os.environ.get("RT_AB_UNIFORM"):  # identical records: index hacks in experimental builds leave the physics unchanged
        for k, v in (("g", 0.6), ("albedo", 0.7), ("scale", 1.0)):
            cfg[k] = np.full_like(cfg[k], v)
        cfg["wl"] = np.full_like(cfg["wl"], 0.5)
    eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
    prm = rt.make_params(cfg["tau_list"], cfg["n_photons"], h=cfg["h"], rng=os.environ.get("RT_AB_RNG", "philox52"))
    for _ in range(2):
        eng.run(prm)
    eng.kernel_time(reset=True)
    for _ in range(5):
        res = eng.run(prm)
    ms, n = eng.kernel_time()
    print(f"{os.environ.get('RT_AMD_LIB', 'in-tree'):40s} {ms / n:8.3f} ms  events {int(res.counts[:, 0].sum()):.4e}", flush=True)
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one()
    else:
        for _ in range(2):
            for lib in sys.argv[1:]:
                subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, RT_AMD_LIB=os.path.abspath(lib)), check=True)
