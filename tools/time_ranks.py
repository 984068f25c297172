"""Strong-scaling shape of the headline sweep measured on ONE GPU: the share of every rank r of a W-rank job
(make_params(rank=r, world=W)) is walked alone and timed with the library's own HIP events; the projected
efficiency of W GPUs is T1 / (W * max_r T_rW).  (The collective, 100 KB once per step, is not in it.)

    python tools/time_ranks.py [photons per atmosphere] > profiles/<round>_strong_scaling.txt
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")
workloads = importlib.import_module("radiative-transfer_amd.workloads")
cfg = workloads.WORKLOADS["mix_2e7"]()
n = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["n_photons"]
eng = rt.Engine(0)
eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])


def share(rank, world, reps=5):
    prm = rt.make_params(cfg["tau_list"], n, h=cfg["h"], rank=rank, world=world)
    eng.run(prm)
    eng.kernel_time(reset=True)
    for _ in range(reps):
        eng.run(prm)
    ms, k = eng.kernel_time()
    return ms


t1 = share(0, 1)
print(f"mix_2e7, {n} photons per atmosphere in total, walk-kernel time per rank share (ms), 8 slabs per atmosphere")
print(f"W=1: {t1:.3f}")
for world in (2, 4, 8):
    ts = [share(r, world) for r in range(world)]
    print(f"W={world}: " + " ".join(f"{t:.3f}" for t in ts) + f"   max {max(ts):.3f}  projected efficiency T1/(W*max) = {t1 / (world * max(ts)):.3f}")
eng.close()
