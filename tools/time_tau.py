"""Walk-kernel time per optical depth: 50 atmospheres of ONE tau each on the headline tables, so that every wave
sees the same mix of emitting and scattering lanes (median of 7 launches on a warmed-up GPU).  python tools/time_tau.py [photons per atmosphere] ["dict(flag=value, ...)"]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _warm import median_kernel_ms, warm  # noqa: E402  (tools/ is the script's own directory)

rt = importlib.import_module("radiative-transfer_amd")
workloads = importlib.import_module("radiative-transfer_amd.workloads")
cfg = workloads.WORKLOADS["mix_2e7"]()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4000000
kw = eval(sys.argv[2]) if len(sys.argv) > 2 else {}  # other flags, e.g. "dict(n_depth_bins=32)"
eng = rt.Engine(0)
eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
warm(rt, eng, cfg)
print(f"{'tau':>8s} {'flights/photon':>15s} {'ms':>9s} {'Gflights/s':>11s} {'Gevents/s':>10s}")
for tau in (1e-3, 1e-2, 0.1, 0.3, 1.0, 3.0, 10.0, 100.0):
    prm = rt.make_params(np.full(50, tau), n, **({"h": cfg["h"]} | kw))
    eng.run(prm)
    ms, res = median_kernel_ms(eng, prm)
    c = res.counts
    fl, ph, ev = (c[:, k].sum() for k in (rt._abi.CNT_FLIGHTS, rt._abi.CNT_PHOTONS, rt._abi.CNT_SCATTER))
    print(f"{tau:8.3g} {fl / ph:15.3f} {ms:9.3f} {fl / ms / 1e6:11.1f} {ev / ms / 1e6:10.1f}", flush=True)
eng.close()
