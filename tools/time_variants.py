"""Kernel time of the headline sweep under other flag combinations (which kernel serves them and how fast)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")
workloads = importlib.import_module("radiative-transfer_amd.workloads")
cfg = workloads.WORKLOADS["mix_2e7"]()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
eng = rt.Engine(0)
eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
A = rt._abi
for kw in (dict(), dict(boundary="absorb"), dict(boundary="reemit"), dict(scatter="iso"), dict(order="gray", h=1.0),
           dict(n_depth_bins=32), dict(rand_mu=1), dict(abs_only=1), dict(scatter="draine_exact", alpha=0.3),
           dict(geometry="shell", shell_inner=1.0), dict(geometry="shell", shell_inner=0.05, n_depth_bins=32),
           dict(geometry="shell", shell_inner=1.0, scatter="draine_exact", alpha=0.3)):
    prm = rt.make_params(cfg["tau_list"], n, **({"h": cfg["h"]} | kw))
    eng.run(prm)
    eng.kernel_time(reset=True)
    res = eng.run(prm)
    ms, _ = eng.kernel_time()
    ev, fl = int(res.counts[:, A.CNT_SCATTER].sum()), int(res.counts[:, A.CNT_FLIGHTS].sum())
    print(f"{str(kw):72s} {eng.last_kernel()[12:-1]:28s} {ms:8.3f} ms  flights {fl:.3e}  events {ev:.3e}  {fl / ms / 1e6:.1f} Gflights/s  {ev / ms / 1e6:.1f} Gevents/s")
eng.close()
