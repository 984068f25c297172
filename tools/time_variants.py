"""Kernel time of the headline sweep under other flag combinations (which kernel serves them and how fast): median of
7 launches each on a warmed-up GPU, relative to the first case (the production configuration)."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["n_photons"]
eng = rt.Engine(0)
eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
A = rt._abi
CASES = [dict(), dict(runtime_kernel=True), dict(rand_mu=1), dict(abs_only=1), dict(escaped_mu=True)]
# every slab combination: loop order x boundary x {hg, iso} x depth bins
for order in ("mc", "gray"):
    for boundary in ("reflect", "absorb", "reemit"):
        for scatter in ("hg", "iso"):
            for nd in (0, 32):
                kw = dict(boundary=boundary, scatter=scatter, n_depth_bins=nd)
                if order == "gray":
                    kw.update(order="gray", h=1.0)
                if kw != dict(boundary="reflect", scatter="hg", n_depth_bins=0):
                    CASES.append(kw)
# the exact Draine law and the shell
for boundary in ("reflect", "absorb", "reemit"):
    for nd in (0, 32):
        CASES.append(dict(scatter="draine_exact", alpha=0.3, boundary=boundary, n_depth_bins=nd))
for scatter in ("hg", "draine_exact"):
    for boundary in ("reflect", "absorb", "reemit"):
        for nd in (0, 32):
            CASES.append(dict(geometry="shell", shell_inner=1.0, scatter=scatter, alpha=0.3, boundary=boundary, n_depth_bins=nd))
# what still has no instantiation of its own
CASES += [dict(abs_only=1, boundary="absorb"), dict(escaped_mu=True, n_depth_bins=32), dict(order="gray", h=1.0, scatter="draine")]
ONLY = sys.argv[2].split(";") if len(sys.argv) > 2 else None  # only these kernels, e.g. "mc,reflect,hg,depth;gray,reemit,hg"
base_ms = None
warm(rt, eng, cfg)
for kw in CASES:
    prm = rt.make_params(cfg["tau_list"], n, **({"h": cfg["h"]} | kw))
    if ONLY and kw:
        eng.run(rt.make_params(cfg["tau_list"][:1], 100, **({"h": cfg["h"]} | kw)))
        if eng.last_kernel()[12:-1] not in ONLY:
            continue
    eng.run(prm)
    ms, res = median_kernel_ms(eng, prm)
    ev, fl = int(res.counts[:, A.CNT_SCATTER].sum()), int(res.counts[:, A.CNT_FLIGHTS].sum())
    rate = fl / ms / 1e6
    base_ms = base_ms or rate
    print(f"{str(kw):100s} {eng.last_kernel()[12:-1]:44s} {ms:8.3f} ms  flights {fl:.3e}  events {ev:.3e}  {rate:6.1f} Gflights/s ({rate / base_ms - 1:+.0%})  {ev / ms / 1e6:.1f} Gevents/s")
eng.close()
