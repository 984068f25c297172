"""Scatter-quorum experiment with an experiment build of the library (-DRT_ABLATION; RT_AMD_LIB names it,
RT_AMD_ALLOW_ABLATION=1): walk-kernel time of the headline sweep and of single-tau sweeps for several quorums.
    python tools/quorum_sweep.py 0 8 16 24 32"""
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
    eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])

    def t(prm, reps):
        for _ in range(2):
            eng.run(prm)
        eng.kernel_time(reset=True)
        for _ in range(reps):
            eng.run(prm)
        return eng.kernel_time()[0]

    out = [f"quorum {os.environ.get('RT_AMD_QUORUM', 'default'):>7s}: sweep {t(rt.make_params(cfg['tau_list'], cfg['n_photons'], h=cfg['h']), 6):7.3f} ms |"]
    for tau in (1e-3, 1e-2, 0.03, 0.1, 0.3, 1.0, 10.0):
        out.append(f"tau {tau:g}: {t(rt.make_params(np.full(50, tau), 4000000, h=cfg['h']), 3):6.3f}")
    print(" ".join(out), flush=True)
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one()
    else:
        for q in sys.argv[1:]:
            subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, RT_AMD_QUORUM=q), check=True)
