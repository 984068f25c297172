"""Chunk-length experiment with an experiment build of the library (-DRT_ABLATION; RT_AMD_LIB names it,
RT_AMD_ALLOW_ABLATION=1): RT_AMD_CHUNK=thin,thick is read at load, so one subprocess per rule:
walk-kernel time of the headline sweep on one rank and of two of the eight shares of an 8-rank job."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
    rt = importlib.import_module("radiative-transfer_amd")
    workloads = importlib.import_module("radiative-transfer_amd.workloads")
    cfg = workloads.WORKLOADS["mix_2e7"]()
    eng = rt.Engine(0)
    eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])

    def share(rank, world, reps):
        prm = rt.make_params(cfg["tau_list"], cfg["n_photons"], h=cfg["h"], rank=rank, world=world)
        eng.run(prm)
        eng.kernel_time(reset=True)
        for _ in range(reps):
            eng.run(prm)
        return eng.kernel_time()[0]

    share(0, 1, 3)  # clocks up
    a = share(0, 1, 6)
    w8 = [share(r, 8, 6) for r in (0, 3, 6)]
    w4 = [share(r, 4, 4) for r in (1,)]
    b = share(0, 1, 6)
    t1 = min(a, b)
    print(f"{os.environ.get('RT_AMD_CHUNK', 'default'):14s} W=1 {a:7.3f} {b:7.3f}   W=8 " + " ".join(f"{t:6.3f}" for t in w8)
          + f"  eff {t1 / 8 / max(w8):.3f}   W=4 {w4[0]:6.3f} eff {t1 / 4 / w4[0]:.3f}", flush=True)
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one()
    else:
        for rule in sys.argv[1:]:
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_ranks.py"), "20000000", "8"], env=dict(os.environ, RT_AMD_CHUNK=rule), check=True)
