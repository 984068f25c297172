"""Run-to-run spread of the kernel time: the headline sweep launch after launch from a cold start (the GPU's clock
ramps up over the first launches), then a few flag combinations 12 times each on the warm GPU."""
import importlib
import sys

sys.path.insert(0, ".")
rt = importlib.import_module("radiative-transfer_amd")
w = importlib.import_module("radiative-transfer_amd.workloads")
cfg = w.WORKLOADS["mix_2e7"]()
eng = rt.Engine(0)
eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])


def times(n, reps, **kw):
    ts = []
    for i in range(reps):
        prm = rt.make_params(cfg["tau_list"], n, **({"h": cfg["h"]} | kw), photon_id_base=i * n)
        eng.kernel_time(reset=True)
        eng.run(prm)
        ts.append(eng.kernel_time()[0])
    return eng.last_kernel(), " ".join("%.2f" % t for t in ts)


print("2e7 photons, 30 launches from cold:", *times(20000000, 30), flush=True)
for kw in (dict(), dict(boundary="reemit"), dict(boundary="reemit", order="gray", h=1.0), dict(boundary="reemit", scatter="iso"),
           dict(n_depth_bins=32), dict(boundary="absorb"), dict()):
    print("4e6 photons:", *times(4000000, 12, **kw), flush=True)
