"""One launch of the headline sweep per flag combination (for counter passes: rocprofv3 reports per kernel name).
python3 tools/run_cases.py [photons per atmosphere]; prints kernel name, flights and events of each case."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")
workloads = importlib.import_module("radiative-transfer_amd.workloads")
cfg = workloads.WORKLOADS["mix_2e7"]()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
eng = rt.Engine(0)
eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
A = rt._abi
CASES = [dict(), dict(n_depth_bins=32), dict(boundary="reemit"), dict(boundary="reemit", n_depth_bins=32), dict(boundary="absorb"),
         dict(scatter="iso"), dict(order="gray", h=1.0), dict(order="gray", h=1.0, scatter="iso"),
         dict(geometry="shell", shell_inner=1.0), dict(scatter="draine_exact", alpha=0.3), dict(escaped_mu=True), dict(abs_only=1)]
for kw in CASES:
    res = eng.run(rt.make_params(cfg["tau_list"], n, **({"h": cfg["h"]} | kw)))
    print(eng.last_kernel(), int(res.counts[:, A.CNT_FLIGHTS].sum()), int(res.counts[:, A.CNT_SCATTER].sum()), flush=True)
eng.close()
