import importlib, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from _warm import median_kernel_ms, warm
rt = importlib.import_module("radiative-transfer_amd"); w = importlib.import_module("radiative-transfer_amd.workloads")
cfg = w.WORKLOADS["mix_2e7"](); eng = rt.Engine(0); eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
A = rt._abi
warm(rt, eng, cfg)
for kw in (dict(), dict(geometry="shell", shell_inner=1.0), dict(scatter="draine_exact", alpha=0.3), dict(abs_only=1), dict(boundary="absorb", n_depth_bins=32),
           dict(boundary="reemit"), dict(n_depth_bins=32), dict(order="gray", h=1.0), dict(escaped_mu=True)):
    out = []
    for rk in (True, False):
        prm = rt.make_params(cfg["tau_list"], cfg["n_photons"], **({"h": cfg["h"]} | kw), runtime_kernel=rk)
        eng.run(prm)
        ms, res = median_kernel_ms(eng, prm, 5)
        out.append("%s %.2f ms %.1f Gfl/s" % (eng.last_kernel()[12:-1], ms, res.counts[:, A.CNT_FLIGHTS].sum() / ms / 1e6))
    print(kw, " | ".join(out), flush=True)
