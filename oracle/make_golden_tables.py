"""Golden vectors for the table builder, from the UNMODIFIED reference functions
``GetKappas`` / ``GetTauScaling`` (read_BPASS.py:56-184).  TEST INFRASTRUCTURE ONLY; runs in
the build container (needs /root/reference).

The reference lists the grain files with ``os.listdir`` and integrates in that order
(read_BPASS.py:75,159-162); on this file system the order is arbitrary and the result is
wrong (negative opacities), so the module's ``os`` is replaced by a proxy whose ``listdir``
returns the names sorted by grain size -- the order on the author's machine.

Writes tests/golden/tables.npz: Kappa_RP, Kappa_F, Scale, ExtScale on every 50th point of the
0.001..10 um grid and the SED-averaged opacities, for Sil / Gra / SiC, a in [0.001, 1] um,
flat and Planck(3e4 K) spectra; plus the constant-Q known answer the reference's own
'abs 1 lambda 0_6 0_001 1 Sil.csv' holds (79318.98815495615 cm^2/g).
"""
import importlib
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_loader  # noqa: E402

tables = importlib.import_module("radiative-transfer_amd.tables")


class _OsProxy:
    def __getattr__(self, name):
        return getattr(os, name)

    @staticmethod
    def listdir(folder):
        names = os.listdir(folder)
        return sorted(names, key=lambda n: float(n.split("_")[-2] + "." + n.split("_")[-1]))


def main():
    warnings.simplefilter("ignore")
    mod = ref_loader.load_read_bpass()
    mod.os = _OsProxy()
    wl = tables.reference_wavelength_grid()
    seds = {"flat": np.ones(len(wl)), "planck": tables.planck_sed(wl)}
    out = {"wl_stride": np.array(50), "abs1_kappa_sil": np.array(
        pd.read_csv("/root/reference/abs 1 lambda 0_6 0_001 1 Sil.csv").Kappa.iloc[0])}
    for species in ("Sil", "Gra", "SiC"):
        folder = f"/root/reference/Draine data {species}/"
        for sname, sed in seds.items():
            spectra = pd.DataFrame({"WL": wl, "6.0": sed})
            df, kav_rp, kav_f = mod.GetTauScaling(folder, 0.001, 1, wl, spectra, 6.0, 0.656)
            p = f"{species}/{sname}/"
            for col in ("Kappa_RP", "Kappa_F", "Scale", "ExtScale"):
                out[p + col] = df[col].to_numpy()[::50]
            out[p + "kappa_av_RP"] = np.array(kav_rp)
            out[p + "kappa_av_F"] = np.array(kav_f)
            i = int(np.where(np.round(wl, 4) == 0.656)[0][0])
            print(species, sname, "Kappa_RP(0.656)", df.Kappa_RP[i], "Kappa_F", df.Kappa_F[i], kav_rp, kav_f)
    # single grain size branch (len(a) == 1)
    spectra = pd.DataFrame({"WL": wl, "6.0": seds["flat"]})
    # (GetTauScaling itself fails for one size: GetKappas then returns (1, n_wl) arrays, read_BPASS.py:166-178)
    _, _, k_rp, k_f, _, a = mod.GetKappas("/root/reference/Draine data Sil/", 0.1, 0.1, wl, spectra, 6.0)
    out["Sil/single0p1/Kappa_RP"] = np.asarray(k_rp).reshape(-1)[::50]
    out["Sil/single0p1/Kappa_F"] = np.asarray(k_f).reshape(-1)[::50]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tables.npz"), **out)


if __name__ == "__main__":
    main()
