"""Golden vectors for the Grain_data frame the walk consumes (WL, g, Albedo, Scale), from the reference's own
chain GetGrainKappas -> GetGrainData -> GetTauScaling -> GetKappas (MC code.py:204-216, read_BPASS.py:48-184), run
UNMODIFIED.  TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference).

GetGrainData reads '<species> Average Values.csv' from the working directory.  Those files are absent from the
reference checkout; the reference wrote them with the commented-out block MC code.py:270-324.  This script writes
them the same way -- the block's function (make_golden_draine_average.load_recipe: the reference's text with its
comment markers removed) on the full wavelength grid, the albedo of :320 and the column names of :322, `to_csv(...,
index=False)` as at :324 -- into a temporary directory that also links the Draine folders, and runs the reference's
GetGrainKappas there for grain_type 'mix' (weights 0.5 / 0.45 / 0.05, a in [0.001, 1] um, wl_ref 0.656) and for
'Sil', on the Planck(3e4 K) spectrum of the benchmark.  The two frames it returns are rounded and merged on WL as
at MC code.py:1226-1229.  As in make_golden_tables.py, the modules' `os.listdir` lists a folder in order of grain
size.

Writes tests/golden/grain_frame.npz: every 50th row of g, Albedo, Scale, Kappa_RP, Kappa_F, ExtScale for both.
"""
import importlib
import os
import sys
import tempfile
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden_draine_average as recipe  # noqa: E402
import ref_loader  # noqa: E402
from make_golden_tables import _OsProxy  # noqa: E402

tables = importlib.import_module("radiative-transfer_amd.tables")
STRIDE = 50


def main():
    warnings.simplefilter("ignore")
    wl = tables.reference_wavelength_grid()
    frame_r = pd.DataFrame({"WL": wl, "6.0": tables.planck_sed(wl)})  # BPASS_data_r: the spectrum on the micron grid
    source, _ = recipe.load_recipe()
    rb = ref_loader.load_read_bpass()
    rb.os = _OsProxy()
    mc = ref_loader.load_mc_code()
    mc.GetGrainData = rb.GetGrainData  # MC code.py takes it from read_BPASS (an import the loader cannot execute: hoki)
    out = {"wl_stride": np.array(STRIDE)}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        for species in ("Sil", "Gra", "SiC"):
            os.symlink(os.path.join(ref_loader.REFERENCE, f"Draine data {species}"), os.path.join(tmp, f"Draine data {species}"))
            av = recipe.run(source, recipe.by_size, os.path.join(ref_loader.REFERENCE, f"Draine data {species}"), wl)
            pd.DataFrame({"WL": frame_r.WL, "Sigma-Abs": av["Sigma-Abs"], "Sigma-Scatt": av["Sigma-Scatt"], "g": av["g"],
                          "Albedo": av["Albedo"], "Q_ext": av["Q_ext"]}).to_csv(os.path.join(tmp, f"{species} Average Values.csv"),
                                                                                index=False)  # MC code.py:322-324
        os.chdir(tmp)
        try:
            for grain_type in ("mix", "Sil"):
                grain, kappa = mc.GetGrainKappas(None, 6.0, frame_r, 0.656, [0.5, 0.45, 0.05], 0.001, 1, grain_type)
                kappa.WL = kappa.WL.round(4)      # MC code.py:1226-1229
                grain.WL = grain.WL.round(4)
                merged = pd.merge(grain, kappa, left_on="WL", right_on="WL")
                assert len(merged) == len(wl) and np.array_equal(merged.WL.to_numpy(), wl)
                for col in ("g", "Albedo", "Scale", "Kappa_RP", "Kappa_F", "ExtScale"):
                    out[f"{grain_type}/{col}"] = merged[col].to_numpy()[::STRIDE]
                i = int(np.where(wl == 0.656)[0][0])
                print(grain_type, "g, Albedo, Scale, Kappa_RP at 0.656 um:", merged.g[i], merged.Albedo[i], merged.Scale[i], merged.Kappa_RP[i])
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "grain_frame.npz"), **out)


if __name__ == "__main__":
    main()
