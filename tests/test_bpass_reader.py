"""The BPASS spectra reader (mc.read_bpass_spectra).  No spectra file ships with the reference (they are listed
as missing large files), so the reader cannot be pinned to one; what can be pinned is pinned: the 51 age names
are the `time` column of the reference's own 'L_Edd dataframe.csv' -- which the reference filled from the
column names of a spectra frame (read_BPASS.py:221-247) -- kept as l_ages in tests/golden/shell.npz, and the
number format is that of the BPASS tables the reference does ship (ionizing-*.dat: '   0.6000000E+01 ...')."""
import importlib
import os

import numpy as np
import pytest

mc = importlib.import_module("radiative-transfer_amd.mc")
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "shell.npz"))


def write_table(path, cols):
    with open(path, "w") as fh:
        for row in cols:
            fh.write("".join("%16.7E" % v for v in row) + "\n")


def test_age_names_are_the_reference_tables():
    assert [str(a) for a in Z["l_ages"]] == list(mc.BPASS_LOG_AGES)
    assert [float(a) for a in mc.BPASS_LOG_AGES] == list(Z["l_ages"])


def test_reads_a_table_in_bpass_format(tmp_path):
    wl = np.arange(1, 501, dtype=float)
    rs = np.random.RandomState(3)
    flux = 10 ** rs.uniform(-30, 8, (len(wl), 51))
    flux[7, 3] = 0.0
    p = tmp_path / "spectra-bin-imf135_300.z020.dat"
    write_table(p, np.column_stack([wl, flux]))
    spec = mc.read_bpass_spectra(str(p))
    assert list(spec) == ["WL"] + list(mc.BPASS_LOG_AGES)
    assert np.array_equal(spec["WL"], wl)
    np.testing.assert_allclose(spec["6.3"], flux[:, 3], rtol=6e-8)   # 7 digits in the file
    np.testing.assert_allclose(spec["11.0"], flux[:, 50], rtol=6e-8)
    assert spec["6.3"][7] == 0.0


@pytest.mark.parametrize("bad", ["columns", "order", "nan"])
def test_refuses_what_is_not_a_spectra_table(tmp_path, bad):
    wl = np.arange(1, 21, dtype=float)
    cols = np.column_stack([wl] + [wl * (i + 1) for i in range(51)])
    if bad == "columns":
        cols = cols[:, :5]      # e.g. an ionizing-*.dat table
    elif bad == "order":
        cols[[3, 4]] = cols[[4, 3]]
    else:
        cols[2, 9] = np.nan
    p = tmp_path / "x.dat"
    write_table(p, cols)
    with pytest.raises(ValueError):
        mc.read_bpass_spectra(str(p))
