"""The device table builder (rt_build_grain_tables) against the host NumPy builder, which the CPU tests pin to
the reference's own CSV outputs (tests/test_tables.py).  Interpolation and the size integrals follow NumPy's
operation order, so the per-wavelength columns agree to rounding; tolerance 1e-13 relative.  The last test compares
the device-built frame with the reference's own directly."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

tables = importlib.import_module("radiative-transfer_amd.tables")
RTOL = 1e-13


def _close(a, b, rtol=RTOL):
    a, b = np.asarray(a), np.asarray(b)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    assert np.all(np.abs(a[ok] - b[ok]) <= rtol * np.abs(b[ok]) + 1e-300), float(np.nanmax(np.abs(a - b) / np.abs(b)))


@pytest.mark.parametrize("species", ["Sil", "Gra", "SiC", "SUV", "PAH"])
def test_species_columns(engine, species):
    gs = tables.load_species(species)
    wl = tables.reference_wavelength_grid()
    sed = tables.planck_sed(wl)
    dev = tables.grain_tables_device(engine, gs, 0.001, 1, wl, sed)
    k = tables.get_kappas(gs, 0.001, 1, wl, sed)
    av = tables.get_draine_average(gs, wl)
    _close(dev["Kappa_RP"], k["Kappa_RP"])
    _close(dev["Kappa_F"], k["Kappa_F"])
    for key in ("Sigma-Abs", "Sigma-Scatt", "g", "Q_ext", "Albedo"):
        _close(dev[key], av[key])
    # the luminosity-weighted means integrate in the other order (size first, then wavelength)
    assert dev["kappa_av_RP"] == pytest.approx(k["kappa_av_RP"], rel=1e-11)
    assert dev["kappa_av_F"] == pytest.approx(k["kappa_av_F"], rel=1e-11)
    assert dev["L"] == pytest.approx(k["L"], rel=1e-13)


def test_size_window_and_off_grid_wavelengths(engine):
    gs = tables.load_species("Sil")
    rs = np.random.RandomState(3)
    # beyond both ends of the tabulated range (np.interp clamps) and exactly on tabulated wavelengths
    wl = np.sort(np.concatenate([rs.uniform(1e-4, 2e3, 5000), gs.wl[3][::7], [gs.wl[0][0], gs.wl[0][-1]]]))
    dev = tables.grain_tables_device(engine, gs, 0.005, 0.25, wl, None, average_over_all_sizes=False)
    k = tables.get_kappas(gs, 0.005, 0.25, wl)
    av = tables.get_draine_average(gs, wl, 0.005, 0.25)
    _close(dev["Kappa_RP"], k["Kappa_RP"])
    _close(dev["Kappa_F"], k["Kappa_F"])
    for key in ("Sigma-Abs", "Sigma-Scatt", "g", "Q_ext"):
        _close(dev[key], av[key])


def test_single_size(engine):
    gs = tables.load_species("Gra")
    a0 = gs.a[len(gs.a) // 2]
    wl = tables.reference_wavelength_grid()[::10]
    dev = tables.grain_tables_device(engine, gs, a0, a0, wl, None, average_over_all_sizes=False)
    k = tables.get_kappas(gs, a0, a0, wl)
    av = tables.get_draine_average(gs, wl, a0, a0)
    _close(dev["Kappa_RP"], k["Kappa_RP"])
    for key in ("Sigma-Abs", "Sigma-Scatt", "g", "Q_ext"):
        _close(dev[key], av[key])


def test_grain_data_mix_matches_host(engine):
    wl = tables.reference_wavelength_grid()
    host = tables.grain_data("mix", wl_list=wl)
    dev = tables.grain_data("mix", wl_list=wl, engine=engine)
    for key in ("g", "Albedo", "Scale", "Kappa_RP", "Kappa_F"):
        _close(dev[key], host[key], 1e-12)


def test_rejects_bad_input(engine):
    gs = tables.load_species("Sil")
    rt = importlib.import_module("radiative-transfer_amd")
    with pytest.raises(rt.RtError):
        tables.grain_tables_device(engine, gs, 500.0, 600.0, np.array([0.5, 0.6]))


def test_l_edd_dataframe_on_device(engine):
    wl = tables.reference_wavelength_grid()[::10]
    spectra = {"6.0": tables.planck_sed(wl, 4.0e4), "6.5": tables.planck_sed(wl, 3.0e4), "9.0": tables.planck_sed(wl, 5.0e3)}
    host = tables.l_edd_dataframe(spectra, wl)
    dev = tables.l_edd_dataframe(spectra, wl, engine=engine)
    for key in host:
        _close(dev[key], host[key], 1e-12)


@pytest.mark.parametrize("grain_type", ["mix", "Sil"])
def test_device_built_frame_matches_the_reference_chain(engine, grain_type):
    """g, Albedo, Scale built on the device against the frame the reference's unmodified GetGrainKappas chain returns
    (tests/golden/grain_frame.npz, generator oracle/make_golden_grain_frame.py), with no host builder in between."""
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "grain_frame.npz"))
    stride = int(z["wl_stride"])
    wl = tables.reference_wavelength_grid()
    gd = tables.grain_data(grain_type, 0.001, 1, wl, sed=tables.planck_sed(wl), engine=engine)
    for col in ("Scale", "Kappa_RP", "Kappa_F", "ExtScale", "Albedo", "g"):
        _close(np.asarray(gd[col])[::stride], z[f"{grain_type}/{col}"], rtol=3e-12)
