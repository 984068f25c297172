"""Concrete synthetic inputs for the benchmark / parity configurations.

The reference's production inputs (BPASS spectra, '<species> Average Values.csv') are not part of
its checkout, so every configuration is defined from what is available: the Draine tables and a
closed-form Planck spectrum run through the reference's own SED -> PDF conversion
(MC code.py:1024-1051).  No random seed or data file beyond the Draine tables is involved.
"""
import numpy as np

from . import tables

TAU_LIST_MC = np.logspace(-3, 2, 100, base=10)     # MC code.py:72
TAU_LIST_GRAY = np.logspace(-2, 2, 10, base=10)    # Numpy_random_walk.py:18 with num_atm = 10


def _finish(name, grain, tau_list, n_photons, **extra):
    wl = np.asarray(grain["WL"], dtype=float)
    sed = tables.planck_sed(wl)
    pdf, _ = tables.photon_pdf(wl, sed)
    out = dict(name=name, describe=extra.pop("describe", name), wl=wl, g=np.asarray(grain["g"], dtype=float), albedo=np.asarray(grain["Albedo"], dtype=float),
               scale=np.asarray(grain["Scale"], dtype=float), pdf=pdf, tau_list=np.asarray(tau_list, dtype=float),
               n_photons=int(n_photons), order="mc", boundary="reflect", scatter="hg", h=tables.H_PLANCK,
               monochromatic=0.0)
    out.update(extra)
    return out


def gray_1e5():
    """Config 1 (plumbing): gray dust, albedo 0.9 (Numpy_random_walk.py:13), g = 0.5, 10 atmospheres."""
    return dict(name="gray_1e5", wl=np.array([1.0]), g=np.array([0.5]), albedo=np.array([0.9]), scale=np.array([1.0]),
                pdf=None, tau_list=TAU_LIST_GRAY, n_photons=100000, order="gray", boundary="reflect", scatter="hg",
                h=1.0, monochromatic=1.0, h_g=None,
                describe="gray dust albedo 0.9 g 0.5, gray-script loop order, reflect + hg")


def sil_single_1e6(wl_list=None):
    """Config 2: one silicate grain size (0.1 um), wavelength-dependent albedo / g / Scale, 1e6 photons per
    atmosphere, 100 atmospheres; shell radius h_g = 5 pc only converts tau to a column density afterwards."""
    grain = tables.single_grain_data("Sil", 0.1, wl_list)
    return _finish("sil_single_1e6", grain, TAU_LIST_MC, 1000000, h_g=5,
                   describe="one silicate grain size (0.1 um), Planck(3e4 K) emission, reflect + hg")


def mix_2e7(wl_list=None, n_photons=20000000):
    """Config 3 (headline): Sil/Gra/SiC mix [0.5, 0.45, 0.05], MRN sizes 0.001-1 um, 2e7 photons per
    atmosphere, 100 atmospheres, h_g = 10 pc."""
    grain = tables.grain_data("mix", 0.001, 1, wl_list)
    return _finish("mix_2e7", grain, TAU_LIST_MC, n_photons, h_g=10,
                   describe="Sil/Gra/SiC MRN mix, Planck(3e4 K) emission, reflect + hg")


def five_species_1e9(wl_list=None, grain_mix=(0.2, 0.2, 0.2, 0.2, 0.2)):
    """Config 4: five species (Sil, Gra, SiC, SUV, PAH), 1e9 photons per atmosphere, h_g = 500 pc.
    SUV and PAH have no counterpart in the reference's table code (extension)."""
    grain = tables.grain_data("mix", 0.001, 1, wl_list, grain_mix=grain_mix)
    return _finish("five_species_1e9", grain, TAU_LIST_MC, 1000000000, h_g=500,
                   describe="Sil/Gra/SiC/SUV/PAH equal mix, Planck(3e4 K) emission, reflect + hg")


def species_radius_batch(wl_list=None, radii_pc=(5, 10, 15, 20, 500), total_histories=10 ** 10):
    """Config 5: 5 shell radii x {Sil, Gra, SiC, mix} x 100 optical depths as ONE sweep of 2000 atmospheres over
    four grain tables (set_tables with 2-D columns + table_of_atm); 1e10 photon histories split evenly.  The
    radius only enters the conversion tau -> column density afterwards (tau_to_radius), so each radius repeats
    the tau list; the atmosphere index still gives every atmosphere its own random streams."""
    kinds = ["Sil", "Gra", "SiC", "mix"]
    grains = [tables.grain_data(k, 0.001, 1, wl_list) for k in kinds]
    out = _finish("species_radius_batch", grains[0], np.tile(TAU_LIST_MC, len(radii_pc) * len(kinds)), 1,
                  describe="5 radii x {Sil, Gra, SiC, mix} x 100 optical depths in one sweep, Planck(3e4 K) emission, reflect + hg")
    for key, col in (("g", "g"), ("albedo", "Albedo"), ("scale", "Scale")):
        out[key] = np.stack([np.asarray(gr[col], dtype=float) for gr in grains])
    n_atm = len(out["tau_list"])
    out["table_of_atm"] = np.tile(np.repeat(np.arange(len(kinds)), len(TAU_LIST_MC)), len(radii_pc)).astype(np.int32)
    out["radius_of_atm"] = np.repeat(np.asarray(radii_pc, dtype=float), len(kinds) * len(TAU_LIST_MC))
    out["n_photons"] = int(total_histories // n_atm)
    out["h_g"] = None
    return out


WORKLOADS = {"gray_1e5": gray_1e5, "sil_single_1e6": sil_single_1e6, "mix_2e7": mix_2e7,
             "five_species_1e9": five_species_1e9, "species_radius_batch": species_radius_batch}


def tau_to_radius(tau, sigma_g, kappa_rp_ref, h_g, f_dg=1 / 100):
    """Shell radius (pc) at which a shell of initial column sigma_g (g cm^-2 at radius h_g) has
    reference optical depth tau: tau = Sigma_g (h_g / r)^2 kappa_RP f_dg  (MC code.py:924, :1350)."""
    return h_g * np.sqrt(sigma_g * kappa_rp_ref * f_dg / np.asarray(tau, dtype=float))
