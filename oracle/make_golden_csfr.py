"""Golden vectors for the continuous-star-formation composite spectrum.

TEST INFRASTRUCTURE ONLY; runs only where the reference checkout is mounted.  Calls the UNMODIFIED reference
function ContinuousStarFormationSED (MC code.py:1075-1088) with the bin widths of MC code.py:1233-1239 on a
seeded synthetic spectra table (the BPASS files are absent) and stores a subsample of the rows.  The test
regenerates the inputs from the same seed.
"""
import os

import numpy as np

from oracle import ref_loader

N_WL, N_AGE, SEED, STRIDE = 100000, 51, 20260111, 997   # the reference hard-codes 100000 wavelengths


def synthetic_spectra():
    rng = np.random.default_rng(SEED)
    base = rng.random((N_AGE, N_WL)) * np.logspace(3, -2, N_AGE)[:, None]
    return {f"{6.0 + 0.1 * j:.1f}": base[j] for j in range(N_AGE)}


def main():
    ref = ref_loader.load_mc_code()
    spectra = synthetic_spectra()
    time_list = np.array(sorted(spectra, key=float))
    time_list_exp = np.power(10, time_list.astype(float))
    delta_t = np.zeros_like(time_list_exp)
    for i in range(len(delta_t)):
        delta_t[i] = 10 ** 6.05 if i == 0 else 10 ** (6.15 + 0.1 * (i - 1)) - 10 ** (6.05 + 0.1 * (i - 1))
    out = {"delta_t": delta_t, "stride": STRIDE}
    for rate in (1, 3.5):
        cols = [ref.ContinuousStarFormationSED(spectra, time_list_exp, time_list, delta_t, age, rate)[::STRIDE]
                for age in time_list_exp]
        out[f"csfr_rate_{rate}"] = np.stack(cols)
    ages_off_grid = np.array([2.0e6, 10 ** 7.33, 5.0e5, 3e11])
    out["off_grid_ages"] = ages_off_grid
    out["off_grid"] = np.stack([ref.ContinuousStarFormationSED(spectra, time_list_exp, time_list, delta_t, a, 1)[::STRIDE]
                                for a in ages_off_grid])
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "csfr.npz"), **out)


if __name__ == "__main__":
    main()
