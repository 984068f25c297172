"""Generates tests/golden/shell.npz by running the UNMODIFIED reference functions ODEs / GetMC /
get_tau_IR / GetRatios (MC code.py:781-955, 1090-1157) on real rows of the reference's own tables.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference).  Inputs taken from the
reference checkout as DATA: 16 regions of 'M51 all region velocity 5 pc starting 1 Myr.csv', the curve
'detailed MC mix spectra-bin-imf135_300 z020.csv' and 'L_Edd dataframe.csv'.  The BPASS spectra are absent,
so the summed luminosity per age that ODEs takes from BPASS_data (MC code.py:917) is fed from the
L_bol_BPASS column through a one-row stand-in frame.

Stored: the inputs, the reference's right-hand side at sample points for five flag combinations, and, for
three (region, aging) cases, trajectories of the reference right-hand side integrated with scipy's DOP853
(rtol 1e-9, max_step 1 pc) sampled at 12 radii.  The reference function costs milliseconds per call (pandas)
and the curve is piecewise linear in tau (thousands of kinks, each one a rejected step), so more cases than
that are integrated from the NumPy restatement oracle/shell_ode.py, which the same file pins.

Usage:  python oracle/make_golden_shell.py
"""
import os
import sys
import warnings

import numpy as np
import pandas as pd
from scipy import integrate

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = ref_loader.REFERENCE
FLAGS = ref_loader._MC_FLAGS | {"wl_ref", "time_slice", "grain_type", "grain_mix", "tau_cutoff", "h_g", "f_dg", "v_0",
                                "t_0", "h_Hi", "includeWholeGalaxy", "static_momentum", "num_bins", "region_ID"}
N_OUT, SPAN_PC = 12, 60.0


class RegionData:
    pass


def main():
    warnings.simplefilter("ignore")
    m = ref_loader.load("MC code.py", flags=FLAGS)
    cm_pc = m.cm_pc
    mc = pd.read_csv(os.path.join(REF, "detailed MC mix spectra-bin-imf135_300 z020.csv"))
    ledd = pd.read_csv(os.path.join(REF, "L_Edd dataframe.csv"))
    gal = pd.read_csv(os.path.join(REF, "M51 all region velocity 5 pc starting 1 Myr.csv"))
    m.MC_data = mc
    m.L_Edd_DF = ledd
    ages_l = ledd.time.to_numpy()
    bpass = pd.DataFrame({"WL": [1.0], **{str(a): [v] for a, v in zip(ages_l, ledd.L_bol_BPASS.to_numpy())}})
    grain = pd.DataFrame({"WL": [0.656], "Kappa_F": [12861.43], "Kappa_RP": [6504.17]})

    moving = gal[gal.velocity_no_aging > 0].head(9)
    still = gal[gal.velocity_no_aging == 0].head(3)
    rows = pd.concat([moving, still])
    cols = {k: rows[k].to_numpy(float) for k in ("Mass_g", "Mass_new", "Mass_old", "H_old", "tau", "L_Bol", "Sigma_g")}
    # four made-up regions: strongly super-Eddington, with molecular / atomic gas mass, thick, compact old disc
    extra = dict(Mass_g=[2e36, 5e36, 1e37, 3e36], Mass_new=[8e35, 3e36, 1e36, 2e36], Mass_old=[1e35, 4e35, 2e36, 1e34],
                 H_old=[3e21, 1e21, 5e20, 4e19], tau=[0.3, 2.5, 40.0, 0.02], L_Bol=[3e40, 2e41, 5e41, 8e40],
                 Sigma_g=[8e-4, 6e-3, 9e-2, 5e-5])
    cols = {k: np.concatenate([cols[k], np.asarray(extra[k], float)]) for k in cols}
    n = len(cols["tau"])
    cols["Mass_CO_Hi"] = np.concatenate([np.zeros(n - 4), [0.0, 2e36, 5e36, 1e35]])
    ref_out = {k: np.concatenate([rows[k].to_numpy(float), np.full(4, np.nan)])
               for k in ("velocity_no_aging", "distance_no_aging", "time_no_aging", "velocity_aging", "distance_aging",
                         "time_aging")}

    def region(i):
        q = RegionData()
        q.M_g, q.M_new, q.M_old = cols["Mass_g"][i], cols["Mass_new"][i], cols["Mass_old"][i]
        q.Sigma_g, q.tau_i, q.L_Bol, q.H_old = cols["Sigma_g"][i], cols["tau"][i], cols["L_Bol"][i], cols["H_old"][i]
        q.h_g_i = m.h_g * cm_pc
        q.M_CO_Hi = cols["Mass_CO_Hi"][i]
        return q

    def rhs(r, X, q, aging, cutoff):
        # the argument list of ODEs, MC code.py:905
        dv, dt = m.ODEs(r, X, m.grain_type, 0.001, 1, bpass, m.time_slice, q, cutoff, grain, aging, None)
        return [float(np.ravel(dv)[0]), float(dt)]  # interpn hands back a one-element array in the aging branch

    out = dict(mc_tau=mc.tau.to_numpy(), mc_ages=np.array([float(c) for c in mc.columns if c != "tau"]),
               mc_values=mc.drop("tau", axis=1).to_numpy(), l_ages=ages_l, l_values=ledd.L_bol_BPASS.to_numpy(),
               kappa_rp_ref=6504.17, h_g=float(m.h_g), h_Hi=float(m.h_Hi), time_slice=float(m.time_slice),
               v_0=float(m.v_0), t_0=float(m.t_0), cm_pc=cm_pc, n_out=N_OUT, span_pc=SPAN_PC,
               **{"kappa_av_RP_" + s: ledd["kappa_av_RP_" + s].to_numpy() for s in ("Sil", "Gra", "SiC")},
               **{"reg_" + k: v for k, v in cols.items()}, **{"refcsv_" + k: v for k, v in ref_out.items()})

    # ---- right-hand side samples ----
    rs = np.random.RandomState(7)
    per = 8
    idx = np.repeat(np.arange(n), per)
    r_s = m.h_g * cm_pc * 10 ** rs.uniform(0, 2.5, n * per)
    v_s = 10 ** rs.uniform(1, 6.5, n * per)
    t_s = 10 ** rs.uniform(6.0, 9.2, n * per)
    t_s[::per] = 1e7           # log10 = 7.0 exactly: a tabulated age
    t_s[1::per] = 10 ** 6.0    # the first age
    r_s[2::per] = 12000 * cm_pc  # beyond the escape radius
    out.update(rhs_region=idx, rhs_r=r_s, rhs_v=v_s, rhs_t=t_s)
    cases = {"plain": (False, 0, False), "aging": (True, 0, False), "cutoff": (False, 0.5, False),
             "aging_cutoff": (True, 0.5, False), "aging_static": (True, 0, True)}
    for name, (aging, cutoff, static) in cases.items():
        m.static_momentum = static
        vals = np.full((len(idx), 2), np.nan)
        for j in range(len(idx)):
            try:
                vals[j] = rhs(r_s[j], [v_s[j], t_s[j]], region(idx[j]), aging, cutoff)
            except ValueError:  # interpn refuses points outside the curve's grid; stored as NaN = no reference value
                pass
        out["rhs_" + name] = vals
        print("rhs", name, np.nanmin(vals[:, 0]), np.nanmax(vals[:, 0]))
    m.static_momentum = False
    out["rhs_case_names"] = np.array(list(cases))
    out["rhs_case_flags"] = np.array([[int(a), c, int(s)] for a, c, s in cases.values()], dtype=float)

    # ---- trajectories of the reference right-hand side ----
    traj_cases = [(0, False), (1, False), (0, True)]
    v_tr = np.full((len(traj_cases), N_OUT), np.nan)
    t_tr = np.full((len(traj_cases), N_OUT), np.nan)
    r_stop = np.full(len(traj_cases), np.nan)
    for k, (i, aging) in enumerate(traj_cases):
        q = region(i)
        r0, r1 = q.h_g_i, q.h_g_i + SPAN_PC * cm_pc
        grid = r0 + (np.arange(N_OUT) + 1.0) / N_OUT * (r1 - r0)
        sol = integrate.solve_ivp(rhs, [r0, r1], [m.v_0, m.t_0], args=(q, aging, 0), method="DOP853", rtol=1e-9,
                                  atol=[1e-6, 1e-6], max_step=cm_pc, dense_output=True)
        r_stop[k] = sol.t[-1]
        # keep only radii comfortably inside what was integrated (a stalling shell ends early)
        ok = grid <= r0 + 0.98 * (sol.t[-1] - r0) if sol.status != 0 else np.ones(N_OUT, bool)
        if ok.any():
            y = sol.sol(grid[ok])
            v_tr[k, ok], t_tr[k, ok] = y[0], y[1]
        print("trajectory", i, aging, "status", sol.status, "steps", len(sol.t), "r_stop/pc %.3f" % (sol.t[-1] / cm_pc),
              "vmax km/s %.4f" % (np.max(sol.y[0]) / 1e5), flush=True)
    out.update(traj_region=np.array([c[0] for c in traj_cases]), traj_aging=np.array([int(c[1]) for c in traj_cases]),
               traj_v=v_tr, traj_t=t_tr, traj_rstop=r_stop)

    np.savez_compressed(os.path.join(GOLDEN, "shell.npz"), **out)
    print("wrote", os.path.join(GOLDEN, "shell.npz"), os.path.getsize(os.path.join(GOLDEN, "shell.npz")))


if __name__ == "__main__":
    main()
