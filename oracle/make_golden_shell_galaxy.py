"""Generates tests/golden/shell_galaxy.npz: the UNMODIFIED reference ODEs (MC code.py:905-955) with the module
flag includeWholeGalaxy set, i.e. with the light and the gravity of the galaxy's other regions (:919-920, :940-941).

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference).  gal_data is the first 48 rows
of the reference's 'M51 all region velocity 5 pc starting 1 Myr.csv' (DATA).  Its two per-view columns,
LHa_ratio_f and dist_to_ref, exist in the reference only as commented set-up lines (MC code.py:1678-1683); they are
formed here by the same recipe for three choices of region_ID and stored, so what this file pins is ODEs itself,
not that recipe.  The other inputs (curve, luminosity per age, region columns) are those of shell.npz and are
read from it.

Stored: the views, the reference's right-hand side at sample points for {no aging, aging}, and one trajectory of
the reference right-hand side (DOP853, rtol 1e-9, max_step 1 pc) sampled at 12 radii.

Usage:  python oracle/make_golden_shell_galaxy.py      (after make_golden_shell.py)
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
from make_golden_shell import FLAGS, GOLDEN, N_OUT, REF, RegionData  # noqa: E402

N_GAL = 48
VIEW_ROWS = (0, 5, 20)       # rows of gal_data taken as region_ID in turn
DISTANCE_PC = 8.34 * 10 ** 6  # M51, MC code.py:1178
SPAN_PC = 40.0


def main():
    warnings.simplefilter("ignore")
    m = ref_loader.load("MC code.py", flags=FLAGS)
    cm_pc = m.cm_pc
    base = np.load(os.path.join(GOLDEN, "shell.npz"))
    mc = pd.read_csv(os.path.join(REF, "detailed MC mix spectra-bin-imf135_300 z020.csv"))
    ledd = pd.read_csv(os.path.join(REF, "L_Edd dataframe.csv"))
    gal = pd.read_csv(os.path.join(REF, "M51 all region velocity 5 pc starting 1 Myr.csv")).head(N_GAL).copy()
    m.MC_data = mc
    m.L_Edd_DF = ledd
    ages_l = ledd.time.to_numpy()
    bpass = pd.DataFrame({"WL": [1.0], **{str(a): [v] for a, v in zip(ages_l, ledd.L_bol_BPASS.to_numpy())}})
    grain = pd.DataFrame({"WL": [0.656], "Kappa_F": [12861.43], "Kappa_RP": [float(base["kappa_rp_ref"])]})
    _, _, ratio = m.GetRatios(ledd, grain, m.time_slice, m.grain_type, m.wl_ref)
    ratio = float(np.ravel(ratio)[0])
    m.ratio_rp_lambda_rp = ratio
    m.includeWholeGalaxy = True

    def view(row):
        # MC code.py:1682-1683 (commented there)
        ref = gal.iloc[row]
        lha = gal.LHa * np.exp(-gal.tau * ratio) / ref.LHa
        dist = DISTANCE_PC * np.sqrt(np.tan(np.radians(gal["RA"] - float(ref.RA))) ** 2
                                     + np.tan(np.radians(gal["Dec"] - float(ref.Dec))) ** 2)
        return lha.to_numpy(float), dist.to_numpy(float)

    views = [view(r) for r in VIEW_ROWS]

    def region(row):
        g = gal.iloc[row]
        q = RegionData()
        q.M_g, q.M_new, q.M_old = float(g.Mass_g), float(g.Mass_new), float(g.Mass_old)
        q.Sigma_g, q.tau_i, q.L_Bol, q.H_old = float(g.Sigma_g), float(g.tau), float(g.L_Bol), float(g.H_old)
        q.h_g_i = m.h_g * cm_pc
        q.M_CO_Hi = 0.0
        return q

    def rhs(r, X, k, aging):
        row = VIEW_ROWS[k]
        gd = gal[["ID", "Mass_tot"]].copy()
        gd["LHa_ratio_f"], gd["dist_to_ref"] = views[k]
        m.region_ID = int(gal.ID.iloc[row])
        dv, dt = m.ODEs(r, X, m.grain_type, 0.001, 1, bpass, m.time_slice, region(row), 0, grain, aging, gd)
        return [float(np.ravel(dv)[0]), float(dt)]

    out = dict(n_gal=N_GAL, view_rows=np.array(VIEW_ROWS), ratio_rp_lambda_rp=ratio, distance_pc=DISTANCE_PC,
               gal_ID=gal.ID.to_numpy(), gal_RA=gal.RA.to_numpy(float), gal_Dec=gal.Dec.to_numpy(float),
               gal_LHa=gal.LHa.to_numpy(float), gal_Mass_tot=gal.Mass_tot.to_numpy(float),
               view_LHa_ratio_f=np.stack([v[0] for v in views]), view_dist_to_ref=np.stack([v[1] for v in views]),
               **{"gal_" + k: gal[k].to_numpy(float) for k in ("Mass_g", "Mass_new", "Mass_old", "H_old", "tau", "L_Bol",
                                                               "Sigma_g")})

    rs = np.random.RandomState(11)
    per = 10
    idx = np.repeat(np.arange(len(VIEW_ROWS)), per)
    r_s = m.h_g * cm_pc * 10 ** rs.uniform(0, 2.5, len(idx))
    v_s = 10 ** rs.uniform(1, 6.5, len(idx))
    t_s = 10 ** rs.uniform(6.0, 9.2, len(idx))
    t_s[::per] = 1e7
    r_s[1::per] = 12000 * cm_pc
    out.update(rhs_view=idx, rhs_r=r_s, rhs_v=v_s, rhs_t=t_s)
    for name, aging in (("plain", False), ("aging", True)):
        vals = np.full((len(idx), 2), np.nan)
        for j in range(len(idx)):
            try:
                vals[j] = rhs(r_s[j], [v_s[j], t_s[j]], idx[j], aging)
            except ValueError:  # interpn refuses points outside the curve's grid
                pass
        out["rhs_" + name] = vals
        print("rhs", name, np.nanmin(vals[:, 0]), np.nanmax(vals[:, 0]))

    # the same points with the flag off: how much the two terms change the right-hand side
    m.includeWholeGalaxy = False
    off = np.array([rhs(r_s[j], [v_s[j], t_s[j]], idx[j], False) for j in range(len(idx))])
    m.includeWholeGalaxy = True
    out["rhs_plain_flag_off"] = off
    print("largest relative change of dv/dr by the flag:",
          np.nanmax(np.abs(out["rhs_plain"][:, 0] - off[:, 0]) / np.abs(off[:, 0])))

    # one trajectory of the reference right-hand side
    q = region(VIEW_ROWS[0])
    r0, r1 = q.h_g_i, q.h_g_i + SPAN_PC * cm_pc
    grid = r0 + (np.arange(N_OUT) + 1.0) / N_OUT * (r1 - r0)
    sol = integrate.solve_ivp(rhs, [r0, r1], [m.v_0, m.t_0], args=(0, False), method="DOP853", rtol=1e-9,
                              atol=[1e-6, 1e-6], max_step=cm_pc, dense_output=True)
    ok = grid <= r0 + 0.98 * (sol.t[-1] - r0) if sol.status != 0 else np.ones(N_OUT, bool)
    v_tr, t_tr = np.full(N_OUT, np.nan), np.full(N_OUT, np.nan)
    if ok.any():
        y = sol.sol(grid[ok])
        v_tr[ok], t_tr[ok] = y[0], y[1]
    print("trajectory status", sol.status, "steps", len(sol.t), "r_stop/pc %.3f" % (sol.t[-1] / cm_pc),
          "vmax km/s %.4f" % (np.max(sol.y[0]) / 1e5))
    out.update(traj_v=v_tr, traj_t=t_tr, traj_rstop=sol.t[-1], traj_span_pc=SPAN_PC)

    path = os.path.join(GOLDEN, "shell_galaxy.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
