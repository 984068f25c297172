"""Generates tests/golden/*.npz by running the UNMODIFIED reference functions.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (it needs the
reference checkout at /root/reference); the GPU box and the tests read the
committed .npz files, never the reference.

Two kinds of vectors are produced for the photon walk
(``RunPhotons``, MC code.py:620-779 and Numpy_random_walk.py:280-408):

walk_replay.npz
    Every photon consumes its own pre-drawn uniform stream
    xi(id, j) = oracle.philox.walk_uniform(id, j, atm=0, seed).  The reference
    module's ``np`` is replaced by a proxy whose ``random.rand(n)`` hands the
    next stream element of each LIVE photon, and ``RemovePhotons`` is wrapped
    so the list of live ids is compacted exactly as the state array is.  The
    wrappers also note at which test each photon left the run, which gives
    per-photon fates, final (z, mu) and the number of draws consumed.
walk_seeded.npz
    Plain ``np.random.seed(s)`` runs of the reference in monochromatic mode;
    pins oracle/numpy_walk.py bitwise (same draws in the same order).

Usage:  python oracle/make_golden.py
"""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import philox  # noqa: E402
import ref_loader  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")

FATE_ESCAPED, FATE_ESCAPED_UNCOUNTED, FATE_ABSORBED, FATE_BOUNDARY = 1, 2, 3, 4


class _NumpyProxy:
    """Stands in for the ``np`` global of the reference module."""

    def __init__(self, rand):
        self.random = self
        self._rand = rand

    def rand(self, *shape):
        return self._rand(*shape)

    def __getattr__(self, name):
        return getattr(np, name)


def replay_reference(order, case):
    """Run the reference RunPhotons with per-photon streams; returns outputs + trace."""
    mod = ref_loader.load_gray() if order == "gray" else ref_loader.load_mc_code()
    n = case["n"]
    seed = case["seed"]
    lam, g, alb, scale, mu0 = (case[k] for k in ("lam", "g", "albedo", "scale", "mu0"))
    st = dict(live=np.arange(n), ctr=np.zeros(n, dtype=np.int64), stage=None, esc_calls=0,
              reemit_ids=None)
    fate = np.zeros(n, dtype=np.int32)
    final_mu = np.zeros(n)
    final_z = np.zeros(n)

    def rand(k):
        ids = st["live"]
        assert k == len(ids), (k, len(ids))
        j = st["ctr"][ids]
        st["ctr"][ids] = j + 1
        draw = {"philox32x4": philox.walk_uniform_packed, "philox40": philox.walk_uniform_p40}.get(case.get("stream"), philox.walk_uniform)
        return draw(ids, j, 0, seed)

    mod.np = _NumpyProxy(rand)

    remove0 = mod.RemovePhotons

    def remove(photon_state, idx):
        flat = np.asarray(idx).ravel()
        ids = st["live"][flat]
        fate[ids] = st["stage"]
        final_mu[ids] = photon_state[flat, 4]
        final_z[ids] = photon_state[flat, 2]
        st["live"] = np.delete(st["live"], flat)
        return remove0(photon_state, idx)

    mod.RemovePhotons = remove

    esc0, bnd0, abs0 = mod.CheckForEscape, mod.CheckBoundary, mod.Absorb

    def esc(*a, **k):
        st["esc_calls"] += 1
        second = (order == "mc") and (st["esc_calls"] % 2 == 0)
        st["stage"] = FATE_ESCAPED_UNCOUNTED if second else FATE_ESCAPED
        return esc0(*a, **k)

    def bnd(photon_state, *a, **k):
        st["stage"] = FATE_BOUNDARY
        st["reemit_ids"] = st["live"][np.where(photon_state[:, 2] <= 0)[0]]
        return bnd0(photon_state, *a, **k)

    def absorb(*a, **k):
        st["stage"] = FATE_ABSORBED
        return abs0(*a, **k)

    mod.CheckForEscape, mod.CheckBoundary, mod.Absorb = esc, bnd, absorb

    if order == "mc":
        created = dict(first=True)

        ex = case.get("extra", {})
        n_emitted = np.ones(n, dtype=np.uint64)  # emission blocks used so far, per photon

        def create(num, init, *rest):
            first = created["first"]
            ids = np.arange(n) if first else st["reemit_ids"]
            created["first"] = False
            assert len(ids) == num
            ps = np.array([init, ] * num, dtype=float)
            if first or "table_wl" not in ex:
                ps[:, 4] = mu0[ids]
                ps[:, 5] = lam[ids]
                ps[:, 6] = g[ids]
                ps[:, 7] = alb[ids]
                ps[:, 8] = scale[ids]
            else:
                # re-emission on a table: the rows the production sweep emits from the photon's NEXT emission block
                from oracle import c_oracle
                u_wl, u_mu = philox.block_uniforms(ids.astype(np.uint64), np.uint64(0x80000000) + n_emitted[ids], 0, seed)
                n_emitted[ids] += np.uint64(1)
                idx = c_oracle.sample_indices(ex["table_pdf"], u_wl)
                ps[:, 4] = u_mu if ex["rand_mu"] else 1.0
                ps[:, 5] = ex["table_wl"][idx]
                ps[:, 6] = ex["table_g"][idx]
                ps[:, 7] = ex["table_albedo"][idx]
                ps[:, 8] = ex["table_scale"][idx]
            return ps

        mod.CreatePhotons = create
        mod.abs_only = case["abs_only"]
        emu, B, E, I, count = mod.RunPhotons(case["tau"], n, case["boundary"], case["scatter"], None, None, None)
    else:
        emu, B, E, I = mod.RunPhotons(case["tau"], n, case["boundary"], float(alb[0]), case["scatter"], float(g[0]))
        count = -1
    return dict(escaped_mu=np.asarray(emu, dtype=float), boundary_momentum=float(B), escaped_momentum=float(E),
                initial_momentum=float(I), scatter_count=int(count), fate=fate, final_mu=final_mu,
                final_z=final_z, draws=st["ctr"].copy())


def make_case(name, order, n, tau, boundary, scatter, seed, abs_only=0, rand_mu0=False, gray_albedo=0.9, gray_g=0.5,
              tiny_g=False, uniform=None, table=None, stream="philox52"):
    rng = np.random.default_rng(seed)
    extra = {}
    if table is not None:
        # A table of several wavelengths: every photon gets the constants of the wavelength the PRODUCTION sweep emits
        # for its id under this seed (first uniform of the id's emission block through the sweep's own sampler,
        # oracle/c_oracle.sample_indices; rand_mu: the block's second uniform is the direction), so that the
        # reference, handed these rows, walks the histories of that sweep.
        if os.path.dirname(HERE) not in sys.path:
            sys.path.insert(0, os.path.dirname(HERE))
        from oracle import c_oracle
        wl_t, g_t, alb_t, scale_t, pdf_t = (np.asarray(v, dtype=float) for v in table)
        ids = np.arange(n, dtype=np.uint64)
        # (the packed stream takes the emission draw from words 0-1 of walk block 0, which no scattering precedes)
        u_wl, u_mu = philox.block_uniforms(ids, np.uint64(0 if stream != "philox52" else 0x80000000), 0, seed)
        assert not (rand_mu0 and stream != "philox52")
        idx = c_oracle.sample_indices(pdf_t, u_wl)
        lam, g, alb, scale = wl_t[idx], g_t[idx], alb_t[idx], scale_t[idx]
        mu0 = u_mu if rand_mu0 else np.ones(n)
        extra = dict(stream=stream, table_wl=wl_t, table_g=g_t, table_albedo=alb_t, table_scale=scale_t, table_pdf=pdf_t, emitted_index=idx,
                     rand_mu=int(rand_mu0))
        return dict(name=name, order=order, n=n, tau=float(tau), boundary=boundary, scatter=scatter, seed=int(seed),
                    abs_only=int(abs_only), lam=lam, g=g, albedo=alb, scale=scale, mu0=mu0, extra=extra, stream=stream)
    if uniform is not None:  # (lambda, g, albedo, scale) the same for every photon: what a one-wavelength table gives, so
        # that a PRODUCTION sweep under the same seed walks the very same histories (tests/test_gpu_replay.py)
        lam, g, alb, scale = (np.full(n, float(v)) for v in uniform)
    elif order == "mc":
        lam = np.round(rng.uniform(0.1, 2.0, n), 4)
        g = rng.uniform(0.05, 0.95, n) * np.where(rng.random(n) < 0.15, -1.0, 1.0)
        if tiny_g:  # number-weighted MRN averages give g as small as 1e-7 (cancellation in the HG formula)
            g = 10.0 ** rng.uniform(-7, -2, n)
        alb = rng.uniform(0.2, 0.95, n)
        scale = rng.uniform(0.3, 3.0, n)
    else:
        lam, scale = np.ones(n), np.ones(n)
        g, alb = np.full(n, float(gray_g)), np.full(n, float(gray_albedo))
    mu0 = rng.random(n) if rand_mu0 else np.ones(n)
    extra = dict(extra, stream=stream)
    return dict(name=name, order=order, n=n, tau=float(tau), boundary=boundary, scatter=scatter, seed=int(seed),
                abs_only=int(abs_only), lam=lam, g=g, albedo=alb, scale=scale, mu0=mu0, extra=extra, stream=stream)


def _table(kind):
    n = 64
    x = np.arange(n) / (n - 1.0)
    wl = np.round(0.1 + 0.9 * x, 4)
    pdf = (1.0 - 0.9 * x) ** 2 / n
    pdf[5:9] = 0.0
    if kind == "plain":
        return wl, (0.3 + 0.5 * x) * np.where(np.arange(n) % 7 == 0, -1, 1), 0.4 + 0.55 * x, 2.0 - 1.5 * x, pdf
    # asymmetry parameters from 0.9 down to 1e-7, as number-weighted MRN averages give them
    return wl, 0.9 * 10.0 ** (-7.0 * x), 0.45 + 0.5 * x, 0.5 + 1.5 * x, pdf


REPLAY_CASES = [
    make_case("mc_reflect_hg_tau3", "mc", 3000, 3.0, "reflect", "hg", 101),
    make_case("mc_reflect_hg_tau0p3_randmu", "mc", 3000, 0.3, "reflect", "hg", 102, rand_mu0=True),
    make_case("mc_absorb_hg_tau2", "mc", 3000, 2.0, "absorb", "hg", 103),
    make_case("mc_reemit_hg_tau2", "mc", 3000, 2.0, "reemit", "hg", 104),
    make_case("mc_reflect_iso_tau5", "mc", 3000, 5.0, "reflect", "iso", 105),
    make_case("mc_reflect_hg_absonly_tau1", "mc", 3000, 1.0, "reflect", "hg", 106, abs_only=1),
    make_case("mc_reflect_hg_tau30", "mc", 1500, 30.0, "reflect", "hg", 107),
    make_case("mc_reflect_hg_tau1em3", "mc", 3000, 1e-3, "reflect", "hg", 108),
    make_case("mc_reemit_hg_tau4_randmu", "mc", 2000, 4.0, "reemit", "hg", 109, rand_mu0=True),
    make_case("gray_reflect_hg_tau3", "gray", 3000, 3.0, "reflect", "hg", 201, gray_albedo=0.9, gray_g=0.5),
    make_case("mc_reflect_hg_tau2_tiny_g", "mc", 3000, 2.0, "reflect", "hg", 110, tiny_g=True),
    make_case("gray_reemit_iso_tau10", "gray", 600, 10.0, "reemit", "iso", 202, gray_albedo=1.0, gray_g=0.0),
    make_case("gray_absorb_hg_tau1", "gray", 3000, 1.0, "absorb", "hg", 203, gray_albedo=0.7, gray_g=-0.6),
    make_case("gray_reflect_hg_g0_tau2", "gray", 3000, 2.0, "reflect", "hg", 204, gray_albedo=0.8, gray_g=0.0),
    # the wavelength-dependent script with one wavelength: the production kernels' own cases
    make_case("mcu_reflect_hg_tau3", "mc", 3000, 3.0, "reflect", "hg", 301, uniform=(0.55, 0.63, 0.82, 1.3)),
    make_case("mcu_absorb_hg_tau2", "mc", 3000, 2.0, "absorb", "hg", 302, uniform=(0.25, 0.4, 0.85, 0.8)),
    make_case("mcu_reemit_hg_tau2", "mc", 3000, 2.0, "reemit", "hg", 303, uniform=(1.5, 0.3, 0.5, 2.0)),
    make_case("mcu_reflect_iso_tau5", "mc", 3000, 5.0, "reflect", "iso", 304, uniform=(0.0912, 0.7, 0.6, 1.0)),
    make_case("mcu_reflect_hg_tau30", "mc", 1500, 30.0, "reflect", "hg", 305, uniform=(0.55, 0.9, 0.95, 0.7)),
    make_case("mcu_reflect_hg_absonly_tau1", "mc", 3000, 1.0, "reflect", "hg", 306, abs_only=1, uniform=(0.55, 0.63, 0.82, 1.3)),
    make_case("mcu_reflect_hg_negg_tau4", "mc", 3000, 4.0, "reflect", "hg", 307, uniform=(2.0, -0.45, 0.9, 1.6)),
    make_case("mcu_reflect_hg_tinyg_tau2", "mc", 3000, 2.0, "reflect", "hg", 308, uniform=(9.5, 3.7e-6, 0.55, 0.4)),
    # ... and with a table of 64 wavelengths, every photon carrying the row the production sweep emits for its id
    make_case("mct_reflect_hg_tau3", "mc", 4000, 3.0, "reflect", "hg", 401, table=_table("plain")),
    make_case("mct_absorb_hg_tau2", "mc", 4000, 2.0, "absorb", "hg", 402, table=_table("plain")),
    make_case("mct_reflect_iso_tau5", "mc", 4000, 5.0, "reflect", "iso", 403, table=_table("plain")),
    make_case("mct_reflect_hg_tau0p3_randmu", "mc", 4000, 0.3, "reflect", "hg", 404, rand_mu0=True, table=_table("plain")),
    make_case("mct_reflect_hg_tau10_smallg", "mc", 4000, 10.0, "reflect", "hg", 405, table=_table("small g")),
]


# The packed stream's own cases (a file of their own: the tests of walk_replay.npz rebuild the 52-bit stream)
PACKED_CASES = [
    make_case("mcp_reflect_hg_tau3", "mc", 3000, 3.0, "reflect", "hg", 501, uniform=(0.55, 0.63, 0.82, 1.3), stream="philox32x4"),
    make_case("mcp_absorb_hg_tau2", "mc", 3000, 2.0, "absorb", "hg", 502, uniform=(0.25, 0.4, 0.85, 0.8), stream="philox32x4"),
    make_case("mcp_reflect_iso_tau5", "mc", 3000, 5.0, "reflect", "iso", 503, uniform=(0.0912, 0.7, 0.6, 1.0), stream="philox32x4"),
    make_case("grayp_reflect_hg_tau3", "gray", 3000, 3.0, "reflect", "hg", 504, gray_albedo=0.9, gray_g=0.5, stream="philox32x4"),
    make_case("mcpt_reflect_hg_tau3", "mc", 4000, 3.0, "reflect", "hg", 505, table=_table("plain"), stream="philox32x4"),
    make_case("mcpt_reflect_hg_tau10_smallg", "mc", 4000, 10.0, "reflect", "hg", 506, table=_table("small g"), stream="philox32x4"),
]


# ... and of the other one-block stream (24 | 40 | 40 | 24 bits)
P40_CASES = [
    make_case("mcq_reflect_hg_tau3", "mc", 3000, 3.0, "reflect", "hg", 701, uniform=(0.55, 0.63, 0.82, 1.3), stream="philox40"),
    make_case("mcq_absorb_hg_tau2", "mc", 3000, 2.0, "absorb", "hg", 702, uniform=(0.25, 0.4, 0.85, 0.8), stream="philox40"),
    make_case("mcq_reflect_iso_tau5", "mc", 3000, 5.0, "reflect", "iso", 703, uniform=(0.0912, 0.7, 0.6, 1.0), stream="philox40"),
    make_case("grayq_reflect_hg_tau3", "gray", 3000, 3.0, "reflect", "hg", 704, gray_albedo=0.9, gray_g=0.5, stream="philox40"),
    make_case("mcqt_reflect_hg_tau3", "mc", 4000, 3.0, "reflect", "hg", 705, table=_table("plain"), stream="philox40"),
    make_case("mcqt_reflect_hg_tau10_smallg", "mc", 4000, 10.0, "reflect", "hg", 706, table=_table("small g"), stream="philox40"),
]


# Re-emission at the floor on a table: the re-emitted photon draws a new wavelength (and direction) from its next
# emission block, so these cases have no fixed per-photon constants and only the production sweeps are compared with them.
REEMIT_TABLE_CASES = [
    make_case("mct_reemit_hg_tau2", "mc", 4000, 2.0, "reemit", "hg", 601, table=_table("plain")),
    make_case("mct_reemit_hg_tau4_randmu", "mc", 4000, 4.0, "reemit", "hg", 602, rand_mu0=True, table=_table("plain")),
    make_case("mct_reemit_iso_tau1", "mc", 4000, 1.0, "reemit", "iso", 603, table=_table("small g")),
]


def seeded_reference(order, tau, n, boundary, scatter, seed, lam, g, alb, scale, alpha=0.3):
    if order == "mc":
        mod = ref_loader.load_mc_code()
        mod.monochromatic = lam
        grain = pd.DataFrame({"WL": [lam], "g": [g], "Albedo": [alb], "Scale": [scale]})
        np.random.seed(seed)
        emu, B, E, I, count = mod.RunPhotons(tau, n, boundary, scatter, None, None, grain)
    else:
        mod = ref_loader.load_gray()
        mod.alpha = alpha  # the gray script reads it as a global (Numpy_random_walk.py:16)
        np.random.seed(seed)
        emu, B, E, I = mod.RunPhotons(tau, n, boundary, alb, scatter, g)
        count = -1
    return np.asarray(emu, dtype=float), float(B), float(E), float(I), int(count)


SEEDED_CASES = [
    # name, order, tau, n, boundary, scatter, seed, lam, g, albedo, scale
    ("mc_reflect_hg", "mc", 2.0, 20000, "reflect", "hg", 7, 0.5, 0.6, 0.7, 1.3),
    ("mc_absorb_hg", "mc", 5.0, 20000, "absorb", "hg", 8, 0.25, 0.4, 0.85, 0.8),
    ("mc_reemit_iso", "mc", 1.0, 20000, "reemit", "iso", 9, 1.5, 0.3, 0.5, 2.0),
    ("mc_reflect_hg_thick", "mc", 50.0, 5000, "reflect", "hg", 10, 0.0912, 0.7, 0.6, 1.0),
    ("gray_reflect_hg", "gray", 3.0, 20000, "reflect", "hg", 11, 1.0, 0.5, 0.9, 1.0),
    ("gray_reflect_draine", "gray", 3.0, 20000, "reflect", "draine", 12, 1.0, 0.5, 0.9, 1.0),
    ("gray_reemit_iso", "gray", 10.0, 20000, "reemit", "iso", 13, 1.0, 0.0, 1.0, 1.0),
    # scatter='draine' with g == 0 and alpha == 0 takes the script's IsoScatter branch (Numpy_random_walk.py:388-391)
    ("gray_reflect_draine_g0_alpha0", "gray", 2.0, 20000, "reflect", "draine", 14, 1.0, 0.0, 0.9, 1.0),
]
SEEDED_ALPHA = {"gray_reflect_draine_g0_alpha0": 0.0}


def write_cases(cases, filename):
    out = {"names": np.array([c["name"] for c in cases])}
    for c in cases:
        res = replay_reference(c["order"], c)
        p = c["name"] + "/"
        for k in ("order", "boundary", "scatter"):
            out[p + k] = np.array(c[k])
        for k in ("n", "tau", "seed", "abs_only", "lam", "g", "albedo", "scale", "mu0"):
            out[p + k] = np.asarray(c[k])
        for k, v in res.items():
            out[p + "ref_" + k] = np.asarray(v)
        for k, v in c["extra"].items():
            out[p + k] = np.asarray(v)
        print(c["name"], "events", res["scatter_count"], "max draws", res["draws"].max(),
              "fates", np.bincount(res["fate"], minlength=5))
    np.savez_compressed(os.path.join(GOLDEN, filename), **out)


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    out = {"names": np.array([c["name"] for c in REPLAY_CASES])}
    for c in REPLAY_CASES:
        res = replay_reference(c["order"], c)
        p = c["name"] + "/"
        for k in ("order", "boundary", "scatter"):
            out[p + k] = np.array(c[k])
        for k in ("n", "tau", "seed", "abs_only", "lam", "g", "albedo", "scale", "mu0"):
            out[p + k] = np.asarray(c[k])
        for k, v in res.items():
            out[p + "ref_" + k] = np.asarray(v)
        for k, v in c["extra"].items():
            out[p + k] = np.asarray(v)
        print(c["name"], "events", res["scatter_count"], "max draws", res["draws"].max(),
              "fates", np.bincount(res["fate"], minlength=5))
    np.savez_compressed(os.path.join(GOLDEN, "walk_replay.npz"), **out)
    write_cases(PACKED_CASES, "walk_replay_packed.npz")
    write_cases(P40_CASES, "walk_replay_philox40.npz")
    write_cases(REEMIT_TABLE_CASES, "walk_reemit_table.npz")

    out = {"names": np.array([c[0] for c in SEEDED_CASES])}
    for (name, order, tau, n, boundary, scatter, seed, lam, g, alb, scale) in SEEDED_CASES:
        alpha = SEEDED_ALPHA.get(name, 0.3)
        emu, B, E, I, count = seeded_reference(order, tau, n, boundary, scatter, seed, lam, g, alb, scale, alpha)
        p = name + "/"
        for k, v in dict(order=order, tau=tau, n=n, boundary=boundary, scatter=scatter, seed=seed, lam=lam, g=g, alpha=alpha,
                         albedo=alb, scale=scale, ref_escaped_mu=emu, ref_boundary_momentum=B,
                         ref_escaped_momentum=E, ref_initial_momentum=I, ref_scatter_count=count).items():
            out[p + k] = np.asarray(v)
        print(name, B, E, I, count, len(emu))
    np.savez_compressed(os.path.join(GOLDEN, "walk_seeded.npz"), **out)


if __name__ == "__main__":
    main()
