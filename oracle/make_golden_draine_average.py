"""Golden vectors for the g / Albedo averages, from the reference's own ``GetDraineAverage`` recipe.
TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference).

The recipe survives in the reference only as a commented-out block (MC code.py:270-314: "Very old function, best
left as a reference"), and the '<species> Average Values.csv' files it wrote are absent from the checkout.  Here the
block is read from the reference's file at generation time, its comment markers are removed (and with them the
remark that follows the colon of the `def` line) -- nothing else is changed -- and the resulting function is executed
on the Draine folders that are present.  What it returns, and
the albedo formed from it as at MC code.py:320, is stored for tests/test_tables.py.  No reference text is kept.

As with GetKappas (make_golden_tables.py) the recipe integrates over the grain sizes in ``os.listdir`` order, which
is arbitrary on this file system; the function's ``os`` lists a folder in order of grain size.  For Gra and SiC
that is also the alphabetical order of the author's machine.  The Sil folder has a 10 um file whose name sorts
before the 1 um one alphabetically; the vectors are generated in order of grain size there too, which is what the
product integrates in (radiative-transfer_amd/tables.py), and the alphabetical result is stored beside it for the
record (it differs: the last trapezoid then runs backwards).

Writes tests/golden/draine_average.npz.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = "/root/reference"
STRIDE = 50


def load_recipe():
    """The commented-out GetDraineAverage of MC code.py as a function: the lines from '# def GetDraineAverage(' to
    its return statement with the leading '# ' (or bare '#') of every line removed."""
    lines = open(os.path.join(REFERENCE, "MC code.py")).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("# def GetDraineAverage("))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("#     return Sigma_abs_av"))
    body = []
    for l in lines[start:end + 1]:
        if l.startswith("# "):
            body.append(l[2:])
        elif l.strip() in ("#", ""):
            body.append("")
        else:
            raise RuntimeError(f"uncommented line inside the block: {l!r}")
    body[0] = body[0][:body[0].index(":") + 1]  # 'def ...(folder, wavelengths):' without the remark behind it
    return "\n".join(body), (start + 1, end + 1)


class _Os:
    """os for the recipe: listdir in a chosen order."""

    def __init__(self, key):
        self.key = key
        self.path = os.path

    def listdir(self, folder):
        return sorted(os.listdir(folder), key=self.key)


def by_size(name):
    return float(name.split("_")[-2] + "." + name.split("_")[-1])


def run(source, key, folder, wl):
    ns = {"np": np, "os": _Os(key)}
    exec(compile(source, "MC code.py (GetDraineAverage, uncommented)", "exec"), ns)
    sig_abs, sig_sca, g_av, q_ext = ns["GetDraineAverage"](folder, wl)
    with np.errstate(invalid="ignore", divide="ignore"):
        albedo = sig_sca / (sig_sca + sig_abs)  # MC code.py:320
    return {"Sigma-Abs": sig_abs, "Sigma-Scatt": sig_sca, "g": g_av, "Albedo": albedo, "Q_ext": q_ext}


def main():
    warnings.simplefilter("ignore")
    source, (first, last) = load_recipe()
    print(f"recipe: MC code.py:{first}-{last}")
    wl = np.round(0.001 + 1e-4 * np.arange(99991), 4)[::STRIDE]
    out = {"wl": wl, "lines": np.array([first, last])}
    for species in ("Sil", "Gra", "SiC"):
        folder = os.path.join(REFERENCE, f"Draine data {species}")
        res = run(source, by_size, folder, wl)
        for k, v in res.items():
            out[f"{species}/{k}"] = v
        print(species, "g(0.551)", res["g"][11], "albedo", res["Albedo"][11], "Q_ext", res["Q_ext"][11])
        alpha = run(source, None, folder, wl)  # plain alphabetical listing
        same = all(np.array_equal(alpha[k], res[k], equal_nan=True) for k in res)
        out[f"{species}/alphabetical_equals_by_size"] = np.array(same)
        if not same:
            for k, v in alpha.items():
                out[f"{species}/alphabetical/{k}"] = v
        print(species, "alphabetical listing gives the same:", same)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "draine_average.npz"), **out)


if __name__ == "__main__":
    main()
