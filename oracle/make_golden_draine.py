"""Golden histograms of the reference's Draine phase-function sampler.

TEST INFRASTRUCTURE ONLY; runs only where the reference checkout is mounted.  Calls the UNMODIFIED reference
function DraineScatter (Numpy_random_walk.py:106-130) for 1e6 photons at g = 0.5 and g = 0.85 with alpha = 0.3
(Numpy_random_walk.py:16) under a fixed NumPy seed and stores the histogram of the returned cosines (40 equal
bins on [-1, 1]).  The reference pairs accepted cosines to photons by acceptance order, batch after batch; every
accepted cosine is an independent draw from the same (clipped) law, which is what the per-photon samplers of the
device and of oracle/rt_oracle.c must reproduce -- the histograms pin that law to the reference's own output.
"""
import os

import numpy as np

from oracle import ref_loader

N, BINS, ALPHA, SEED = 1000000, 40, 0.3, 20260112
G_VALUES = (0.5, 0.85)


def main():
    ref = ref_loader.load_gray()
    out = {"n": N, "alpha": ALPHA, "g": np.array(G_VALUES), "edges": np.linspace(-1.0, 1.0, BINS + 1)}
    hists = []
    for g in G_VALUES:
        ref.np.random.seed(SEED)
        phi, mu = ref.DraineScatter(N, g, ALPHA)
        assert len(phi) == N and len(mu) >= N
        hists.append(np.histogram(mu[:N], bins=out["edges"])[0])
    out["hist"] = np.stack(hists)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "draine_reference_hist.npz"), **out)
    print(out["hist"])


if __name__ == "__main__":
    main()
