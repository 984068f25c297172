"""The CPU-side pieces of bench.py (no GPU): the bounded CPU baseline on one and on several processes, the
statistical comparison that goes on the benchmark line, the lookup of committed profile summaries, and the
launcher that turns `bench.py --gpus N` into N rank processes."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np

import bench
from conftest import ROOT

workloads = importlib.import_module("radiative-transfer_amd.workloads")
engine_mod = importlib.import_module("radiative-transfer_amd.engine")
A = importlib.import_module("radiative-transfer_amd._abi")


def test_cpu_baseline_single_and_forked():
    cfg = workloads.WORKLOADS["gray_1e5"]()
    one, rows1 = bench.cpu_baseline(cfg, 2000, 1, seconds_cap=20.0)
    many, rows2 = bench.cpu_baseline(cfg, 2000, 2, seconds_cap=20.0)
    for r, cores in ((one, 1), (many, 2)):
        assert r["kind"] == "port" and r["cores"] == cores and r["value"] > 0 and "atmospheres" in r["sample"]
    # same seeds per atmosphere, so both walk the same photons
    assert one["sample"].split("(")[1].split(" events")[0] == many["sample"].split("(")[1].split(" events")[0]
    by_atm = {r["atm"]: r for r in rows2}
    for r in rows1:
        assert r["events"] == by_atm[r["atm"]]["events"] and r["escaped"] == by_atm[r["atm"]]["escaped"]
        assert r["hist"].shape == (bench.N_MU_BINS + 1,) and r["initial"] > 0


def _sweep_like(rows, n):
    """A SweepResult filled from CPU rows as if a device had produced them (n photons per atmosphere)."""
    res = engine_mod.SweepResult(len(rows), bench.N_MU_BINS, 0, 1)
    for r in rows:
        a = r["atm"]
        res.sums[a, A.SUM_INITIAL], res.sums[a, A.SUM_ESCAPED], res.sums[a, A.SUM_BOUNDARY] = r["initial"], r["escaped"], r["boundary"]
        m = (r["initial"] - r["escaped"] + r["boundary"]) / r["initial"]
        p0 = r["initial"] / n
        # per-photon transfers of variance v around m*p0: sum t^2 = n (m^2 + v) p0^2, sum t p0 = n m p0^2
        v = 0.25
        res.sums[a, A.SUM_TRANSFER_SQ] = n * (m * m + v) * p0 * p0
        res.sums[a, A.SUM_TRANSFER_P0] = n * m * p0 * p0
        res.sums[a, A.SUM_P0_SQ] = n * p0 * p0
        res.counts[a, A.CNT_PHOTONS] = n
        res.mu_hist[a] = r["hist"]
    return res


def test_stat_parity_fields():
    cfg = workloads.WORKLOADS["gray_1e5"]()
    _, rows = bench.cpu_baseline(cfg, 4000, 1, seconds_cap=20.0)
    res = _sweep_like(rows, 4000)
    sp = bench.stat_parity(res, rows, 4000)
    assert sp["dof"] == len(rows) and sp["chi2"] == 0.0 and sp["max_abs_z"] == 0.0   # identical samples
    assert sp["ks_d"] == 0.0 and sp["ks_crit"] > 0 and sp["ks_alpha"] == 0.001
    shifted = _sweep_like(rows, 4000)
    shifted.sums[:, A.SUM_ESCAPED] *= 0.9                                        # a real disagreement shows
    assert bench.stat_parity(shifted, rows, 4000)["max_abs_z"] > 1.0
    assert bench.stat_parity(res, [], 4000) is None


def test_committed_profile_lookup():
    t = bench.committed_json("*_traffic.json", {"workload": "mix_2e7", "photons_per_atmosphere": 20000000})
    assert t is not None and t["bytes_per_launch"] > 1e9
    assert bench.committed_json("*_traffic.json", {"workload": "mix_2e7", "photons_per_atmosphere": 123}) is None
    assert np.isclose(bench.FP64_VECTOR_PEAK_TFLOPS, 256 * 128 * 2.4e9 / 1e12, rtol=1e-3)


def test_gpus_n_without_launcher_starts_rank_processes(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start two rank processes (here they stop at once:
    this container has no GPU) instead of printing a usage message; the parent never imports torch."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu", "--workload", "gray_1e5", "--backend", "gloo"], env=env, capture_output=True, text=True,
                       timeout=600)
    import torch
    if torch.cuda.is_available():
        line = json.loads(p.stdout.strip().splitlines()[-1])
        assert p.returncode == 0 and line["n_gpus"] == 2 and "1 all-reduce per step" in line["config"]["parallelism"]
    else:
        assert p.returncode != 0
        assert "bench.py needs an MI355X" in p.stderr and "launch with" not in p.stderr
