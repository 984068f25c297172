"""Strong-scaling shape of the headline sweep measured on ONE GPU: the share of every rank r of a W-rank job
(make_params(rank=r, world=W)) is walked alone, enqueued back to back like the benchmark's steps (no host
synchronisation between launches, so the GPU does not idle and clock down between them), and timed with the
library's own HIP events; the projected efficiency of W GPUs is T1 / (W * max_r T_rW).  (The collective, one
100 KB all-reduce per step, is not in it.)

    python tools/time_ranks.py [photons per atmosphere] > profiles/<round>_strong_scaling.txt
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")
D = importlib.import_module("radiative-transfer_amd.distributed")
workloads = importlib.import_module("radiative-transfer_amd.workloads")
cfg = workloads.WORKLOADS["mix_2e7"]()
n = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["n_photons"]
only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (2, 4, 8)
eng = rt.Engine(0)
eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
buf = D.TallyBuffer(torch, len(cfg["tau_list"]), 8, 10, torch.device("cuda", 0))
stream = torch.cuda.current_stream().cuda_stream


def share(rank, world, reps=20):
    prm = rt.make_params(cfg["tau_list"], n, h=cfg["h"], rank=rank, world=world)
    for _ in range(3):
        eng.run_device(prm, *buf.pointers(), stream)
    torch.cuda.synchronize()
    eng.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.run_device(prm, *buf.pointers(), stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    return eng.kernel_time()[0], wall


share(0, 1, 10)
t1, w1 = share(0, 1)
print(f"mix_2e7, {n} photons per atmosphere in total, 8 slabs per atmosphere, {os.environ.get('RT_AMD_CHUNK', 'default chunk rule')}")
print(f"walk-kernel time per rank share (ms; in brackets: wall time per step of 20 back-to-back steps)")
print(f"W=1: {t1:.3f} [{w1:.3f}]")
for world in only:
    ts = [share(r, world) for r in range(world)]
    k, w = max(t[0] for t in ts), max(t[1] for t in ts)
    print(f"W={world}: " + " ".join(f"{t[0]:.3f}" for t in ts) + f"   max {k:.3f} [{w:.3f}]  projected efficiency T1/(W*max) = "
          f"{t1 / (world * k):.3f} [{w1 / (world * w):.3f}]", flush=True)
eng.close()
