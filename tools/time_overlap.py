"""Do consecutive sweeps overlap usefully when they alternate between two contexts / streams?  The share of rank 0 of
a W-rank job (headline sweep) is enqueued 40 times, once on one stream and once alternating between two contexts on
two streams; prints the wall time per step of both.
    python tools/time_overlap.py [world]"""
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
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
engs, bufs, streams = [], [], []
for _ in range(2):
    e = rt.Engine(0)
    e.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
    engs.append(e)
    bufs.append(D.TallyBuffer(torch, len(cfg["tau_list"]), 8, 10, dev))
    streams.append(torch.cuda.Stream(dev))
prm = rt.make_params(cfg["tau_list"], cfg["n_photons"], h=cfg["h"], rank=0, world=world)


def run(n_streams, reps=40):
    for k in range(4):
        engs[k % n_streams].run_device(prm, *bufs[k % n_streams].pointers(), streams[k % n_streams].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        engs[k % n_streams].run_device(prm, *bufs[k % n_streams].pointers(), streams[k % n_streams].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(2):
    a, b = run(1), run(2)
    print(f"W={world}: one stream {a:.3f} ms per step, two streams {b:.3f} ms per step ({(1 - b / a) * 100:+.1f} %)", flush=True)
for e in engs:
    e.close()
