"""Soak: the headline sweep N times back to back on one stream, every result compared on the device with the first
one (same seed and id base, so every bit must repeat); alternates the allocating and the static entry point.
python tools/soak.py [sweeps]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")
D = importlib.import_module("radiative-transfer_amd.distributed")
workloads = importlib.import_module("radiative-transfer_amd.workloads")
n_sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = workloads.WORKLOADS["mix_2e7"]()
eng = rt.Engine(0)
eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
dev = torch.device("cuda", 0)
buf = D.TallyBuffer(torch, len(cfg["tau_list"]), 8, 10, dev)
stream = torch.cuda.current_stream().cuda_stream
prm = rt.make_params(cfg["tau_list"], cfg["n_photons"], h=cfg["h"])
eng.run_device(prm, *buf.pointers(), stream)
first = buf.words.clone()
bad = torch.zeros((), dtype=torch.int64, device=dev)
t0 = time.perf_counter()
for i in range(n_sweeps):
    eng.run_device(prm, *buf.pointers(), stream, static=bool(i & 1))
    bad += (buf.words != first).sum()
    if i % 200 == 199:
        print(f"{i + 1} sweeps, {int(bad.item())} words differ, {time.perf_counter() - t0:.1f} s", flush=True)
torch.cuda.synchronize()
print(f"{n_sweeps} sweeps of {int(first.numel())} words: {int(bad.item())} words differed from the first sweep")
eng.close()
sys.exit(1 if int(bad.item()) else 0)
