"""When do the waves of the persistent walk kernel finish?  Needs the tracing build of the library:
    hipcc ... -DRT_ABLATION -DRT_ABL_WAVE_TRACE -o radiative-transfer_amd/csrc/variants/trace.so
    RT_AMD_ALLOW_ABLATION=1 RT_AMD_LIB=.../trace.so python tools/wave_trace.py [world] [photons]
Prints, for one rank's share of the headline sweep, the distribution of the waves' end times and of the length of
their last chunk, relative to the launch's span (100 MHz clock)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("radiative-transfer_amd")
workloads = importlib.import_module("radiative-transfer_amd.workloads")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = workloads.WORKLOADS["mix_2e7"]()
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["n_photons"]
eng = rt.Engine(0)
eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
prm = rt.make_params(cfg["tau_list"], n, h=cfg["h"], rank=0, world=world)
for _ in range(3):
    eng.run(prm)
eng.kernel_time(reset=True)
eng.run(prm)
ms = eng.kernel_time()[0]
nw = 5120
buf = np.zeros((nw, 8), dtype=np.uint64)
lib = eng._lib
rc = lib.rt_debug_wave_trace(buf.ctypes.data_as(C.POINTER(C.c_uint64)), nw)
assert rc == 0
t = buf.astype(np.float64)
live = t[:, 2] > 0
t = t[live]
t0 = t[:, 0].min()
begin, last, end = (t[:, 0] - t0) / 1e5, (t[:, 1] - t0) / 1e5, (t[:, 2] - t0) / 1e5   # ms
span = end.max()
print(f"W={world} n={n} {os.environ.get('RT_AMD_CHUNK', 'default')}: kernel {ms:.3f} ms, trace span {span:.3f} ms, {len(t)} waves, "
      f"chunks per wave {t[:, 3].mean():.1f} (min {t[:, 3].min():.0f}, max {t[:, 3].max():.0f})")
print("wave start (ms)  p50 %.3f  p99 %.3f  max %.3f" % tuple(np.percentile(begin, [50, 99, 100])))
q = [0, 1, 10, 25, 50, 75, 90, 99, 100]
print("wave end   (ms)  " + "  ".join(f"p{p} {v:.3f}" for p, v in zip(q, np.percentile(end, q))))
print("last chunk (ms)  " + "  ".join(f"p{p} {v:.3f}" for p, v in zip(q, np.percentile(end - last, q))))
idle = (span - end).sum() / (span * len(t))
print(f"idle wave-time after a wave's end, share of waves x span: {idle:.4f}")
it, drain, few, lanes = (t[:, k].sum() for k in (4, 5, 6, 7))
print(f"loop iterations {it:.3e}; after a chunk's last id was handed out {drain / it:.4f} of them, with at most 4 lanes at work "
      f"{few / it:.4f}; mean lanes at work per iteration {lanes / it:.2f}")
# how full is the machine over time: waves still running at time x
for x in np.linspace(0.8, 1.0, 11):
    print(f"  t = {x:.2f} span: {np.mean(end > x * span):.3f} of the waves still running")
eng.close()
