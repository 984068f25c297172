#!/usr/bin/env python3
"""Benchmark of the dust photon random walk on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* is one full sweep of the headline configuration: 100 atmospheres
(tau = logspace(-3, 2, 100)) x 2e7 photons each through the Sil/Gra/SiC dust mix, tables and
emission law resident in HBM before the timed region.  For N > 1 (one process per GPU, launched by
torch.distributed.run) every rank walks its own slabs of the photon-id range -- per-GPU work is
fixed, so the global run has N x 2e7 photons per atmosphere ("weak" scaling) -- and the only
collective is one all-reduce (RCCL) of the slab tallies per step, inside the timed region.

Prints ONE JSON line on rank 0.  `value` = scatter events of all ranks / wall time of the K steps
(max over ranks).  `roofline` is for the walk kernel: algorithmic FP64 operations per launch
(35 per scatter event, as the reference writes them: MC code.py:427-441, 493-508, 576-618) divided by
the kernel's average duration measured with HIP events on its stream, against the FP64 vector peak.
`cpu_baseline` times the NumPy restatement of the reference loop (oracle/numpy_walk.py, pinned to
the reference by tests/golden) on a bounded sample of the same workload on this host.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_OPS_PER_EVENT = 35          # algorithmic FP64 operations of one scatter event
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X: 256 CU x 128 FLOP/clk x 2.4 GHz


_CPU_CFG = None  # the workload, inherited by forked workers


def _cpu_atmosphere(task):
    """One atmosphere of the sweep through the NumPy restatement of RunPhotons; returns its scatter count."""
    from oracle import numpy_walk as nw
    k, n_cpu = task
    cfg = _CPU_CFG
    rs = np.random.RandomState(12345 + k)
    pdf = np.ones(len(cfg["wl"])) if cfg["pdf"] is None else cfg["pdf"]
    cdf = np.cumsum(np.clip(np.nan_to_num(pdf), 0, 1))
    idx = np.minimum(np.searchsorted(cdf, rs.rand(n_cpu) * cdf[-1], side="right"), len(cdf) - 1)
    row = (lambda a: a if a.ndim == 1 else a[cfg["table_of_atm"][k]])  # several tables: this atmosphere's
    r = nw.run_photons(cfg["tau_list"][k], cfg["wl"][idx], row(cfg["g"])[idx], row(cfg["albedo"])[idx], row(cfg["scale"])[idx],
                       np.ones(n_cpu), boundary=cfg["boundary"], scatter=cfg["scatter"], order=cfg["order"],
                       stream=nw.GlobalStream(rs), h=cfg["h"])
    return int(r["scatter_count"])


def cpu_baseline(cfg, n_cpu, procs=1, seconds_cap=30.0):
    """NumPy restatement of RunPhotons over the same tau list with n_cpu photons per atmosphere, on `procs`
    host processes (atmospheres dealt out one at a time, thickest first so the cap cuts the cheap end)."""
    global _CPU_CFG
    _CPU_CFG = cfg
    n_atm = len(cfg["tau_list"])
    tasks = [(k, n_cpu) for k in range(n_atm)] if procs == 1 else [(k, n_cpu) for k in reversed(range(n_atm))]
    events, done, t0 = 0, 0, time.perf_counter()
    if procs == 1:
        for t in tasks:
            events += _cpu_atmosphere(t)
            done += 1
            if time.perf_counter() - t0 > seconds_cap:
                break
    else:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:  # forked before anything touches the GPU
            for ev in pool.imap_unordered(_cpu_atmosphere, tasks):
                events += ev
                done += 1
                if time.perf_counter() - t0 > seconds_cap:
                    pool.terminate()
                    break
    dt = time.perf_counter() - t0
    return dict(value=events / dt, unit="scatter events/s", cores=procs, kind="port",
                sample=f"{n_cpu} photons/atmosphere x {done} of {n_atm} atmospheres of the same sweep "
                       f"({events} events in {dt:.1f} s, oracle/numpy_walk.py, {procs} process(es))")


def measured_traffic(workload, per_gpu):
    """HBM-side bytes per kernel launch from the committed PMC passes (profiles/*_traffic.json), if one
    was taken for exactly this workload; rocprofv3 cannot be run from inside the benchmark."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload and int(d.get("photons_per_atmosphere", -1)) == int(per_gpu):
            best = d
    return None if best is None else float(best["bytes_per_launch"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="mix_2e7")
    ap.add_argument("--photons", type=int, default=0, help="photons per atmosphere per GPU (0 = the workload's own)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--cpu-photons", type=int, default=1000000)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the all-cores CPU baseline (0 = min(16, cores))")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--rng", choices=["philox52", "philox32x4"], default="philox52",
                    help="philox32x4 = the opt-in one-block-per-iteration stream with 32-bit uniforms (not the headline)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo stages the tallies through host memory; for rehearsing N > 1 on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    rt = importlib.import_module("radiative-transfer_amd")
    workloads = importlib.import_module("radiative-transfer_amd.workloads")
    cfg = workloads.WORKLOADS[args.workload]()

    # CPU baselines first: the worker processes are forked while this process has not touched the GPU
    cpu = {}
    if world == 1 and not args.no_cpu:
        cpu["cpu_baseline"] = cpu_baseline(cfg, args.cpu_photons, 1)
        procs = args.cpu_procs or min(16, len(os.sched_getaffinity(0)))
        if procs > 1:
            cpu["cpu_baseline_all_cores"] = cpu_baseline(cfg, args.cpu_photons, procs)

    import torch
    import torch.distributed as dist

    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {local_rank} has no GPU ({n_dev} visible); RCCL needs one GPU per rank")
    local_dev = local_rank % n_dev  # only ever differs from local_rank in a gloo rehearsal
    torch.cuda.set_device(local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group("gloo")

    A = rt._abi
    per_gpu = args.photons or cfg["n_photons"]
    n_slabs = 8
    total_photons = per_gpu * world if args.scaling == "weak" else per_gpu
    tau = cfg["tau_list"]
    n_atm = len(tau)

    eng = rt.Engine(local_dev)
    eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
    dev = torch.device("cuda", local_dev)
    slab = torch.zeros(n_atm * n_slabs * A.NSUM, dtype=torch.float64, device=dev)
    counts = torch.zeros(n_atm * A.NCNT, dtype=torch.int64, device=dev)
    hist = torch.zeros(n_atm * 11, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        prm = rt.make_params(tau, total_photons, boundary=cfg["boundary"], scatter=cfg["scatter"], order=cfg["order"],
                             n_mu_bins=10, n_slabs=n_slabs, rank=rank, world=world, h=cfg["h"],
                             monochromatic=cfg["monochromatic"], photon_id_base=i * total_photons, rng=args.rng,
                             table_of_atm=cfg.get("table_of_atm"))
        eng.run_device(prm, slab.data_ptr(), counts.data_ptr(), hist.data_ptr(), None, stream)
        if world > 1:
            # every slab entry is non-zero on exactly one rank, so the sum is exact and order-free
            if args.backend == "nccl":
                dist.all_reduce(slab)
                dist.all_reduce(counts)
                dist.all_reduce(hist)
            else:
                for t in (slab, counts, hist):
                    h = t.cpu()
                    dist.all_reduce(h)
                    t.copy_(h)
        # scatter events of this step (whole job after the all-reduce), kept on the device
        return counts.view(n_atm, A.NCNT)[:, A.CNT_SCATTER].sum()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev_dev = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(args.warmup):
        ev_dev = ev_dev + step(i)  # same device work as a timed step (also loads torch's own kernels)
    fence()
    eng.kernel_time(reset=True)
    ev_dev = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev_dev = ev_dev + step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    events = int(ev_dev.item())          # after the all-reduce this is the whole job's count
    kernel_ms, launches = eng.kernel_time()
    last = counts.view(n_atm, A.NCNT).cpu().numpy()
    events_last = int(last[:, A.CNT_SCATTER].sum())
    events_rank_last = events_last / world if world > 1 else events_last  # equal shares by construction

    fast = (cfg["boundary"], cfg["scatter"], cfg["order"]) == ("reflect", "hg", "mc")  # the kernel rt_run picks
    packed = args.rng == "philox32x4"
    plain = fast and not packed and not cfg["monochromatic"]  # no workload sets rand_mu
    kernel_name = "walk_kernel<%s, %s, %s>" % tuple("true" if b else "false" for b in (fast, packed, plain))
    if rank == 0:
        value = events / dt
        achieved = events_rank_last * FP64_OPS_PER_EVENT / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "photon-scatter events/sec", "value": value, "unit": "events/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{cfg['name']}: {n_atm} atmospheres tau={tau[0]:.3g}..{tau[-1]:.3g} (log-spaced) x "
                                   f"{per_gpu:.3g} photons per atmosphere per GPU, {cfg['describe']}",
                       "photons_per_atmosphere_global": total_photons, "photon_histories_per_step": int(last[:, A.CNT_PHOTONS].sum()),
                       "flights_per_step": int(last[:, A.CNT_FLIGHTS].sum()), "events_per_step": events_last,
                       "rng": "Philox4x32-10, " + ("52-bit uniforms, 2 blocks per loop iteration" if args.rng == "philox52"
                                                   else "32-bit uniforms, 1 block per loop iteration (opt-in mode)"),
                       "parallelism": f"photon-id slabs over {world} GPU(s), 1 all-reduce of tallies per step"},
            "photon_histories_per_s": float(last[:, A.CNT_PHOTONS].sum()) * args.steps / dt,
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                         "traffic": measured_traffic(cfg["name"], per_gpu),
                         "kernel": kernel_name, "kernel_avg_ms": kernel_ms, "kernel_launches": launches,
                         "flop_per_event": FP64_OPS_PER_EVENT},
        }
        out.update(cpu)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
