#!/usr/bin/env python3
"""Benchmark of the dust photon random walk on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* is one full sweep of the headline configuration: 100 atmospheres
(tau = logspace(-3, 2, 100)) x 2e7 photons each through the Sil/Gra/SiC dust mix, tables and
emission law resident in HBM before the timed region.  For N > 1 there is one process per GPU.
Started by torch.distributed.run the script is one rank; started directly with --gpus N > 1 (no
WORLD_SIZE in the environment) it first launches N fresh rank processes of itself -- before this
process makes any GPU call -- and relays rank 0's line.  Every rank walks its own slabs of every
atmosphere's photon-id range; by default the JOB is fixed (2e7 photons per atmosphere in total,
"strong" scaling: the quantity the reference's '20 million photon runs' fixes), with --scaling weak
the per-GPU work is fixed instead.  The only collective is ONE all-reduce (RCCL) per step of one
buffer holding the slab sums, integer tallies and escape histogram, inside the timed region.

Prints ONE JSON line on rank 0.  `value` = scatter events of all ranks / wall time of the K steps
(max over ranks).  `roofline` is for the walk kernel: algorithmic FP64 operations per launch
(35 per scatter event, as the reference writes them: MC code.py:427-441, 493-508, 576-618) divided by
the kernel's average duration measured with HIP events on its stream, against the FP64 vector peak;
`valu_issue_frac` and `lane_util` beside it are the issue-side figures of the committed counter pass
(profiles/*_pmc.json).  `cpu_baseline` times the NumPy restatement of the reference loop
(oracle/numpy_walk.py, pinned to the reference by tests/golden) on a bounded sample of the same
workload on this host, and `stat_parity` compares that sample's momentum fractions and escape
cosines with the GPU sweep's (chi-square over atmospheres, two-sample Kolmogorov-Smirnov distance).
`opt_in_streams` (one GPU, after the timed region) repeats the sweep on the two opt-in random streams.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_OPS_PER_EVENT = 35          # algorithmic FP64 operations of one scatter event
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X: 256 CU x 128 FLOP/clk x 2.4 GHz
N_MU_BINS = 10


_CPU_CFG = None  # the workload, inherited by forked workers


def _cpu_atmosphere(task):
    """One atmosphere of the sweep through the NumPy restatement of RunPhotons: scatter count, the three momenta
    of runMC's estimator (MC code.py:1062) and the escape cosines binned like the device histogram."""
    from oracle import numpy_walk as nw
    k, n_cpu = task
    cfg = _CPU_CFG
    rs = np.random.RandomState(12345 + k)
    pdf = np.ones(len(cfg["wl"])) if cfg["pdf"] is None else cfg["pdf"]
    row = (lambda a: a if a.ndim == 1 else a[cfg["table_of_atm"][k]])  # several tables: this atmosphere's
    cdf = np.cumsum(np.clip(np.nan_to_num(row(pdf)), 0, 1))
    idx = np.minimum(np.searchsorted(cdf, rs.rand(n_cpu) * cdf[-1], side="right"), len(cdf) - 1)
    r = nw.run_photons(cfg["tau_list"][k], cfg["wl"][idx], row(cfg["g"])[idx], row(cfg["albedo"])[idx], row(cfg["scale"])[idx],
                       np.ones(n_cpu), boundary=cfg["boundary"], scatter=cfg["scatter"], order=cfg["order"],
                       stream=nw.GlobalStream(rs), h=cfg["h"])
    bins = np.clip(np.floor(N_MU_BINS * r["escaped_mu"]).astype(np.int64), 0, N_MU_BINS)
    return dict(atm=k, events=int(r["scatter_count"]), initial=float(r["initial_momentum"]),
                escaped=float(r["escaped_momentum"]), boundary=float(r["boundary_momentum"]),
                hist=np.bincount(bins, minlength=N_MU_BINS + 1))


def cpu_baseline(cfg, n_cpu, procs=1, seconds_cap=30.0):
    """NumPy restatement of RunPhotons over the same tau list with n_cpu photons per atmosphere, on `procs`
    host processes (atmospheres dealt out one at a time, thickest first so the cap cuts the cheap end).
    Returns the baseline record and the per-atmosphere results of the atmospheres that finished."""
    global _CPU_CFG
    _CPU_CFG = cfg
    n_atm = len(cfg["tau_list"])
    tasks = [(k, n_cpu) for k in range(n_atm)] if procs == 1 else [(k, n_cpu) for k in reversed(range(n_atm))]
    rows, events, t0 = [], 0, time.perf_counter()
    if procs == 1:
        for t in tasks:
            rows.append(_cpu_atmosphere(t))
            events += rows[-1]["events"]
            if time.perf_counter() - t0 > seconds_cap:
                break
    else:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:  # forked before anything touches the GPU
            for r in pool.imap_unordered(_cpu_atmosphere, tasks):
                rows.append(r)
                events += r["events"]
                if time.perf_counter() - t0 > seconds_cap:
                    pool.terminate()
                    break
    dt = time.perf_counter() - t0
    rec = dict(value=events / dt, unit="scatter events/s", cores=procs, kind="port",
               sample=f"{n_cpu} photons/atmosphere x {len(rows)} of {n_atm} atmospheres of the same sweep "
                      f"({events} events in {dt:.1f} s, oracle/numpy_walk.py, {procs} process(es))")
    return rec, rows


def stat_parity(res, cpu_rows, n_cpu):
    """GPU sweep against the CPU sample, atmosphere by atmosphere.  z = difference of the momentum fractions
    (initial - escaped + boundary) / initial over its standard error; the per-photon variance is the GPU's
    (RT_SUM_TRANSFER_*), scaled to each side's photon count.  ks_d: largest two-sample Kolmogorov-Smirnov
    distance between the binned escape-cosine distributions, with its critical value at alpha = 0.001."""
    if not cpu_rows:
        return None
    m_gpu, sig_gpu = res.momentum_fraction(), res.momentum_fraction_sigma()
    n_gpu = res.counts[:, 7].astype(float)  # RT_CNT_PHOTONS
    z, ks = [], None
    for r in cpu_rows:
        a = r["atm"]
        m_cpu = (r["initial"] - r["escaped"] + r["boundary"]) / r["initial"]
        sig2 = sig_gpu[a] ** 2 * (1.0 + n_gpu[a] / n_cpu)
        if sig2 > 0:
            z.append((m_gpu[a] - m_cpu) / np.sqrt(sig2))
        hg, hc = res.mu_hist[a].astype(float), r["hist"].astype(float)
        ng, nc = hg.sum(), hc.sum()
        if ng >= 100 and nc >= 100:
            d = float(np.max(np.abs(np.cumsum(hg) / ng - np.cumsum(hc) / nc)))
            crit = 1.9495 * np.sqrt((ng + nc) / (ng * nc))
            if ks is None or d / crit > ks[0] / ks[1]:
                ks = (d, float(crit), a)
    z = np.asarray(z)
    out = dict(chi2=float(np.sum(z * z)), dof=int(len(z)), max_abs_z=float(np.max(np.abs(z))) if len(z) else None,
               cpu_photons_per_atmosphere=int(n_cpu), atmospheres=len(cpu_rows))
    if ks is not None:
        out.update(ks_d=ks[0], ks_crit=ks[1], ks_atmosphere=int(ks[2]), ks_alpha=0.001)
    return out


def committed_json(pattern, match):
    """Latest profiles/<pattern> whose entries agree with `match` (rocprofv3 cannot run inside the benchmark)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if all(d.get(k) == v for k, v in match.items()):
            best = d
    return best


def launch_ranks(n, argv):
    """--gpus N > 1 without a launcher: N fresh rank processes of this script, started before this process has made
    any GPU call; rank 0's JSON line is relayed and the children's exit code returned."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    return rc if rc else (0 if line else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="mix_2e7")
    ap.add_argument("--photons", type=int, default=0, help="photons per atmosphere (0 = the workload's own): of the whole job "
                                                          "with --scaling strong, per GPU with --scaling weak")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--slabs", type=int, default=8, help="fixed photon-id slabs per atmosphere (a multiple of --gpus; results depend on it, not on --gpus)")
    ap.add_argument("--cpu-photons", type=int, default=1000000)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the all-cores CPU baseline (0 = every core this process may use)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--rng", choices=["philox52", "philox32x4", "philox40"], default="philox52",
                    help="philox32x4 / philox40 = the opt-in one-block-per-iteration streams (32-bit uniforms; 24 | 40 | 40 | 24 "
                         "bits); not the headline")
    ap.add_argument("--no-other-streams", action="store_true", help="skip the untimed pass over the opt-in random streams (--no-cpu skips it too)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo stages the tallies through host memory; for rehearsing N > 1 on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    rt = importlib.import_module("radiative-transfer_amd")
    workloads = importlib.import_module("radiative-transfer_amd.workloads")
    D = importlib.import_module("radiative-transfer_amd.distributed")
    cfg = workloads.WORKLOADS[args.workload]()

    # CPU baselines first: the worker processes are forked while this process has not touched the GPU
    cpu, cpu_rows = {}, []
    if world == 1 and not args.no_cpu:
        cpu["cpu_baseline"], cpu_rows = cpu_baseline(cfg, args.cpu_photons, 1)
        host_cores = len(os.sched_getaffinity(0))
        procs = args.cpu_procs or host_cores
        if procs > 1:
            cpu["cpu_baseline_all_cores"], rows_all = cpu_baseline(cfg, args.cpu_photons, procs)
            cpu["cpu_baseline_all_cores"]["host_cores"] = host_cores
            if len(rows_all) > len(cpu_rows):
                cpu_rows = rows_all

    import torch
    import torch.distributed as dist

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
    base_photons = args.photons or cfg["n_photons"]
    n_slabs = args.slabs
    if n_slabs % world:
        raise SystemExit(f"--gpus must divide the {n_slabs} photon-id slabs of an atmosphere")
    total_photons = base_photons * world if args.scaling == "weak" else base_photons
    per_gpu = total_photons / world
    tau = cfg["tau_list"]
    n_atm = len(tau)

    eng = rt.Engine(local_dev)
    eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"])
    dev = torch.device("cuda", local_dev)
    buf = D.TallyBuffer(torch, n_atm, n_slabs, N_MU_BINS, dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i, rng=args.rng):
        prm = rt.make_params(tau, total_photons, boundary=cfg["boundary"], scatter=cfg["scatter"], order=cfg["order"],
                             n_mu_bins=N_MU_BINS, n_slabs=n_slabs, rank=rank, world=world, h=cfg["h"],
                             monochromatic=cfg["monochromatic"], photon_id_base=i * total_photons, rng=rng,
                             table_of_atm=cfg.get("table_of_atm"))
        eng.run_device(prm, *buf.pointers(), stream)
        buf.all_reduce()  # the step's only collective: every slab entry is non-zero on exactly one rank
        # scatter events of this step (whole job after the all-reduce), kept on the device
        return buf.counts.view(n_atm, A.NCNT)[:, A.CNT_SCATTER].sum()

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
    res = buf.result()                   # the last step's tallies (whole job)
    last = res.counts.astype(np.int64)
    events_last = int(last[:, A.CNT_SCATTER].sum())
    events_rank_last = events_last / world  # equal shares by construction

    kernel_name = eng.last_kernel()  # the instantiation the library picked for these flags

    # Outside the timed region, one GPU only: the same sweep on the two opt-in streams that spend one Philox block per
    # loop iteration, each with its own comparison against the CPU sample -- reported beside the headline, never as it.
    other_streams = {}
    if world == 1 and args.rng == "philox52" and not (args.no_other_streams or args.no_cpu):
        n_alt = max(1, min(args.steps, 5))
        for alt in ("philox40", "philox32x4"):
            step(0, alt)
            fence()
            eng.kernel_time(reset=True)
            n_ev = torch.zeros((), dtype=torch.int64, device=dev)
            t1 = time.perf_counter()
            for i in range(n_alt):
                n_ev = n_ev + step(1 + i, alt)
            fence()
            dt_alt = time.perf_counter() - t1
            ms_alt, _ = eng.kernel_time()
            other_streams[alt] = {"value": int(n_ev.item()) / dt_alt, "unit": "events/s", "steps": n_alt,
                                  "ms_per_step": dt_alt / n_alt * 1e3, "kernel_avg_ms": ms_alt, "kernel": eng.last_kernel(),
                                  "stat_parity": stat_parity(buf.result(), cpu_rows, args.cpu_photons)}
    if rank == 0:
        value = events / dt
        achieved = events_rank_last * FP64_OPS_PER_EVENT / (kernel_ms * 1e-3) / 1e12
        # counters cannot be collected inside this process: the figures of the latest committed counter passes of this
        # kernel on this workload, each with the photon count it was taken at (traffic scaled to this launch's)
        traffic = committed_json("*_traffic.json", {"workload": cfg["name"]})
        pmc = committed_json("*_pmc.json", {"workload": cfg["name"], "kernel": kernel_name})
        out = {
            "metric": "photon-scatter events/sec", "value": value, "unit": "events/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{cfg['name']}: {n_atm} atmospheres tau={tau[0]:.3g}..{tau[-1]:.3g} (log-spaced) x "
                                   f"{total_photons:.3g} photons per atmosphere in total ({args.scaling} scaling: "
                                   f"{per_gpu:.3g} per GPU), {cfg['describe']}",
                       "photons_per_atmosphere_global": total_photons, "photon_histories_per_step": int(last[:, A.CNT_PHOTONS].sum()),
                       "flights_per_step": int(last[:, A.CNT_FLIGHTS].sum()), "events_per_step": events_last,
                       "rng": "Philox4x32-10, " + {"philox52": "52-bit uniforms, 2 blocks per loop iteration",
                                                   "philox32x4": "32-bit uniforms, 1 block per loop iteration (opt-in mode)",
                                                   "philox40": "1 block per loop iteration cut as azimuth 24 | cosine 40 | step 40 | "
                                                               "absorption 24 bits (opt-in mode)"}[args.rng],
                       "parallelism": f"photon-id slabs ({n_slabs} per atmosphere) over {world} GPU(s), "
                                      + ("no collective" if world == 1 else "1 all-reduce per step (one int64 buffer: slab sums "
                                                                           f"as bit patterns, counts, histogram; {buf.words.numel() * 8} bytes, {args.backend})")},
            "photon_histories_per_s": float(last[:, A.CNT_PHOTONS].sum()) * args.steps / dt,
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                         "traffic": None if traffic is None else float(traffic["bytes_per_launch"]) * per_gpu / traffic["photons_per_atmosphere"],
                         "traffic_source": None if traffic is None else
                         f"L2-to-fabric bytes of one launch at {traffic['photons_per_atmosphere']:.3g} photons per atmosphere "
                         f"({traffic['bytes_per_launch']:.4g} B), scaled to {per_gpu:.3g}",
                         "kernel": kernel_name, "kernel_avg_ms": kernel_ms, "kernel_launches": launches,
                         "flop_per_event": FP64_OPS_PER_EVENT,
                         # issue side, from the committed counter pass of this kernel: share of the SIMDs' cycles that
                         # issue a vector instruction, and the share of the 64 lanes those instructions use
                         "valu_issue_frac": None if pmc is None else pmc.get("valu_issue_frac"),
                         "lane_util": None if pmc is None else pmc.get("lane_util"),
                         "valu_instr_per_64_flights": None if pmc is None else pmc.get("valu_instr_per_64_flights"),
                         "pmc_source": None if pmc is None else pmc.get("source")},
        }
        out.update(cpu)
        out["stat_parity"] = stat_parity(res, cpu_rows, args.cpu_photons)
        if other_streams:
            out["opt_in_streams"] = other_streams
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
