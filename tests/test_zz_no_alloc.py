"""rt_run_device_static: the sweep entry point that allocates nothing (its work memory is a static device array).
Called through the C ABI and compared with rt_run_device on the same inputs: identical bits, on two calls in a row,
the second with a smaller input (the array keeps the first call's contents), and once more after a larger one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(n=96):
    wl = np.round(np.linspace(0.1, 1.0, n), 4)
    return wl, np.linspace(0.3, 0.8, n), np.linspace(0.4, 0.95, n), np.linspace(2.0, 0.5, n), np.linspace(1.0, 0.1, n) / n


def _run(rt, eng, torch, tau, n_photons, static, n_depth_bins=0, **kw):
    A = rt._abi
    prm = rt.make_params(tau, n_photons, n_mu_bins=10, n_depth_bins=n_depth_bins, **kw)
    na, ns = prm.n_atm, 8
    dev = torch.device("cuda", 0)
    # buffers start as garbage: the entry point must clear what has to start at zero
    slab = torch.full((na * ns * A.NSUM,), 7.0, dtype=torch.float64, device=dev)
    counts = torch.full((na * A.NCNT,), 7, dtype=torch.int64, device=dev)
    hist = torch.full((na * 11,), 7, dtype=torch.int64, device=dev)
    depth = torch.full((max(na * ns * 2 * n_depth_bins, 1),), 7.0, dtype=torch.float64, device=dev)
    eng.run_device(prm, slab.data_ptr(), counts.data_ptr(), hist.data_ptr(), depth.data_ptr() if n_depth_bins else None,
                   torch.cuda.current_stream().cuda_stream, static=static)
    torch.cuda.synchronize()
    return slab.cpu().numpy(), counts.cpu().numpy(), hist.cpu().numpy(), depth.cpu().numpy()


def test_static_entry_point_matches_the_allocating_one(rt):
    import torch
    eng = rt.Engine(0)
    eng.set_tables(*_tables())
    big = (np.logspace(-2, 1.5, 24), 400000)
    small = (np.array([0.2, 3.0, 30.0]), 5003)          # fewer atmospheres, fewer and ragged chunks
    want = {k: _run(rt, eng, torch, *v, static=False) for k, v in (("big", big), ("small", small))}
    for name in ("big", "small", "big", "small", "small"):  # back to back: the array holds the previous call's data
        got = _run(rt, eng, torch, *(big if name == "big" else small), static=True)
        for g, w in zip(got[:3], want[name][:3]):
            assert np.array_equal(g, w), name
    assert int(want["small"][1].reshape(-1, 10)[:, rt._abi.CNT_PHOTONS].sum()) == 3 * 5003
    # other kernels and options go through the same work memory: depth bins (per-chunk rows of their own), the
    # run-time kernel, a rank's share
    for kw in (dict(boundary="absorb", n_depth_bins=8), dict(n_depth_bins=32, runtime_kernel=True), dict(rank=1, world=4),
               dict(order="gray", h=1.0)):
        a = _run(rt, eng, torch, *small, static=False, **kw)
        b = _run(rt, eng, torch, *small, static=True, **kw)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        if kw.get("n_depth_bins"):
            assert np.array_equal(a[3], b[3]) and np.any(a[3] != 0)
    eng.close()


def test_static_entry_point_refuses_what_does_not_fit(rt):
    import torch
    eng = rt.Engine(0)
    eng.set_tables(*_tables())
    with pytest.raises(rt.RtError, match="at most 4096 atmospheres"):
        _run(rt, eng, torch, np.full(4097, 1.0), 10, static=True)
    with pytest.raises(rt.RtError, match="work memory"):
        _run(rt, eng, torch, np.full(100, 1.0), 2 * 10 ** 9, static=True)      # 12 million chunks
    # nothing was enqueued by the refused calls; the next one is unaffected
    a = _run(rt, eng, torch, np.array([1.0]), 1000, static=True)
    b = _run(rt, eng, torch, np.array([1.0]), 1000, static=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    eng.close()


def test_static_entry_point_leaves_device_memory_alone(rt):
    """Free device memory is the same before and after static sweeps of growing size on a fresh context, while the
    allocating entry point takes its work memory from the device on its first large call.  (What is compared is the
    device's free memory, so the test sees allocations by anyone in the process, the runtime included.)"""
    import torch
    A = rt._abi
    eng = rt.Engine(0)
    eng.set_tables(*_tables())
    tau, n = np.logspace(-2, 1.5, 40), 2000000
    dev = torch.device("cuda", 0)
    slab = torch.zeros(len(tau) * 8 * A.NSUM, dtype=torch.float64, device=dev)
    counts = torch.zeros(len(tau) * A.NCNT, dtype=torch.int64, device=dev)
    hist = torch.zeros(len(tau) * 11, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # one small call first: the HIP runtime sets up its own pools (kernel arguments, signals) at a process's first launches
    eng.run_device(rt.make_params(tau[:1], 1000, n_mu_bins=10), slab.data_ptr(), counts.data_ptr(), hist.data_ptr(), None,
                   stream, static=True)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for k in (1, 4, 40):   # 1, 4, 40 atmospheres: ever more plan entries and chunk partial sums
        p = rt.make_params(tau[:k], n, n_mu_bins=10)
        eng.run_device(p, slab.data_ptr(), counts.data_ptr(), hist.data_ptr(), None, stream, static=True)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free0
    # the allocating entry point on a sweep whose work memory is tens of MB (40 atmospheres x 2e7 photons: 97 920 chunks)
    big = rt.make_params(tau, 20000000, n_mu_bins=10)
    eng.run_device(big, slab.data_ptr(), counts.data_ptr(), hist.data_ptr(), None, stream, static=False)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] < free0
    eng.run_device(big, slab.data_ptr(), counts.data_ptr(), hist.data_ptr(), None, stream, static=True)   # fits the array: 8.9 MB
    torch.cuda.synchronize()
    eng.close()


def test_static_sweeps_from_two_streams_and_contexts_take_turns(rt):
    """All static sweeps of a GPU share one work array.  Sweeps enqueued from two contexts on two streams, with no
    host synchronisation and no event of the caller's between them, must still give the allocating entry point's
    bits: the library makes each wait for the previous one's last kernel.  (A build without that wait passes this test
    too on most runs: a walk launch fills every wave slot of the GPU, so the next sweep's first operations only get
    onto the device while the last waves of the previous one retire, and the window in which they can hurt each other
    is a few microseconds.  The test pins the results of the pattern, the ordering itself is by construction.)"""
    import torch
    A = rt._abi
    dev = torch.device("cuda", 0)
    engs = [rt.Engine(0), rt.Engine(0)]
    for e in engs:
        e.set_tables(*_tables())
    # the first and third sweep run for milliseconds, the host needs tens of microseconds to enqueue the next one
    cases = [(np.logspace(-1, 1.3, 12), 20000000, 0), (np.array([0.05, 2.0, 20.0, 7.0]), 90001, 1),
             (np.logspace(-2, 1.0, 30), 6000000, 2), (np.array([5.0]), 1200000, 3)]
    want = [_run(rt, engs[0], torch, tau, n, static=False, seed=7 + s) for tau, n, s in cases]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    outs = []
    for rep in range(3):
        for k, (tau, n, s) in enumerate(cases):
            prm = rt.make_params(tau, n, n_mu_bins=10, seed=7 + s)
            slab = torch.full((prm.n_atm * 8 * A.NSUM,), 7.0, dtype=torch.float64, device=dev)
            counts = torch.full((prm.n_atm * A.NCNT,), 7, dtype=torch.int64, device=dev)
            hist = torch.full((prm.n_atm * 11,), 7, dtype=torch.int64, device=dev)
            outs.append((k, slab, counts, hist))
            # alternate context and stream; the long first case is still running when the next one is enqueued
            engs[(k + rep) % 2].run_device(prm, slab.data_ptr(), counts.data_ptr(), hist.data_ptr(), None,
                                           streams[k % 2].cuda_stream, static=True)
    torch.cuda.synchronize()
    for k, slab, counts, hist in outs:
        assert np.array_equal(slab.cpu().numpy(), want[k][0])
        assert np.array_equal(counts.cpu().numpy(), want[k][1])
        assert np.array_equal(hist.cpu().numpy(), want[k][2])
    for e in engs:
        e.close()
