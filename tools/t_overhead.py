import importlib, time, numpy as np, torch, sys
sys.path.insert(0,'.')
rt = importlib.import_module("radiative-transfer_amd"); w = importlib.import_module("radiative-transfer_amd.workloads")
A = rt._abi
cfg = w.mix_2e7()
t0 = time.perf_counter(); eng = rt.Engine(0); t1 = time.perf_counter()
eng2 = rt.Engine(0); t2 = time.perf_counter(); eng2.close()
print("rt_create: first context %.1f ms (loads the code object, asks the occupancy of every instantiation), second %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
t0 = time.perf_counter(); eng.set_tables(cfg["wl"], cfg["g"], cfg["albedo"], cfg["scale"], cfg["pdf"]); print("rt_set_tables (99 991 wavelengths): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
tau = cfg["tau_list"]; n_atm=len(tau)
dev = torch.device("cuda",0)
slab = torch.zeros(n_atm*8*A.NSUM, dtype=torch.float64, device=dev); counts = torch.zeros(n_atm*A.NCNT, dtype=torch.int64, device=dev); hist = torch.zeros(n_atm*11, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for N in (20000000, 2000000):
    for it in range(4):
        torch.cuda.synchronize()
        t0=time.perf_counter()
        prm = rt.make_params(tau, N, n_mu_bins=10, photon_id_base=it*N)
        t1=time.perf_counter()
        eng.run_device(prm, slab.data_ptr(), counts.data_ptr(), hist.data_ptr(), None, stream)
        t2=time.perf_counter()
        torch.cuda.synchronize()
        t3=time.perf_counter()
        print(N, "make_params %.3f ms  run_device(host) %.3f ms  wait %.3f ms total %.3f" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t3-t0)*1e3), "kernel", eng.kernel_time(reset=True))
