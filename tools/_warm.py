"""Shared by the timing scripts: the GPU's clock takes two to three 20 ms launches to ramp up from idle (the first
launches of a cold process measure up to 15 % slow), so every script first keeps the GPU busy for > 100 ms and then
reports the MEDIAN of several launches per case."""
import statistics


def warm(rt, eng, cfg, launches=6):
    prm = rt.make_params(cfg["tau_list"], cfg["n_photons"], h=cfg["h"])
    for _ in range(launches):
        eng.run(prm)


def median_kernel_ms(eng, prm, reps=7):
    """Median walk-kernel time of `reps` launches of `prm` (library's own HIP events) and the last result."""
    ts, res = [], None
    for _ in range(reps):
        eng.kernel_time(reset=True)
        res = eng.run(prm)
        ts.append(eng.kernel_time()[0])
    return statistics.median(ts), res
