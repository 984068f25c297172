"""Sums rocprofv3 --pmc CSV rows of the walk kernel over the passes written by tools/profile_pmc.sh.

    python tools/summarize_pmc.py gpurun_out/<tag> > profiles/<name>.md
"""
import csv
import glob
import json
import os
import sys


def main(root):
    totals, ndisp = {}, {}
    for path in sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv"))):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if "rt::walk_" not in row["Kernel_Name"]:
                    continue
                name = row["Counter_Name"]
                totals[name] = totals.get(name, 0.0) + float(row["Counter_Value"])
                ndisp.setdefault(name, set()).add(row["Dispatch_Id"])
    bench = None
    for log in sorted(glob.glob(os.path.join(root, "*.log"))):
        for line in open(log):
            if line.startswith("{"):
                bench = json.loads(line)
    print(f"# walk_kernel hardware counters ({root})\n")
    if bench:
        c = bench["config"]
        print(f"bench: {c['workload']}; events/step {c['events_per_step']}, flights/step {c['flights_per_step']}, "
              f"histories/step {c['photon_histories_per_step']}, kernel avg {bench['roofline']['kernel_avg_ms']:.3f} ms (profiled run)\n")
    print("| counter | total over walk_kernel dispatches | dispatches | per dispatch |")
    print("|---|---|---|---|")
    for k in sorted(totals):
        n = len(ndisp[k])
        print(f"| {k} | {totals[k]:.6g} | {n} | {totals[k] / n:.6g} |")
    t = {k: totals[k] / len(ndisp[k]) for k in totals}
    print("\n## derived (per dispatch)\n")
    if "SQ_THREAD_CYCLES_VALU" in t and "SQ_ACTIVE_INST_VALU" in t:
        print(f"- VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = "
              f"{t['SQ_THREAD_CYCLES_VALU'] / (64 * t['SQ_ACTIVE_INST_VALU']):.3f}")
    if "SQ_ACTIVE_INST_VALU" in t and "SQ_WAVE_CYCLES" in t:
        print(f"- share of wave-cycles issuing VALU = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {t['SQ_ACTIVE_INST_VALU'] / t['SQ_WAVE_CYCLES']:.3f}")
        print(f"- share of wave-cycles waiting (SQ_WAIT_ANY) = {t.get('SQ_WAIT_ANY', 0) / t['SQ_WAVE_CYCLES']:.3f}; "
              f"issue-stalled (SQ_WAIT_INST_ANY) = {t.get('SQ_WAIT_INST_ANY', 0) / t['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_ACTIVE_INST_VALU" in t and "SQ_BUSY_CYCLES" in t:
        print(f"- SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {t['SQ_ACTIVE_INST_VALU'] / t['SQ_BUSY_CYCLES']:.3f}")
    if bench and "SQ_INSTS_VALU" in t:
        fl = bench["config"]["flights_per_step"]
        ev = bench["config"]["events_per_step"]
        print(f"- VALU wave-instructions per 64 flights = {t['SQ_INSTS_VALU'] / (fl / 64):.1f}; per 64 events = {t['SQ_INSTS_VALU'] / (ev / 64):.1f}")
    if "FETCH_SIZE" in t:
        print(f"- FETCH_SIZE = {t['FETCH_SIZE']:.6g} KB per dispatch (gfx950: x2 for wide coalesced streams; gathers uncalibrated)")
    if "WRITE_SIZE" in t:
        print(f"- WRITE_SIZE = {t['WRITE_SIZE']:.6g} KB per dispatch")
    if "TCC_HIT_sum" in t and "TCC_MISS_sum" in t:
        print(f"- L2 hit rate = {t['TCC_HIT_sum'] / (t['TCC_HIT_sum'] + t['TCC_MISS_sum']):.4f}")


if __name__ == "__main__":
    main(sys.argv[1])
