"""profiles/<round>_* from one profiling pass:  python tools/write_profiles.py gpurun_out/<tag> <round, e.g. r02> "<one-line description of the kernel>"

<tag>/trace   rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu`   (+ trace_bench.json, its JSON line)
<tag>/pmc     tools/profile_pmc.sh        <tag>/mem      tools/pmc_mem.sh
<tag>/traffic tools/pmc_traffic.sh        <tag>/tau.log  tools/time_tau.py      <tag>/variants.log  tools/time_variants.py
"""
import csv
import glob
import io
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def summary(path):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), path], check=True,
                          capture_output=True, text=True).stdout.split("\n")


def val(lines, name):
    for l in lines:
        if l.startswith("| " + name + " "):
            return float(l.split("|")[4])  # per dispatch: a counter collected in several passes has several dispatches
    raise KeyError(name)


def main(root, rnd, what):
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(glob.glob(os.path.join(root, "trace", "*", "*kernel_stats.csv"))[0], os.path.join(prof, f"{rnd}_kernel_stats.csv"))
    with open(os.path.join(prof, f"{rnd}_traffic.json"), "w") as fh:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_json.py"), os.path.join(root, "traffic"), "mix_2e7",
                        "20000000"], check=True, stdout=fh)
    shutil.copy(os.path.join(root, "tau.log"), os.path.join(prof, f"{rnd}_per_tau.txt"))
    shutil.copy(os.path.join(root, "variants.log"), os.path.join(prof, f"{rnd}_variants.txt"))
    a, b = summary(os.path.join(root, "pmc")), summary(os.path.join(root, "mem"))
    a[0] = f"# walk_kernel hardware counters (tools/profile_pmc.sh, {what})"
    g, act, ins = val(a, "GRBM_GUI_ACTIVE"), val(a, "SQ_ACTIVE_INST_VALU"), val(a, "SQ_INSTS_VALU")
    ms = float(re.search(r"kernel avg ([0-9.]+) ms", a[2]).group(1))
    out = "\n".join(a).rstrip() + "\n"
    out += f"- VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = {act * 4 / (1024 * g / 8):.3f}\n"
    out += f"- mean issue cost of a VALU wave instruction = 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = {4 * act / ins:.2f} cycles\n"
    out += (f"- shader clock in this PROFILED run = (GRBM_GUI_ACTIVE / 8) / kernel time = {g / 8 / (ms * 1e-3) / 1e9:.2f} GHz "
            "(counter collection slows the kernel; the variants file has the same launch unprofiled)\n")
    out += "\n## memory path (tools/pmc_mem.sh, same workload)\n\n"
    i = [k for k, l in enumerate(b) if l.startswith("| counter")][0]
    j = [k for k, l in enumerate(b) if l.startswith("## derived")][0]
    out += "\n".join(l for l in b[i:j] if l.startswith("|") and re.search(r"TA_|TCP_|TD_|GRBM|counter|---", l)) + "\n"
    reqs, acc, lat = val(b, "TCP_TCC_READ_REQ_sum"), val(b, "TCP_TOTAL_CACHE_ACCESSES_sum"), val(b, "TCP_TCC_READ_REQ_LATENCY_sum")
    hist = float(re.search(r"histories/step (\d+)", b[2]).group(1))
    out += f"\n- L1 misses (TCP_TCC_READ_REQ) per photon history = {reqs / hist:.2f} (one alias line + one record line); L1 hit rate = {1 - reqs / acc:.2f}\n"
    out += f"- mean L2 request latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ = {lat / reqs:.0f} cycles\n"
    open(os.path.join(prof, f"{rnd}_pmc.md"), "w").write(out)
    flights_pmc = float(re.search(r"flights/step (\d+)", a[2]).group(1))
    kern = json.loads(open(os.path.join(root, "trace_bench.json")).read().strip().splitlines()[-1])["roofline"]["kernel"]
    with open(os.path.join(prof, f"{rnd}_pmc.json"), "w") as fh:
        json.dump({"workload": "mix_2e7", "kernel": kern,
                   "valu_issue_frac": act * 4 / (1024 * g / 8), "lane_util": val(a, "SQ_THREAD_CYCLES_VALU") / 64 / act,
                   "valu_instr_per_64_flights": ins / (flights_pmc / 64), "cycles_per_valu_instr": 4 * act / ins,
                   "source": f"profiles/{rnd}_pmc.md (rocprofv3 --pmc passes of tools/profile_pmc.sh at 4e6 photons per atmosphere): "
                             "valu_issue_frac = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), "
                             "lane_util = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)"}, fh, indent=1)
        fh.write("\n")
    row = next(csv.DictReader(open(os.path.join(prof, f"{rnd}_kernel_stats.csv"))))
    bench = json.loads(open(os.path.join(root, "trace_bench.json")).read().strip().splitlines()[-1])
    flights = float(re.search(r"flights/step (\d+)", a[2]).group(1))
    print(f"trace: {row['Calls']} launches, avg {float(row['AverageNs']) / 1e6:.2f} ms, min {float(row['MinNs']) / 1e6:.2f} ms; "
          f"bench {bench['roofline']['kernel_avg_ms']:.2f} ms kernel, {bench['ms_per_step']:.2f} ms/step, {bench['value']:.3g} events/s, "
          f"{bench['photon_histories_per_s']:.3g} histories/s, {bench['roofline']['achieved']:.3f} TFLOP/s = {bench['roofline']['frac']:.4f}; "
          f"VALU instr per 64 flights {ins / (flights / 64):.1f}, lane utilisation {val(a, 'SQ_THREAD_CYCLES_VALU') / 64 / act:.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "final kernel of the round")
