"""Per-kernel table of the counter passes written by tools/pmc_variants.sh: python tools/summarize_pmc_variants.py gpurun_out/<tag>"""
import csv
import glob
import os
import sys

root = sys.argv[1]
flights = {}
for line in open(os.path.join(root, "sq1.log")):
    p = line.split()
    if len(p) == 3 and p[0].startswith("walk_kernel<"):
        flights[p[0]] = int(p[1])
tot = {}
for path in sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv"))):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if "rt::walk_" not in row["Kernel_Name"]:
                continue
            d = tot.setdefault(int(row["Dispatch_Id"]), {})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
# dispatches of the walk kernel come in the order of the cases; the two passes see the same order
per_pass = {}
for path in sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv"))):
    ids = sorted({int(r["Dispatch_Id"]) for r in csv.DictReader(open(path)) if "rt::walk_" in r["Kernel_Name"]})
    per_pass[path] = ids
names = list(flights)
rows = [dict() for _ in names]
for path, ids in per_pass.items():
    assert len(ids) == len(names), (path, len(ids), len(names))
    for k, i in enumerate(ids):
        rows[k].update(tot[i])
print("| kernel | VALU instr / 64 flights | cycles per VALU instr | lane util | wave-cycles issuing VALU | wave-cycles issue-stalled | LDS instr / 64 flights | LDS atomics / 64 fl | bank-conflict cycles / LDS active | SALU / 64 fl | VMEM rd / 64 fl | branches / 64 fl |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, t in zip(names, rows):
    f64 = flights[name] / 64
    g = lambda k: t.get(k, float("nan"))  # noqa: E731
    print(f"| {name[12:-1]} | {g('SQ_INSTS_VALU') / f64:.1f} | {4 * g('SQ_ACTIVE_INST_VALU') / g('SQ_INSTS_VALU'):.2f} | "
          f"{g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_ACTIVE_INST_VALU')):.3f} | {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.3f} | {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f} | "
          f"{g('SQ_INSTS_LDS') / f64:.1f} | {g('SQ_INSTS_LDS_ATOMIC') / f64:.1f} | {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_ACTIVE_INST_LDS'), 1):.2f} | "
          f"{g('SQ_INSTS_SALU') / f64:.1f} | {g('SQ_INSTS_VMEM_RD') / f64:.1f} | {g('SQ_INSTS_BRANCH') / f64:.1f} |")
