"""profiles/<name>_traffic.json from the passes of tools/pmc_traffic.sh:
    python tools/traffic_json.py gpurun_out/traffic mix_2e7 20000000 > profiles/r01_traffic.json
"""
import csv
import glob
import json
import os
import sys


def main(root, workload, photons):
    tot = {}
    for path in sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv"))):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if "rt::walk_" in row["Kernel_Name"]:
                    tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    g = lambda k: tot.get(k, 0.0)
    out = {
        "workload": workload, "photons_per_atmosphere": int(photons),
        "source": "tools/pmc_traffic.sh (rocprofv3 --pmc, one pass per counter group, one launch of the production walk kernel each)",
        "FETCH_SIZE_KB": g("FETCH_SIZE"), "WRITE_SIZE_KB": g("WRITE_SIZE"),
        "TCC_EA0_RDREQ": g("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_32B": g("TCC_EA0_RDREQ_32B_sum"),
        "TCC_EA0_WRREQ": g("TCC_EA0_WRREQ_sum"), "TCC_EA0_WRREQ_64B": g("TCC_EA0_WRREQ_64B_sum"),
        "TCC_HIT": g("TCC_HIT_sum"), "TCC_MISS": g("TCC_MISS_sum"),
        "bytes_per_launch": (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0,
        "note": "L2-to-fabric traffic (FETCH_SIZE = TCC_EA0_RDREQ x 64 B). The 4.8 MB of tables (1.6 MB alias table + "
                "3.2 MB records) stay in the Infinity Cache, so almost none of this reaches HBM. gfx950 FETCH_SIZE is "
                "calibrated only for wide coalesced streams (where it reads half the bytes); for these 16-byte gathers the "
                "figure may be low by up to 2x. Algorithmic gather volume at the L2: 48 B per photon history.",
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:4])
