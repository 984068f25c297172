"""Packs the per-grain-size Draine tables of a checkout of the reference repository into one
.npz per species under radiative-transfer_amd/data/ (same numbers as the text files, parsed with
float()).  Data preparation, run once in the build container:

    python tools/pack_draine.py /root/reference
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tables = importlib.import_module("radiative-transfer_amd.tables")


def main(ref):
    for species, folder in tables.SPECIES_DIR.items():
        gs = tables.read_draine_folder(os.path.join(ref, folder), species)
        out = os.path.join(tables.DATA_DIR, f"draine_{species}.npz")
        gs.save(out)
        print(species, len(gs.a), "sizes", gs.a.min(), "..", gs.a.max(), "rows", len(gs.wl[0]), os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
