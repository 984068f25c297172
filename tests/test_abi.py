"""The C-ABI library loads and exports every symbol include/rt_amd.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported(rt):
    lib = rt._abi.load_library()
    names = declared_functions()
    assert set(names) == set(rt._abi.EXPORTS)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.rt_abi_version() == rt._abi.RT_ABI_VERSION


def test_struct_layout_matches_header(rt):
    A = rt._abi
    # the size word + 12 int32 + a reserved one, 2 uint64, 3 doubles, 2 pointers, 2 int32, 1 pointer, 1 double, 1 pointer
    assert ctypes.sizeof(A.RtParams) == 14 * 4 + 2 * 8 + 3 * 8 + 2 * 8 + 2 * 4 + 8 + 8 + 8
    assert ctypes.sizeof(A.RtTallies) == 7 * 8
    assert ctypes.sizeof(A.RtPhotonTrace) == 32
    assert A.RtParams.struct_size.offset == 0 and A.RtParams.seed.offset == 56 and A.RtParams.tau_atm.offset == 96


def test_header_is_plain_c_and_layouts_match_ctypes(rt, tmp_path):
    """include/rt_amd.h compiles as C11 with gcc, and every struct has the size and field offsets of its
    ctypes mirror."""
    import subprocess
    A = rt._abi
    pairs = {"rt_params": A.RtParams, "rt_tallies": A.RtTallies, "rt_photon_trace": A.RtPhotonTrace,
             "rt_grain_set": A.RtGrainSet, "rt_grain_tables": A.RtGrainTables, "rt_shell_model": A.RtShellModel,
             "rt_shell_galaxy": A.RtShellGalaxy,
             "rt_shell_regions": A.RtShellRegions, "rt_shell_control": A.RtShellControl, "rt_shell_result": A.RtShellResult}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rt_amd.h"', 'int main(void) {']
    for cname, ct in pairs.items():
        lines.append(f'printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for line in filter(None, out):
        cname, field, value = line.split()
        ct = pairs[cname]
        want = ctypes.sizeof(ct) if field == "size" else getattr(ct, field).offset
        assert int(value) == want, line
        seen += 1
    assert seen == sum(len(ct._fields_) + 1 for ct in pairs.values())


def test_no_cpu_fallback(rt):
    """Without a GPU the engine must refuse to start rather than compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError, match="no HIP device|no CPU fallback|failed"):
        rt.Engine(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "radiative-transfer_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(base, f)).read()
                assert "oracle" not in text.replace("# oracle", ""), f
