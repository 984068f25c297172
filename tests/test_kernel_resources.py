"""The production instantiations of the walk kernel keep their register budget: no scratch memory (a spill inside the
loop costs 15-20 % of the headline rate and no parity test sees it) and at least 5 waves per SIMD.  The figures are
the compiler's own remarks for rt_walk_base.hip (device code only, no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "radiative-transfer_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_production_kernels_do_not_spill(tmp_path):
    import __graft_entry__ as entry
    out = subprocess.run(["hipcc", *entry.HIP_FLAGS, "--cuda-device-only", "-c", "rt_walk_base.hip", "-o", str(tmp_path / "base.o"),
                          "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True, check=True).stderr
    kernels = {}
    for block in out.split("Function Name: ")[1:]:
        name = block.split()[0]
        num = lambda key: int(re.search(re.escape(key) + r":? (\d+)", block).group(1))  # noqa: E731
        kernels[name] = dict(vgprs=num("VGPRs"), scratch=num("ScratchSize [bytes/lane]"), waves=num("Occupancy [waves/SIMD]"))
    # Fixed<mc, reflect, hg, no depth bins>: every stream, plain and not
    lean = {k: v for k, v in kernels.items() if "5FixedILi0ELi0ELi0ELb0ELb0ELb0ELb0EEE" in k}
    assert len(lean) == 6, sorted(kernels)
    for name, r in lean.items():
        assert r["scratch"] == 0, (name, r)
    plain = {k: v for k, v in lean.items() if k.split("EEE")[1].startswith(("Li0ELb1", "Li1ELb1", "Li2ELb1"))}
    assert len(plain) == 3, sorted(lean)
    for name, r in plain.items():
        assert r["waves"] >= 5 and r["vgprs"] <= 96, (name, r)
    assert all(r["scratch"] == 0 for r in kernels.values()), kernels  # the run-time kernels as well
