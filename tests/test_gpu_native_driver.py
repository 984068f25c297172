"""The C ABI from a caller without Python: tests/native/abi_driver (C++, built by __graft_entry__.build() against
include/rt_amd.h and librt_amd.so) calls every entry point and checks what a caller can check on its own --
histories end once, reruns are identical, the static sweep equals the allocating one bit for bit, rank shares add up
to the whole, the library's collective with a communicator of one, refusals carry their error codes."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
DRIVER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "abi_driver")


def test_native_driver_passes():
    assert os.path.exists(DRIVER), "tests/native/abi_driver is not built: python -c 'import __graft_entry__ as g; g.build()'"
    out = subprocess.run([DRIVER], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
