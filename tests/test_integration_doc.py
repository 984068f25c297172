"""The integration guide's hand-written ctypes binding is executed as it stands: its structs must have the size and the
field offsets of include/rt_amd.h (through the package's own mirror, which tests/test_abi.py compares with the header
compiled by gcc), so a binding typed from the document cannot hand the library a struct of another layout."""
import ctypes
import os
import re

from conftest import ROOT


def python_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def struct_block():
    blocks = [b for b in python_blocks() if "class RtParams(C.Structure)" in b]
    assert len(blocks) == 1
    return blocks[0]


def call_block():
    blocks = [b for b in python_blocks() if "lib.rt_run(" in b]
    assert len(blocks) == 1
    return blocks[0]


def test_documented_structs_match_the_header(rt):
    ns = {}
    exec(struct_block(), ns)
    A = rt._abi
    for doc, own in ((ns["RtParams"], A.RtParams), (ns["RtTallies"], A.RtTallies)):
        assert ctypes.sizeof(doc) == ctypes.sizeof(own)
        assert [n for n, _ in doc._fields_] == [n for n, _ in own._fields_]
        for name, _ in own._fields_:
            assert getattr(doc, name).offset == getattr(own, name).offset, name
            assert getattr(doc, name).size == getattr(own, name).size, name


def test_documented_calls_name_the_current_abi(rt):
    block = call_block()
    assert f"lib.rt_abi_version() == {rt._abi.RT_ABI_VERSION}" in block      # the snippet checks the version it was typed from
    assert "struct_size=C.sizeof(RtParams)" in block
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    assert f"#define RT_ABI_VERSION {rt._abi.RT_ABI_VERSION}" in header
    assert f"(ABI {rt._abi.RT_ABI_VERSION})" in struct_block()
