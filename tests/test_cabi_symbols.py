"""CPU tier: libvlnce_hip.so builds for gfx950, loads, and exports every
function include/vlnce_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

import __graft_entry__ as ge
from vlnce_amd import _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include",
                      "vlnce_hip.h")


@pytest.fixture(scope="module")
def dll():
    if not os.path.exists("/opt/rocm/bin/hipcc") and not os.path.exists(ge.OUT):
        pytest.skip("no hipcc and no prebuilt library")
    ge.build()
    return ctypes.CDLL(ge.OUT)


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vlnce_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == _lib.exported_symbols()


def test_every_declared_symbol_is_exported(dll):
    for name in declared_functions():
        assert hasattr(dll, name), name
    dll.vlnce_version.restype = ctypes.c_int
    assert dll.vlnce_version() >= 100


def test_kernels_are_gfx950_code_objects(dll):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", ge.OUT],
                         capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("llvm-readelf unavailable")
    blob = open(ge.OUT, "rb").read()
    assert b"gfx950" in blob


def test_binding_refuses_a_library_of_another_abi(dll, monkeypatch):
    lib = _lib.HipLib()  # loads and checks vlnce_version() (no GPU needed)
    assert lib.dll.vlnce_version() == _lib.HipLib.ABI
    monkeypatch.setattr(_lib.HipLib, "ABI", _lib.HipLib.ABI + 1)
    with pytest.raises(RuntimeError, match="rebuild"):
        _lib.HipLib()
