"""C-ABI surface (no GPU needed): the library loads, exports every symbol include/naqs_hip.h
declares, and its host-only logic (term grouping, argument validation) behaves."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden
from naqs_amd import _lib, packing


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "naqs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(naqs_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_what_the_binding_binds():
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.lib_path())
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert _lib.load_library().naqs_abi_version() == _lib.ABI_VERSION == 9


def test_strerror_and_device_count():
    lib = _lib.load_library()
    assert lib.naqs_strerror(0) == b"ok"
    assert b"invalid" in lib.naqs_strerror(-1)
    assert lib.naqs_device_count() >= 0


@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2", "Li2O", "N2_1.5"])
def test_terms_group_matches_reference_dedupe(mol):
    """naqs_terms_group == np.unique(XY, return_inverse) of hamiltonian.py:248 turned into a CSR."""
    h = golden(f"ham_{mol}.npz")
    ham = packing.load_packed(os.path.join(ROOT, "tests", "golden", f"ham_{mol}.npz"))
    g = ham.grouped()
    assert np.array_equal(g["xy_g"], h["unique_xy"])
    counts = np.bincount(h["unique2all_xy"], minlength=len(h["unique_xy"]))
    assert np.array_equal(np.diff(g["row_ptr"]), counts)
    # terms keep ascending original order inside a group (the reference's summation order)
    for gi in range(len(g["xy_g"])):
        o = g["order"][g["row_ptr"][gi]:g["row_ptr"][gi + 1]]
        assert np.all(np.diff(o) > 0) and np.all(h["unique2all_xy"][o] == gi)
    assert np.array_equal(g["yz_t"], h["yz"][g["order"]]) and np.array_equal(g["c_t"], h["coeff"][g["order"]])


def test_terms_group_empty_and_invalid():
    lib = _lib.load_library()
    kxy = ctypes.c_int64(-1)
    rp = np.zeros(1, np.int32)
    assert lib.naqs_terms_group(0, None, None, None, ctypes.byref(kxy), None, rp.ctypes.data, None, None, None) == 0
    assert kxy.value == 0 and rp[0] == 0
    assert lib.naqs_terms_group(-1, None, None, None, ctypes.byref(kxy), None, rp.ctypes.data, None, None, None) == -1
    assert lib.naqs_terms_group(3, None, None, None, ctypes.byref(kxy), None, rp.ctypes.data, None, None, None) == -1


def test_argument_validation_without_device():
    lib = _lib.load_library()
    out = ctypes.c_void_p()
    xy = np.zeros(1, np.uint64)
    c = np.zeros(1, np.float64)
    assert lib.naqs_ham_create(0, 1, 1, 1, xy.ctypes.data, xy.ctypes.data, c.ctypes.data, 0, ctypes.byref(out)) == -1
    assert lib.naqs_ham_create(65, 1, 1, 1, xy.ctypes.data, xy.ctypes.data, c.ctypes.data, 0, ctypes.byref(out)) == -4
    assert lib.naqs_ham_create(4, 1, 1, 1, xy.ctypes.data, xy.ctypes.data, c.ctypes.data, 0, None) == -1
    bad = np.array([1 << 10], np.uint64)     # mask outside n_qubits
    assert lib.naqs_ham_create(4, 1, 1, 1, bad.ctypes.data, xy.ctypes.data, c.ctypes.data, 0, ctypes.byref(out)) == -1
    assert lib.naqs_eloc(None, 1, None, None, 0, 0, 1, None, None) == -1
    assert lib.naqs_ham_destroy(None) == 0
    assert lib.naqs_popcount_parity(None, 3, 4, None, None) == -1
    if lib.naqs_device_count() == 0:
        assert lib.naqs_ham_create(4, 1, 1, 1, xy.ctypes.data, xy.ctypes.data, c.ctypes.data, 0,
                                   ctypes.byref(out)) == -5


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setenv("NAQS_HIP_LIB", "/nonexistent/libnaqs_hip.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.NaqsError):
        _lib.load_library()


@pytest.mark.gpu
def test_torch_free_cpp_client_through_the_c_abi(tmp_path):
    """tools/abi_smoke.cpp: a C++ program that includes only include/naqs_hip.h and links libnaqs_hip.so — no Python, no
    torch in the process — creates a Hamiltonian handle from a dumped golden case (LiH and N2, the reference's own
    E_loc values), runs naqs_eloc and compares.  The binary is built by __graft_entry__.build() (or here if missing)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    exe = ge.build_abi_smoke()
    for mol, tag in (("LiH", "c1"), ("N2", "small")):
        dump = str(tmp_path / f"{mol}.bin")
        subprocess.check_call([sys.executable, os.path.join(root, "tools", "dump_case.py"), mol, tag, dump])
        r = subprocess.run([exe, dump], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert "-> OK" in r.stdout, r.stdout
