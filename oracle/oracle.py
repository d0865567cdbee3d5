"""ctypes wrapper of oracle/libnaqs_oracle.so (+ numpy glue).

TEST INFRASTRUCTURE — see the header of naqs_oracle.c.  Only tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnaqs_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "naqs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnaqs_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_eloc_staged.restype = ctypes.c_int
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def set_threads(n):
    lib().oracle_set_threads(int(n))


def max_threads():
    return lib().oracle_max_threads()


def popcount_parity(arr):
    """Restates src.utils.hamiltonian_math.popcount_parity incl. the 1-D reshape and the TypeError."""
    arr = np.ascontiguousarray(arr)
    if arr.ndim == 1:
        arr = arr.reshape(-1, 1)
    fn = {np.dtype(np.int16): "oracle_popcount_parity_i16", np.dtype(np.int32): "oracle_popcount_parity_i32",
          np.dtype(np.int64): "oracle_popcount_parity_i64"}.get(arr.dtype)
    if fn is None:
        raise TypeError(f"Unsupported array dtype for popcount_parity(...): {arr.dtype}.")
    out = np.empty(arr.shape, np.int8)
    getattr(lib(), fn)(_p(arr), _p(out), ctypes.c_int64(arr.shape[0]), ctypes.c_int64(arr.shape[1]))
    return out


def dedupe(xy, yz):
    """hamiltonian.py:248-252."""
    uxy, u2a_xy = np.unique(xy, return_inverse=True)
    uyz, u2a_yz = np.unique(yz, return_inverse=True)
    return uxy.astype(np.uint64), u2a_xy.astype(np.int64), uyz.astype(np.uint64), u2a_yz.astype(np.int64)


def get_hij(keys, xy, yz, coeff):
    """popcount table + get_Hij_cy on the deduped masks -> dense [M*Kxy] like the reference."""
    keys = np.ascontiguousarray(keys, np.uint64)
    uxy, u2a_xy, uyz, u2a_yz = dedupe(xy, yz)
    P = popcount_parity((keys[:, None] & uyz[None, :]).astype(np.int64))
    M, Kxy, K, Kyz = len(keys), len(uxy), len(xy), len(uyz)
    out = np.empty(M * Kxy, np.float64)
    c = np.ascontiguousarray(coeff, np.float64)
    lib().oracle_get_hij(ctypes.c_int64(M), ctypes.c_int64(Kxy), ctypes.c_int64(K), ctypes.c_int64(Kyz),
                         _p(u2a_xy), _p(P), _p(u2a_yz), _p(c), _p(out))
    return out, P


def csr_mv(data, indices, indptr, v):
    data = np.ascontiguousarray(data, np.float64)
    indices = np.ascontiguousarray(indices, np.int32)
    indptr = np.ascontiguousarray(indptr, np.int32)
    vv = np.ascontiguousarray(np.stack([v.real, v.imag], -1), np.float64)
    out = np.empty((len(indptr) - 1, 2), np.float64)
    lib().oracle_csr_mv(ctypes.c_int64(len(indptr) - 1), _p(data), _p(indices), _p(indptr), _p(vv), _p(out))
    return out[:, 0] + 1j * out[:, 1]


def _psi64(psi):
    psi = np.asarray(psi)
    if np.iscomplexobj(psi):
        psi = np.stack([psi.real, psi.imag], -1)
    return np.ascontiguousarray(psi, np.float64)


def eloc_staged(n_qubits, n_alpha, n_beta, xy, yz, coeff, keys, psi):
    """calculate_local_energy restated stage by stage.  keys must be ascending & unique."""
    keys = np.ascontiguousarray(keys, np.uint64)
    assert np.all(keys[1:] > keys[:-1]), "oracle_eloc_staged wants ascending unique keys"
    uxy, u2a_xy, uyz, u2a_yz = dedupe(xy, yz)
    c = np.ascontiguousarray(coeff, np.float64)
    p = _psi64(psi)
    out = np.empty((len(keys), 2), np.float64)
    st = lib().oracle_eloc_staged(int(n_qubits), int(n_alpha), int(n_beta), ctypes.c_int64(len(xy)),
                                  ctypes.c_int64(len(uxy)), ctypes.c_int64(len(uyz)), _p(uxy), _p(u2a_xy),
                                  _p(uyz), _p(u2a_yz), _p(c), ctypes.c_int64(len(keys)), _p(keys), _p(p), _p(out))
    if st != 0:
        raise MemoryError("oracle_eloc_staged")
    return out[:, 0] + 1j * out[:, 1]


def group_terms(xy, yz, coeff):
    """CSR by unique xy, numpy restatement (stable: ascending original index inside a group)."""
    xy = np.asarray(xy, np.uint64)
    order = np.argsort(xy, kind="stable")
    xs = xy[order]
    starts = np.flatnonzero(np.r_[True, xs[1:] != xs[:-1]]) if len(xs) else np.zeros(0, np.int64)
    row_ptr = np.r_[starts, len(xs)].astype(np.int32)
    return (xs[starts].astype(np.uint64), row_ptr, np.asarray(yz, np.uint64)[order].copy(),
            np.asarray(coeff, np.float64)[order].copy(), order.astype(np.int64))


def eloc_matrix_free(xy, yz, coeff, keys, psi, row_begin=0, n_rows=None):
    """Direct formula; keys in ANY order (sorted internally), rows refer to the given order."""
    keys = np.ascontiguousarray(keys, np.uint64)
    perm = np.argsort(keys, kind="stable")
    ks = np.ascontiguousarray(keys[perm])
    ps = np.ascontiguousarray(_psi64(psi)[perm])
    xy_g, row_ptr, yz_t, c_t, _ = group_terms(xy, yz, coeff)
    full = np.empty((len(keys), 2), np.float64)
    lib().oracle_eloc_matrix_free(ctypes.c_int64(len(xy_g)), _p(xy_g), _p(row_ptr), _p(yz_t), _p(c_t),
                                  ctypes.c_int64(len(ks)), _p(ks), _p(ps), ctypes.c_int64(0),
                                  ctypes.c_int64(len(ks)), _p(full))
    e = np.empty(len(keys), np.complex128)
    e[perm] = full[:, 0] + 1j * full[:, 1]
    if n_rows is None:
        n_rows = len(keys) - row_begin
    return e[row_begin:row_begin + n_rows]


def eloc_reduce(w, eloc):
    w = np.ascontiguousarray(w, np.float64)
    e = np.ascontiguousarray(np.stack([eloc.real, eloc.imag], -1), np.float64)
    out = np.empty(4, np.float64)
    lib().oracle_eloc_reduce(ctypes.c_int64(len(w)), _p(w), _p(e), _p(out))
    return out
