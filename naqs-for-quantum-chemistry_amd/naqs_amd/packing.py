"""Pauli-term pre-processing and the input formats either side of it.

Mirrors ``_PauliHamiltonianDynamic.__calc_coupling_info`` (reference
src/optimizer/hamiltonian.py:373-430) and the unique-mask dedupe (:248-252), and reads the
reference's ``<molecule>_qubit_hamiltonian.pkl`` (a pickled openfermion ``QubitOperator``,
src/utils/system.py:14-62) without openfermion.
"""
import ctypes
import os
import pickle
from dataclasses import dataclass

import numpy as np

from . import _lib


@dataclass
class PackedHamiltonian:
    """K Pauli strings as bit-masks + real couplings, in the reference's term order."""
    n_qubits: int
    n_alpha: int
    n_beta: int
    xy: np.ndarray      # uint64 [K]  bit q set iff Pauli on q is X or Y
    yz: np.ndarray      # uint64 [K]  bit q set iff Pauli on q is Y or Z
    coeff: np.ndarray   # float64 [K] Re(i^nY) * coefficient

    @property
    def K(self):
        return int(self.xy.shape[0])

    def grouped(self):
        """CSR by unique XY mask via the library's host helper -> dict of arrays."""
        lib = _lib.load_library()
        K = self.K
        xy_g = np.empty(max(K, 1), np.uint64)
        row_ptr = np.empty(K + 1, np.int32)
        yz_t = np.empty(max(K, 1), np.uint64)
        c_t = np.empty(max(K, 1), np.float64)
        order = np.empty(max(K, 1), np.int64)
        kxy = ctypes.c_int64(0)
        xy, yz, c = (np.ascontiguousarray(self.xy, np.uint64), np.ascontiguousarray(self.yz, np.uint64),
                     np.ascontiguousarray(self.coeff, np.float64))
        st = lib.naqs_terms_group(K, xy.ctypes.data, yz.ctypes.data, c.ctypes.data, ctypes.byref(kxy),
                                  xy_g.ctypes.data, row_ptr.ctypes.data, yz_t.ctypes.data, c_t.ctypes.data,
                                  order.ctypes.data)
        _lib.check(st, "naqs_terms_group")
        n = kxy.value
        return dict(xy_g=xy_g[:n].copy(), row_ptr=row_ptr[:n + 1].copy(), yz_t=yz_t[:K].copy(),
                    c_t=c_t[:K].copy(), order=order[:K].copy())


def pack_qubit_hamiltonian(terms, n_qubits, n_alpha, n_beta, n_excitations_max=None, n_occ=0):
    """``terms``: mapping ``((qubit, 'X'|'Y'|'Z'), ...) -> complex`` (openfermion QubitOperator.terms).

    Same rules as the reference (hamiltonian.py:383-430): XY mask from X/Y, YZ mask from Y/Z,
    coupling = Re(i^nY) * coefficient cast to real (imaginary part dropped), terms that flip a
    frozen qubit (< n_occ) or exceed ``n_excitations_max`` flips are skipped, dict order kept.
    """
    xy, yz, cf = [], [], []
    for term, coupling in terms.items():
        x = y = 0
        n_y = n_exc = 0
        valid = True
        for qubit, pauli in term:
            if pauli in ("X", "Y"):
                x |= 1 << qubit
                if pauli == "Y":
                    n_y += 1
                    y |= 1 << qubit
                if qubit < n_occ:
                    valid = False
                    break
                if n_excitations_max is not None:
                    n_exc += 1
                    if n_exc > n_excitations_max:
                        valid = False
                        break
            elif pauli == "Z":
                y |= 1 << qubit
        if valid:
            xy.append(x)
            yz.append(y)
            cf.append(((1j ** n_y).real * complex(coupling)).real)
    return PackedHamiltonian(int(n_qubits), int(n_alpha), int(n_beta), np.array(xy, np.uint64),
                             np.array(yz, np.uint64), np.array(cf, np.float64))


class _QubitOperatorUnpickler(pickle.Unpickler):
    """Resolves every ``openfermion.*`` global to an empty stand-in class, so the pickled
    QubitOperator loads as a plain object with a ``.terms`` dict when openfermion is absent."""

    def find_class(self, module, name):
        if module.split(".")[0] == "openfermion":
            return type(name, (), {})
        return super().find_class(module, name)


def load_qubit_hamiltonian_pkl(path):
    """Read ``<molecule>_qubit_hamiltonian.pkl`` -> object with ``.terms``."""
    with open(path, "rb") as f:
        try:
            import openfermion  # noqa: F401
            return pickle.load(f)
        except ImportError:
            return _QubitOperatorUnpickler(f).load()


def n_qubits_of_terms(terms):
    return 1 + max((q for term in terms for q, _ in term), default=-1)


def load_packed(path):
    """Load a packed-term ``.npz`` (keys n_qubits, n_alpha, n_beta, xy, yz, coeff)."""
    with np.load(path) as z:
        return PackedHamiltonian(int(z["n_qubits"]), int(z["n_alpha"]), int(z["n_beta"]),
                                 z["xy"].astype(np.uint64), z["yz"].astype(np.uint64),
                                 z["coeff"].astype(np.float64))


def save_packed(path, ham):
    np.savez_compressed(path, n_qubits=np.int64(ham.n_qubits), n_alpha=np.int64(ham.n_alpha),
                        n_beta=np.int64(ham.n_beta), xy=ham.xy, yz=ham.yz, coeff=ham.coeff)


def data_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
