"""Host logic: Pauli-term pre-processing and the .pkl input format (no GPU)."""
import os

import numpy as np

from conftest import GOLDEN, golden
from naqs_amd import packing


def test_pkl_loads_without_openfermion_and_packs_like_the_reference():
    qh = packing.load_qubit_hamiltonian_pkl(os.path.join(GOLDEN, "molecules", "LiH", "LiH_qubit_hamiltonian.pkl"))
    assert packing.n_qubits_of_terms(qh.terms) == 12
    ham = packing.pack_qubit_hamiltonian(qh.terms, 12, 2, 2)
    ref = golden("ham_LiH.npz")
    assert ham.K == 631
    assert np.array_equal(ham.xy, ref["xy"]) and np.array_equal(ham.yz, ref["yz"])
    assert np.array_equal(ham.coeff, ref["coeff"])         # bit-identical couplings, same term order


def test_pack_rules():
    terms = {(): 1.5 + 0j, ((0, "X"), (1, "Y")): 2.0 + 0j, ((0, "Y"), (1, "Y")): 1.0 + 0j,
             ((2, "Z"),): -0.5 + 0.25j}
    ham = packing.pack_qubit_hamiltonian(terms, 4, 1, 1)
    assert ham.xy.tolist() == [0, 3, 3, 0] and ham.yz.tolist() == [0, 2, 3, 4]
    # Re(i^nY)*c: one Y -> 0, two Y -> -1; imaginary parts are dropped (hamiltonian.py:416,424)
    assert ham.coeff.tolist() == [1.5, 0.0, -1.0, -0.5]
    # frozen-qubit and excitation limits (hamiltonian.py:394-401)
    assert packing.pack_qubit_hamiltonian(terms, 4, 1, 1, n_occ=1).K == 2
    assert packing.pack_qubit_hamiltonian(terms, 4, 1, 1, n_excitations_max=1).K == 2


def test_save_load_roundtrip(tmp_path):
    ham = packing.load_packed(os.path.join(GOLDEN, "ham_H2O.npz"))
    p = str(tmp_path / "h.npz")
    packing.save_packed(p, ham)
    again = packing.load_packed(p)
    assert again.n_qubits == 14 and (again.n_alpha, again.n_beta) == (5, 5)
    assert np.array_equal(again.xy, ham.xy) and np.array_equal(again.coeff, ham.coeff)
