"""Host logic: Pauli-term pre-processing and the .pkl input format (no GPU)."""
import os

import numpy as np
import pytest

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


REF_MOLECULES = "/root/reference/molecules"


@pytest.mark.skipif(not os.path.isdir(REF_MOLECULES), reason="build container only: needs the reference's molecule folders")
def test_packing_matches_the_reference_for_every_molecule_folder():
    """Every molecule folder the reference ships with a qubit-Hamiltonian pickle (32: 4 to 30 qubits, up to 28 393 Pauli
    strings, closed and open shell) through this repository's own readers (stub unpickler, built-in HDF5 reader) and
    packing rule, against SHA-256 digests of the reference's own ``__calc_coupling_info`` output
    (``hamiltonian.py:373-430``; ``tests/golden/make_golden.py packing``): xy | yz | coeff bit for bit, term order included."""
    import hashlib
    import json
    from naqs_amd import system
    with open(os.path.join(GOLDEN, "packing_sha256.json")) as f:
        want = json.load(f)
    assert len(want) >= 32
    for mol, rec in sorted(want.items()):
        molecule, qh = system.load_molecule(os.path.join(REF_MOLECULES, mol), verbose=False)
        n_qubits = packing.n_qubits_of_terms(qh.terms)
        ham = packing.pack_qubit_hamiltonian(qh.terms, n_qubits, molecule.get_n_alpha_electrons(), molecule.get_n_beta_electrons())
        got = hashlib.sha256(ham.xy.astype(np.uint64).tobytes() + ham.yz.astype(np.uint64).tobytes()
                             + ham.coeff.astype(np.float64).tobytes()).hexdigest()
        assert (n_qubits, ham.K) == (rec["n_qubits"], rec["K"]), mol
        assert got == rec["sha256"], mol
