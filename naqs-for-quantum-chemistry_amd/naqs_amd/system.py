"""Molecule loading and seeding: counterpart of the reference's src/utils/system.py:14-79.

``load_molecule`` reads ``<dir>/<name>.hdf5`` (OpenFermion MolecularData) through the built-in
minimal HDF5 reader and ``<dir>/<name>_qubit_hamiltonian.pkl`` through the stub unpickler, so the
reference's ``molecules/*`` directories work unchanged without openfermion or h5py installed.
The Jordan-Wigner fallback of the reference (system.py:35-46, when the pickle is missing) needs
openfermion and is not reproduced.
"""
import os
import random

import numpy as np
import torch

from .hdf5_lite import read_hdf5
from .packing import load_qubit_hamiltonian_pkl, n_qubits_of_terms

_SCALARS = ("n_electrons", "multiplicity", "n_orbitals", "n_qubits", "n_atoms", "hf_energy", "mp2_energy",
            "cisd_energy", "ccsd_energy", "fci_energy", "nuclear_repulsion", "name", "basis", "description")


class Molecule:
    """The scalar part of OpenFermion's MolecularData that the run path touches."""

    def __init__(self, filename):
        self.filename = filename
        vals = read_hdf5(filename + ".hdf5" if not filename.endswith(".hdf5") else filename, keys=None)
        for k in _SCALARS:
            v = vals.get(k)
            if isinstance(v, (bool, np.bool_)) and not v:       # OpenFermion stores None as False
                v = None
            setattr(self, k, v)

    def get_n_alpha_electrons(self):
        return int((self.n_electrons + (self.multiplicity - 1)) // 2)

    def get_n_beta_electrons(self):
        return int((self.n_electrons - (self.multiplicity - 1)) // 2)


class PackedMolecule:
    """Molecule metadata carried by a packed-term ``.npz`` (tests/golden/ham_<mol>.npz)."""

    def __init__(self, path):
        with np.load(path) as z:
            self.name = os.path.splitext(os.path.basename(path))[0]
            self.n_qubits = int(z["n_qubits"])
            self.n_electrons = int(z["n_electrons"]) if "n_electrons" in z else int(z["n_alpha"]) + int(z["n_beta"])
            self.multiplicity = int(z["multiplicity"]) if "multiplicity" in z else 1
            self.n_orbitals = self.n_qubits // 2
            for k in ("hf_energy", "ccsd_energy", "fci_energy", "mp2_energy"):
                setattr(self, k, float(z[k]) if k in z else None)
        self.basis = self.description = None

    get_n_alpha_electrons = Molecule.get_n_alpha_electrons
    get_n_beta_electrons = Molecule.get_n_beta_electrons


def load_molecule(fname, hamiltonian_fname=None, verbose=True):
    if fname.endswith(".npz"):
        # packed fixture instead of <dir>/<name>.hdf5 + <name>_qubit_hamiltonian.pkl
        from .packing import load_packed
        return PackedMolecule(fname), load_packed(fname)
    if os.path.isdir(fname):
        fname = os.path.join(fname, os.path.split(os.path.normpath(fname))[-1])
    print(f"Loading molecule from {fname}.hdf5", end="...")
    molecule = Molecule(fname)
    print("done.")
    if hamiltonian_fname is None:
        hamiltonian_fname = fname + "_qubit_hamiltonian.pkl"
    print(f"Loading molecule from {hamiltonian_fname}", end="...")
    if not os.path.exists(hamiltonian_fname):
        raise FileNotFoundError(f"{hamiltonian_fname}: the Jordan-Wigner fallback needs openfermion and is out of scope")
    qubit_hamiltonian = load_qubit_hamiltonian_pkl(hamiltonian_fname)
    print("done.")
    if molecule.n_qubits is None:
        molecule.n_qubits = n_qubits_of_terms(qubit_hamiltonian.terms)
    if verbose:
        print(f"{fname}.hdf5 has:")
        print(f"\tHartree-Fock energy of {molecule.hf_energy} Hartree.")
        print(f"\tMP2 energy of {molecule.mp2_energy} Hartree.")
        print(f"\tCCSD energy of {molecule.ccsd_energy} Hartree.")
        print(f"\tFCI energy of {molecule.fci_energy} Hartree.")
        print(f"\nHamiltonian for {fname}.hdf5 has:")
        print(f"\t{n_qubits_of_terms(qubit_hamiltonian.terms)} qubits (orbitals), with {molecule.n_electrons} electrons "
              f"({molecule.get_n_alpha_electrons()}/{molecule.get_n_beta_electrons()} alpha/beta).")
    return molecule, qubit_hamiltonian


def set_global_seed(seed=-1):
    """Same draw order as the reference (system.py:64-79): python ``random`` is seeded with
    ``seed`` and hands out the sub-seeds of numpy (twice: the old ``scipy.random`` alias), torch and the
    device generators, in that order.  Returns the seed used."""
    if seed < 0:
        seed = random.randint(0, 2 ** 32)
    print("\n------------------------------------------")
    print(f"\tSetting global seed using {seed}.")
    print("------------------------------------------\n")
    random.seed(seed)
    np.random.seed(random.randint(0, 2 ** 32) % (2 ** 32))
    np.random.seed(random.randint(0, 2 ** 32) % (2 ** 32))
    torch.manual_seed(random.randint(0, 2 ** 32))
    a, b = random.randint(0, 2 ** 32), random.randint(0, 2 ** 32)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(a)
        torch.cuda.manual_seed_all(b)
    return seed


def mk_dir(d, quiet=False):
    if not os.path.exists(d):
        os.makedirs(d, exist_ok=True)
        if not quiet:
            print("created directory: ", d)
