"""Import shim for the *reference* implementation (this container only).

TEST INFRASTRUCTURE.  Used by ``make_golden.py`` to import
tomdbar/naqs-for-quantum-chemistry from a scratch copy of ``/root/reference``
and dump golden input/output vectors into ``tests/golden/*.npz``.  Nothing in
the product, ``-m gpu`` tests, ``smoke()`` or ``bench.py`` imports this file;
the reference itself never travels to the GPU box.

Recipe:
  1. copy ``src/ src_cpp/ experiments/`` to a scratch dir (the mount is
     read-only and the Cython build writes ``.so`` files into ``src/utils``);
  2. patch ``prange(2**N`` -> ``prange(1<<N`` in ``hilbert_math.pyx`` (Cython 3
     rejects a C-double loop bound);
  3. ``python3 src_cpp/setup.py build_ext --inplace --force``;
  4. install the import shims below, then ``import src.*``.
"""
import os
import pickle
import shutil
import subprocess
import sys
import types

import numpy as np

REFERENCE = "/root/reference"
SCRATCH = os.environ.get("NAQS_REF_SCRATCH", "/tmp/ref")


def build_scratch():
    if not os.path.isdir(REFERENCE):
        raise RuntimeError("reference tree not mounted; golden vectors can only be regenerated "
                           "in the build container")
    so = os.path.join(SCRATCH, "src", "utils")
    if os.path.isdir(so) and any(f.startswith("hamiltonian_math") and f.endswith(".so")
                                 for f in os.listdir(so)):
        return
    os.makedirs(SCRATCH, exist_ok=True)
    for d in ("src", "src_cpp", "experiments"):
        dst = os.path.join(SCRATCH, d)
        if os.path.exists(dst):
            shutil.rmtree(dst)
        shutil.copytree(os.path.join(REFERENCE, d), dst)
    subprocess.check_call(["chmod", "-R", "u+w", SCRATCH])
    pyx = os.path.join(SCRATCH, "src_cpp", "hilbert_math.pyx")
    txt = open(pyx).read().replace("prange(2**N", "prange(1<<N")
    open(pyx, "w").write(txt)
    subprocess.check_call([sys.executable, "src_cpp/setup.py", "build_ext", "--inplace", "--force"],
                          cwd=SCRATCH, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def install_shims():
    import scipy
    import torch  # noqa: F401

    np.long = np.int64                                     # removed in NumPy >= 1.24
    six = types.ModuleType("torch._six")
    six.inf = float("inf")
    sys.modules["torch._six"] = six                        # removed from torch
    scipy.random = np.random                               # old SciPy alias (double seeding of numpy)

    of = types.ModuleType("openfermion")
    ofh = types.ModuleType("openfermion.hamiltonians")
    oft = types.ModuleType("openfermion.transforms")

    class MolecularData:                                   # h5py/openfermion are absent here
        def __init__(self, *a, **k):
            raise RuntimeError("openfermion is not installed")

    ofh.MolecularData = MolecularData
    oft.get_fermion_operator = oft.jordan_wigner = lambda *a, **k: None
    of.hamiltonians, of.transforms = ofh, oft
    sys.modules.update({"openfermion": of, "openfermion.hamiltonians": ofh,
                        "openfermion.transforms": oft})
    if SCRATCH not in sys.path:
        sys.path.insert(0, SCRATCH)


def patch_reference():
    """SciPy >= 1.15 rejects torch tensors as fancy indices (hamiltonian.py:93-94)."""
    import torch
    import src.optimizer.hamiltonian as H

    base = H._PauliHamiltonianDynamic.__mro__[1]

    def _sub(self, idxs):
        if torch.is_tensor(idxs):
            idxs = idxs.numpy()
        idxs = np.asarray(idxs).astype(np.int64)
        return self.H[idxs[:, None], idxs]

    setattr(base, "_" + base.__name__.lstrip("_") + "__get_new_H_subspace", _sub)
    # name-mangled private: class is "__PauliHamiltonianBase" -> "_PauliHamiltonianBase__get..."
    setattr(base, "_PauliHamiltonianBase__get_new_H_subspace", _sub)


class _StubUnpickler(pickle.Unpickler):
    """Loads an openfermion QubitOperator pickle without openfermion."""

    def find_class(self, module, name):
        if module.startswith("openfermion"):
            return type(name, (), {})
        return super().find_class(module, name)


def load_qubit_hamiltonian(molecule):
    path = os.path.join(REFERENCE, "molecules", molecule, f"{molecule}_qubit_hamiltonian.pkl")
    with open(path, "rb") as f:
        return _StubUnpickler(f).load()


# closed-shell electron counts (the HDF5 metadata cannot be read without h5py)
ELECTRONS = {"LiH": (2, 2), "H2O": (5, 5), "N2": (7, 7), "Li2O": (7, 7), "H2": (1, 1)}


def n_qubits_of(qh):
    return 1 + max(q for term in qh.terms for q, _ in term)


def setup():
    build_scratch()
    install_shims()
    patch_reference()
