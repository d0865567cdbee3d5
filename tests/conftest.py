import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_unavailable():
    try:
        import torch
        if not torch.cuda.is_available():
            return "no HIP device"
    except Exception as e:                                   # pragma: no cover
        return f"torch unavailable: {e}"
    # (a GPU box with a missing libnaqs_hip.so is NOT a reason to skip: there the tests must fail loudly)
    return None


def pytest_collection_modifyitems(config, items):
    """``-m gpu`` tests need a real MI355X: on a host without one a plain
    ``pytest tests`` skips them (with the reason) instead of failing one by one."""
    # no test may hang a run (the GPU box is charged by the minute and a wedged suite is killed without a report): a
    # per-test limit through pytest-timeout where it is installed (it is, here and on the GPU box)
    if config.pluginmanager.hasplugin("timeout"):
        for it in items:
            if not it.get_closest_marker("timeout"):
                it.add_marker(pytest.mark.timeout(300))
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    why = _gpu_unavailable()
    if why:
        skip = pytest.mark.skip(reason=f"gpu test: {why}")
        for it in gpu_items:
            it.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def dense_pauli_case(N=6, n_terms=60, seed=5):
    """A random real-symmetric qubit Hamiltonian as (terms dict, dense 2^N x 2^N matrix): Pauli strings with an even number
    of Y, qubit q <-> bit q of the basis index, |1> = occupied; the dense matrix is the sum of Kronecker products."""
    import numpy as np
    rs = np.random.RandomState(seed)
    sig = {"I": np.eye(2, dtype=complex), "X": np.array([[0, 1], [1, 0]], complex),
           "Y": np.array([[0, -1j], [1j, 0]], complex), "Z": np.array([[1, 0], [0, -1]], complex)}
    terms, dense = {}, np.zeros((1 << N, 1 << N), complex)
    while len(terms) < n_terms:
        ops = rs.choice(list("IXYZ"), size=N, p=[0.4, 0.2, 0.2, 0.2])
        if (ops == "Y").sum() % 2:
            continue
        key = tuple((q, str(o)) for q, o in enumerate(ops) if o != "I")
        if key in terms:
            continue
        c = float(rs.normal())
        terms[key] = c + 0j
        m = np.eye(1, dtype=complex)
        for q in range(N):                                  # bit q is the q-th least significant: kron from the top down
            m = np.kron(sig[str(ops[q])], m)
        dense += c * m
    assert np.max(np.abs(dense.imag)) == 0.0 and np.max(np.abs(dense - dense.T)) < 1e-12
    return terms, dense.real, rs
