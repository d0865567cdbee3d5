"""naqs_amd — MI355X-native local-energy / log-psi hot path of NAQS variational Monte Carlo.

Host-side Python mirror of the reference's interface for that path (the reference is
Python, so the host side above the C ABI is Python too).  The compute is in
``libnaqs_hip.so`` (hand-written HIP for gfx950, ``csrc/``); PyTorch is used for device
memory, streams and ``torch.distributed`` only.
"""
from ._lib import NaqsError, lib_path, load_library  # noqa: F401
from .packing import (PackedHamiltonian, load_packed, load_qubit_hamiltonian_pkl,  # noqa: F401
                      pack_qubit_hamiltonian)

__version__ = "0.1.0"
