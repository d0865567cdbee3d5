"""Inner-ring shims: importable modules with the names and signatures of the reference's three Cython
extensions — numpy in, numpy out — backed by ``libnaqs_hip.so``.

A reference checkout switches its hot kernels to the MI355X by importing these instead of its own
``src/utils/*.so`` (``INTEGRATION.md`` section 1)::

    from naqs_amd.compat.hamiltonian_math import get_Hij_cy, popcount_parity     # src/optimizer/hamiltonian.py:13
    from naqs_amd.compat.sparse_math import sparse_dense_mv                      # src/optimizer/energy.py:27

Arrays are copied to the device, the kernel runs on the current HIP stream, the result is copied back: the
per-call PCIe traffic is the price of keeping the reference's host-array interface — the outer ring
(``naqs_amd.optimizer``) keeps everything on the device instead.  There is no CPU fallback: without a HIP
device or the library these raise ``NaqsError``.
"""
from . import hamiltonian_math, sparse_math  # noqa: F401
