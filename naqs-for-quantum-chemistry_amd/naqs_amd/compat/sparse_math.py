"""``src.utils.sparse_math.sparse_dense_mv`` of the reference (src_cpp/sparse_math.pyx:13-83), on the MI355X."""
import numpy as np
import torch

from .. import _lib
from ..hamiltonian import _stream_ptr
from .hamiltonian_math import _device


def sparse_dense_mv(m, v, par=None):
    """CSR matrix (scipy, float32 / float64) times dense vector (real or complex) -> complex ndarray ``[m.shape[1]]``.

    Type rules of the reference's ``__type_mv`` (sparse_math.pyx:13-41): float64 matrices give complex128; float32
    matrices give complex64 unless ``v`` is complex128; any other matrix dtype raises ``Exception``.  ``par`` (the
    reference's OpenMP switch) is accepted and ignored — one wavefront per row either way.  The product is
    accumulated in float64 on the device (the 32-bit combination is rounded to complex64 at the end, the reference
    accumulates it in complex64)."""
    if m.dtype == np.dtype(np.float64):
        n_bit = 64
    elif m.dtype == np.dtype(np.float32):
        n_bit = 32
    else:
        raise Exception("m must have dtype of np.float32 or np.float64.")
    v = np.asarray(v)
    if np.iscomplexobj(v) and v.dtype == np.dtype(np.complex128):
        n_bit = 64
    out_dtype = np.complex128 if n_bit == 64 else np.complex64
    m = m.tocsr() if getattr(m, "format", "csr") != "csr" else m
    rows = m.shape[0]
    # the reference writes out[j] for j < m.shape[1] reading indptr[j], i.e. it assumes a square matrix
    n_out = m.shape[1]
    if rows != n_out:
        raise ValueError("sparse_dense_mv expects a square matrix (the reference indexes rows by m.shape[1])")
    if v.shape[0] != m.shape[1]:
        raise ValueError(f"dimension mismatch: matrix {m.shape}, vector {v.shape}")
    if rows == 0:
        return np.zeros(0, dtype=out_dtype)
    lib, dev = _lib.load_library(), _device()
    vc = np.ascontiguousarray(v.astype(np.complex128))
    d_v = torch.from_numpy(vc.view(np.float64).reshape(-1, 2)).to(dev)
    d_data = torch.from_numpy(np.ascontiguousarray(m.data.astype(np.float64))).to(dev)
    d_idx = torch.from_numpy(np.ascontiguousarray(m.indices.astype(np.int32))).to(dev)
    d_ptr = torch.from_numpy(np.ascontiguousarray(m.indptr.astype(np.int32))).to(dev)
    out = torch.empty((rows, 2), dtype=torch.float64, device=dev)
    st = lib.naqs_csr_mv(rows, d_data.data_ptr(), d_idx.data_ptr(), d_ptr.data_ptr(), d_v.data_ptr(), out.data_ptr(),
                         _stream_ptr(dev))
    _lib.check(st, "naqs_csr_mv")
    res = out.cpu().numpy()
    return (res[:, 0] + 1j * res[:, 1]).astype(out_dtype)
