"""``src.utils.hamiltonian_math`` of the reference (src_cpp/hamiltonian_math.pyx), on the MI355X.

``get_Hij_cy`` (:198-288, reachable bodies ``__inner_*_float/double`` :85-100) and ``popcount_parity``
(:455-484 + the typed bodies :295-453) with the reference's argument lists, dtypes and error behaviour."""
import numpy as np
import torch

from .. import _lib
from ..hamiltonian import _stream_ptr

_PARITY_DTYPES = (np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64)


def _device():
    if not torch.cuda.is_available():
        raise _lib.NaqsError("naqs_amd.compat needs a HIP device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def popcount_parity(arr):
    """``1 - 2 * (popcount(arr) % 2)`` element-wise -> int8 array of the same shape; 1-D input becomes ``[n, 1]``
    and unsupported dtypes raise ``TypeError`` like the reference (hamiltonian_math.pyx:455-484)."""
    arr = np.asarray(arr)
    if len(arr.shape) == 1:
        arr = arr.reshape(-1, 1)
    if arr.dtype.type not in _PARITY_DTYPES:
        raise TypeError(f"Unsupported array dtype for popcount_parity(...): {arr.dtype}.")
    lib, dev = _lib.load_library(), _device()
    # unsigned arrays travel as their signed bit patterns: sign extension adds an even number of set bits
    signed = np.ascontiguousarray(arr).view(np.dtype(f"int{8 * arr.dtype.itemsize}"))
    a = torch.from_numpy(signed).to(dev)
    out = torch.empty(a.shape, dtype=torch.int8, device=dev)
    st = lib.naqs_popcount_parity(a.data_ptr(), a.element_size(), a.numel(), out.data_ptr(), _stream_ptr(dev))
    _lib.check(st, "naqs_popcount_parity")
    return out.cpu().numpy()


def get_Hij_cy(state_i_idx, _unique_XY_sites_idx, _unique2all_XY_sites_idx, P_k_by_unique_YZ_sites,
               _unique2all_YZ_sites_idx, couplings):
    """``H_ij[i * Kxy + unique2all_XY[k]] += P[i, unique2all_YZ[k]] * couplings[k]`` for every sample i and term k
    (k ascending) -> flat array ``[M * Kxy]`` of the couplings' dtype (float32 / float64), bit-identical to the
    reference's loop (same addends, same order, same arithmetic type).  ``P`` is the parity table
    ``popcount_parity(state_i_idx[:, None] & unique_YZ[None, :])`` the caller already has (hamiltonian.py:301-305)."""
    M = len(state_i_idx)
    Kxy = len(_unique_XY_sites_idx)
    u2a_xy = np.asarray(_unique2all_XY_sites_idx).astype(np.int64).reshape(-1)
    u2a_yz = np.asarray(_unique2all_YZ_sites_idx).astype(np.int64).reshape(-1)
    K = len(u2a_xy)
    couplings = np.asarray(couplings).squeeze()
    if couplings.dtype not in (np.float32, np.float64):
        raise TypeError(f"get_Hij_cy on the MI355X: couplings must be float32 or float64, got {couplings.dtype} "
                        "(the reference's long-double branch has no device counterpart)")
    couplings = couplings.reshape(-1)
    P = np.asarray(P_k_by_unique_YZ_sites)
    if P.ndim != 2 or P.shape[0] != M:
        raise ValueError(f"P_k_by_unique_YZ_sites must be [M, Kyz] with M = {M}, got {P.shape}")
    if P.dtype != np.int8:
        if P.size and (P.min() < -128 or P.max() > 127):
            raise ValueError("parity table entries must fit int8 (they are +-1 in the reference's call)")
        P = P.astype(np.int8)
    Kyz = P.shape[1]
    if K and (u2a_xy.min() < 0 or u2a_xy.max() >= Kxy or u2a_yz.min() < 0 or u2a_yz.max() >= Kyz):
        raise IndexError("unique2all index outside its table")
    out_dtype = couplings.dtype
    if M == 0 or Kxy == 0:
        return np.zeros(M * Kxy, dtype=out_dtype)
    # terms grouped by output column, ascending term index inside a column (stable): the reference's summation order
    order = np.argsort(u2a_xy, kind="stable")
    group_ptr = np.zeros(Kxy + 1, np.int32)
    np.cumsum(np.bincount(u2a_xy, minlength=Kxy), out=group_ptr[1:])
    lib, dev = _lib.load_library(), _device()
    d_P = torch.from_numpy(np.ascontiguousarray(P)).to(dev)
    d_gp = torch.from_numpy(group_ptr).to(dev)
    d_yt = torch.from_numpy(u2a_yz[order].astype(np.int32)).to(dev)
    d_ct = torch.from_numpy(np.ascontiguousarray(couplings[order])).to(dev)
    out = torch.empty(M * Kxy, dtype=torch.float64 if out_dtype == np.float64 else torch.float32, device=dev)
    st = lib.naqs_hij_from_parity(M, Kxy, Kyz, K, d_P.data_ptr(), d_gp.data_ptr(), d_yt.data_ptr(), d_ct.data_ptr(),
                                  d_ct.element_size(), out.data_ptr(), _stream_ptr(dev))
    _lib.check(st, "naqs_hij_from_parity")
    return out.cpu().numpy()
