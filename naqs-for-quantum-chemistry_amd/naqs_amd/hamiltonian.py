"""Device-resident Pauli Hamiltonian: the drop-in counterpart of the reference's
``PauliHamiltonian`` (src/optimizer/hamiltonian.py:32-216, :218-370) for the local-energy path.

The reference caches a scipy CSR matrix over the restricted Hilbert space and grows it lazily
(``update_H``); here nothing is cached — ``libnaqs_hip.so`` regenerates matrix elements on the
fly from the packed terms (matrix-free), so ``update_H`` / ``freeze_H`` keep their names and
argument meaning but only validate / record state.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .packing import PackedHamiltonian, pack_qubit_hamiltonian

_KIND = {("psi", torch.float32): _lib.PSI_F32, ("psi", torch.float64): _lib.PSI_F64,
         ("log_psi", torch.float32): _lib.LOGPSI_F32, ("log_psi", torch.float64): _lib.LOGPSI_F64}


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def keys_to_device(keys, device):
    """Any integer array/tensor of bit-string keys -> contiguous device int64 tensor holding the
    uint64 bit pattern (torch has no uint64 arithmetic; only the bits matter)."""
    if isinstance(keys, np.ndarray):
        keys = torch.from_numpy(np.ascontiguousarray(keys).astype(np.uint64).view(np.int64))
    keys = keys.reshape(-1)
    if keys.dtype != torch.int64:
        keys = keys.to(torch.int64)
    return keys.to(device).contiguous()


class DevicePauliHamiltonian:
    """Packed terms on one GPU + the kernels that consume them (handle of ``naqs_ham_create``)."""

    def __init__(self, packed: PackedHamiltonian, device=None):
        self._h = ctypes.c_void_p(None)
        self._lib = _lib.load_library()           # raises NaqsError when the HIP library is missing
        if not torch.cuda.is_available():
            raise _lib.NaqsError("DevicePauliHamiltonian needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.packed = packed
        xy = np.ascontiguousarray(packed.xy, np.uint64)
        yz = np.ascontiguousarray(packed.yz, np.uint64)
        cf = np.ascontiguousarray(packed.coeff, np.float64)
        st = self._lib.naqs_ham_create(packed.n_qubits, packed.n_alpha, packed.n_beta, packed.K,
                                       xy.ctypes.data, yz.ctypes.data, cf.ctypes.data,
                                       self.device.index, ctypes.byref(self._h))
        _lib.check(st, "naqs_ham_create")
        info = (ctypes.c_int64 * 8)()
        _lib.check(self._lib.naqs_ham_info(self._h, ctypes.byref(info)), "naqs_ham_info")
        self.K, self.Kxy, self.n_qubits = int(info[0]), int(info[1]), int(info[2])
        self.key_bits, self.diag_terms = int(info[5]), int(info[6])

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.naqs_ham_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the hot path -------------------------------------------------------------------------
    def reserve(self, M):
        _lib.check(self._lib.naqs_ham_reserve(self._h, int(M)), "naqs_ham_reserve")

    def local_energy(self, keys, wf, kind="psi", row_begin=0, n_rows=None, out=None, weights=None, sums_out=None):
        """E_loc for table rows [row_begin, row_begin+n_rows) -> float64 tensor [n_rows, 2] (Re, Im).

        keys : int64 device tensor [M] (bit pattern of the uint64 keys), unique, physical
        wf   : device tensor [M, 2], float32/float64; (Re psi, Im psi) for kind="psi",
               (log|psi|, phase) for kind="log_psi"
        """
        M = keys.shape[0]
        if n_rows is None:
            n_rows = M - row_begin
        if wf.shape != (M, 2):
            raise ValueError(f"wave function must have shape [{M}, 2], got {tuple(wf.shape)}")
        if not (keys.is_cuda and wf.is_cuda and keys.dtype == torch.int64):
            raise ValueError("keys (int64) and wf must be device tensors")
        wf = wf.contiguous()
        keys = keys.contiguous()
        code = _KIND.get((kind, wf.dtype))
        if code is None:
            raise TypeError(f"unsupported wave-function kind/dtype: {kind}/{wf.dtype}")
        if out is None:
            out = torch.empty((n_rows, 2), dtype=torch.float64, device=self.device)
        if weights is not None:
            # fused: E_loc and (sum w Re E, sum w Im E, sum w Re(E)^2, sum w) of the produced rows in one launch
            w = weights.to(device=self.device, dtype=torch.float64).contiguous()
            if w.shape[0] != n_rows:
                raise ValueError("weights must cover exactly the produced rows")
            if sums_out is None:
                sums_out = torch.empty(4, dtype=torch.float64, device=self.device)
            st = self._lib.naqs_eloc_reduced(self._h, M, keys.data_ptr(), wf.data_ptr(), code, int(row_begin),
                                             int(n_rows), w.data_ptr(), out.data_ptr(), sums_out.data_ptr(),
                                             _stream_ptr(self.device))
            _lib.check(st, "naqs_eloc_reduced")
            return out, sums_out
        st = self._lib.naqs_eloc(self._h, M, keys.data_ptr(), wf.data_ptr(), code, int(row_begin), int(n_rows),
                                 out.data_ptr(), _stream_ptr(self.device))
        _lib.check(st, "naqs_eloc")
        return out

    def reduce(self, weights, eloc, out=None):
        """-> float64 device tensor [4] = (sum w Re E, sum w Im E, sum w Re(E)^2, sum w)."""
        w = weights.to(device=self.device, dtype=torch.float64).contiguous()
        if out is None:
            out = torch.empty(4, dtype=torch.float64, device=self.device)
        st = self._lib.naqs_eloc_reduce(self._h, eloc.shape[0], w.data_ptr(), eloc.contiguous().data_ptr(),
                                        out.data_ptr(), _stream_ptr(self.device))
        _lib.check(st, "naqs_eloc_reduce")
        return out

    # ---- H restricted to the sampled states, matrix-free ----------------------------------------
    def matvec(self, keys, v, out=None):
        """out_i = sum_j H_ij v_j over the sampled keys (``naqs_hmatvec``).  v: real [M] or complex-as-pairs [M, 2]
        float64 device tensor; returns the same shape."""
        M = keys.shape[0]
        real = v.dim() == 1
        vv = torch.stack([v, torch.zeros_like(v)], -1) if real else v
        vv = vv.to(device=self.device, dtype=torch.float64).contiguous()
        res = torch.empty((M, 2), dtype=torch.float64, device=self.device) if out is None or real else out
        st = self._lib.naqs_hmatvec(self._h, M, keys.contiguous().data_ptr(), vv.data_ptr(), 0, M, res.data_ptr(),
                                    _stream_ptr(self.device))
        _lib.check(st, "naqs_hmatvec")
        return res[:, 0].contiguous() if real else res

    @torch.no_grad()
    def lowest_eigenpair(self, keys, max_iter=400, tol=1e-10, seed=0):
        """Lowest eigenpair of H restricted to the sampled keys by Lanczos with full re-orthogonalisation, every
        product matrix-free on the device (no M x M or M x Kxy matrix is formed): the scalable form of the
        sampled-subspace diagonalisation of solve_H (energy.py:762-786).  -> (eigenvalue, eigenvector float64 [M])."""
        M = keys.shape[0]
        if M == 0:
            raise ValueError("empty sample set")
        gen = torch.Generator(device=self.device).manual_seed(int(seed))
        v = torch.randn(M, dtype=torch.float64, device=self.device, generator=gen)
        v /= v.norm()
        m_max = int(min(max_iter, M))
        V = torch.empty((m_max, M), dtype=torch.float64, device=self.device)
        from scipy.linalg import eigh_tridiagonal
        alpha, beta = [], []
        theta, s_vec, k = None, None, 0
        for k in range(m_max):
            V[k] = v
            w = self.matvec(keys, v)
            a = torch.dot(v, w)
            w = w - a * v - (beta[-1] * V[k - 1] if k > 0 else 0.0)
            for _ in range(2):                                   # full re-orthogonalisation, twice is enough
                w = w - V[:k + 1].t() @ (V[:k + 1] @ w)
            b = w.norm()
            a_h, b_h = torch.stack([a, b]).tolist()              # the iteration's one host synchronisation
            alpha.append(a_h)
            # lowest Ritz pair of the tridiagonal matrix only (O(k) bisection + inverse iteration, not a full eigh),
            # and not on every iteration once the basis is large: the test costs more than the product it saves
            last = k == m_max - 1 or b_h < 1e-14
            if k < 32 or k % 8 == 0 or last:
                if k == 0:
                    theta, s_vec = alpha[0], np.ones(1)
                else:
                    ev, evec = eigh_tridiagonal(np.array(alpha), np.array(beta), select="i", select_range=(0, 0))
                    theta, s_vec = float(ev[0]), evec[:, 0]
                resid = abs(b_h * s_vec[-1])
                if resid < tol * max(1.0, abs(theta)) or last:
                    break
            beta.append(b_h)
            v = w / b
        vec = torch.as_tensor(s_vec, dtype=torch.float64, device=self.device) @ V[:k + 1]
        vec = vec / vec.norm()
        return float(theta), vec

    # ---- inner ring ----------------------------------------------------------------------------
    def dense_hij(self, keys):
        """Device counterpart of get_Hij_cy: float64 [M, Kxy], H[i, key_i ^ xy_g]."""
        M = keys.shape[0]
        out = torch.empty((M, self.Kxy), dtype=torch.float64, device=self.device)
        st = self._lib.naqs_get_hij(self._h, M, keys.contiguous().data_ptr(), out.data_ptr(),
                                    _stream_ptr(self.device))
        _lib.check(st, "naqs_get_hij")
        return out

    # ---- measurement ---------------------------------------------------------------------------
    def prof_enable(self, max_records, stride=1):
        _lib.check(self._lib.naqs_prof_enable(self._h, int(max_records)), "naqs_prof_enable")
        _lib.check(self._lib.naqs_prof_stride(self._h, int(stride)), "naqs_prof_stride")

    def prof_read(self):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        _lib.check(self._lib.naqs_prof_read(self._h, ctypes.byref(ms), ctypes.byref(n)), "naqs_prof_read")
        return ms.value, n.value

    def last_kernel(self):
        """Name (with template arguments) of the kernel the most recent call launched (measurement aid)."""
        buf = ctypes.create_string_buffer(128)
        _lib.check(self._lib.naqs_ham_last_kernel(self._h, buf, 128), "naqs_ham_last_kernel")
        return buf.value.decode()


def popcount_parity_device(arr):
    """Device counterpart of src.utils.hamiltonian_math.popcount_parity: int8 tensor, same shape.
    Raises TypeError on unsupported dtypes like the reference (hamiltonian_math.pyx:484)."""
    if arr.dtype not in (torch.int16, torch.int32, torch.int64):
        raise TypeError(f"Unsupported array dtype for popcount_parity(...): {arr.dtype}.")
    lib = _lib.load_library()
    a = arr.contiguous()
    out = torch.empty(a.shape if a.dim() > 1 else (a.numel(), 1), dtype=torch.int8, device=a.device)
    st = lib.naqs_popcount_parity(a.data_ptr(), a.element_size(), a.numel(), out.data_ptr(), _stream_ptr(a.device))
    _lib.check(st, "naqs_popcount_parity")
    return out


def csr_mv_device(data, indices, indptr, v):
    """Device counterpart of src.utils.sparse_math.sparse_dense_mv (f64 CSR x complex128 [n,2])."""
    lib = _lib.load_library()
    rows = indptr.numel() - 1
    out = torch.empty((rows, 2), dtype=torch.float64, device=v.device)
    st = lib.naqs_csr_mv(rows, data.contiguous().data_ptr(), indices.contiguous().data_ptr(),
                         indptr.contiguous().data_ptr(), v.contiguous().data_ptr(), out.data_ptr(),
                         _stream_ptr(v.device))
    _lib.check(st, "naqs_csr_mv")
    return out


class PauliHamiltonian:
    """Factory with the reference's signature (hamiltonian.py:47-61)."""

    @staticmethod
    def get(hilbert, qubit_hamiltonian, hamiltonian_fname=None, restricted_idxs=None,
            n_excitations_max=None, verbose=False, dtype=np.float64, device=None):
        if isinstance(qubit_hamiltonian, PackedHamiltonian):
            packed = qubit_hamiltonian
        else:
            packed = pack_qubit_hamiltonian(qubit_hamiltonian.terms, hilbert.N, hilbert.N_alpha, hilbert.N_beta,
                                            n_excitations_max=n_excitations_max, n_occ=getattr(hilbert, "N_occ", 0))
        return MatrixFreePauliHamiltonian(hilbert, packed, verbose=verbose, device=device)


class MatrixFreePauliHamiltonian(DevicePauliHamiltonian):
    """``_PauliHamiltonianDynamic`` counterpart: same method names, nothing cached."""

    def __init__(self, hilbert, packed, verbose=False, device=None):
        super().__init__(packed, device=device)
        self.hilbert = hilbert
        self.verbose = verbose
        self._frozen_H = False

    def update_H(self, state_idx=None, check_unseen=True, assume_unique=False):
        return None          # matrix elements are regenerated per call; there is no cache to update

    def freeze_H(self):
        self._frozen_H = True

    def unfreeze_H(self):
        self._frozen_H = False

    def is_frozen(self):
        return self._frozen_H

    def get_H(self, idxs):
        """H restricted to the given states as a scipy CSR in *sample order* (hamiltonian.py:93-111,
        without the full-sample reordering quirk Q1).  Diagnostic path (solve_H); builds on
        ``naqs_get_hij`` and keeps explicit zeros like the reference."""
        from scipy.sparse import csr_matrix
        keys = keys_to_device(idxs, self.device)
        hij = self.dense_hij(keys).cpu().numpy()
        k = keys.cpu().numpy().view(np.uint64)
        xy_g = np.unique(self.packed.xy)
        order = np.argsort(k, kind="stable")
        ks = k[order]
        j = k[:, None] ^ xy_g[None, :]
        pos = np.searchsorted(ks, j)
        pos[pos == len(ks)] = 0
        hit = ks[pos] == j
        rows = np.broadcast_to(np.arange(len(k))[:, None], j.shape)[hit]
        cols = order[pos[hit]]
        return csr_matrix((hij[hit], (rows, cols)), shape=(len(k), len(k)))
