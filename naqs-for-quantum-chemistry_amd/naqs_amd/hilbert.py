"""Particle-number-restricted Hilbert space: the counterpart of the reference's
``Hilbert.get`` / ``_HilbertRestricted`` (src/utils/hilbert.py:28-37, :381-640).

Same public names and conventions — qubit q <-> bit q of the key, even q = alpha spin-orbitals,
odd q = beta (:446-449), states are int8 +-1 under ``Encoding.SIGNED`` (:573-581), idx dtype
int16/int32/int64 by N (:405-410) — but nothing is enumerated up front: the reference builds
the whole determinant list with itertools and a dense 2^N ``full2restricted`` table (:429-469,
8 GiB + 8 GiB at N = 30); here keys are converted with bit arithmetic, the restricted index
is a combinatorial rank from two 2^(N/2)-entry tables, and the basis is materialised lazily
and only on request (``get_subspace`` / ``get_basis``).
"""
import math
from enum import Enum
from itertools import combinations

import numpy as np
import torch


class Encoding(Enum):
    BINARY = 0
    SIGNED = 1


class Hilbert:
    @staticmethod
    def get(N, N_alpha=None, N_beta=None, *args, **kwargs):
        if N_alpha is None or N_beta is None:
            raise NotImplementedError("only the (N_alpha, N_beta)-restricted space is on the MI355X path "
                                      "(reference _HilbertFull / _HilbertPartiallyRestricted are out of scope)")
        if isinstance(N_alpha, (list, tuple, np.ndarray)) or isinstance(N_beta, (list, tuple, np.ndarray)):
            raise NotImplementedError("open-shell multi-sector spaces (_HilbertPartiallyRestricted) are out of scope")
        return HilbertRestricted(N, int(N_alpha), int(N_beta), *args, **kwargs)


def _half_rank_table(n_orb, n_set):
    """table[m] = lexicographic rank of the occupied-orbital combination encoded by the n_orb-bit
    mask m among itertools.combinations(range(n_orb), n_set) — the reference's enumeration order
    (hilbert.py:448-449) — or -1 when popcount(m) != n_set."""
    table = np.full(1 << n_orb, -1, np.int64)
    for r, comb_ in enumerate(combinations(range(n_orb), n_set)):
        table[sum(1 << o for o in comb_)] = r
    return table


def _compress_bits(keys, offset, n_orb):
    """bits offset, offset+2, ... of each key -> contiguous n_orb-bit integers."""
    out = np.zeros(keys.shape, np.int64)
    for o in range(n_orb):
        out |= ((keys >> np.uint64(2 * o + offset)) & np.uint64(1)).astype(np.int64) << o
    return out


class HilbertRestricted:
    def __init__(self, N, N_alpha=None, N_beta=None, encoding=Encoding.BINARY, make_basis=True, verbose=False):
        self.N = int(N)
        self.N_alpha, self.N_beta = int(N_alpha), int(N_beta)
        self.N_up = self.N_alpha + self.N_beta
        self.N_occ = 0
        self.n_orb_alpha, self.n_orb_beta = math.ceil(self.N / 2), self.N // 2
        assert 0 <= self.N_alpha <= self.n_orb_alpha and 0 <= self.N_beta <= self.n_orb_beta
        self.size = math.comb(self.n_orb_alpha, self.N_alpha) * math.comb(self.n_orb_beta, self.N_beta)
        if encoding not in (Encoding.BINARY, Encoding.SIGNED):
            raise ValueError("{} is not a recognised encoding.".format(encoding))
        self.encoding = encoding
        self.verbose = verbose
        self._state_torch_dtype, self._state_np_dtype = torch.int8, np.int8
        if N < 16:
            self._idx_torch_dtype, self._idx_np_dtype = torch.int16, np.int16
        elif N < 30:
            self._idx_torch_dtype, self._idx_np_dtype = torch.int32, np.int32
        else:
            self._idx_torch_dtype, self._idx_np_dtype = torch.int64, np.int64
        self._idx_basis_vec = torch.tensor([2 ** n for n in range(N)], dtype=torch.int64)
        self.alpha_mask = sum(1 << q for q in range(0, N, 2))
        self.beta_mask = sum(1 << q for q in range(1, N, 2))
        self._rank_a = self._rank_b = None
        self._unrank_a = self._unrank_b = None
        self._basis_keys = None

    # ---- dtype helpers (same names as the reference's _HilbertBase) ----
    def get_idx_dtype(self, type="torch"):
        type = type.lower()
        if type == "torch":
            return self._idx_torch_dtype
        if type in ("np", "numpy"):
            return self._idx_np_dtype
        raise ValueError("type must be 'torch' or 'numpy'.")

    def get_state_dtype(self, type="torch"):
        type = type.lower()
        if type == "torch":
            return self._state_torch_dtype
        if type in ("np", "numpy"):
            return self._state_np_dtype
        raise ValueError("type must be 'torch' or 'numpy'.")

    def to_idx_tensor(self, idx):
        return idx.to(self._idx_torch_dtype) if torch.is_tensor(idx) else torch.tensor(idx, dtype=self._idx_torch_dtype)

    def to_idx_array(self, idx):
        if torch.is_tensor(idx):
            idx = idx.cpu().numpy()
        return np.asarray(idx).astype(self._idx_np_dtype)

    def to_state_tensor(self, state):
        return state.to(self._state_torch_dtype) if torch.is_tensor(state) else torch.tensor(state, dtype=self._state_torch_dtype)

    # ---- key <-> state (bit arithmetic; works on any device) ----
    def state2idx(self, state, use_restricted_idxs=False):
        """[B, N] occupations (+-1 or 0/1) -> [B, 1] keys, idx = sum_q [s_q > 0] 2^q (hilbert.py:573-581)."""
        if isinstance(state, np.ndarray):
            state = torch.from_numpy(state)
        bits = (state > 0).to(torch.int64)
        idx = (bits * self._idx_basis_vec.to(state.device)).sum(dim=-1, keepdim=True)
        if use_restricted_idxs:
            idx = self.full2restricted_idx(idx)
        return self.to_idx_tensor(idx)

    def idx2state(self, idx, use_restricted_idxs=False):
        if not torch.is_tensor(idx):
            idx = torch.as_tensor(np.asarray(idx).astype(np.int64))
        idx = idx.reshape(-1).to(torch.int64)
        if use_restricted_idxs:
            idx = self.restricted2full_idx(idx).to(torch.int64)
        bits = (idx.unsqueeze(-1) >> torch.arange(self.N, device=idx.device)) & 1
        if self.encoding == Encoding.SIGNED:
            bits = 2 * bits - 1
        return bits.to(self._state_torch_dtype)

    def is_physical(self, idx):
        k = np.asarray(idx.cpu() if torch.is_tensor(idx) else idx).astype(np.int64).astype(np.uint64).reshape(-1)
        pa = np.array([bin(int(x) & self.alpha_mask).count("1") for x in k])
        pb = np.array([bin(int(x) & self.beta_mask).count("1") for x in k])
        return (pa == self.N_alpha) & (pb == self.N_beta)

    # ---- restricted index = rank in the reference's enumeration order ----
    def _tables(self):
        if self._rank_a is None:
            self._rank_a = _half_rank_table(self.n_orb_alpha, self.N_alpha)
            self._rank_b = _half_rank_table(self.n_orb_beta, self.N_beta)
        return self._rank_a, self._rank_b

    def full2restricted_idx(self, idx):
        """keys -> position in product(alpha combinations, beta combinations) (hilbert.py:446-469);
        -1 for keys outside the restricted space (the reference's physicality test, :607-623)."""
        np_out = not torch.is_tensor(idx)
        k = np.asarray(idx if np_out else idx.cpu().numpy()).astype(np.int64).astype(np.uint64)
        ra, rb = self._tables()
        a = ra[_compress_bits(k, 0, self.n_orb_alpha)]
        b = rb[_compress_bits(k, 1, self.n_orb_beta)]
        out = np.where((a >= 0) & (b >= 0), a * math.comb(self.n_orb_beta, self.N_beta) + b, -1)
        return self.to_idx_array(out) if np_out else self.to_idx_tensor(torch.from_numpy(out))

    def restricted2full_idx(self, idx):
        np_out = not torch.is_tensor(idx)
        r = np.asarray(idx if np_out else idx.cpu().numpy()).astype(np.int64)
        if self._unrank_a is None:
            ra, rb = self._tables()
            self._unrank_a = np.argsort(np.where(ra >= 0, ra, np.iinfo(np.int64).max), kind="stable")[:(ra >= 0).sum()]
            self._unrank_b = np.argsort(np.where(rb >= 0, rb, np.iinfo(np.int64).max), kind="stable")[:(rb >= 0).sum()]
        nb = math.comb(self.n_orb_beta, self.N_beta)
        ma, mb = self._unrank_a[r // nb], self._unrank_b[r % nb]
        out = np.zeros(r.shape, np.int64)
        for o in range(self.n_orb_alpha):
            out |= ((ma >> o) & 1) << (2 * o)
        for o in range(self.n_orb_beta):
            out |= ((mb >> o) & 1) << (2 * o + 1)
        return self.to_idx_array(out) if np_out else self.to_idx_tensor(torch.from_numpy(out))

    # ---- whole-space enumeration (diagnostics only; guarded) ----
    def _all_keys(self):
        if self._basis_keys is None:
            if self.size > 5_000_000:
                raise MemoryError(f"refusing to enumerate a {self.size}-state space; sample it instead")
            self._basis_keys = self.restricted2full_idx(np.arange(self.size)).astype(np.int64)
        return self._basis_keys

    def get_subspace(self, N_up=None, N_alpha=None, N_beta=None, N_occ=None, N_exc_max=None,
                     ret_states=True, ret_idxs=False, use_restricted_idxs=False):
        if N_occ not in (None, 0) or N_exc_max is not None:
            raise NotImplementedError("frozen-core / excitation-limited subspaces are out of scope")
        # the space IS the (N_alpha, N_beta) sector: the reference's arguments (energy.py:93-97) select it again or nothing
        for name, got, have in (("N_alpha", N_alpha, self.N_alpha), ("N_beta", N_beta, self.N_beta),
                                ("N_up", N_up, self.N_alpha + self.N_beta)):
            if got is not None and int(got) != int(have):
                raise NotImplementedError(f"get_subspace({name}={got}) on a space restricted to {name}={have}")
        keys = torch.from_numpy(self._all_keys())
        idxs = self.to_idx_tensor(torch.arange(self.size)) if use_restricted_idxs else self.to_idx_tensor(keys)
        if ret_states and ret_idxs:
            return self.idx2state(keys), idxs
        if ret_states:
            return self.idx2state(keys)
        return idxs

    def get_basis(self, ret_states=True, ret_idxs=False, use_restricted_idxs=False):
        return self.get_subspace(ret_states=ret_states, ret_idxs=ret_idxs, use_restricted_idxs=use_restricted_idxs)
