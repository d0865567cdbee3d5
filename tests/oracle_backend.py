"""TEST-ONLY stand-in for DevicePauliHamiltonian backed by the CPU oracle, so that the host logic
around the kernels (optimiser step, row sharding, collectives) can run without a GPU.  Never
imported by the product."""
import numpy as np
import torch

from naqs_amd.packing import pack_qubit_hamiltonian, PackedHamiltonian
from oracle import oracle


class OracleHamiltonian:
    def __init__(self, packed):
        self.packed = packed
        self.device = torch.device("cpu")

    def local_energy(self, keys, wf, kind="psi", row_begin=0, n_rows=None, out=None, weights=None, sums_out=None):
        k = keys.cpu().numpy().astype(np.int64).view(np.uint64)
        v = wf.detach().cpu().numpy().astype(np.float64)
        psi = np.exp(v[:, 0] + 1j * v[:, 1]) if kind == "log_psi" else v[:, 0] + 1j * v[:, 1]
        e = oracle.eloc_matrix_free(self.packed.xy, self.packed.yz, self.packed.coeff, k, psi,
                                    row_begin=row_begin, n_rows=n_rows)
        out = torch.tensor(np.stack([e.real, e.imag], -1))
        if weights is not None:
            return out, self.reduce(weights, out)
        return out

    def reduce(self, weights, eloc):
        w = weights.double()
        return torch.stack([(w * eloc[:, 0]).sum(), (w * eloc[:, 1]).sum(), (w * eloc[:, 0] ** 2).sum(), w.sum()])

    def get_H(self, idxs):
        from scipy.sparse import csr_matrix
        k = (idxs.cpu().numpy() if torch.is_tensor(idxs) else np.asarray(idxs)).astype(np.int64).view(np.uint64).reshape(-1)
        hij, _ = oracle.get_hij(k, self.packed.xy, self.packed.yz, self.packed.coeff)
        xy_g = np.unique(self.packed.xy)
        hij = hij.reshape(len(k), len(xy_g))
        order = np.argsort(k, kind="stable")
        ks = k[order]
        j = k[:, None] ^ xy_g[None, :]
        pos = np.searchsorted(ks, j)
        pos[pos == len(ks)] = 0
        hit = ks[pos] == j
        rows = np.broadcast_to(np.arange(len(k))[:, None], j.shape)[hit]
        return csr_matrix((hij[hit], (rows, order[pos[hit]])), shape=(len(k), len(k)))


def install(monkeypatch_or_module):
    """Route naqs_amd.optimizer.PauliHamiltonian.get to the oracle-backed stand-in."""
    import naqs_amd.optimizer as opt

    class _Factory:
        @staticmethod
        def get(hilbert, qubit_hamiltonian, **kw):
            packed = qubit_hamiltonian if isinstance(qubit_hamiltonian, PackedHamiltonian) else \
                pack_qubit_hamiltonian(qubit_hamiltonian.terms, hilbert.N, hilbert.N_alpha, hilbert.N_beta)
            return OracleHamiltonian(packed)

    if hasattr(monkeypatch_or_module, "setattr"):
        monkeypatch_or_module.setattr(opt, "PauliHamiltonian", _Factory)
    else:
        opt.PauliHamiltonian = _Factory
    opt.keys_to_device = lambda keys, device: (torch.from_numpy(np.ascontiguousarray(keys).astype(np.uint64).view(np.int64))
                                               if isinstance(keys, np.ndarray) else keys.reshape(-1).to(torch.int64))
