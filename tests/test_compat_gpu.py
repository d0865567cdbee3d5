"""The inner-ring shims (naqs_amd.compat: the reference's Cython module names and signatures, numpy in -> numpy
out through libnaqs_hip.so) against the reference's own intermediates (``ring_*`` / ``pp_*`` vectors of
tests/golden/eloc_<mol>.npz, recorded from src.utils.hamiltonian_math / sparse_math by make_golden.py)."""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

from conftest import golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2"])
def test_popcount_parity_and_get_hij_like_the_reference_calls_them(mol):
    """update_H's call sequence (hamiltonian.py:301-337) with the shims in place of the Cython functions."""
    from naqs_amd.compat.hamiltonian_math import get_Hij_cy, popcount_parity
    z, h = golden(f"eloc_{mol}.npz"), golden(f"ham_{mol}.npz")
    idx_dtype = np.int16 if int(h["n_qubits"]) < 16 else np.int32                       # hilbert.py:405-410
    keys = z["ring_keys"].astype(np.int64).astype(idx_dtype)
    uyz = h["unique_yz"].astype(np.int64).astype(idx_dtype)
    P_bits = np.bitwise_and(keys[:, None], uyz[None, :])
    P = popcount_parity(P_bits)
    assert P.dtype == np.int8 and np.array_equal(P, z["ring_P"])
    couplings = h["coeff"].reshape(-1, 1)                                               # [K, 1] like the reference
    Hij = get_Hij_cy(keys, h["unique_xy"], h["unique2all_xy"], P, h["unique2all_yz"], couplings)
    assert Hij.dtype == np.float64 and Hij.shape == z["ring_Hij"].shape
    assert np.array_equal(Hij, z["ring_Hij"])                                           # bit-identical
    H32 = get_Hij_cy(keys, h["unique_xy"], h["unique2all_xy"], P.astype(np.int64), h["unique2all_yz"],
                     couplings.astype(np.float32))
    assert H32.dtype == np.float32 and np.max(np.abs(H32 - z["ring_Hij"])) < 1e-5 * np.abs(z["ring_Hij"]).max()


def test_popcount_parity_dtypes_shapes_and_errors():
    from naqs_amd.compat.hamiltonian_math import popcount_parity
    z = golden("eloc_N2.npz")
    for name in ("int16", "int32", "int64"):
        arr = z[f"pp_in_{name}"]
        assert np.array_equal(popcount_parity(arr), z[f"pp_out_{name}"])
        neg = -arr - 1                                                                  # negative values: sign extension
        want = (1 - 2 * (np.array([bin(int(x) & (2 ** 64 - 1)).count("1") for x in neg.ravel()]) % 2)).astype(np.int8)
        assert np.array_equal(popcount_parity(neg).ravel(), want)
        assert np.array_equal(popcount_parity(arr.view(np.dtype("u" + arr.dtype.name))), z[f"pp_out_{name}"])
    small = np.arange(-128, 128, dtype=np.int8)
    got = popcount_parity(small)
    assert got.shape == (256, 1)                                                        # 1-D input -> [n, 1]
    assert np.array_equal(got.ravel(), [1 - 2 * (bin(int(x) & 0xFF).count("1") % 2) for x in small])
    assert np.array_equal(popcount_parity(small.view(np.uint8)).ravel(), got.ravel())
    with pytest.raises(TypeError, match="Unsupported array dtype for popcount_parity"):
        popcount_parity(np.ones((2, 2), np.float32))


@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2"])
def test_sparse_dense_mv_like_the_reference_calls_it(mol):
    from naqs_amd.compat.sparse_math import sparse_dense_mv
    z = golden(f"eloc_{mol}.npz")
    n = len(z["ring_keys"])
    H = csr_matrix((z["ring_csr_data"], z["ring_csr_indices"], z["ring_csr_indptr"]), shape=(n, n))
    v = z["ring_v"]                                                                     # complex64 psi, energy.py:241-243
    out = sparse_dense_mv(H, v)
    assert out.dtype == np.complex128 and np.max(np.abs(out - z["ring_mv"])) < 1e-13 * max(1, np.abs(z["ring_mv"]).max())
    assert np.array_equal(sparse_dense_mv(H, v, par=False), out)
    # the type table of __type_mv (sparse_math.pyx:13-41)
    H32 = H.astype(np.float32)
    assert sparse_dense_mv(H32, v.astype(np.complex64)).dtype == np.complex64
    assert sparse_dense_mv(H32, v.astype(np.complex128)).dtype == np.complex128
    assert sparse_dense_mv(H32, v.real.astype(np.float32)).dtype == np.complex64
    real = sparse_dense_mv(H, v.real)
    assert real.dtype == np.complex128 and np.max(np.abs(real - H @ v.real)) < 1e-12
    H64i = csr_matrix((H.data, H.indices.astype(np.int64), H.indptr.astype(np.int64)), shape=H.shape)
    assert np.array_equal(sparse_dense_mv(H64i, v), out)
    with pytest.raises(Exception, match="m must have dtype of np.float32 or np.float64"):
        sparse_dense_mv(H.astype(np.int32), v)


def test_local_energy_assembled_from_the_shims_equals_the_fused_kernel():
    """E_loc the reference's way — parity table, dense H_ij, CSR over the sampled states, SpMV, conj(./psi)
    (hamiltonian.py:301-363 + energy.py:248) — with every native call going through the shims, against the
    reference's complex128 result and the fused matrix-free kernel."""
    from naqs_amd import hamiltonian, packing
    from naqs_amd.compat.hamiltonian_math import get_Hij_cy, popcount_parity
    from naqs_amd.compat.sparse_math import sparse_dense_mv
    import os
    from conftest import GOLDEN
    z, h = golden("eloc_LiH.npz"), golden("ham_LiH.npz")
    keys = z["c1_keys"].astype(np.int64)
    M, Kxy = len(keys), len(h["unique_xy"])
    P = popcount_parity(np.bitwise_and(keys[:, None], h["unique_yz"].astype(np.int64)[None, :]))
    Hij = get_Hij_cy(keys, h["unique_xy"], h["unique2all_xy"], P, h["unique2all_yz"], h["coeff"])
    j = np.bitwise_xor(keys[:, None], h["unique_xy"].astype(np.int64)[None, :]).ravel()
    pos = np.searchsorted(keys, j)
    pos[pos == M] = 0
    hit = keys[pos] == j
    H = csr_matrix((Hij[hit], (np.repeat(np.arange(M), Kxy)[hit], pos[hit])), shape=(M, M))
    psi = z["c1_psi_f32"].astype(np.float64)
    v = psi[:, 0] + 1j * psi[:, 1]
    e = (sparse_dense_mv(H, v) / v).conj()
    ref = z["c1_eloc_c128"]
    assert np.max(np.abs(e - ref) / np.maximum(1, np.abs(ref))) < 1e-12
    ham = hamiltonian.DevicePauliHamiltonian(packing.load_packed(os.path.join(GOLDEN, "ham_LiH.npz")), device="cuda:0")
    f = ham.local_energy(hamiltonian.keys_to_device(z["c1_keys"], ham.device),
                         torch.as_tensor(z["c1_psi_f32"], device=ham.device), kind="psi").cpu().numpy()
    assert np.max(np.abs(e - (f[:, 0] + 1j * f[:, 1])) / np.maximum(1, np.abs(ref))) < 1e-10
