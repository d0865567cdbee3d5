"""Parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle
and the golden vectors dumped from the reference.  Run on the GPU box with ``-m gpu``.

Tolerances: matrix elements / parities are bit-exact; E_loc is f64 arithmetic with a different
summation order than the reference's SpMV, asserted to 1e-10 relative (north star: 1e-6 Ha).
"""
import ctypes
import os
from itertools import combinations

import numpy as np
import pytest

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def rel_err(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) if len(a) else 0.0


@pytest.fixture(scope="module")
def env():
    from naqs_amd import _lib, hamiltonian, packing
    from oracle import oracle
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.load_library()
    return dict(lib=_lib, H=hamiltonian, P=packing, O=oracle)


def dev_ham(env, mol):
    return env["H"].DevicePauliHamiltonian(env["P"].load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz")))


def run_eloc(env, ham, keys, wf, kind="psi", dtype=torch.float32, **kw):
    k = env["H"].keys_to_device(np.asarray(keys, np.uint64), ham.device)
    w = torch.as_tensor(np.asarray(wf), dtype=dtype, device=ham.device)
    e = ham.local_energy(k, w, kind=kind, **kw)
    torch.cuda.synchronize()
    e = e.cpu().numpy()
    return e[:, 0] + 1j * e[:, 1]


def all_keys(N, na, nb):
    al = [sum(1 << b for b in c) for c in combinations(range(0, N, 2), na)]
    be = [sum(1 << b for b in c) for c in combinations(range(1, N, 2), nb)]
    return np.sort(np.array([a | b for a in al for b in be], np.uint64))


def random_physical_keys(N, na, nb, M, seed):
    rs = np.random.RandomState(seed)
    keys = set()
    ev, od = np.arange(0, N, 2), np.arange(1, N, 2)
    while len(keys) < M:
        a = rs.choice(ev, na, replace=False)
        b = rs.choice(od, nb, replace=False)
        keys.add(int(sum(1 << int(q) for q in a) | sum(1 << int(q) for q in b)))
    return np.sort(np.array(list(keys), np.uint64))


def synth_logpsi(M, seed, sigma=2.0):
    rs = np.random.RandomState(seed)
    return np.stack([rs.normal(-0.5 * np.log(max(M, 1)), sigma, M), rs.uniform(0, 2 * np.pi, M)], -1)


# ------------------------------------------------------------------ golden vectors (reference outputs)
@pytest.mark.parametrize("mol,tag", [("LiH", "c1"), ("LiH", "half"), ("H2O", "c1"), ("N2", "small"), ("N2", "c2")])
def test_eloc_matches_reference_golden(env, mol, tag):
    z = golden(f"eloc_{mol}.npz")
    ham = dev_ham(env, mol)
    e = run_eloc(env, ham, z[f"{tag}_keys"], z[f"{tag}_psi_f32"])
    want = z[f"{tag}_eloc_c128"]
    assert rel_err(e, want) < 1e-10
    # variational energy for fixed samples: north-star bound 1e-6 Ha, we hold 1e-9
    w = np.abs(z[f"{tag}_psi_f32"][:, 0] + 1j * z[f"{tag}_psi_f32"][:, 1]) ** 2
    w /= w.sum()
    assert abs(np.sum(w * e.real) - np.sum(w * want.real)) < 1e-9
    # float32 tensor the reference hands to the optimiser
    assert np.max(np.abs(e.real - z[f"{tag}_eloc_f32"][:, 0]) / np.maximum(1, np.abs(e.real))) < 2e-6


def test_li2o_subset_matches_reference_cython_kernels(env):
    """Config-4 Hamiltonian, golden from the reference's Cython kernels (int64 idx dtype path)."""
    z = golden("eloc_Li2O_subset.npz")
    ham = dev_ham(env, "Li2O")
    e = run_eloc(env, ham, z["keys"], z["psi_f32"])
    assert rel_err(e, z["eloc_c128"]) < 1e-10
    e = run_eloc(env, ham, z["keys"], z["log_psi_f32"], kind="log_psi")
    assert rel_err(e, z["eloc_c128"]) < 2e-5          # log-psi route recomputes psi from float32 (log|psi|, phase)


@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2"])
def test_inner_ring_bit_exact(env, mol):
    z = golden(f"eloc_{mol}.npz")
    ham = dev_ham(env, mol)
    k = env["H"].keys_to_device(z["ring_keys"], ham.device)
    hij = ham.dense_hij(k).cpu().numpy().ravel()
    assert np.array_equal(hij, z["ring_Hij"])                      # get_Hij_cy, identical bits
    for name, tdt in (("int16", torch.int16), ("int32", torch.int32), ("int64", torch.int64)):
        a = torch.as_tensor(z[f"pp_in_{name}"], dtype=tdt, device=ham.device)
        got = env["H"].popcount_parity_device(a).cpu().numpy()
        assert np.array_equal(got, z[f"pp_out_{name}"])
    dev = ham.device
    mv = env["H"].csr_mv_device(torch.as_tensor(z["ring_csr_data"], device=dev),
                                torch.as_tensor(z["ring_csr_indices"], device=dev),
                                torch.as_tensor(z["ring_csr_indptr"], device=dev),
                                torch.as_tensor(np.stack([z["ring_v"].real, z["ring_v"].imag], -1).astype(np.float64),
                                                device=dev)).cpu().numpy()
    assert rel_err(mv[:, 0] + 1j * mv[:, 1], z["ring_mv"]) < 1e-13


def test_popcount_parity_negative_1d_typeerror(env):
    a = torch.tensor([-1, -2, 5, 0, -32768], dtype=torch.int16, device="cuda")
    got = env["H"].popcount_parity_device(a).cpu().numpy()
    assert got.shape == (5, 1) and np.array_equal(got, env["O"].popcount_parity(a.cpu().numpy()))
    with pytest.raises(TypeError):
        env["H"].popcount_parity_device(torch.zeros(3, device="cuda"))


# ------------------------------------------------------------------ oracle on seeded inputs
@pytest.mark.parametrize("mol,M,seed", [("LiH", 37, 1), ("LiH", 224, 2), ("H2O", 440, 3), ("N2", 3000, 4)])
def test_eloc_vs_oracle_all_input_kinds(env, mol, M, seed):
    h = golden(f"ham_{mol}.npz")
    ham = dev_ham(env, mol)
    space = all_keys(int(h["n_qubits"]), int(h["n_alpha"]), int(h["n_beta"]))
    rs = np.random.RandomState(seed)
    keys = rs.permutation(rs.choice(space, M, replace=False))      # deliberately unsorted
    lp = synth_logpsi(M, seed)
    psi64 = np.exp(lp[:, 0]) * np.exp(1j * lp[:, 1])
    want = env["O"].eloc_matrix_free(h["xy"], h["yz"], h["coeff"], keys, psi64)
    e = run_eloc(env, ham, keys, np.stack([psi64.real, psi64.imag], -1), dtype=torch.float64)
    assert rel_err(e, want) < 1e-10
    e = run_eloc(env, ham, keys, lp, kind="log_psi", dtype=torch.float64)
    assert rel_err(e, want) < 1e-9                                  # device exp/sincos vs numpy
    lp32 = lp.astype(np.float32)
    psi32 = np.exp(lp32[:, 0].astype(np.float64)) * np.exp(1j * lp32[:, 1].astype(np.float64))
    want32 = env["O"].eloc_matrix_free(h["xy"], h["yz"], h["coeff"], keys, psi32)
    e = run_eloc(env, ham, keys, lp32, kind="log_psi", dtype=torch.float32)
    assert rel_err(e, want32) < 1e-9


def test_row_shards_tile_the_full_result(env):
    """The multi-GPU shape: each rank evaluates a contiguous slice of rows against the whole table."""
    h, z = golden("ham_N2.npz"), golden("eloc_N2.npz")
    ham = dev_ham(env, "N2")
    keys, psi, want = z["small_keys"], z["small_psi_f32"], z["small_eloc_c128"]
    parts = []
    for r in range(3):
        b = len(keys) * r // 3
        n = len(keys) * (r + 1) // 3 - b
        parts.append(run_eloc(env, ham, keys, psi, row_begin=b, n_rows=n))
    assert rel_err(np.concatenate(parts), want) < 1e-10
    assert len(run_eloc(env, ham, keys, psi, row_begin=5, n_rows=0)) == 0


def test_staging_variants_agree_bitwise(env):
    z = golden("eloc_N2.npz")
    ham = dev_ham(env, "N2")
    res = []
    for stage in ("0", "1", "2"):
        os.environ["NAQS_STAGE"] = stage
        try:
            res.append(run_eloc(env, ham, z["small_keys"], z["small_psi_f32"]))
        finally:
            del os.environ["NAQS_STAGE"]
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])
    for rpb in ("4", "64", "1000"):
        os.environ["NAQS_ROWS_PER_BLOCK"] = rpb
        try:
            assert np.array_equal(run_eloc(env, ham, z["small_keys"], z["small_psi_f32"]), res[0])
        finally:
            del os.environ["NAQS_ROWS_PER_BLOCK"]


@pytest.mark.parametrize("mol,tag", [("LiH", "c1"), ("H2O", "c1"), ("N2", "c2")])
def test_single_and_double_compaction_kernels_agree(env, mol, tag):
    """NAQS_ELOC_V=1 (one compaction: a lane owns a group through filter, probe and push; heavy hits summed by the whole
    wave) and the default eloc_kernel2 / eloc_kernel3 (filter pass queue -> dense probe passes, heavy groups as <= 8-term
    chunks) are the same sums in a different order: 1e-12 relative, and all within the golden tolerance.  eloc_kernel3 is
    eloc_kernel2 with a leaner instruction stream — the same sums in the SAME order: bit-identical."""
    z = golden(f"eloc_{mol}.npz")
    ham = dev_ham(env, mol)
    e = {}
    for v in ("1", "3"):
        os.environ["NAQS_ELOC_V"] = v
        try:
            e[v] = run_eloc(env, ham, z[f"{tag}_keys"], z[f"{tag}_psi_f32"])
            assert f"eloc_kernel{'' if v == '1' else v}<" in ham.last_kernel()
        finally:
            del os.environ["NAQS_ELOC_V"]
    e["2"] = run_eloc(env, ham, z[f"{tag}_keys"], z[f"{tag}_psi_f32"])
    assert "eloc_kernel2<" in ham.last_kernel()
    assert rel_err(e["1"], e["3"]) < 1e-12
    assert np.array_equal(e["2"], e["3"])
    for v in e:
        assert rel_err(e[v], z[f"{tag}_eloc_c128"]) < 1e-10, v


def test_eloc_against_dense_pauli_algebra(env):
    """The HIP path against linear algebra, with neither the oracle nor the reference in between: a random real-symmetric
    qubit Hamiltonian as a dense matrix of Kronecker products, restricted to a random sample set (no particle filter)."""
    from conftest import dense_pauli_case
    N = 6
    terms, dense, rs = dense_pauli_case(N)
    ham = env["H"].DevicePauliHamiltonian(env["P"].pack_qubit_hamiltonian(terms, N, -1, -1))
    for M in (1 << N, 40):
        keys = np.sort(rs.choice(1 << N, size=M, replace=False)).astype(np.uint64)
        psi = rs.normal(size=M) + 1j * rs.normal(size=M)
        sub = dense[np.ix_(keys.astype(int), keys.astype(int))]
        want = np.conj(sub @ psi / psi)
        e = run_eloc(env, ham, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
        assert np.max(np.abs(e - want)) < 1e-12 * max(1.0, np.max(np.abs(want)))


def test_edge_cases(env):
    h = golden("ham_LiH.npz")
    ham = dev_ham(env, "LiH")
    # a single sample: only the diagonal group couples
    key = all_keys(12, 2, 2)[:1]
    e = run_eloc(env, ham, key, np.array([[0.3, -0.4]]), dtype=torch.float64)
    want = env["O"].eloc_matrix_free(h["xy"], h["yz"], h["coeff"], key, np.array([0.3 - 0.4j]))
    assert rel_err(e, want) < 1e-12 and abs(e[0].imag) < 1e-12
    # M = 0
    k = torch.empty(0, dtype=torch.int64, device=ham.device)
    w = torch.empty((0, 2), dtype=torch.float32, device=ham.device)
    assert ham.local_energy(k, w).shape == (0, 2)
    # bad shapes / ranges raise
    k = env["H"].keys_to_device(all_keys(12, 2, 2)[:8], ham.device)
    with pytest.raises(ValueError):
        ham.local_energy(k, torch.zeros((7, 2), device=ham.device))
    with pytest.raises(env["lib"].NaqsError):
        ham.local_energy(k, torch.ones((8, 2), device=ham.device), row_begin=4, n_rows=8)
    # scratch growth: small call, then a larger one, then small again
    for M in (10, 200, 10):
        ks = all_keys(12, 2, 2)[:M]
        lp = synth_logpsi(M, 9)
        psi = np.exp(lp[:, 0] + 1j * lp[:, 1])
        e = run_eloc(env, ham, ks, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
        assert rel_err(e, env["O"].eloc_matrix_free(h["xy"], h["yz"], h["coeff"], ks, psi)) < 1e-10


def test_hamiltonian_without_diagonal_and_empty(env):
    P, H, O = env["P"], env["H"], env["O"]
    full = P.load_packed(os.path.join(GOLDEN, "ham_LiH.npz"))
    nd = full.xy != 0
    ham = H.DevicePauliHamiltonian(P.PackedHamiltonian(12, 2, 2, full.xy[nd], full.yz[nd], full.coeff[nd]))
    keys = all_keys(12, 2, 2)[::2]
    lp = synth_logpsi(len(keys), 5)
    psi = np.exp(lp[:, 0] + 1j * lp[:, 1])
    e = run_eloc(env, ham, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
    assert rel_err(e, O.eloc_matrix_free(full.xy[nd], full.yz[nd], full.coeff[nd], keys, psi)) < 1e-10
    empty = H.DevicePauliHamiltonian(P.PackedHamiltonian(12, 2, 2, np.zeros(0, np.uint64), np.zeros(0, np.uint64),
                                                         np.zeros(0)))
    e = run_eloc(env, empty, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
    assert np.all(e == 0)


@pytest.mark.parametrize("filtered", [False, True])
def test_64bit_keys(env, filtered):
    """n_qubits > 32 takes the 64-bit mask path; synthetic Hamiltonian (no reference molecule is that big)."""
    P, H, O = env["P"], env["H"], env["O"]
    N, na, nb = 40, 6, 6
    rs = np.random.RandomState(17)
    keys = random_physical_keys(N, na, nb, 1500, 17)
    # XY masks that connect sampled states: xor of random key pairs (+ the diagonal), a few terms each
    pairs = rs.randint(0, len(keys), size=(300, 2))
    xys = np.unique(np.r_[np.uint64(0), keys[pairs[:, 0]] ^ keys[pairs[:, 1]]])
    xy = np.repeat(xys, rs.randint(1, 6, size=len(xys)))
    yz = rs.randint(0, 1 << 20, size=len(xy)).astype(np.uint64) | (rs.randint(0, 1 << 20, size=len(xy)).astype(np.uint64) << np.uint64(20))
    cf = rs.normal(size=len(xy))
    perm = rs.permutation(len(xy))
    xy, yz, cf = xy[perm], yz[perm], cf[perm]
    ham = H.DevicePauliHamiltonian(P.PackedHamiltonian(N, na if filtered else -1, nb if filtered else -1, xy, yz, cf))
    assert ham.key_bits == 64
    lp = synth_logpsi(len(keys), 3)
    psi = np.exp(lp[:, 0] + 1j * lp[:, 1])
    e = run_eloc(env, ham, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
    want = O.eloc_matrix_free(xy, yz, cf, keys, psi)
    assert np.count_nonzero(np.abs(want) > 0) > 100
    assert rel_err(e, want) < 1e-10
    # the other code paths of the 64-bit instantiations: Bloom filter in LDS (needs the 1024-thread workgroup), the
    # single-compaction kernel, both
    for extra in ({"NAQS_BLOOM": "1", "NAQS_BLOCK": "1024"}, {"NAQS_ELOC_V": "1"},
                  {"NAQS_ELOC_V": "1", "NAQS_BLOOM": "1", "NAQS_BLOCK": "1024"}):
        os.environ.update(extra)
        try:
            e2 = run_eloc(env, ham, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
        finally:
            for k in extra:
                del os.environ[k]
        assert rel_err(e2, want) < 1e-10, extra
        if "NAQS_ELOC_V" not in extra:
            assert np.array_equal(e2, e), extra                      # the filter may only skip look-ups that would miss


def test_li2o_subset_vs_oracle(env):
    """Config 4 shape (30 qubits, 20 558 terms) at a size the oracle finishes in seconds."""
    h = golden("ham_Li2O.npz")
    ham = dev_ham(env, "Li2O")
    rs = np.random.RandomState(1234)
    # clustered samples (single/double excitations of one determinant) so that couplings do hit
    base = random_physical_keys(30, 7, 7, 1, 7)[0]
    keys = {int(base)}
    uxy = np.unique(h["xy"])
    frontier = [int(base)]
    while len(keys) < 4000:
        k = frontier[rs.randint(len(frontier))]
        j = k ^ int(uxy[rs.randint(len(uxy))])
        if bin(j & 0x15555555).count("1") == 7 and bin(j & 0x2AAAAAAA).count("1") == 7 and j not in keys:
            keys.add(j)
            frontier.append(j)
    keys = np.sort(np.array(list(keys), np.uint64))
    lp = synth_logpsi(len(keys), 11)
    psi = np.exp(lp[:, 0] + 1j * lp[:, 1])
    e = run_eloc(env, ham, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
    want = env["O"].eloc_matrix_free(h["xy"], h["yz"], h["coeff"], keys, psi)
    assert rel_err(e, want) < 1e-10


def clustered_keys(h, M, seed):
    """Up to M physical keys grown from one determinant by the Hamiltonian's own flip masks, so that couplings do hit.
    (A molecule's point-group symmetry splits a sector into components the Hamiltonian does not connect — CH2's (5, 3)
    sector of 735 states has components of ~390 — so fewer than M come back when the component is smaller.)"""
    N, na, nb = int(h["n_qubits"]), int(h["n_alpha"]), int(h["n_beta"])
    am = sum(1 << q for q in range(0, N, 2))
    bm = sum(1 << q for q in range(1, N, 2))
    rs = np.random.RandomState(seed)
    base = int(random_physical_keys(N, na, nb, 1, seed)[0])
    uxy = [int(x) for x in np.unique(h["xy"]) if x]
    keys, frontier = {base}, [base]
    while frontier and len(keys) < M:
        k = frontier.pop(rs.randint(len(frontier)))
        for x in rs.permutation(uxy)[:64]:                      # a random subset of the neighbours keeps the cluster ragged
            j = k ^ int(x)
            if len(keys) < M and j not in keys and bin(j & am).count("1") == na and bin(j & bm).count("1") == nb:
                keys.add(j)
                frontier.append(j)
        if len(frontier) == 0 and len(keys) < M:                # walk exhausted its random subsets: full neighbourhoods
            for k2 in list(keys):
                for x in uxy:
                    j = k2 ^ x
                    if len(keys) < M and j not in keys and bin(j & am).count("1") == na and bin(j & bm).count("1") == nb:
                        keys.add(j)
                        frontier.append(j)
            if len(frontier) == 0:
                break
    return np.sort(np.array(list(keys), np.uint64))


@pytest.mark.parametrize("mol", ["PH3", "H4O2", "C2", "CH2", "O2"])
def test_hamiltonian_shapes_beyond_baseline_vs_oracle(env, mol):
    """Shapes BASELINE never selects: PH3 (24 qubits, 24 369 Pauli strings — more terms than Li2O), H4O2 (28 qubits,
    28 393 strings: the term tables no longer fit the LDS budget -> STAGE 1), C2 (20 qubits), and the open-shell
    triplets CH2 (5 alpha / 3 beta) and O2 (9 / 7) whose sectors have n_alpha != n_beta.  Packing of each is pinned to
    the reference's by tests/test_packing.py; E_loc at M = 2 000 against the pinned oracle."""
    h = golden(f"ham_{mol}.npz")
    ham = dev_ham(env, mol)
    keys = clustered_keys(h, 2000, 21)
    assert len(keys) >= 250, len(keys)
    lp = synth_logpsi(len(keys), 12)
    psi = np.exp(lp[:, 0] + 1j * lp[:, 1])
    e = run_eloc(env, ham, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
    print(mol, ham.last_kernel())
    want = env["O"].eloc_matrix_free(h["xy"], h["yz"], h["coeff"], keys, psi)
    assert rel_err(e, want) < 1e-10
    n_conn = np.count_nonzero(np.abs(want - want.real.mean()) > 0)        # the test is vacuous if nothing couples
    assert n_conn > len(keys) // 2
    # every LDS-staging tier the size selects or allows must agree bit for bit
    for stage in ("0", "1"):
        os.environ["NAQS_STAGE"] = stage
        try:
            e2 = run_eloc(env, ham, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
        finally:
            del os.environ["NAQS_STAGE"]
        assert np.array_equal(e, e2), (mol, stage)


# ------------------------------------------------------------------ size-independent properties at full size
@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2"])
def test_exact_eigenvector_gives_constant_local_energy(env, mol):
    """Physics KAT on the whole restricted space: H psi0 = E0 psi0  =>  E_loc[i] == E0 for every i."""
    import json
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    h = golden(f"ham_{mol}.npz")
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    ham = dev_ham(env, mol)
    keys = all_keys(int(h["n_qubits"]), int(h["n_alpha"]), int(h["n_beta"]))
    k = env["H"].keys_to_device(keys, ham.device)
    hij = ham.dense_hij(k).cpu().numpy()
    uxy = np.unique(h["xy"])
    j = keys[:, None] ^ uxy[None, :]
    pos = np.searchsorted(keys, j)
    pos[pos == len(keys)] = 0
    hit = keys[pos] == j
    rows = np.broadcast_to(np.arange(len(keys))[:, None], j.shape)[hit]
    Hm = sp.csr_matrix((hij[hit], (rows, pos[hit])), shape=(len(keys),) * 2)
    w, v = spla.eigsh(Hm, k=1, which="SA")
    assert abs(w[0] - kat["fci"][mol]) < 1e-8
    psi = v[:, 0] * np.exp(0.7j)                                   # global phase must not matter
    big = np.abs(psi) > 1e-6 * np.abs(psi).max()
    e = run_eloc(env, ham, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
    assert np.max(np.abs(e[big] - w[0])) < 1e-6 * max(1.0, abs(w[0]))


def test_full_size_properties_li2o_50k(env):
    """BASELINE config 4 size (M = 50 000, Li2O).  No oracle at this size; properties instead:
    <psi|H|psi> = sum |psi_i|^2 conj(E_loc_i) is real for a symmetric H, E_loc is invariant under a
    global rescaling/rephasing of psi, and a row shard equals the same rows of the full result."""
    h = golden("ham_Li2O.npz")
    ham = dev_ham(env, "Li2O")
    keys = random_physical_keys(30, 7, 7, 50000, 1234)
    lp = synth_logpsi(len(keys), 4321, sigma=1.0)
    psi = np.exp(lp[:, 0] + 1j * lp[:, 1])
    wf = np.stack([psi.real, psi.imag], -1)
    e = run_eloc(env, ham, keys, wf, dtype=torch.float64)
    assert np.all(np.isfinite(e.real)) and np.all(np.isfinite(e.imag))
    expect = np.sum(np.abs(psi) ** 2 * np.conj(e))
    assert abs(expect.imag) < 1e-9 * abs(expect.real)
    c = 3.0 * np.exp(1.1j)
    e2 = run_eloc(env, ham, keys, np.stack([(c * psi).real, (c * psi).imag], -1), dtype=torch.float64)
    assert rel_err(e2, e) < 1e-11
    part = run_eloc(env, ham, keys, wf, dtype=torch.float64, row_begin=12345, n_rows=6250)
    assert np.array_equal(part, e[12345:12345 + 6250])
    # spot-check 64 random rows against the oracle's direct formula
    rows = np.random.RandomState(0).choice(len(keys), 64, replace=False)
    for r in rows[:8]:
        want = env["O"].eloc_matrix_free(h["xy"], h["yz"], h["coeff"], keys, psi, row_begin=int(r), n_rows=1)
        assert rel_err(e[r:r + 1], want) < 1e-10


def test_reduce_matches_oracle(env):
    z = golden("nade_N2.npz")
    ham = dev_ham(env, "N2")
    e = z["sgd_eloc_c128"]
    w = z["samp_counts"].astype(np.float64)
    et = torch.as_tensor(np.stack([e.real, e.imag], -1), device=ham.device)
    got = ham.reduce(torch.as_tensor(w, device=ham.device), et).cpu().numpy()
    want = env["O"].eloc_reduce(w, e)
    assert np.max(np.abs(got - want) / np.maximum(1, np.abs(want))) < 1e-12
    E = got[0] / got[3]
    assert abs(E - z["sgd_E"]) < 2e-5 * abs(E)


def test_fused_reduction_and_epoch_wrap(env):
    """naqs_eloc_reduced == naqs_eloc + naqs_eloc_reduce; and 600 consecutive calls on one handle with
    alternating sample sets cross the 8-bit epoch wrap of the never-cleared hash table twice."""
    h = golden("ham_N2.npz")
    z = golden("eloc_N2.npz")
    ham = dev_ham(env, "N2")
    keys_a, psi_a, want_a = z["small_keys"], z["small_psi_f32"], z["small_eloc_c128"]
    space = all_keys(20, 7, 7)
    keys_b = np.sort(np.random.RandomState(77).choice(space, 3000, replace=False))
    lp = synth_logpsi(3000, 78)
    psi_b = np.exp(lp[:, 0] + 1j * lp[:, 1])
    want_b = env["O"].eloc_matrix_free(h["xy"], h["yz"], h["coeff"], keys_b, psi_b)
    ka = env["H"].keys_to_device(keys_a, ham.device)
    kb = env["H"].keys_to_device(keys_b, ham.device)
    wa = torch.as_tensor(psi_a, device=ham.device)
    wb = torch.as_tensor(np.stack([psi_b.real, psi_b.imag], -1), device=ham.device)
    w = torch.rand(len(keys_a), dtype=torch.float64, device=ham.device)
    e, sums = ham.local_energy(ka, wa, weights=w)
    ref = ham.reduce(w, e)
    torch.cuda.synchronize()
    assert torch.max(torch.abs(sums - ref) / ref.abs().clamp(min=1)).item() < 1e-12
    for it in range(600):
        if it % 2:
            e = ham.local_energy(kb, wb)
        else:
            e, sums = ham.local_energy(ka, wa, weights=w)
    torch.cuda.synchronize()
    eb = ham.local_energy(kb, wb).cpu().numpy()
    ea, sums = ham.local_energy(ka, wa, weights=w)
    ea = ea.cpu().numpy()
    assert rel_err(eb[:, 0] + 1j * eb[:, 1], want_b) < 1e-10
    assert rel_err(ea[:, 0] + 1j * ea[:, 1], want_a) < 1e-10
    assert torch.max(torch.abs(sums - ref) / ref.abs().clamp(min=1)).item() < 1e-12
    # row shard + weights of the shard
    e2, s2 = ham.local_energy(ka, wa, row_begin=500, n_rows=700, weights=w[500:1200])
    assert torch.max(torch.abs(s2 - ham.reduce(w[500:1200], e2)) / s2.abs().clamp(min=1)).item() < 1e-12


def test_bloom_filter_variant_is_bit_identical(env):
    """Large batches take the LDS Bloom-filter variant of eloc_kernel (NAQS_BLOOM forces it on/off): it may
    only skip look-ups that would miss, so E_loc must not change by a single bit."""
    ham = dev_ham(env, "Li2O")
    z = golden("eloc_Li2O_subset.npz")
    keys = random_physical_keys(30, 7, 7, 24000, 5)
    keys = np.unique(np.r_[keys, z["keys"]])                      # a connected cluster inside a random cloud
    lp = synth_logpsi(len(keys), 6, sigma=1.0)
    psi = np.exp(lp[:, 0] + 1j * lp[:, 1])
    wf = np.stack([psi.real, psi.imag], -1)
    res = {}
    for flag in ("0", "1"):
        os.environ["NAQS_BLOOM"] = flag
        try:
            res[flag] = run_eloc(env, ham, keys, wf, dtype=torch.float64)
        finally:
            del os.environ["NAQS_BLOOM"]
    assert np.array_equal(res["0"], res["1"])
    auto = run_eloc(env, ham, keys, wf, dtype=torch.float64)      # M >= 20000 -> Bloom variant by default
    assert np.array_equal(auto, res["0"])
    assert np.count_nonzero(np.abs(res["0"].imag) > 1e-12) > 500  # the cluster really couples
    # small molecule with the filter forced on (all tables + filter in LDS)
    z2 = golden("eloc_N2.npz")
    ham2 = dev_ham(env, "N2")
    os.environ["NAQS_BLOOM"] = "1"
    try:
        e = run_eloc(env, ham2, z2["c2_keys"], z2["c2_psi_f32"])
    finally:
        del os.environ["NAQS_BLOOM"]
    assert rel_err(e, z2["c2_eloc_c128"]) < 1e-10


@pytest.mark.parametrize("mol,n", [("LiH", None), ("H2O", 300), ("N2", 3000)])
def test_matvec_and_lanczos_against_the_explicit_submatrix(mol, n):
    """naqs_hmatvec == get_H(keys) @ v (the reference's sub-matrix, built from naqs_get_hij), and the device Lanczos
    finds scipy's lowest eigenpair of that matrix; on the whole LiH space that is the FCI energy of kat.json."""
    import json
    import scipy.sparse.linalg as spla
    from naqs_amd import hamiltonian, packing
    from naqs_amd.hilbert import Encoding, Hilbert
    from test_nade import ELECTRONS
    N, na, nb = ELECTRONS[mol]
    hil = Hilbert.get(N, na, nb, encoding=Encoding.SIGNED)
    packed = packing.load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz"))
    ham = hamiltonian.PauliHamiltonian.get(hil, packed, device="cuda")
    all_keys = np.sort(hil._all_keys())
    keys_np = all_keys if n is None else np.sort(np.random.RandomState(5).choice(all_keys, n, replace=False))
    keys = hamiltonian.keys_to_device(keys_np.astype(np.int64), "cuda")
    H = ham.get_H(keys).astype(np.float64)
    assert abs(H - H.T).max() < 1e-12
    v = np.random.RandomState(1).randn(len(keys_np))
    got = ham.matvec(keys, torch.as_tensor(v, device="cuda")).cpu().numpy()
    want = H @ v
    assert np.max(np.abs(got - want)) < 1e-10 * max(1.0, np.abs(want).max())
    val, vec = ham.lowest_eigenpair(keys)
    w, u = spla.eigsh(H, k=1, which="SA")
    assert abs(val - w[0]) < 1e-8
    vec = vec.cpu().numpy()
    assert min(np.abs(vec - u[:, 0]).max(), np.abs(vec + u[:, 0]).max()) < 1e-5
    if n is None:
        kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
        fci = kat["fci"][mol] if "fci" in kat and mol in kat["fci"] else None
        if fci is not None:
            assert abs(val - fci) < 1e-8


@pytest.mark.parametrize("mol", ["H2O", "N2"])
def test_device_lanczos_reproduces_fci_on_the_whole_space(mol):
    """Matrix-free Lanczos over every physical state == the FCI energy recorded from the reference Hamiltonian."""
    import json
    from naqs_amd import hamiltonian, packing
    from naqs_amd.hilbert import Encoding, Hilbert
    from test_nade import ELECTRONS
    N, na, nb = ELECTRONS[mol]
    hil = Hilbert.get(N, na, nb, encoding=Encoding.SIGNED)
    ham = hamiltonian.PauliHamiltonian.get(hil, packing.load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz")), device="cuda")
    keys = hamiltonian.keys_to_device(np.sort(hil._all_keys()).astype(np.int64), "cuda")
    val, vec = ham.lowest_eigenpair(keys)
    fci = json.load(open(os.path.join(GOLDEN, "kat.json")))["fci"][mol]
    assert abs(val - fci) < 1e-8, (val, fci)
    hv = ham.matvec(keys, vec)
    assert float((hv - val * vec).norm()) < 1e-6


@pytest.mark.parametrize("world", [2, 3, 8])
def test_eloc_from_the_gathered_table_equals_the_compact_call(env, world):
    """ABI 6, the sharded training step's second call: `naqs_eloc_gathered` reads (log|psi|, phase) straight from the
    all-gather's layout — `world` equal padded shards [world][S_pad][2], shard r = rows r S .. of the table — and returns
    my rows' E_loc, the weighted sums and the same-table proof.  Against `naqs_eloc_reduced` on the compacted table: the
    same E_loc bit for bit and the same four sums; ext8[4:8] = (M, M^2, c, c^2) with c = 20 low bits of the key sum."""
    import ctypes
    from naqs_amd.fused import _stream_ptr
    lib = env["lib"].load_library()
    z = golden("eloc_N2.npz")
    ham = dev_ham(env, "N2")
    keys_np = z["small_keys"]
    M = len(keys_np)
    rs = np.random.RandomState(5)
    lp = np.stack([rs.normal(-0.5 * np.log(M), 1.5, M), rs.uniform(0, 2 * np.pi, M)], -1).astype(np.float32)
    w = rs.uniform(0.1, 1.0, M)
    w /= w.sum()
    dev = ham.device
    keys = env["H"].keys_to_device(keys_np, dev)
    lp_d = torch.as_tensor(lp, device=dev)
    w_d = torch.as_tensor(w, dtype=torch.float64, device=dev)
    S = -(-M // world)
    S_pad = S + 37                                             # padded contributions, as sized from n_unq_samples_max
    table = torch.full((world, S_pad, 2), float("nan"), dtype=torch.float32, device=dev)     # padding must never be read
    for r in range(world):
        b, e = min(M, r * S), min(M, (r + 1) * S)
        table[r, :e - b] = lp_d[b:e]
    for rank in range(world):
        b, e = min(M, rank * S), min(M, (rank + 1) * S)
        e_ref, sums_ref = ham.local_energy(keys, lp_d, kind="log_psi", row_begin=b, n_rows=e - b, weights=w_d[b:e])
        eloc = torch.zeros((max(e - b, 1), 2), dtype=torch.float64, device=dev)
        ext = torch.zeros(8, dtype=torch.float64, device=dev)
        st = lib.naqs_eloc_gathered(ham._h, M, keys.data_ptr(), table.data_ptr(), S, S_pad, b, e - b,
                                    w_d[b:].data_ptr() if e > b else None, eloc.data_ptr(), ext.data_ptr(), _stream_ptr(dev))
        env["lib"].check(st, "naqs_eloc_gathered")
        torch.cuda.synchronize()
        assert torch.equal(eloc[:e - b], e_ref)
        assert torch.equal(ext[:4], sums_ref if e > b else torch.zeros(4, dtype=torch.float64, device=dev))
        c = float(int(keys_np.astype(np.uint64).sum(dtype=np.uint64)) & 0xFFFFF)
        assert ext[4:].tolist() == [float(M), float(M) ** 2, c, c * c]


def test_first_call_on_a_side_stream_of_a_busy_gpu(env):
    """A handle's allocation-time zero fills (hash table, training scratch) run on the null stream, which is not ordered
    against torch's non-blocking side streams: unless the library waits for them, a fill that is late because the GPU is
    busy wipes what the first call's kernels have already written (found as a wrong FIRST energy in one of ~80 runs of
    `experiments.run --farm --per-gpu 4`).  First call of a fresh handle on a side stream, under load from another stream,
    must be the answer of a quiet GPU."""
    z = golden("eloc_N2.npz")
    keys, psi = z["c2_keys"], z["c2_psi_f32"]
    quiet = run_eloc(env, dev_ham(env, "N2"), keys, psi)
    load = torch.randn(6144, 6144, device="cuda")
    busy, side = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(6):
        ham = dev_ham(env, "N2")                              # fresh handle: the first call allocates and zero-fills
        with torch.cuda.stream(busy):
            for _ in range(12):
                load @ load
        with torch.cuda.stream(side):
            got = run_eloc(env, ham, keys, psi)
        assert np.array_equal(got, quiet), rep
    torch.cuda.synchronize()
