"""Host logic of the ansatz (runs on CPU here; the same torch modules run on the GPU):
teacher-forced log psi against the reference's outputs, and the on-device tree sampler against
exact probabilities (the numpy RNG stream of the reference cannot be reproduced -> statistics)."""
import numpy as np
import pytest
import torch

from conftest import golden
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.nade import NadeMasking
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals

ELECTRONS = {"LiH": (12, 2, 2), "H2O": (14, 5, 5), "N2": (20, 7, 7), "CH2": (14, 5, 3)}   # CH2: triplet restricted to m_s = S


def make_wf(mol, z, device="cpu", masking=None):
    """The ansatz a ``nade_*.npz`` fixture was recorded with (variant fixtures carry masking / aggregate_phase /
    use_amp_spin_sym; the base ones are the published single-phase, spin-symmetric, PARTIAL-masked network)."""
    N, na, nb = ELECTRONS["N2" if mol.startswith("N2") else mol]
    hil = Hilbert.get(N, na, nb, encoding=Encoding.SIGNED, make_basis=True)
    if masking is None:
        masking = NadeMasking(int(z["cfg_masking"])) if "cfg_masking" in z.files else NadeMasking.PARTIAL
    agg = bool(z["cfg_aggregate_phase"]) if "cfg_aggregate_phase" in z.files else False
    sym = bool(z["cfg_use_amp_spin_sym"]) if "cfg_use_amp_spin_sym" in z.files else True
    psym = bool(z["cfg_use_phase_spin_sym"]) if "cfg_use_phase_spin_sym" in z.files else False
    comb = bool(z["cfg_combined_amp_phase_blocks"]) if "cfg_combined_amp_phase_blocks" in z.files else False
    wf = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, masking=masking,
                                   amp_hidden_size=[int(z["cfg_n_hid"])],
                                   phase_hidden_size=[int(z["cfg_n_hid_phase"])] * int(z["cfg_n_layer_phase"]),
                                   use_amp_spin_sym=sym, use_phase_spin_sym=psym, aggregate_phase=agg,
                                   combined_amp_phase_blocks=comb,
                                   n_alpha_electrons=na, n_beta_electrons=nb, device=device)
    sd = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd:")}
    assert set(sd) == set(wf.model.state_dict()), "state_dict keys must match the reference's"
    wf.model.load_state_dict(sd)
    return hil, wf


@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2"])
def test_log_psi_matches_reference(mol):
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z)
    s = torch.tensor(z["eval_states"])
    with torch.no_grad():
        cond = wf._evaluate_log_psi(s, gather_state=False).numpy()
        lp = wf.log_psi(s).numpy()
    ref_cond = z["eval_cond"]
    finite = np.isfinite(ref_cond)
    assert np.array_equal(np.isfinite(cond), finite)               # same -inf pattern (masks)
    assert np.max(np.abs(cond[finite] - ref_cond[finite])) < 2e-5
    assert np.max(np.abs(lp - z["eval_log_psi"])) < 5e-5            # float32 network, different GEMM order
    # key <-> state round trip with the reference's convention
    assert np.array_equal(hil.state2idx(s).squeeze().numpy().astype(np.uint64), z["eval_keys"])
    assert np.array_equal(hil.idx2state(z["eval_keys"].astype(np.int64)).numpy(), z["eval_states"])


@pytest.mark.parametrize("mol", ["LiH", "H2O"])
def test_log_psi_on_reference_samples_and_sgd_gradient(mol):
    """log psi of the reference's own sampled states, then the VMC loss of _SGD_step
    (energy.py:328-329) with the reference's E_loc: gradients must match the reference's."""
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z)
    s = torch.tensor(z["samp_states"])
    lp = wf.log_psi(s)
    assert np.max(np.abs(lp.detach().numpy() - z["samp_log_psi"])) < 5e-5
    w = torch.tensor(z["samp_counts"], dtype=torch.float32)
    w = (w / w.sum()).unsqueeze(-1)
    e = torch.tensor(z["sgd_eloc_f32"])
    e_corr = e - (w * e).sum(0)
    re = lp[:, 0] * e_corr[:, 0] - lp[:, 1] * e_corr[:, 1]
    loss = 2 * (w.squeeze() * re).sum()
    assert abs(loss.item() - float(z["sgd_loss"])) < 1e-4 * max(1, abs(float(z["sgd_loss"])))
    loss.backward()
    for name, p in wf.model.named_parameters():
        g_ref = z["grad:" + name]
        scale = max(1e-3, np.abs(g_ref).max())
        assert np.max(np.abs(p.grad.numpy() - g_ref)) < 2e-3 * scale, name


def test_masking_modes():
    z = golden("nade_LiH.npz")
    s = torch.tensor(z["eval_states"])
    _, wf_full = make_wf("LiH", z, masking=NadeMasking.FULL)
    _, wf_none = make_wf("LiH", z, masking=NadeMasking.NONE)
    with torch.no_grad():
        full = wf_full._evaluate_log_psi(s, gather_state=False)[..., 0]
        none = wf_none._evaluate_log_psi(s, gather_state=False)[..., 0]
    assert torch.isfinite(none).all()
    # FULL masking: the 4 conditional probabilities of every block still sum to one over allowed outcomes
    assert torch.allclose(full.exp().pow(2).sum(-1), torch.ones(full.shape[:2]), atol=1e-5)
    assert torch.isinf(full).any()


@pytest.mark.parametrize("mol", ["LiH", "H2O"])
def test_sampler_statistics(mol):
    """Tree sampler: unique sorted physical states, counts add up (minus discarded unphysical draws),
    returned probs == |psi|^2, and empirical frequencies follow |psi|^2 (chi-square-ish bound)."""
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z)
    g = torch.Generator().manual_seed(7)
    n = 400000
    states, counts, probs, lp = wf.sample(n, generator=g)
    keys = hil.state2idx(states).squeeze().numpy().astype(np.int64)
    assert np.all(np.diff(keys) > 0)                                  # unique and ascending (qubit_ordering=-1)
    assert hil.is_physical(keys).all()
    assert counts.dtype == torch.int64 and 0 < counts.sum().item() <= n
    p_model = lp[:, 0].detach().exp().pow(2).numpy()
    assert np.allclose(probs.numpy(), p_model, rtol=2e-4, atol=1e-9)
    # exact distribution over the whole restricted space (PARTIAL masking leaks a little mass outside)
    with torch.no_grad():
        all_states, all_keys = hil.get_subspace(ret_states=True, ret_idxs=True)
        p_all = wf.log_psi(all_states)[:, 0].exp().pow(2).numpy().astype(np.float64)
    kept = counts.sum().item()
    assert abs(kept / n - p_all.sum()) < 5 * np.sqrt(p_all.sum() * (1 - p_all.sum()) / n) + 1e-3
    freq = np.zeros(len(p_all))
    pos = np.searchsorted(all_keys.numpy().astype(np.int64), keys)
    order = np.argsort(all_keys.numpy().astype(np.int64))
    freq[order[np.searchsorted(all_keys.numpy().astype(np.int64)[order], keys)]] = counts.numpy()
    expect = p_all * n
    big = expect > 50
    zscore = (freq[big] - expect[big]) / np.sqrt(expect[big])
    assert np.abs(zscore).max() < 6 and abs(zscore.mean()) < 0.5


def test_sampler_max_batch_size():
    from naqs_amd.nade import MaxBatchSizeExceededError
    z = golden("nade_LiH.npz")
    _, wf = make_wf("LiH", z)
    with pytest.raises(MaxBatchSizeExceededError):
        wf.sample(100000, max_batch_size=5)


def test_checkpoint_roundtrip(tmp_path):
    z = golden("nade_LiH.npz")
    hil, wf = make_wf("LiH", z)
    f = wf.save(str(tmp_path / "wf"), quiet=True)
    ck = torch.load(f, weights_only=False)
    assert set(ck) == {"model:state_dict", "wavefunction:permute_qubits", "wavefunction:qubit2model_permutation",
                       "wavefunction:model2qubit_permutation"}
    _, wf2 = make_wf("LiH", z)
    for p in wf2.model.parameters():
        p.data.zero_()
    wf2.load(f)
    s = torch.tensor(z["eval_states"][:16])
    with torch.no_grad():
        assert torch.equal(wf.log_psi(s), wf2.log_psi(s))


def test_hilbert_rank_matches_reference_enumeration():
    hil = Hilbert.get(12, 2, 2, encoding=Encoding.SIGNED)
    keys = hil.restricted2full_idx(np.arange(hil.size))
    assert hil.size == 225 and len(np.unique(keys)) == 225 and hil.is_physical(keys).all()
    assert np.array_equal(hil.full2restricted_idx(keys), np.arange(225))
    assert hil.full2restricted_idx(np.array([0, 1, 3]))[0] == -1
    # first states of product(combinations(alpha), combinations(beta)) (hilbert.py:446-469)
    assert keys[:3].tolist() == [0b1111, 0b100111, 0b10000111]
    assert hil.get_idx_dtype("np") == np.int16 and Hilbert.get(20, 7, 7).get_idx_dtype("np") == np.int32
    assert Hilbert.get(30, 7, 7).get_idx_dtype("np") == np.int64
