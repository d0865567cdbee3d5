"""The ansatz variants the reference's scripts actually run, against vectors recorded from the reference
(tests/golden/make_golden.py `variants`): the run.py default with one phase block per orbital pair
(``aggregate_phase=True``, experiments/run.py:31, nade.py:556-569), ``-no_amp_sym``
(batch_train_no_amp_sym.sh:14), ``-no_mask_psi`` (batch_train_no_mask.sh:14) and ``-full_mask_psi`` on two
geometries of the N2 sweep (N2_energy_surface.sh:5-8 -> batch_train_full_mask.sh:14).  CPU: the torch modules +
host logic (E_loc by the oracle-backed stand-in); the same fixtures drive the HIP path in test_variants_gpu.py."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden
from naqs_amd import packing
from test_nade import ELECTRONS, make_wf

VARIANT_FIXTURES = ["LiH_aggphase", "LiH_noampsym", "LiH_fullmask", "N2_aggphase", "N2_noampsym", "N2_nomask",
                    "N2_0.75_fullmask", "N2_2.25_fullmask",
                    # open shell restricted to m_s = S (experiments/_base.py:101-123): CH2 triplet, 5 alpha / 3 beta electrons
                    "CH2_noampsym", "CH2_fullmask_noampsym"]
TAGS = ["fullmask_noampsym", "aggphase", "noampsym", "nomask", "fullmask", "phasesym_agg", "phasesym", "combampphase"]
# round 4: the live options no published script uses — -phase_sym (nade.py:281, 593-610) and -comb_amp_phase
# (nade.py:257-262, 294-303) — as PyTorch modules, against the reference's own vectors (make_golden.py `widen`)
WIDEN_FIXTURES = ["LiH_phasesym", "LiH_phasesym_agg", "LiH_combampphase"]
ADAM = [{'lr': 1e-3, 'betas': (0.9, 0.99), 'weight_decay': 0, 'eps': 1e-15, 'amsgrad': False}, {'lr': 1e-2}]


def split(fix):
    for tag in TAGS:
        if fix.endswith("_" + tag):
            return fix[:-len(tag) - 1], tag
    raise ValueError(fix)


@pytest.mark.parametrize("fix", VARIANT_FIXTURES + WIDEN_FIXTURES)
def test_variant_log_psi_matches_reference(fix):
    mol, tag = split(fix)
    z = golden(f"nade_{fix}.npz")
    hil, wf = make_wf(mol, z)
    agg = tag in ("aggphase", "phasesym_agg", "combampphase")
    assert wf.model.aggregate_phase == agg and wf.model.use_amp_spin_sym == ("noampsym" not in tag)
    assert len(wf.model.phase_layers) == (0 if tag == "combampphase" else (wf.model.P if agg else 1))
    if fix in WIDEN_FIXTURES:
        assert wf.model.use_phase_spin_sym and wf.model._n_out_phase == 3      # (-comb_amp_phase forces it to follow the amplitude's)
    s = torch.tensor(z["eval_states"])
    with torch.no_grad():
        cond = wf._evaluate_log_psi(s, gather_state=False).numpy()
        lp = wf.log_psi(s).numpy()
    ref = z["eval_cond"]
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(cond), finite)
    if "fullmask" in tag:
        assert (~finite).any()
    if tag == "nomask":
        assert finite.all()
    assert np.max(np.abs(cond[finite] - ref[finite])) < 2e-5
    assert np.max(np.abs(lp - z["eval_log_psi"])) < 5e-5
    # the reference's own sampled states (its sampler's accumulated log psi, nade.py:714-723)
    ss = torch.tensor(z["samp_states"])
    with torch.no_grad():
        lps = wf.log_psi(ss).numpy()
    assert np.max(np.abs(lps - z["samp_log_psi"])) < 5e-5


@pytest.mark.parametrize("fix", ["LiH_aggphase", "LiH_noampsym", "LiH_fullmask", "N2_aggphase", "N2_noampsym",
                                 "CH2_noampsym", "CH2_fullmask_noampsym"] + WIDEN_FIXTURES)
def test_variant_sgd_step_matches_reference_step(fix, tmp_path, monkeypatch):
    """energy, variance, loss, every gradient and every parameter after the reference's own _SGD_step."""
    import oracle_backend
    from naqs_amd.optimizer import PartialSamplingOptimizer
    mol, tag = split(fix)
    oracle_backend.install(monkeypatch)
    z = golden(f"nade_{fix}.npz")
    hil, wf = make_wf(mol, z)
    N, na, nb = ELECTRONS["N2" if mol.startswith("N2") else mol]
    ham = packing.load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz"))
    opt = PartialSamplingOptimizer(
        n_samples=2000, n_samples_max=1e12, n_unq_samples_min=10, n_unq_samples_max=1e5, log_exact_energy=False,
        wavefunction=wf, qubit_hamiltonian=ham, pre_compute_H=False, n_electrons=na + nb, n_alpha_electrons=na,
        n_beta_electrons=nb, normalise_psi=True, grad_clip_factor=None, optimizer=torch.optim.Adam,
        optimizer_args=[dict(a) for a in ADAM], save_loc=str(tmp_path), pauli_hamiltonian_dtype=np.float64, seed=3)
    states = torch.tensor(z["samp_states"])
    counts = torch.tensor(z["samp_counts"])
    keys = hil.state2idx(states).squeeze(-1)
    grads = {}
    real_step = opt.optimizer.step

    def spy(*a, **k):
        grads.update({n: p.grad.detach().clone().numpy() for n, p in wf.model.named_parameters()})
        return real_step(*a, **k)

    opt.optimizer.step = spy
    E, var = opt._SGD_step(states, keys, None, sample_weights=counts.double() / counts.sum().double())
    assert abs(E - float(z["sgd_E"])) < 2e-5 * max(1, abs(E))            # reference rounds E_loc to float32
    assert abs(var - float(z["sgd_Var"])) < 1e-3 * max(1, abs(var))
    assert abs(opt.last_loss.item() - float(z["sgd_loss"])) < 2e-4 * max(1, abs(float(z["sgd_loss"])))
    for name, p in wf.model.named_parameters():
        g_ref = z["grad:" + name]
        assert np.max(np.abs(grads[name] - g_ref)) < 2e-3 * max(1e-3, np.abs(g_ref).max()), name
        assert np.max(np.abs(p.detach().numpy() - z["sd_after:" + name])) < 2e-5, name


@pytest.mark.parametrize("mol", ["N2_0.75", "N2_2.25"])
def test_oracle_on_sweep_geometry_eloc(mol):
    """config 5: E_loc at M = 10 000 on the two end geometries of the sweep, oracle vs the reference's complex128."""
    from oracle import oracle
    z = golden(f"eloc_{mol}.npz")
    ham = packing.load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz"))
    psi = z["c2_psi_f32"].astype(np.float64)
    e = oracle.eloc_matrix_free(ham.xy, ham.yz, ham.coeff, z["c2_keys"], psi[:, 0] + 1j * psi[:, 1])
    ref = z["c2_eloc_c128"]
    assert np.max(np.abs(e - ref) / np.maximum(1, np.abs(ref))) < 1e-11


def test_variant_sampler_statistics_no_amp_sym():
    """The PyTorch sampler without the amplitude symmetry: probs == |psi|^2 and the kept fraction of the draws."""
    z = golden("nade_LiH_noampsym.npz")
    hil, wf = make_wf("LiH", z)
    g = torch.Generator().manual_seed(5)
    n = 200000
    states, counts, probs, lp = wf.sample(n, generator=g)
    assert np.allclose(probs.numpy(), lp[:, 0].detach().exp().pow(2).numpy(), rtol=2e-4, atol=1e-9)
    with torch.no_grad():
        p_all = wf.log_psi(hil.get_subspace(ret_states=True))[:, 0].exp().pow(2).double().sum().item()
    kept = counts.sum().item()
    assert abs(kept / n - p_all) < 5 * np.sqrt(max(p_all * (1 - p_all), 1e-9) / n) + 1e-3


def test_pre_flatten_follows_the_reference(tmp_path, monkeypatch, capsys):
    """-n_pretrain n = OptimizerBase.pre_flatten as experiments/_base.py:284-289 calls it (energy.py:840-904): n supervised
    epochs towards the uniform amplitude over the restricted space.  Parameters after three epochs against the reference's
    (one batch: the order torch.randperm visits the states in does not enter the mean-squared error beyond rounding)."""
    import oracle_backend
    from naqs_amd.optimizer import PartialSamplingOptimizer
    oracle_backend.install(monkeypatch)
    z = golden("pretrain_LiH.npz")
    hil, wf = make_wf("LiH", z)
    N, na, nb = ELECTRONS["LiH"]
    ham = packing.load_packed(os.path.join(GOLDEN, "ham_LiH.npz"))
    opt = PartialSamplingOptimizer(
        n_samples=1000, n_samples_max=1e12, n_unq_samples_min=10, n_unq_samples_max=1e5, log_exact_energy=False,
        wavefunction=wf, qubit_hamiltonian=ham, pre_compute_H=False, n_electrons=na + nb, n_alpha_electrons=na,
        n_beta_electrons=nb, normalise_psi=True, grad_clip_factor=None, optimizer=torch.optim.Adam,
        optimizer_args=[dict(a) for a in ADAM], save_loc=str(tmp_path), pauli_hamiltonian_dtype=np.float64, seed=3)
    opt.pre_flatten(int(z["n_epochs"]), 1000, optimizer_args={'lr': 1e-3}, output_freq=25, use_sampling=False,
                    max_batch_size=550000, flatten_phase=False)
    out = capsys.readouterr().out
    assert "Pre-flattening NAQS amplitudes...using 1 batch(es) of size of at most 550000." in out and "Epoch 1 : loss = " in out
    for name, p in wf.model.named_parameters():
        assert np.max(np.abs(p.detach().numpy() - z["sd_after:" + name])) < 2e-6, name
    with torch.no_grad():
        la = wf.log_psi(hil.get_subspace(ret_states=True))[..., 0].numpy()
    assert np.max(np.abs(la - z["log_amp_after"])) < 2e-5
    opt.pre_flatten(0)                                                    # -n_pretrain 0 (every published script): nothing to do
    with pytest.raises(NotImplementedError):
        opt.pre_flatten(1, use_sampling=True)                             # the reference's own sampling branch cannot run
