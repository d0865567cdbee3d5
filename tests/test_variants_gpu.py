"""The ansatz variants the reference's scripts run, on the MI355X, against the reference's own vectors
(tests/golden/make_golden.py ``variants``; CPU counterpart: test_variants.py):

* ``-no_amp_sym`` / ``-no_mask_psi`` / ``-full_mask_psi`` stay inside the fused HIP family — the kernels' ``sym = 0``
  and masking branches run here (sampler, matrix-core log psi, training forward/backward);
* the run.py default (``aggregate_phase=True``: one phase block per orbital pair, 128 hidden units) is a second fused
  family: the per-pair phase blocks run through the amplitude kernels in raw mode (forward, training backward with
  128-unit blocks);  an ansatz outside both families falls back to the PyTorch modules — announced on stdout — with
  E_loc on the HIP kernels;
* config 5: E_loc at M = 10 000 and the FULL-masked network on the two end geometries of the N2 sweep, and a short
  training run of one geometry.
"""
import os

import numpy as np
import pytest
from scipy import stats

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FUSED = ["LiH_noampsym", "LiH_fullmask", "N2_noampsym", "N2_nomask", "N2_0.75_fullmask", "N2_2.25_fullmask",
         "LiH_aggphase", "N2_aggphase",
         # -phase_sym with the single phase block (round 5): spin-ordered inputs, 3 outputs, the sign shift — on the kernels
         "LiH_phasesym", "LiH_phasesym_agg",
         # open shell restricted to m_s = S (experiments/_base.py:101-123): CH2 triplet, 5 alpha / 3 beta electrons, no amp symmetry
         "CH2_noampsym", "CH2_fullmask_noampsym"]
EAGER = []


def _wf(fix):
    from test_nade import make_wf
    from test_variants import split
    mol = split(fix)[0]
    z = golden(f"nade_{fix}.npz")
    hil, wf = make_wf(mol, z, device="cuda")
    return mol, z, hil, wf


def _opt(mol, wf, tmp, **kw):
    from naqs_amd import packing
    from naqs_amd.optimizer import PartialSamplingOptimizer
    from test_optimizer import ADAM
    ham = packing.load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz"))
    na, nb = int(ham.n_alpha), int(ham.n_beta)
    args = dict(n_samples=100000, n_samples_max=1e12, n_unq_samples_min=10, n_unq_samples_max=1e5, log_exact_energy=False,
                wavefunction=wf, qubit_hamiltonian=ham, pre_compute_H=False, n_electrons=na + nb, n_alpha_electrons=na,
                n_beta_electrons=nb, normalise_psi=True, grad_clip_factor=None, optimizer=torch.optim.Adam,
                optimizer_args=[dict(a) for a in ADAM], save_loc=str(tmp), pauli_hamiltonian_dtype=np.float64, seed=5)
    args.update(kw)
    return PartialSamplingOptimizer(**args)


@pytest.mark.parametrize("fix", FUSED)
def test_fused_family_log_psi_matches_reference(fix):
    """keys -> log psi through the HIP kernels (amplitudes + phase MLP on the matrix cores) vs the reference's
    wavefunction.log_psi; un-physical-prefix handling and the masks are the variant's."""
    mol, z, hil, wf = _wf(fix)
    fused = wf.fused()
    assert fused is not None, "these variants are inside the fused family"
    for tag in ("eval", "samp"):
        keys = torch.as_tensor(z[f"{tag}_keys"].astype(np.int64), device="cuda")
        lp = fused.log_psi(keys).cpu().numpy()
        assert np.max(np.abs(lp - z[f"{tag}_log_psi"])) < 5e-5, tag
    # the torch modules on the device agree with the kernels
    s = torch.tensor(z["eval_states"], device="cuda")
    with torch.no_grad():
        ref = wf.log_psi(s).cpu().numpy()
    assert np.max(np.abs(ref - z["eval_log_psi"])) < 5e-5


@pytest.mark.parametrize("fix", FUSED + EAGER)
def test_sgd_step_matches_reference_step_on_device(fix, tmp_path, capsys):
    mol, z, hil, wf = _wf(fix)
    opt = _opt(mol, wf, tmp_path)
    from naqs_amd.flat_adam import FlatAdam
    assert wf.fused() is not None and isinstance(opt.optimizer, FlatAdam)
    assert wf.fused().aggregate == (fix.endswith("aggphase") or fix.endswith("phasesym_agg"))
    states = torch.tensor(z["samp_states"], device="cuda")
    counts = torch.tensor(z["samp_counts"], device="cuda")
    keys = hil.state2idx(states).squeeze(-1)
    lp = torch.tensor(z["samp_log_psi"], device="cuda")
    psi = torch.stack([lp[:, 0].exp() * lp[:, 1].cos(), lp[:, 0].exp() * lp[:, 1].sin()], -1)
    e = opt.calculate_local_energy(keys, psi=psi, ret_complex=True)            # HIP E_loc either way
    want = z["sgd_eloc_c128"]
    assert np.max(np.abs(e - want) / np.maximum(1, np.abs(want))) < 2e-5        # psi recomputed in float32 on device
    E, var = opt._SGD_step(states, keys, None, sample_weights=counts.double() / counts.sum().double())
    assert abs(E - float(z["sgd_E"])) < 2e-5 * max(1, abs(E))
    assert abs(var - float(z["sgd_Var"])) < 1e-3 * max(1, abs(var))
    for name, p in wf.model.named_parameters():
        d = np.abs(p.detach().cpu().numpy() - z["sd_after:" + name])
        # the first Adam step moves every parameter by lr * sign(g) (eps = 1e-15): where the reference's gradient is
        # itself rounding noise the sign — hence a 2 lr difference — is not determined.  A handful of such entries per
        # tensor (<= 1e-4 of it), each with a gradient below 1e-3 of the tensor's scale (float32 sums over ~10^4 samples of
        # mixed sign carry ~1e-4 of it as noise in either implementation); everything else agrees to 2e-5
        flipped = d >= 2e-5
        if flipped.any():
            g = np.abs(z["grad:" + name])
            assert flipped.sum() <= max(2, 1e-4 * d.size), (name, int(flipped.sum()))
            assert d[flipped].max() < 2.1e-3 and g[flipped].max() <= 1e-3 * g.max(), (name, d[flipped].max())


@pytest.mark.parametrize("fix", ["LiH_combampphase"])
def test_live_options_outside_the_fused_family_run_on_device(fix, tmp_path, capsys):
    """-comb_amp_phase (no published script uses it): PyTorch modules on the device, announced once, with
    the HIP E_loc kernel underneath — log psi and one whole _SGD_step against the reference's recorded vectors."""
    mol, z, hil, wf = _wf(fix)
    assert wf.fused() is None
    assert "fused HIP network kernels not available for this ansatz" in capsys.readouterr().out
    s = torch.tensor(z["eval_states"], device="cuda")
    with torch.no_grad():
        lp = wf.log_psi(s).cpu().numpy()
    assert np.max(np.abs(lp - z["eval_log_psi"])) < 5e-5
    opt = _opt(mol, wf, tmp_path)
    states = torch.tensor(z["samp_states"], device="cuda")
    counts = torch.tensor(z["samp_counts"], device="cuda")
    keys = hil.state2idx(states).squeeze(-1)
    E, var = opt._SGD_step(states, keys, None, sample_weights=counts.double() / counts.sum().double())
    assert abs(E - float(z["sgd_E"])) < 2e-5 * max(1, abs(E))
    assert abs(var - float(z["sgd_Var"])) < 1e-3 * max(1, abs(var))
    for name, p in wf.model.named_parameters():
        assert np.max(np.abs(p.detach().cpu().numpy() - z["sd_after:" + name])) < 2e-5, name
    opt.run(3, output_freq=10 ** 6)                       # and the training loop (PyTorch sampler, HIP E_loc) runs
    assert opt.n_steps == 3


@pytest.mark.parametrize("fix,fmt", [("LiH_phasesym", "2"), ("LiH_phasesym", "1"), ("LiH_phasesym", "0"), ("LiH_phasesym_agg", "2")])
def test_phase_spin_symmetry_on_the_kernels(fix, fmt, tmp_path, monkeypatch, capsys):
    """-phase_sym (nade.py:281, 507-533, 590-610) inside the fused families — the single phase block (the phase kernels) and the
    per-pair phase blocks of the run.py default (the amplitude kernels in raw mode, every block ordering ITS prefix): spin-ordered inputs
    (alpha / beta strings of the first P-1 pairs exchanged where idx(alpha) > idx(beta)), a 3-output layer whose middle row
    serves |01> and |10>, + pi (N_01 mod 2) where idx(alpha) < idx(beta).  log psi in all three number formats of the phase
    kernels and the training gradients against autograd through the PyTorch modules (themselves held to the reference's
    vectors in test_variants.py); then the whole loop — device sampler, one-call training steps — runs."""
    monkeypatch.setenv("NAQS_PHASE_MODE", fmt)
    mol, z, hil, wf = _wf(fix)
    agg = fix.endswith("_agg")
    fused = wf.fused()
    assert fused is not None and fused.phase_sym and fused.aggregate == agg
    assert all(blk.linears()[-1].out_features == 3 for blk in wf.model.phase_layers) and len(wf.model.phase_layers) == (wf.model.P if agg else 1)
    keys = torch.as_tensor(np.concatenate([z["eval_keys"], z["samp_keys"]]).astype(np.int64), device="cuda")
    states = hil.idx2state(keys)
    # the fixture's rows exercise every branch: inputs exchanged / not, shifted / not, all four outcomes of the last pair
    P = wf.model.P
    bits = (states.reshape(len(keys), -1) > 0).long()[:, wf._q2m]               # model order: pair k = (alpha, beta) at columns 2k, 2k + 1
    pw = (1 << torch.arange(P, device="cuda")).long()
    ia, ib = (bits[:, 0::2] * pw).sum(1), (bits[:, 1::2] * pw).sum(1)
    ia1, ib1 = (bits[:, 0:2 * (P - 1):2] * pw[:P - 1]).sum(1), (bits[:, 1:2 * (P - 1):2] * pw[:P - 1]).sum(1)
    n01 = ((bits[:, 0::2] == 0) & (bits[:, 1::2] == 1)).sum(1)
    assert (ia1 > ib1).any() and (ia1 < ib1).any() and ((ia < ib) & (n01 % 2 == 1)).any() and ((ia < ib) & (n01 % 2 == 0)).any()
    assert len(torch.unique(bits[:, 2 * (P - 1)] + 2 * bits[:, 2 * (P - 1) + 1])) >= 3
    lp_k = fused.log_psi(keys)
    lp_t = wf.log_psi(states).reshape(-1, 2)
    if not agg:
        assert fused.last_kernel().startswith({"2": "phase_kernel_h<", "1": "phase_kernel_h<", "0": "phase_kernel<"}[fmt])
    assert torch.max(torch.abs(lp_t.detach() - lp_k)).item() < 2e-5
    if fmt != "2":
        return
    gen = torch.Generator(device="cuda").manual_seed(3)
    g = torch.randn((len(keys), 2), device="cuda", generator=gen) / len(keys)
    lp_s, saved = fused.forward_saved(keys)
    assert torch.max(torch.abs(lp_s - lp_k)).item() < 1e-6
    for p in wf.model.parameters():
        p.grad = None
    fused.backward_saved(saved, g)
    mine = {n: p.grad.clone() for n, p in wf.model.named_parameters()}
    for p in wf.model.parameters():
        p.grad = None
    (g * lp_t).sum().backward()
    for n, p in wf.model.named_parameters():
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((mine[n] - p.grad).abs().max()) < 2e-5 * scale + 1e-10, n
    assert all(float(blk.linears()[-1].weight.grad.abs().max(1).values.min()) > 0 for blk in wf.model.phase_layers)     # all three rows learn
    for p in wf.model.parameters():
        p.grad = None
    opt = _opt(mol, wf, tmp_path)
    assert opt._can_onecall()
    opt.run(5, output_freq=10 ** 6, save_final=False)
    assert opt.n_steps == 5 and np.isfinite(opt.log[__import__("naqs_amd.optimizer", fromlist=["LogKey"]).LogKey.E_LOC][-1][1])
    assert "fused HIP network kernels not available" not in capsys.readouterr().out


@pytest.mark.parametrize("fix", ["LiH_aggphase", "N2_aggphase"])
def test_aggregate_phase_log_psi_on_device(fix):
    """The PyTorch modules on the device (conditionals of every block) and the HIP kernels' gradients against autograd."""
    mol, z, hil, wf = _wf(fix)
    s = torch.tensor(z["eval_states"], device="cuda")
    with torch.no_grad():
        lp = wf.log_psi(s).cpu().numpy()
        cond = wf._evaluate_log_psi(s, gather_state=False).cpu().numpy()
    ref = z["eval_cond"]
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(cond), finite)
    assert np.max(np.abs(cond[finite] - ref[finite])) < 2e-5 and np.max(np.abs(lp - z["eval_log_psi"])) < 5e-5
    assert np.abs(ref[..., :-1, :, 1]).max() > 0          # every block contributes a phase, not only the last
    # d/d theta sum_i (g0_i log|psi_i| + g1_i phase_i): naqs_net_train_backward (amplitude blocks + raw phase blocks, 128
    # hidden units -> the 128-sample-tile variant of the backward kernel) vs autograd through the modules
    fused = wf.fused()
    keys = torch.as_tensor(z["samp_keys"].astype(np.int64), device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(3)
    g = torch.randn((len(keys), 2), device="cuda", generator=gen)
    lp_k, saved = fused.forward_saved(keys)
    for p in wf.model.parameters():
        p.grad = None
    fused.backward_saved(saved, g)
    mine = {n: p.grad.clone() for n, p in wf.model.named_parameters()}
    for p in wf.model.parameters():
        p.grad = None
    lp_t = wf.log_psi(hil.idx2state(keys)).reshape(-1, 2)
    assert torch.max(torch.abs(lp_t.detach() - lp_k)).item() < 5e-5
    (g * lp_t).sum().backward()
    for n, p in wf.model.named_parameters():
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((mine[n] - p.grad).abs().max()) < 2e-4 * scale + 1e-9, n


def test_ansatz_outside_the_fused_families_falls_back_loudly(capsys):
    """Two phase hidden layers per block with aggregate_phase=True is in neither family: PyTorch modules on the device,
    announced on stdout (not silent), E_loc still on the HIP kernels."""
    from naqs_amd.hilbert import Encoding, Hilbert
    from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
    hil = Hilbert.get(12, 2, 2, encoding=Encoding.SIGNED)
    wf = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[32, 32],
                                   use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=True,
                                   n_alpha_electrons=2, n_beta_electrons=2, device="cuda")
    assert wf.fused() is None
    out = capsys.readouterr().out
    assert "fused HIP network kernels not available" in out and "aggregate_phase" in out
    states, counts, probs, lp = wf.sample(10000)                                   # the PyTorch sampler on the device
    assert lp.shape == (len(states), 2) and counts.sum().item() <= 10000
    wf2 = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, amp_hidden_size=[200], phase_hidden_size=[512, 512],
                                    use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=False,
                                    n_alpha_electrons=2, n_beta_electrons=2, device="cuda")
    assert wf2.fused() is None and "amplitude hidden width 200" in capsys.readouterr().out


@pytest.mark.parametrize("mol", ["N2_0.75", "N2_2.25"])
def test_eloc_on_sweep_geometries(mol):
    """config 5, E_loc at the headline batch size on the two end geometries (K = 2 239 ... 2 951 terms)."""
    from naqs_amd import hamiltonian, packing
    z = golden(f"eloc_{mol}.npz")
    ham = hamiltonian.DevicePauliHamiltonian(packing.load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz")), device="cuda:0")
    keys = hamiltonian.keys_to_device(z["c2_keys"], ham.device)
    ref = z["c2_eloc_c128"]
    for kind, arr in (("psi", z["c2_psi_f32"]), ("log_psi", z["c2_log_psi_f32"])):
        e = ham.local_energy(keys, torch.as_tensor(arr, device=ham.device), kind=kind).cpu().numpy()
        e = e[:, 0] + 1j * e[:, 1]
        tol = 1e-10 if kind == "psi" else 2e-5          # log psi -> psi re-evaluated in f64 on the device vs the reference's f32 exp
        assert np.max(np.abs(e - ref) / np.maximum(1, np.abs(ref))) < tol, kind
    w = (z["c2_psi_f32"].astype(np.float64) ** 2).sum(-1)
    psi = torch.as_tensor(z["c2_psi_f32"], device=ham.device)
    e, sums = ham.local_energy(keys, psi, kind="psi", weights=torch.as_tensor(w, device=ham.device))
    energy = (sums[0] / sums[3]).item()
    assert abs(energy - (w * ref.real).sum() / w.sum()) < 1e-9


def test_open_shell_eloc_matches_reference_golden():
    """CH2 (triplet: 5 alpha / 3 beta electrons): the reference's calculate_local_energy on 400 of the sector's 735 states."""
    from naqs_amd import hamiltonian, packing
    z = golden("eloc_CH2.npz")
    ham_p = packing.load_packed(os.path.join(GOLDEN, "ham_CH2.npz"))
    assert (ham_p.n_alpha, ham_p.n_beta) == (5, 3)
    ham = hamiltonian.DevicePauliHamiltonian(ham_p, device="cuda:0")
    keys = hamiltonian.keys_to_device(z["c1_keys"], ham.device)
    e = ham.local_energy(keys, torch.as_tensor(z["c1_psi_f32"], device=ham.device), kind="psi").cpu().numpy()
    e = e[:, 0] + 1j * e[:, 1]
    ref = z["c1_eloc_c128"]
    assert np.max(np.abs(e - ref) / np.maximum(1, np.abs(ref))) < 1e-10


@pytest.mark.parametrize("fix", ["CH2_noampsym", "CH2_fullmask_noampsym"])
def test_open_shell_sampler_matches_psi_squared(fix):
    """The tree sampler with n_alpha != n_beta electron budgets (PARTIAL and FULL masking): only states of the (5, 3)
    sector come out, in ascending key order, and their counts follow the exact |psi|^2 of the same network."""
    mol, z, hil, wf = _wf(fix)
    assert (hil.N_alpha, hil.N_beta) == (5, 3) and hil.size == 735
    fused = wf.fused()
    n = 2_000_000
    keys, counts, probs = fused.sample(n, seed=31, max_unique=100000)
    k, c = keys.cpu().numpy(), counts.cpu().numpy()
    assert np.all(np.diff(k) > 0) and hil.is_physical(k).all()
    all_keys = np.sort(hil._all_keys())
    with torch.no_grad():
        lp = wf.log_psi(hil.idx2state(torch.as_tensor(all_keys, device="cuda"))).reshape(-1, 2)
    p = np.exp(2.0 * lp[:, 0].double().cpu().numpy())
    pos = np.searchsorted(all_keys, k)
    assert np.allclose(probs.cpu().numpy(), p[pos], rtol=2e-4, atol=1e-12)
    total, p_phys = c.sum(), p.sum()
    if "fullmask" in fix:
        assert total == n and abs(p_phys - 1) < 1e-4                         # FULL masking: every draw is physical
    else:
        assert abs(total - n * p_phys) < 6 * np.sqrt(n * p_phys * (1 - p_phys)) + 1
    obs = np.zeros(len(all_keys))
    obs[pos] = c
    expect = p / p_phys * total
    m = expect >= 5
    chi2 = ((obs[m] - expect[m]) ** 2 / expect[m]).sum()
    assert stats.chi2.sf(chi2, m.sum()) > 1e-4, (chi2, m.sum())


def test_open_shell_cli_run(tmp_path, capsys):
    """`python -m experiments.run -m molecules/CH2 -single_phase ...`: the guard is gone — amplitude symmetry is switched
    off like the reference does (experiments/_base.py:109-114) and the run descends towards the sector's ground state."""
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    from experiments import _base
    res = _base.run(n_hid=64, n_samps=1e6, n_unq_samps_min=10, n_unq_samps_max=1e5,
                    argv=["-m", os.path.join(GOLDEN, "ham_CH2.npz"), "-o", str(tmp_path / "run"), "-single_phase", "-n_hid_phase", "64",
                          "-n_layer_phase", "2", "-n_train", "400", "-output_freq", "200", "-s", "111"])
    txt = capsys.readouterr().out
    assert "turning off use_amp_spin_sym" in txt and "fused HIP network kernels not available" not in txt
    r = res[0]
    assert r["fci"] is not None and r["final"] > r["fci"] - 1e-3 and r["final"] < r["fci"] + 0.15, r
    assert r["eig"] >= r["fci"] - 1e-7


def test_sampler_without_amp_symmetry_matches_psi_squared():
    """naqs_net_sample with sym = 0 (4 raw outputs per block, no spin ordering of the inputs): chi-square against the
    exact |psi|^2 of the same network over the whole LiH space."""
    mol, z, hil, wf = _wf("LiH_noampsym")
    fused = wf.fused()
    n = 2_000_000
    keys, counts, probs = fused.sample(n, seed=77, max_unique=100000)
    k, c = keys.cpu().numpy(), counts.cpu().numpy()
    assert np.all(np.diff(k) > 0) and hil.is_physical(k).all()
    all_keys = np.sort(hil._all_keys())
    with torch.no_grad():
        lp = wf.log_psi(hil.idx2state(torch.as_tensor(all_keys, device="cuda"))).reshape(-1, 2)
    p = np.exp(2.0 * lp[:, 0].double().cpu().numpy())                       # torch modules = independent of the kernels
    pos = np.searchsorted(all_keys, k)
    assert np.allclose(probs.cpu().numpy(), p[pos], rtol=2e-4, atol=1e-12)
    total, p_phys = c.sum(), p.sum()
    assert abs(total - n * p_phys) < 6 * np.sqrt(n * p_phys * (1 - p_phys)) + 1
    obs = np.zeros(len(all_keys))
    obs[pos] = c
    expect = p / p_phys * total
    m = expect >= 5
    chi2 = ((obs[m] - expect[m]) ** 2 / expect[m]).sum()
    assert stats.chi2.sf(chi2, m.sum()) > 1e-4, (chi2, m.sum())


def test_sweep_geometry_short_training_run_full_mask(tmp_path):
    """config 5 end to end on one GPU: N2 at 0.75 A with -full_mask_psi (batch_train_full_mask.sh:14), 300 steps of the
    published schedule's first phase: variational (never below FCI), clearly improving, every draw kept."""
    from naqs_amd.optimizer import LogKey
    mol, z, hil, wf = _wf("N2_0.75_fullmask")
    fci = float(golden("ham_N2_0.75.npz")["fci_energy"])
    opt = _opt(mol, wf, tmp_path, n_samples=1000000, n_unq_samples_min=1000,
               optimizer_args=[{'lr': 2e-3, 'betas': (0.9, 0.99), 'eps': 1e-15}, {'lr': 1e-2}])
    states, counts, _ = opt.get_samples()
    assert counts.sum().item() == opt.n_samples                  # FULL masking: no un-physical draw to discard
    opt.run(n_epochs=300, save_freq=None, save_final=False, output_freq=100)
    e = np.array([x[1] for x in opt.log[LogKey.E_LOC]])
    assert np.mean(e[-10:]) < np.mean(e[:10]) - 1.0
    assert np.mean(e[-10:]) > fci - 1e-3, (np.mean(e[-10:]), fci)


def test_default_ansatz_cli_run_on_device(tmp_path, capsys):
    """`python -m experiments.run -m molecules/LiH` with the reference's own defaults for the ansatz (run.py:11-31:
    aggregate phase, 128 hidden units, amplitude symmetry) end to end on the HIP path: sampler, fused forward with the
    per-pair phase blocks, E_loc, HIP backward of both sets of blocks, FlatAdam, checkpoints, summary."""
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    from experiments import _base
    from naqs_amd.flat_adam import FlatAdam
    out = str(tmp_path / "run")
    lih = os.path.join(GOLDEN, "molecules", "LiH")
    made = {}
    real = _base.PartialSamplingOptimizer

    class Spy(real):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            made["opt"] = self

    _base.PartialSamplingOptimizer = Spy
    try:
        res = _base.run(n_hid=128, n_samps=1e6, n_unq_samps_min=10, n_unq_samps_max=1e5,
                        argv=["-m", lih, "-o", out, "-n_train", "200", "-output_freq", "100", "-s", "111"])
    finally:
        _base.PartialSamplingOptimizer = real
    txt = capsys.readouterr().out
    assert "fused HIP network kernels not available" not in txt          # the default ansatz is a fused family
    opt = made["opt"]
    assert opt.wavefunction.model.aggregate_phase and opt.wavefunction.fused().aggregate
    assert isinstance(opt.optimizer, FlatAdam)
    r = res[0]
    assert r["final"] < -7.0 and r["final"] > r["fci"] - 1e-3             # 200 steps from random init (-2 Ha): variational, most of the way
    assert r["eig"] >= r["fci"] - 1e-8
    assert os.path.exists(os.path.join(out, "summary.txt")) and os.path.exists(os.path.join(out, "energy_optimizer.pth"))


def test_sampler_with_wide_blocks_matches_psi_squared():
    """128 hidden units per amplitude block (the reference's default -n_hid): the tree sampler's quads take 32 units each."""
    mol, z, hil, wf = _wf("LiH_aggphase")
    fused = wf.fused()
    n = 2_000_000
    keys, counts, probs = fused.sample(n, seed=9, max_unique=100000)
    k, c = keys.cpu().numpy(), counts.cpu().numpy()
    all_keys = np.sort(hil._all_keys())
    with torch.no_grad():
        lp = wf.log_psi(hil.idx2state(torch.as_tensor(all_keys, device="cuda"))).reshape(-1, 2)
    p = np.exp(2.0 * lp[:, 0].double().cpu().numpy())
    pos = np.searchsorted(all_keys, k)
    assert np.allclose(probs.cpu().numpy(), p[pos], rtol=2e-4, atol=1e-12)
    obs = np.zeros(len(all_keys))
    obs[pos] = c
    expect = p / p.sum() * c.sum()
    m = expect >= 5
    chi2 = ((obs[m] - expect[m]) ** 2 / expect[m]).sum()
    assert stats.chi2.sf(chi2, m.sum()) > 1e-4, (chi2, m.sum())


@pytest.mark.parametrize("fix", ["LiH_aggphase", "N2_aggphase"])
def test_aggregate_phase_merged_launches_equal_separate_ones(fix, monkeypatch):
    """aggregate_phase networks run their two sets of per-pair blocks (amplitude, phase) as ONE launch each way (forward,
    backward, re-pack): same device functions on the same operands -> log psi and every gradient element bit for bit."""
    mol, z, hil, wf = _wf(fix)
    keys = torch.as_tensor(z["samp_keys"].astype(np.int64), device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(3)
    g = torch.randn((len(keys), 2), device="cuda", generator=gen) / len(keys)
    out = {}
    for mode in ("7", "0"):
        monkeypatch.setenv("NAQS_AGG_MERGE", mode)
        fused = wf.fused()
        assert fused.aggregate
        fused.refresh()
        for p in wf.model.parameters():
            p.grad = None
        fused._grad_flat = None
        lp = fused.log_psi(keys).clone()
        lp_t, saved = fused.forward_saved(keys)
        fused.backward_saved(saved, g)
        torch.cuda.synchronize()
        out[mode] = (lp, lp_t.clone(), torch.cat([p.grad.reshape(-1) for p in wf.model.parameters()]).clone())
    for a, b in zip(out["7"], out["0"]):
        assert torch.equal(a, b)
    assert float(out["7"][2].abs().max()) > 0
