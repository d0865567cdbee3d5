"""The ansatz on the GPU: same checks as tests/test_nade.py on cuda, plus the full hot path
keys -> log psi -> E_loc against the oracle."""
import numpy as np
import pytest

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2"])
def test_log_psi_matches_reference_on_device(mol):
    from test_nade import make_wf
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z, device="cuda")
    s = torch.tensor(z["eval_states"], device="cuda")
    with torch.no_grad():
        lp = wf.log_psi(s)
    assert lp.is_cuda
    assert np.max(np.abs(lp.cpu().numpy() - z["eval_log_psi"])) < 5e-5


def test_sampler_on_device_and_full_hot_path():
    import os
    from test_nade import make_wf
    from naqs_amd import hamiltonian, packing
    from oracle import oracle
    z = golden("nade_H2O.npz")
    hil, wf = make_wf("H2O", z, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    states, counts, probs, lp = wf.sample(200000, generator=g)
    assert states.is_cuda and lp.requires_grad and counts.sum().item() <= 200000
    keys = hil.state2idx(states).squeeze()
    k_np = keys.cpu().numpy().astype(np.int64)
    assert np.all(np.diff(k_np) > 0) and hil.is_physical(k_np).all()
    assert torch.allclose(probs, lp[:, 0].detach().exp().pow(2), rtol=2e-4, atol=1e-9)
    hp = packing.load_packed(os.path.join(GOLDEN, "ham_H2O.npz"))
    ham = hamiltonian.DevicePauliHamiltonian(hp)
    e = ham.local_energy(hamiltonian.keys_to_device(keys, ham.device), lp.detach(), kind="log_psi")
    torch.cuda.synchronize()
    e = e.cpu().numpy()
    lp64 = lp.detach().cpu().numpy().astype(np.float64)
    psi = np.exp(lp64[:, 0] + 1j * lp64[:, 1])
    want = oracle.eloc_matrix_free(hp.xy, hp.yz, hp.coeff, k_np.astype(np.uint64), psi)
    got = e[:, 0] + 1j * e[:, 1]
    assert np.max(np.abs(got - want) / np.maximum(1, np.abs(want))) < 1e-9


@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2"])
def test_fused_log_psi_matches_reference_and_torch(mol):
    """naqs_net_logpsi (amp_kernel + MFMA phase_kernel) vs the reference's log psi and vs the PyTorch
    modules on the same weights: float32 network, different summation order -> 5e-5 absolute."""
    import os
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.hamiltonian import keys_to_device
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z, device="cuda")
    fused = FusedLogPsi(wf)
    keys = keys_to_device(z["eval_keys"], wf.device)
    lp = fused.log_psi(keys)
    torch.cuda.synchronize()
    assert np.max(np.abs(lp.cpu().numpy() - z["eval_log_psi"])) < 5e-5
    with torch.no_grad():
        lp_t = wf.log_psi(torch.tensor(z["eval_states"], device="cuda"))
    assert torch.max(torch.abs(lp - lp_t)).item() < 2e-5
    # every tile height of the MFMA kernel, and a batch that is not a multiple of the tile
    for rb in ("1", "2", "3", "4"):
        os.environ["NAQS_PHASE_RB"] = rb
        try:
            lp_rb = fused.log_psi(keys[:-3])
            torch.cuda.synchronize()
        finally:
            del os.environ["NAQS_PHASE_RB"]
        assert torch.max(torch.abs(lp_rb - lp_t[:-3])).item() < 2e-5, rb


@pytest.mark.parametrize("mol", ["LiH", "N2"])
def test_amplitude_kernel_forms_agree(mol):
    """The amplitude conditionals exist in three forms: inside the phase kernel's prologue (default, NAQS_AMP_MODE=1),
    the same matrix-core items as a kernel of their own (=2: what naqs_net_logamp and the aggregate-phase family use) and
    the VALU amp_kernel (=0).  1 and 2 run identical arithmetic per (tile, pair) item -> identical log|psi|; 0 is exact
    f32 FMA chains -> 2e-5; the phase column does not depend on the form at all."""
    import os
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.hamiltonian import keys_to_device
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z, device="cuda")
    fused = FusedLogPsi(wf)
    keys = keys_to_device(z["eval_keys"], wf.device)
    res = {}
    for mode in ("1", "2", "0"):
        os.environ["NAQS_AMP_MODE"] = mode
        try:
            res[mode] = fused.log_psi(keys).clone()
            torch.cuda.synchronize()
        finally:
            del os.environ["NAQS_AMP_MODE"]
    assert torch.equal(res["1"], res["2"])
    assert torch.max(torch.abs(res["1"] - res["0"])).item() < 2e-5
    assert torch.equal(res["1"][:, 1], res["0"][:, 1])
    assert np.max(np.abs(res["2"].cpu().numpy() - z["eval_log_psi"])) < 5e-5


def test_fused_log_psi_unphysical_and_masking_modes():
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.hamiltonian import keys_to_device
    from naqs_amd.nade import NadeMasking
    z = golden("nade_LiH.npz")
    for masking in (NadeMasking.FULL, NadeMasking.NONE, NadeMasking.PARTIAL):
        hil, wf = make_wf("LiH", z, device="cuda", masking=masking)
        fused = FusedLogPsi(wf)
        keys_np = np.r_[z["eval_keys"][:50].astype(np.int64), np.array([0b111111, 0b1, 0], np.int64)]  # last 3 unphysical
        keys = keys_to_device(keys_np, wf.device)
        lp = fused.log_psi(keys).cpu().numpy()
        with torch.no_grad():
            lp_t = wf.log_psi(hil.idx2state(torch.tensor(keys_np, device="cuda"))).cpu().numpy()
        same_inf = np.array_equal(np.isinf(lp[:, 0]), np.isinf(lp_t[:, 0]))
        fin = np.isfinite(lp_t[:, 0])
        assert same_inf and np.max(np.abs(lp[fin] - lp_t[fin])) < 2e-5, masking


def test_fused_refresh_after_parameter_update():
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.hamiltonian import keys_to_device
    z = golden("nade_H2O.npz")
    hil, wf = make_wf("H2O", z, device="cuda")
    fused = FusedLogPsi(wf)
    keys = keys_to_device(z["eval_keys"], wf.device)
    before = fused.log_psi(keys).clone()
    with torch.no_grad():
        for p in wf.model.parameters():
            p.add_(0.01 * torch.randn_like(p))
    fused.refresh()
    after = fused.log_psi(keys)
    with torch.no_grad():
        want = wf.log_psi(torch.tensor(z["eval_states"], device="cuda"))
    assert torch.max(torch.abs(after - want)).item() < 2e-5
    assert torch.max(torch.abs(after - before)).item() > 1e-3


def _phase_f64_reference(hil, wf, keys_np, P):
    states = hil.idx2state(torch.tensor(keys_np)).double()
    x = states[:, wf.qubit2model_permutation]
    h = torch.cat([x[:, 0:2 * (P - 1):2], x[:, 1:2 * (P - 1):2]], 1)
    lins = [l for l in wf.model.phase_layers[0].linears()]
    for i, lin in enumerate(lins):
        h = h @ lin.weight.detach().double().cpu().T + lin.bias.detach().double().cpu()
        if i + 1 < len(lins):
            h = torch.relu(h)
    occ = ((x[:, 2 * (P - 1)] > 0).long() + 2 * (x[:, 2 * (P - 1) + 1] > 0).long())
    return h.gather(1, occ.view(-1, 1)).squeeze(1).numpy()


def _phase_errors(fused, keys, ref, modes):
    import os
    errs = {}
    for mode in modes:
        os.environ["NAQS_PHASE_MODE"] = mode
        try:
            fused.refresh()                       # each mode packs its own weight format
            lp = fused.log_psi(keys)
            torch.cuda.synchronize()
        finally:
            del os.environ["NAQS_PHASE_MODE"]
        errs[mode] = np.abs(lp[:, 1].cpu().numpy().astype(np.float64) - ref)
    fused.refresh()
    return errs


def test_split_phase_kernels_are_f32_equivalent():
    """The phase MLP runs on the 16-bit matrix cores with every f32 operand split exactly enough: mode 1 = three bf16
    planes (six cross products per multiply), mode 2 (default) = two scaled f16 planes (three), f32 accumulation.
    Against a float64 evaluation of the same network their error must be of the same size as the exact-f32 MFMA
    kernel's (NAQS_PHASE_MODE=0)."""
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.hamiltonian import keys_to_device
    z = golden("nade_N2.npz")
    hil, wf = make_wf("N2", z, device="cuda")
    with torch.no_grad():                       # make the phase outputs O(10) so that relative errors are visible
        for p in wf.model.phase_layers.parameters():
            p.mul_(1.5)
    fused = FusedLogPsi(wf)
    keys_np = z["samp_keys"][:4096].astype(np.int64)
    keys = keys_to_device(keys_np, wf.device)
    ref = _phase_f64_reference(hil, wf, keys_np, 10)
    errs = _phase_errors(fused, keys, ref, ("0", "1", "2"))
    scale = np.abs(ref).max()
    print("phase error vs float64: f32 MFMA max %.3e mean %.3e | bf16x3 max %.3e mean %.3e | f16x2 max %.3e mean %.3e | scale %.3f"
          % (errs["0"].max(), errs["0"].mean(), errs["1"].max(), errs["1"].mean(), errs["2"].max(), errs["2"].mean(), scale))
    for mode in ("0", "1", "2"):
        assert errs[mode].max() < 5e-6 * scale, (mode, errs[mode].max(), scale)
    assert errs["1"].mean() < 2.0 * errs["0"].mean() + 1e-9, (errs["0"].mean(), errs["1"].mean())
    assert errs["2"].mean() < 2.0 * errs["0"].mean() + 1e-9, (errs["0"].mean(), errs["2"].mean())


@pytest.mark.parametrize("case", ["huge", "tiny", "wide", "zero_layer"])
def test_f16x2_phase_kernel_dynamic_range(case):
    """The f16x2 format scales every tensor by a power of two derived from the weights on the device (f16 tops out at
    65 504): activations of order 1e6 (would overflow unscaled), weights of order 1e-7 (would be f16-subnormal
    unscaled), weights spanning 12 orders of magnitude inside one layer, and an all-zero layer must all come out
    with the exact-f32 kernel's accuracy."""
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.hamiltonian import keys_to_device
    z = golden("nade_N2.npz")
    hil, wf = make_wf("N2", z, device="cuda")
    lins = [l for l in wf.model.phase_layers[0].linears()]
    g = torch.Generator(device="cpu").manual_seed(5)
    with torch.no_grad():
        if case == "huge":
            lins[0].weight.mul_(3.0e4); lins[0].bias.mul_(3.0e4); lins[1].weight.mul_(40.0)
            lins[2].weight.mul_(1e-6); lins[2].bias.mul_(1.0)
        elif case == "tiny":
            lins[0].weight.mul_(1e-6); lins[0].bias.mul_(1e-6); lins[1].weight.mul_(1e-6); lins[1].bias.mul_(1e-12)
            lins[2].weight.mul_(1e12)
        elif case == "wide":
            f = torch.pow(10.0, torch.rand(lins[1].weight.shape, generator=g) * 12.0 - 9.0).to(lins[1].weight.device)
            lins[1].weight.mul_(f)
            lins[2].weight.mul_(1e-2)
        else:
            lins[1].weight.zero_()
    fused = FusedLogPsi(wf)
    keys_np = z["samp_keys"][:2048].astype(np.int64)
    keys = keys_to_device(keys_np, wf.device)
    ref = _phase_f64_reference(hil, wf, keys_np, 10)
    assert np.all(np.isfinite(ref))
    errs = _phase_errors(fused, keys, ref, ("0", "2"))
    scale = max(np.abs(ref).max(), 1e-30)
    print("%s: f32 MFMA max %.3e mean %.3e | f16x2 max %.3e mean %.3e | scale %.3e"
          % (case, errs["0"].max(), errs["0"].mean(), errs["2"].max(), errs["2"].mean(), scale))
    assert np.all(np.isfinite(errs["2"]))
    assert errs["2"].max() < max(3.0 * errs["0"].max(), 2e-6 * scale), (errs["0"].max(), errs["2"].max(), scale)
    assert errs["2"].mean() < 2.0 * errs["0"].mean() + 1e-7 * scale, (errs["0"].mean(), errs["2"].mean())


@pytest.mark.parametrize("mol", ["LiH", "N2"])
def test_fused_logpsi_eloc_equals_separate_calls(mol):
    """naqs_logpsi_eloc (keys inserted by the amplitude kernel, psi written by the phase kernel, no prep kernel)
    == naqs_net_logpsi followed by naqs_eloc(+reduce): bit-identical, also across repeated calls (epochs)."""
    import os
    from test_nade import make_wf
    from naqs_amd import hamiltonian, packing
    from naqs_amd.fused import FusedLogPsi
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z, device="cuda")
    fused = FusedLogPsi(wf)
    ham = hamiltonian.DevicePauliHamiltonian(packing.load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz")))
    keys = hamiltonian.keys_to_device(z["samp_keys"], wf.device)
    w = torch.as_tensor(z["samp_counts"].astype(np.float64), device=wf.device)
    lp = fused.log_psi(keys)
    e, sums = ham.local_energy(keys, lp, kind="log_psi", weights=w)
    for _ in range(3):
        lp2, e2, sums2 = fused.log_psi_and_local_energy(ham, keys, weights=w)
    torch.cuda.synchronize()
    assert torch.equal(lp, lp2) and torch.equal(e, e2) and torch.equal(sums, sums2)
    lp3, e3 = fused.log_psi_and_local_energy(ham, keys)
    assert torch.equal(e, e3)
    # and against the reference's E_loc for these samples (float32 psi there -> 2e-5)
    want = z["sgd_eloc_c128"]
    got = e.cpu().numpy()
    assert np.max(np.abs(got[:, 0] + 1j * got[:, 1] - want) / np.maximum(1, np.abs(want))) < 2e-5


@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2"])
def test_fused_training_forward_backward_matches_autograd(mol):
    """naqs_net_logamp / naqs_net_amp_backward (+ the phase MLP through torch) against PyTorch autograd of the
    module formulation on the same weights: values 5e-5, every parameter gradient to 2e-4 of its scale."""
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.hamiltonian import keys_to_device
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z, device="cuda")
    fused = FusedLogPsi(wf)
    states = torch.tensor(z["eval_states"], device="cuda")
    keys = keys_to_device(z["eval_keys"].astype(np.int64), "cuda")
    gen = torch.Generator(device="cuda").manual_seed(3)
    g = torch.randn((len(keys), 2), device="cuda", generator=gen) / len(keys)
    params = list(wf.model.parameters())

    lp_ref = wf.log_psi(states).reshape(-1, 2)
    grads_ref = torch.autograd.grad((lp_ref * g).sum(), params, allow_unused=True)
    lp = fused.log_psi_train(keys)
    grads = torch.autograd.grad((lp * g).sum(), params, allow_unused=True)
    assert torch.max(torch.abs(lp - lp_ref)).item() < 5e-5
    assert torch.allclose(lp[:, 0], fused.log_psi(keys)[:, 0], rtol=0, atol=2e-6)   # VALU amp_kernel vs the MFMA prologue of the phase kernel
    for (name, _), a, b in zip(wf.model.named_parameters(), grads, grads_ref):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, name
            continue
        scale = float(b.abs().max()) + 1e-12
        assert a is not None and float((a - b).abs().max()) < 2e-4 * scale + 1e-9, (name, float((a - b).abs().max()), scale)
    # determinism of the HIP backward
    grads2 = torch.autograd.grad((fused.log_psi_train(keys) * g).sum(), params, allow_unused=True)
    assert all(torch.equal(a, b) for a, b in zip(grads, grads2) if a is not None)


@pytest.mark.parametrize("mode", ["hip", "blas"])
@pytest.mark.parametrize("mol", ["LiH", "N2"])
def test_graph_free_training_step_matches_autograd(mol, mode):
    """forward_saved / backward_saved (no autograd engine) give the same values and .grad as the Function path."""
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.hamiltonian import keys_to_device
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z, device="cuda")
    fused = FusedLogPsi(wf)
    fused.train_mode = mode
    keys = keys_to_device(z["eval_keys"].astype(np.int64), "cuda")
    gen = torch.Generator(device="cuda").manual_seed(4)
    g = torch.randn((len(keys), 2), device="cuda", generator=gen) / len(keys)
    params = list(wf.model.parameters())
    lp_ref = fused.log_psi_train(keys)
    grads_ref = torch.autograd.grad((lp_ref * g).sum(), params, allow_unused=True)
    for p in params:
        p.grad = None
    lp, saved = fused.forward_saved(keys)
    assert not lp.requires_grad and torch.allclose(lp, lp_ref.detach(), rtol=0, atol=1e-6)
    fused.backward_saved(saved, g)
    for (name, p), b in zip(wf.model.named_parameters(), grads_ref):
        assert p.grad is not None, name
        ref = torch.zeros_like(p) if b is None else b
        scale = float(ref.abs().max()) + 1e-12
        assert float((p.grad - ref).abs().max()) < 2e-5 * scale + 1e-10, (name, float((p.grad - ref).abs().max()), scale)
    fused.backward_saved(saved, g)                               # accumulates like autograd does
    assert torch.allclose(params[-1].grad, 2 * grads_ref[-1], rtol=1e-5, atol=1e-10)


def test_fused_step_calls_equal_their_parts():
    """The training loop's combined library calls against the separate ones they replace, same inputs, bit for bit:
    naqs_vmc_sample_forward_eloc == naqs_net_sample_weighted + naqs_net_train_forward_eloc; naqs_net_train_backward_vmc ==
    naqs_vmc_loss_grad_ev + naqs_net_train_backward; naqs_shard_proof == its definition."""
    import os
    from conftest import GOLDEN
    from test_nade import make_wf
    from naqs_amd import _lib, hamiltonian, packing
    from naqs_amd.fused import _stream_ptr
    z = golden("nade_LiH.npz")
    hil, wf = make_wf("LiH", z, device="cuda")
    fused = wf.fused()
    ham = hamiltonian.DevicePauliHamiltonian(packing.load_packed(os.path.join(GOLDEN, "ham_LiH.npz")), device="cuda:0")
    # sampling + forward + E_loc
    keys, counts, probs, weights, pre = fused.sample_forward_local_energy(ham, 10 ** 6, 77, 1000)
    lp, saved, eloc, sums = pre
    k2, c2, p2, w2 = fused.sample(10 ** 6, 77, 1000, with_weights=True)
    assert torch.equal(keys, k2) and torch.equal(counts, c2) and torch.equal(probs, p2) and torch.equal(weights, w2)
    lp2, saved2, eloc2, sums2 = fused.forward_saved_with_local_energy(ham, k2, w2)
    torch.cuda.synchronize()
    assert torch.equal(lp, lp2) and torch.equal(eloc, eloc2) and torch.equal(sums, sums2)
    # loss gradient + backward
    for p in wf.model.parameters():
        p.grad = None
    g, ev = fused.backward_from_local_energy(saved2, eloc2, w2.contiguous(), sums2)
    grads = [p.grad.clone() for p in wf.model.parameters()]
    for p in wf.model.parameters():
        p.grad = None
    fused._grad_flat = None
    lp3, saved3 = fused.forward_saved(k2)
    g_ref, ev_ref = fused.vmc_loss_grad(eloc2, w2.contiguous(), sums2, with_energy=True)
    fused.backward_saved(saved3, g_ref)
    torch.cuda.synchronize()
    assert torch.equal(g, g_ref) and torch.equal(ev, ev_ref)
    for a, p in zip(grads, wf.model.parameters()):
        assert torch.equal(a, p.grad)
    # the accumulator payload of the multi-GPU step
    lib = _lib.load_library()
    ext = torch.empty(8, dtype=torch.float64, device="cuda")
    _lib.check(lib.naqs_shard_proof(len(k2), k2.data_ptr(), sums2.data_ptr(), ext.data_ptr(), _stream_ptr(k2.device)), "naqs_shard_proof")
    c = float((k2.sum() & 0xFFFFF).item())
    want = torch.cat([sums2, torch.tensor([len(k2), len(k2) ** 2, c, c * c], dtype=torch.float64, device="cuda")])
    assert torch.equal(ext, want)


@pytest.mark.parametrize("mol,n_samples", [("N2", 10 ** 6), ("N2", 300), ("LiH", 10 ** 5)])
def test_backward_as_one_launch_equals_its_pieces(mol, n_samples, monkeypatch):
    """backward_mega_kernel (grad_in + amplitude blocks + the upper layers' weight gradients in one launch) runs the device
    functions of the separate launches on the same operands: every gradient element bit for bit."""
    from test_nade import make_wf
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z, device="cuda")
    fused = wf.fused()
    keys, counts, probs = fused.sample(n_samples, 5, 100000)
    gen = torch.Generator(device="cuda").manual_seed(2)
    g = torch.randn((len(keys), 2), device="cuda", generator=gen) / len(keys)
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NAQS_TRAIN_MEGA", mode)
        for p in wf.model.parameters():
            p.grad = None
        fused._grad_flat = None
        lp, saved = fused.forward_saved(keys)
        fused.backward_saved(saved, g)
        torch.cuda.synchronize()
        grads[mode] = torch.cat([p.grad.reshape(-1) for p in wf.model.parameters()]).clone()
        assert fused.last_kernel() is not None
    assert torch.equal(grads["1"], grads["0"]) and float(grads["1"].abs().max()) > 0


def test_fresh_network_on_a_side_stream_of_a_busy_gpu():
    """The handle's allocation-time zero fills (weight-range words at creation, the training scratch at the first training
    forward) run on the null stream; a side stream is not ordered against it, so the library has to wait for them itself —
    otherwise a fill that is late (busy GPU) wipes the ranges / activations the side stream's kernels have just written.
    A fresh network used for the first time on a side stream, under load from another stream, gives the quiet GPU's answer
    (see test_first_call_on_a_side_stream_of_a_busy_gpu for E_loc's table)."""
    from test_nade import make_wf
    z = golden("nade_N2.npz")
    s = torch.tensor(z["eval_states"], device="cuda")
    hil, wf = make_wf("N2", z, device="cuda")
    with torch.no_grad():
        quiet = wf.log_psi(s).clone()
    quiet_train = wf.log_psi(s).detach().clone()              # (with autograd: the training forward)
    load = torch.randn(6144, 6144, device="cuda")
    busy, side = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(4):
        with torch.cuda.stream(busy):
            for _ in range(12):
                load @ load
        with torch.cuda.stream(side):
            hil, wf2 = make_wf("N2", z, device="cuda")        # fresh handle, created and first used on the side stream
            with torch.no_grad():
                got = wf2.log_psi(s)
            got_train = wf2.log_psi(s).detach()
            side.synchronize()
        assert torch.equal(got, quiet) and torch.equal(got_train, quiet_train), rep
    torch.cuda.synchronize()


def test_big_layer_shared_by_two_workgroups_agrees_with_one(monkeypatch):
    """Small tables (two workgroups per 16-row tile still fit the chip): phase_kernel_ws<1, SAVE, SPLIT> halves the big layer's
    weight stream per workgroup; the producer's partial rows reach the consumer as tagged words.  Against the one-workgroup
    form (NAQS_WS_SPLIT=0), inference and training forward alike: log|psi| identical, the phase within rounding of the
    different final add (2e-6), gradients (they read the activations the two halves saved) to 1e-5 of their scale; repeated
    calls (the call tag moves on) and ragged / single-row / just-too-large tables included."""
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.hamiltonian import keys_to_device
    from test_eloc_gpu import random_physical_keys
    z = golden("nade_N2.npz")
    hil, wf = make_wf("N2", z, device="cuda")
    fused = FusedLogPsi(wf)
    assert fused.train_mode == "hip"
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    all_keys = random_physical_keys(20, 7, 7, 16 * (cu // 2) + 40, 11).astype(np.int64)
    params = list(wf.model.parameters())
    for rep, M in enumerate([1, 15, 16, 17, 333, 1200, 16 * (cu // 2), 16 * (cu // 2) + 1, 1200, 1200, 1200]):
        keys = keys_to_device(np.sort(np.roll(all_keys, 7 * rep)[:M]), "cuda")
        gen = torch.Generator(device="cuda").manual_seed(3)
        g = torch.randn((M, 2), device="cuda", generator=gen) / M
        res = {}
        for split in ("1", "0"):
            monkeypatch.setenv("NAQS_WS_SPLIT", split)
            lp = fused.log_psi(keys).clone()
            name = fused.last_kernel()
            for p in params:
                p.grad = None
            lpt, saved = fused.forward_saved(keys)
            name_t = fused.last_kernel()
            fused.backward_saved(saved, g)
            res[split] = (lp, lpt.clone(), [None if p.grad is None else p.grad.clone() for p in params], name, name_t)
        fits = 2 * ((M + 15) // 16) <= cu
        assert ("SPLIT=1" in res["1"][3]) == fits and ("SPLIT=1" in res["1"][4]) == fits, (M, res["1"][3], res["1"][4])
        assert "SAVE=0" in res["1"][3] and "SAVE=1" in res["1"][4]
        assert "SPLIT" not in res["0"][3] and "SPLIT" not in res["0"][4] and "phase_kernel_ws<RB=1" in res["0"][3]
        for a, b in ((res["1"][0], res["0"][0]), (res["1"][1], res["0"][1])):
            assert torch.equal(a[:, 0], b[:, 0])
            assert torch.max(torch.abs(a[:, 1] - b[:, 1])).item() < 2e-6
            assert fits or torch.equal(a, b)
        assert torch.equal(res["1"][0], res["1"][1])           # inference and training forward: the same kernel, the same values
        for (pname, _), a, b in zip(wf.model.named_parameters(), res["1"][2], res["0"][2]):
            if b is None:
                assert a is None
                continue
            scale = float(b.abs().max()) + 1e-12
            assert float((a - b).abs().max()) < 1e-5 * scale + 1e-10, (M, pname)


def test_transposed_big_layer_kernel_agrees(monkeypatch):
    """phase_kernel_wt (NAQS_PHASE_WT=1, off by default: measured no faster): the big layer as H1^T = W1 . H0^T in 80-row tiles
    shared by two workgroups, layer 0's chunks produced just in time by the amplitude waves into an LDS ring.  Same products,
    another summation order inside an MFMA and in the output layer: log|psi| bit-identical to the default kernel, phase to the
    last bits, both <= 2e-5 of the PyTorch modules — also on a table that is not a multiple of the tile, and chained into E_loc."""
    import os
    from test_nade import make_wf
    from naqs_amd import hamiltonian, packing
    from naqs_amd.fused import FusedLogPsi
    z = golden("nade_N2.npz")
    hil, wf = make_wf("N2", z, device="cuda")
    rs = np.random.RandomState(3)
    space = np.array(sorted(set(z["samp_keys"].astype(np.int64).tolist()) | set(z["eval_keys"].astype(np.int64).tolist())))
    keys_np = np.sort(rs.choice(space, min(len(space), 4171), replace=False))
    keys = hamiltonian.keys_to_device(keys_np, wf.device)
    ref = FusedLogPsi(wf)
    lp0 = ref.log_psi(keys).clone()
    assert "phase_kernel_ws" in ref.last_kernel()
    monkeypatch.setenv("NAQS_PHASE_WT", "1")
    fused = FusedLogPsi(wf)                                   # (packs the transposed copy of the big layer)
    lp1 = fused.log_psi(keys).clone()
    assert "phase_kernel_wt" in fused.last_kernel(), fused.last_kernel()
    with torch.no_grad():
        lp_t = wf.log_psi(hil.idx2state(torch.as_tensor(keys_np, device="cuda")))
    assert torch.equal(lp0[:, 0], lp1[:, 0])
    assert torch.max(torch.abs(lp1 - lp_t)).item() < 2e-5 and torch.max(torch.abs(lp1[:, 1] - lp0[:, 1])).item() < 2e-6
    ham = hamiltonian.DevicePauliHamiltonian(packing.load_packed(os.path.join(GOLDEN, "ham_N2.npz")))
    lp2, e2 = fused.log_psi_and_local_energy(ham, keys)
    e_ref = ham.local_energy(keys, lp1, kind="log_psi")
    torch.cuda.synchronize()
    assert torch.equal(lp2, lp1) and torch.equal(e2, e_ref)
    for _ in range(3):                                        # the call tag of the hand-over words moves on
        assert torch.equal(fused.log_psi(keys), lp1)
