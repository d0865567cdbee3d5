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
