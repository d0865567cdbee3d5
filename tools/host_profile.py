"""cProfile of the optimizer loop (host side) on the GPU box."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import numpy as np, torch
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.optimizer import PartialSamplingOptimizer
from naqs_amd.system import load_molecule, set_global_seed
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
set_global_seed(1)
mol, qh = load_molecule(os.path.join(ROOT, "tests/golden/ham_N2.npz"))
na, nb = mol.get_n_alpha_electrons(), mol.get_n_beta_electrons()
hil = Hilbert.get(N=mol.n_qubits, N_alpha=na, N_beta=nb, encoding=Encoding.SIGNED)
wf = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512], use_amp_spin_sym=True,
                               use_phase_spin_sym=False, aggregate_phase=False, n_alpha_electrons=na, n_beta_electrons=nb, device="cuda")
opt = PartialSamplingOptimizer(n_samples=1000000, n_samples_max=1e12, n_unq_samples_min=1000, n_unq_samples_max=1e5, wavefunction=wf,
                               qubit_hamiltonian=qh, pre_compute_H=False, n_electrons=mol.n_electrons, n_alpha_electrons=na,
                               n_beta_electrons=nb, optimizer=torch.optim.Adam, normalise_psi=True, grad_clip_factor=None,
                               optimizer_args=[{'lr': 1e-3, 'betas': (0.9, 0.99), 'eps': 1e-15}, {'lr': 1e-2}],
                               save_loc="/tmp/host_profile", seed=1, pauli_hamiltonian_dtype=np.float64)
opt.run(50, output_freq=1000)
pr = cProfile.Profile(); pr.enable()
opt.run(300, output_freq=1000)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45); print(st.getvalue()[:9000])
