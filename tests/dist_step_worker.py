"""Child process of tests/test_optimizer_gpu.py: runs the optimiser loop on cuda:0 as one rank of a
``torch.distributed`` group and writes its per-step energies + final parameters to a file.

    python dist_step_worker.py <backend> <rank> <world> <port> <out.pt> [n_steps]

``nccl`` with world 1 exercises RCCL itself; ``gloo`` with world 2 (two processes sharing the one GPU of the
test box — RCCL refuses two ranks on one device, gloo stages device tensors through the host) exercises the
world > 1 branches of ``_SGD_step`` on the HIP path: table log psi through the inference kernel between the
training forward and backward, non-trivial row shards, the accumulator and flat-gradient all-reduces.
``none`` runs the same loop without a process group (the single-process answer)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd"), os.path.join(ROOT, "tests")]
backend, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
n_steps = int(sys.argv[6]) if len(sys.argv) > 6 else 30
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
if backend != "none":
    dist.init_process_group(backend, rank=rank, world_size=world)

from naqs_amd.hilbert import Encoding, Hilbert  # noqa: E402
from naqs_amd.optimizer import LogKey, PartialSamplingOptimizer  # noqa: E402
from naqs_amd.system import load_molecule, set_global_seed  # noqa: E402
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals  # noqa: E402

set_global_seed(1)
mol, qh = load_molecule(os.path.join(ROOT, "tests/golden/ham_H2O.npz"))
na, nb = mol.get_n_alpha_electrons(), mol.get_n_beta_electrons()
hil = Hilbert.get(N=mol.n_qubits, N_alpha=na, N_beta=nb, encoding=Encoding.SIGNED)
torch.manual_seed(3)
wf = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512],
                               use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=False,
                               n_alpha_electrons=na, n_beta_electrons=nb, device="cuda")
opt = PartialSamplingOptimizer(
    n_samples=100000, n_samples_max=1e12, n_unq_samples_min=10, n_unq_samples_max=1e5, wavefunction=wf,
    qubit_hamiltonian=qh, pre_compute_H=False, n_electrons=mol.n_electrons, n_alpha_electrons=na, n_beta_electrons=nb,
    optimizer=torch.optim.Adam, normalise_psi=True, grad_clip_factor=None,
    optimizer_args=[{'lr': 1e-3, 'betas': (0.9, 0.99), 'eps': 1e-15}, {'lr': 1e-2}],
    save_loc=os.path.join(os.path.dirname(out), f"ckpt_{backend}_{rank}"), seed=1, pauli_hamiltonian_dtype=np.float64)
assert wf.fused() is not None, "the HIP path must be the one under test"
opt.run(n_steps, output_freq=10 ** 6)
energies = [x[1] for x in opt.log[LogKey.E_LOC]]
params = torch.cat([p.detach().reshape(-1) for p in wf.model.parameters()]).cpu()
torch.save({"energies": energies, "params": params, "dist_modes": list(opt.dist_mode_log)}, out)
if backend != "none":
    dist.barrier()
    dist.destroy_process_group()
print(f"worker {backend} {rank}/{world}: done, E[-1] = {energies[-1]:.8f}")
