"""Is a training run bit-reproducible?  The same seeded optimiser twice in one process (and the one-call step against the
call-by-call loop): first step at which the logged energies differ, per molecule.
usage: python tools/determinism_probe.py <molecule npz> [steps]"""
import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import numpy as np, torch
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.nade import NadeMasking
from naqs_amd.optimizer import PartialSamplingOptimizer, LogKey
from naqs_amd.system import load_molecule, set_global_seed
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals

mol_f = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda", 0)


shape = (dict(amp_hidden_size=[128], phase_hidden_size=[128], aggregate_phase=True) if os.environ.get("NAQS_PROFILE_DEFAULT_ANSATZ") == "1"
         else dict(amp_hidden_size=[64], phase_hidden_size=[512, 512], aggregate_phase=False))


def run(onecall):
    os.environ["NAQS_TRAIN_ONECALL"] = onecall
    with contextlib.redirect_stdout(io.StringIO()):
        set_global_seed(1)
        mol, qh = load_molecule(mol_f)
    na, nb = mol.get_n_alpha_electrons(), mol.get_n_beta_electrons()
    hil = Hilbert.get(N=mol.n_qubits, N_alpha=na, N_beta=nb, encoding=Encoding.SIGNED)
    wf = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, masking=NadeMasking.PARTIAL, use_amp_spin_sym=(na == nb), use_phase_spin_sym=False,
                                   n_alpha_electrons=na, n_beta_electrons=nb, device=dev, **shape)
    opt = PartialSamplingOptimizer(n_samples=1000000, n_samples_max=1e12, n_unq_samples_min=1000, n_unq_samples_max=1e5,
                                   wavefunction=wf, qubit_hamiltonian=qh, pre_compute_H=False, n_electrons=mol.n_electrons,
                                   n_alpha_electrons=na, n_beta_electrons=nb, optimizer=torch.optim.Adam,
                                   optimizer_args=[{'lr': 1e-3, 'betas': (0.9, 0.99), 'eps': 1e-15}, {'lr': 1e-2}],
                                   save_loc="/tmp/determinism_probe", seed=1, grad_clip_factor=None, log_exact_energy=False,
                                   pauli_hamiltonian_dtype=np.float64, normalise_psi=True)
    with contextlib.redirect_stdout(io.StringIO()):
        opt.run(steps, output_freq=10 ** 9)
    e = np.array([x[1] for x in opt.log[LogKey.E_LOC]])
    n = np.array([x[1] for x in opt.log[LogKey.N_UNIQUE_SAMP]])
    return e, n


modes = os.environ.get("PROBE_MODES", "1,1,0").split(",")          # (PROBE_MODES=0,0: an older library without naqs_vmc_step)
runs = [(f"run {i} (one-call={m})", run(m)) for i, m in enumerate(modes)]
e0, n0 = runs[0][1]
for name, (e, n) in runs[1:]:
    de = np.nonzero(e != e0)[0]
    dn = np.nonzero(n != n0)[0]
    print(f"{os.path.basename(mol_f)}: {name} vs run 0: energies first differ at step {de[0] if len(de) else None} "
          f"(|d| = {abs(e[de[0]] - e0[de[0]]) if len(de) else 0:.3e}), sample counts at {dn[0] if len(dn) else None}; "
          f"M range {n0.min()}-{n0.max()}")
