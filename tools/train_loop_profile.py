"""The optimiser's own training loop (PartialSamplingOptimizer.run) on one GPU, timed; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel launch list of a VMC step.
usage: python tools/train_loop_profile.py [molecule npz] [n_samples] [steps] [warmup]"""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import numpy as np, torch
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.nade import NadeMasking
from naqs_amd.optimizer import PartialSamplingOptimizer
from naqs_amd.system import load_molecule, set_global_seed

from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
mol_f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests/golden/ham_N2.npz")
n_samples = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
warmup = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(io.StringIO()):
    set_global_seed(1)
    mol, qh = load_molecule(mol_f)
na, nb = mol.get_n_alpha_electrons(), mol.get_n_beta_electrons()
hil = Hilbert.get(N=mol.n_qubits, N_alpha=na, N_beta=nb, encoding=Encoding.SIGNED)
# NAQS_PROFILE_DEFAULT_ANSATZ=1: the reference's run.py default (aggregate phase: one phase block per pair, 128 hidden units)
if os.environ.get("NAQS_PROFILE_DEFAULT_ANSATZ") == "1":
    shape = dict(amp_hidden_size=[128], phase_hidden_size=[128], aggregate_phase=True)
else:
    shape = dict(amp_hidden_size=[64], phase_hidden_size=[512, 512], aggregate_phase=False)
wf = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, masking=NadeMasking.PARTIAL, use_amp_spin_sym=True,
                               use_phase_spin_sym=False, n_alpha_electrons=na, n_beta_electrons=nb, device=dev, **shape)
opt = PartialSamplingOptimizer(n_samples=n_samples, n_samples_max=1e12, n_unq_samples_min=1000, n_unq_samples_max=1e5,
                               wavefunction=wf, qubit_hamiltonian=qh, pre_compute_H=False, n_electrons=mol.n_electrons,
                               n_alpha_electrons=na, n_beta_electrons=nb, optimizer=torch.optim.Adam,
                               optimizer_args=[{'lr': 1e-3, 'betas': (0.9, 0.99), 'eps': 1e-15}, {'lr': 1e-2}],
                               save_loc="/tmp/train_loop_profile", seed=1, grad_clip_factor=None, log_exact_energy=False,
                               pauli_hamiltonian_dtype=np.float64, normalise_psi=True)
with contextlib.redirect_stdout(io.StringIO()):
    opt.run(warmup, output_freq=10 ** 9)
if os.environ.get("NAQS_GC") == "freeze":
    import gc; gc.collect(); gc.freeze()
elif os.environ.get("NAQS_GC") == "off":
    import gc; gc.disable()
prof = None
if os.environ.get("NAQS_PROFILE_HOST") == "1":      # cProfile of the timed loop only (host side of a step)
    import cProfile
    prof = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
if prof:
    prof.enable()
with contextlib.redirect_stdout(io.StringIO()):
    opt.run(steps, output_freq=10 ** 9)
if prof:
    prof.disable()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
if prof:
    import pstats
    buf = io.StringIO()
    pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(24)
    print(buf.getvalue()[:5000])
from naqs_amd.optimizer import LogKey
n_unq = opt.log[LogKey.N_UNIQUE_SAMP][-1][1]
print(f"{os.path.basename(mol_f)}: {steps} steps, {dt / steps * 1e3:.3f} ms/step, {n_unq} unique samples in the last step, "
      f"<E_loc> = {opt.log[LogKey.E_LOC][-1][1]:.6f}")
try:      # forwards launched ahead of the host's look at M, and how many of them stood (naqs_net_spec_counts)
    import ctypes
    from naqs_amd import _lib
    fz = wf._fused
    c = (ctypes.c_int64 * 2)()
    _lib.check(_lib.load_library().naqs_net_spec_counts(fz._h, c), "naqs_net_spec_counts")
    print(f"  forward launched ahead of M: {c[0]} times, {c[1]} stood")
except Exception as e:      # (an older library)
    print("  (no spec counts:", e, ")")
