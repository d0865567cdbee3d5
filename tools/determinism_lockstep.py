"""Two identically seeded optimisers stepped in lockstep (call-by-call path): which quantity differs first?
usage: python tools/determinism_lockstep.py <molecule npz> [steps]"""
import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
os.environ["NAQS_TRAIN_ONECALL"] = "0"
import numpy as np, torch
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.nade import NadeMasking
from naqs_amd.optimizer import PartialSamplingOptimizer
from naqs_amd.system import load_molecule, set_global_seed
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals

def eprint(*a):
    print(*a, file=sys.stderr, flush=True)


mol_f = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)


def make():
    with contextlib.redirect_stdout(io.StringIO()):
        set_global_seed(1)
        mol, qh = load_molecule(mol_f)
    na, nb = mol.get_n_alpha_electrons(), mol.get_n_beta_electrons()
    hil = Hilbert.get(N=mol.n_qubits, N_alpha=na, N_beta=nb, encoding=Encoding.SIGNED)
    wf = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, masking=NadeMasking.PARTIAL, use_amp_spin_sym=True, use_phase_spin_sym=False,
                                   n_alpha_electrons=na, n_beta_electrons=nb, device=dev, amp_hidden_size=[64],
                                   phase_hidden_size=[512, 512], aggregate_phase=False)
    return PartialSamplingOptimizer(n_samples=1000000, n_samples_max=1e12, n_unq_samples_min=1000, n_unq_samples_max=1e5,
                                    wavefunction=wf, qubit_hamiltonian=qh, pre_compute_H=False, n_electrons=mol.n_electrons,
                                    n_alpha_electrons=na, n_beta_electrons=nb, optimizer=torch.optim.Adam,
                                    optimizer_args=[{'lr': 1e-3, 'betas': (0.9, 0.99), 'eps': 1e-15}, {'lr': 1e-2}],
                                    save_loc="/tmp/determinism_probe", seed=1, grad_clip_factor=None, log_exact_energy=False,
                                    pauli_hamiltonian_dtype=np.float64, normalise_psi=True)


A, B = make(), make()
for step in range(steps):
    out = []
    for o in (A, B):
        with contextlib.redirect_stdout(io.StringIO()):
            o.get_samples(lazy=True)
            keys, w = o._sample_keys.clone(), o._sample_weights.clone()
            if o is B and not (keys.shape == out[0][0].shape and torch.equal(keys, out[0][0])):
                # same parameters (compared after the last step), same seed, different tables: look closer before stepping on
                pa, pb = A.wavefunction.flatten_parameters(), B.wavefunction.flatten_parameters()
                eprint(f"step {step}: tables differ (M = {len(out[0][0])} / {len(keys)}); parameters equal: {torch.equal(pa, pb)}; "
                      f"n_samples {A.n_samples} / {B.n_samples}; sample calls {A.wavefunction._sample_calls} / {B.wavefunction._sample_calls}")
                fa, fb = A.wavefunction.fused(), B.wavefunction.fused()
                probe = out[0][0][:4096].contiguous()
                eprint("   log psi of A's first keys equal between the two nets:", torch.equal(fa.log_psi(probe), fb.log_psi(probe)))
                for label, f in (("A", fa), ("B", fb)):
                    t = [f.sample(A.n_samples, seed=4242, max_unique=100000) for _ in range(4)]
                    eprint(f"   net {label}: M of four draws with one seed: {[len(x[0]) for x in t]}; equal to the first: "
                          f"{[all(torch.equal(a, b) for a, b in zip(x, t[0])) for x in t]}")
                ta, tb = fa.sample(A.n_samples, seed=4242, max_unique=100000), fb.sample(A.n_samples, seed=4242, max_unique=100000)
                eprint("   A vs B with one seed:", len(ta[0]), len(tb[0]), all(torch.equal(a, b) for a, b in zip(ta, tb)) if len(ta[0]) == len(tb[0]) else False)
                sys.exit(0)
            pre = o._prefused[1] if o._prefused is not None else None
            lp = pre[0].clone() if pre is not None else None
            el = pre[2].clone() if pre is not None else None
            ev = o._SGD_step(None, o._sample_keys, None, sample_weights=o._sample_weights, lazy=True).clone()
        gflat = o.wavefunction.fused()._grad_flat
        out.append((keys, w, lp, el, ev, gflat.clone() if gflat is not None else None, o.wavefunction.flatten_parameters().clone()))
    names = ("keys", "weights", "log psi", "E_loc", "(E, Var)", "gradient", "parameters")
    diff = [n for n, a, b in zip(names, out[0], out[1]) if a is not None and (a.shape != b.shape or not torch.equal(a, b))]
    if diff:
        print(f"step {step}: M = {len(out[0][0])} / {len(out[1][0])}: differ: {diff}")
        for n, a, b in zip(names, out[0], out[1]):
            if n in diff and a.shape == b.shape:
                d = (a.double() - b.double()).abs()
                print(f"   {n}: {int((d > 0).sum())} of {d.numel()} elements, max |d| = {float(d.max()):.3e}")
        break
else:
    print(f"{steps} steps identical (M up to {max(len(o._sample_keys) for o in (A, B))})")
