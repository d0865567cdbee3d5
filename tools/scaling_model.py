"""Scaling model of the TRAINING step without the node: on ONE GPU, rank 0's share of `PartialSamplingOptimizer._SGD_step`
at world sizes 1, 2, 4, 8 — the real world > 1 code path of the optimiser (replicated sampler, forward/backward and E_loc
for the rank's row shard only, the table assembled by an all-gather, two all-reduces) driven through a stand-in for
`torch.distributed` that issues no collective: `all_reduce` multiplies by W (as if every rank had contributed rank 0's
partial sums: ratios such as <E> and the same-table proof are unchanged, the shard gradient stands in for the full one) and `all_gather_into_tensor`
delivers the true table by evaluating log psi of all rows with the inference kernel — GPU work a real rank does not do,
timed separately and subtracted.

usage: python tools/scaling_model.py [molecule npz] [steps] [out.json]
Everything is single-GPU evidence ("unmeasured on hardware" for W > 1); collective latencies are labelled assumptions."""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import numpy as np
import torch
from naqs_amd import optimizer as O
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.nade import NadeMasking
from naqs_amd.system import load_molecule, set_global_seed
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals

mol_f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests/golden/ham_N2.npz")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
out_f = sys.argv[3] if len(sys.argv) > 3 else None
ASSUMED_US = {"all_gather_table": 25.0, "all_reduce_accumulators": 20.0, "all_reduce_gradient_1MB": 35.0}
dev = torch.device("cuda", 0)


class FakeDist:
    """rank 0 of `world` with nobody else there"""

    def __init__(self, world, fused_ref):
        self.world, self.fused_ref, self.keys, self.t_table = world, fused_ref, None, 0.0

    def get_world_size(self): return self.world
    def get_rank(self): return 0
    def is_available(self): return True
    def is_initialized(self): return True
    def get_backend(self): return "emulated"

    def all_reduce(self, t, *a, **k):
        t.mul_(float(self.world))          # W identical contributions: <E> (a ratio of sums) unchanged, the same-table proof holds,
        return None                         # the shard's gradient stands in for the sum over the ranks

    def all_gather_into_tensor(self, out, mine):
        fused = self.fused_ref()
        lp = fused.log_psi(self.keys)                                   # what the other ranks would have delivered
        S = mine.shape[0]
        M = lp.shape[0]
        Sr = -(-M // self.world)
        view = out.view(self.world, S, 2)
        pad = torch.zeros((Sr * self.world, 2), dtype=lp.dtype, device=lp.device)
        pad[:M] = lp
        view[:, :Sr] = pad.view(self.world, Sr, 2)
        view[0] = mine


def build():
    with contextlib.redirect_stdout(io.StringIO()):
        set_global_seed(1)
        mol, qh = load_molecule(mol_f)
    na, nb = mol.get_n_alpha_electrons(), mol.get_n_beta_electrons()
    hil = Hilbert.get(N=mol.n_qubits, N_alpha=na, N_beta=nb, encoding=Encoding.SIGNED)
    wf = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, masking=NadeMasking.PARTIAL, use_amp_spin_sym=True, use_phase_spin_sym=False,
                                   n_alpha_electrons=na, n_beta_electrons=nb, device=dev, amp_hidden_size=[64],
                                   phase_hidden_size=[512, 512], aggregate_phase=False)
    opt = O.PartialSamplingOptimizer(n_samples=1000000, n_samples_max=1e12, n_unq_samples_min=1000, n_unq_samples_max=1e5, wavefunction=wf,
                                     qubit_hamiltonian=qh, pre_compute_H=False, n_electrons=mol.n_electrons, n_alpha_electrons=na,
                                     n_beta_electrons=nb, optimizer=torch.optim.Adam,
                                     optimizer_args=[{'lr': 1e-3, 'betas': (0.9, 0.99), 'eps': 1e-15}, {'lr': 1e-2}],
                                     save_loc="/tmp/scaling_model", seed=1, grad_clip_factor=None, log_exact_energy=False,
                                     pauli_hamiltonian_dtype=np.float64, normalise_psi=True)
    return wf, opt


rows = []
train_first = int(os.environ.get("NAQS_SCALING_TRAIN_FIRST", "1000"))
wf, opt = build()
with contextlib.redirect_stdout(io.StringIO()):
    opt.run(train_first, output_freq=10 ** 9)          # a network part-way into training: peaked distribution, M ~ 10^3
for g in opt.optimizer.param_groups:                 # ... then frozen (lr = 0), so that every world size times the SAME workload
    g["lr"] = 0.0
real_dist, real_step = O._dist, opt._SGD_step
for W in (1, 2, 4, 8):
    fake = FakeDist(W, lambda: wf.fused(need_phase=True)) if W > 1 else None
    if fake is not None:
        O._dist = lambda: fake

        def step(states, states_idx, *a, **k):
            fake.keys = O.keys_to_device(states_idx, dev)
            return real_step(states, states_idx, *a, **k)
        opt._SGD_step = step
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            opt.run(40, output_freq=10 ** 9)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            opt.run(steps, output_freq=10 ** 9)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        n_unq = float(np.mean([x[1] for x in opt.log[O.LogKey.N_UNIQUE_SAMP][-steps:]]))
        # the stand-in's table evaluation (inference kernel on all M rows + the copies), timed alone on the same table size
        t_tab = 0.0
        if fake is not None:
            keys = fake.keys
            mine = torch.zeros((-(-len(keys) // W), 2), dtype=torch.float32, device=dev)
            out = torch.empty((mine.shape[0] * W, 2), dtype=torch.float32, device=dev)
            for _ in range(20):
                fake.all_gather_into_tensor(out, mine)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(200):
                fake.all_gather_into_tensor(out, mine)
            torch.cuda.synchronize(); t_tab = (time.perf_counter() - t1) / 200
    finally:
        O._dist, opt._SGD_step = real_dist, real_step
    rows.append({"world": W, "wall_ms_per_step_incl_standin": dt * 1e3, "standin_table_ms": t_tab * 1e3,
                 "rank0_ms_per_step": (dt - t_tab) * 1e3, "mean_unique_samples": n_unq,
                 "path": "single-process step (forward + E_loc in one library call)" if W == 1 else
                         "sharded step (forward of my rows, all-gather, E_loc of my rows, two all-reduces)"})
    print(rows[-1], flush=True)
base = rows[0]["rank0_ms_per_step"]
coll = sum(ASSUMED_US.values()) * 1e-3
for r in rows:
    r["kernel_only_speedup"] = base / r["rank0_ms_per_step"]
    r["model_ms_per_step"] = r["rank0_ms_per_step"] + (coll if r["world"] > 1 else 0.0)
    r["model_speedup"] = base / r["model_ms_per_step"]
res = {"what": "training step (published N2 network), rank 0's share per world size, measured on ONE GPU; collectives NOT issued",
       "molecule": os.path.basename(mol_f), "steps": steps, "unmeasured_on_hardware": True,
       "assumed_collective_latency_us": ASSUMED_US, "per_world": rows}
print(json.dumps(res))
if out_f:
    with open(out_f, "w") as f:
        json.dump(res, f, indent=1)
