"""Scaling model of the TRAINING step without the node: on ONE GPU, rank 0's share of `PartialSamplingOptimizer._SGD_step`
at world sizes 1, 2, 4, 8 — the real world > 1 code path of the optimiser (replicated sampler, forward/backward and E_loc
for the rank's row shard only, the table assembled by an all-gather, two all-reduces) driven through a stand-in for
`torch.distributed` that issues no collective: `all_reduce` multiplies by W (as if every rank had contributed rank 0's
partial sums: ratios such as <E> and the same-table proof are unchanged, the shard gradient stands in for the full one) and `all_gather_into_tensor`
delivers the true table by evaluating log psi of all rows with the inference kernel — GPU work a real rank does not do,
timed separately and subtracted.

usage: python tools/scaling_model.py [molecule npz] [steps] [out.json]
Everything is single-GPU evidence ("unmeasured on hardware" for W > 1).  The collectives enter as bench.py's model: each
one's MEASURED floor at world size 1 (RCCL on this box) plus ring steps at an assumed per-hop latency and one xGMI link's rate."""
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
import bench                                      # measured world-1 collective floors + the stated per-hop model
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
        if self.keys is None or getattr(fused, "_last_shard_M", None) is not None:
            # the four-call sharded step: the keys of this step sit in the step's buffers (recorded by the wrapper below)
            self.keys = fused._shard_bufs["keys_used"][:fused._last_shard_M]
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
    # NAQS_SCALING_PUBLISHED=1: the sample counts of experiments/run.py (1e7 samples, 1e4 .. 1e5 unique) — the regime in which the
    # step is meant to shard (tables of >= 10^4 rows); default: a part-trained network's ~10^3-row tables
    pub = os.environ.get("NAQS_SCALING_PUBLISHED") == "1"
    opt = O.PartialSamplingOptimizer(n_samples=10000000 if pub else 1000000, n_samples_max=1e12, n_unq_samples_min=10000 if pub else 1000,
                                     n_unq_samples_max=1e5, wavefunction=wf,
                                     qubit_hamiltonian=qh, pre_compute_H=False, n_electrons=mol.n_electrons, n_alpha_electrons=na,
                                     n_beta_electrons=nb, optimizer=torch.optim.Adam,
                                     optimizer_args=[{'lr': 1e-3, 'betas': (0.9, 0.99), 'eps': 1e-15}, {'lr': 1e-2}],
                                     save_loc="/tmp/scaling_model", seed=1, grad_clip_factor=None, log_exact_energy=False,
                                     pauli_hamiltonian_dtype=np.float64, normalise_psi=True)
    return wf, opt


WORLDS = [int(x) for x in os.environ.get("NAQS_SCALING_WORLDS", "1,2,4,8").split(",")]
if len(WORLDS) > 1:
    # one fresh process per world size (a real run never changes its world size either: buffers, modes and the sampler's
    # launch hints of one size must not leak into the timing of the next), results merged here
    import subprocess
    rows = []
    for W in WORLDS:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), mol_f, str(steps)], env=dict(os.environ, NAQS_SCALING_WORLDS=str(W)),
                             capture_output=True, text=True, timeout=900)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('{"what"')]
        if not line:
            raise SystemExit(f"world {W} failed:\n{out.stdout[-2000:]}\n{out.stderr[-2000:]}")
        sub = json.loads(line[-1])
        rows += sub["per_world"]
        print(sub["per_world"][0], flush=True)
    base = rows[0]["rank0_ms_per_step"]
    n_params = sub["n_params"]
    m_big = max(r["mean_unique_samples"] for r in rows)
    floor, how = bench.measure_collective_floor(dev, int(m_big) * 8, n_params * 4)
    table = bench.collective_table(WORLDS, int(m_big) * 8, n_params * 4, floor)
    for r in rows:
        coll = sum(table[str(r["world"])].values()) * 1e-3
        r["kernel_only_speedup"] = base / r["rank0_ms_per_step"]
        r["model_ms_per_step"] = r["rank0_ms_per_step"] + (coll if (r["world"] > 1 and r["mode"] == "sharded") else 0.0)
        r["model_speedup"] = base / r["model_ms_per_step"]
    res = dict(sub, per_world=rows)
    res["collective_latency"] = {"world1_floor_us": floor, "world1_floor_source": how,
                                 "model": f"floor + ring steps x ({bench.ASSUMED_HOP_US} us ASSUMED per xGMI hop + (bytes / W) at {bench.XGMI_LINK_GBS} "
                                          "GB/s per link); steps = W - 1 (all-gather), 2 (W - 1) (all-reduce)",
                                 "bytes": {"all_gather_table": int(m_big) * 8, "all_reduce_accumulators": 64, "all_reduce_gradient": n_params * 4},
                                 "per_world_us": table}
    print(json.dumps(res))
    if out_f:
        with open(out_f, "w") as f:
            json.dump(res, f, indent=1)
    raise SystemExit(0)

rows = []
train_first = int(os.environ.get("NAQS_SCALING_TRAIN_FIRST", "1000"))
wf, opt = build()
with contextlib.redirect_stdout(io.StringIO()):
    opt.run(train_first, output_freq=10 ** 9)          # a network part-way into training: peaked distribution, M ~ 10^3
for g in opt.optimizer.param_groups:                 # ... then frozen (lr = 0), so that every world size times the SAME workload
    g["lr"] = 0.0
real_dist, real_step = O._dist, opt._SGD_step
for W in WORLDS:
    fake = FakeDist(W, lambda: wf.fused(need_phase=True)) if W > 1 else None
    if fake is not None:
        O._dist = lambda: fake

        def step(states, states_idx, *a, **k):
            fake.keys = O.keys_to_device(states_idx, dev)
            return real_step(states, states_idx, *a, **k)
        opt._SGD_step = step
        fz = wf.fused(need_phase=True)
        if not hasattr(fz, "_real_ssf"):
            fz._real_ssf = fz.shard_sample_forward

            def ssf(*a, **k):
                r = fz._real_ssf(*a, **k)
                fz._last_shard_M = r[1]
                return r
            fz.shard_sample_forward = ssf
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
        if fake is not None and fake.keys is not None and opt._dist_mode == "sharded":
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
    mode = opt._dist_mode                                  # the optimiser's own policy picked it (shard_min_rows)
    if mode != "sharded":
        t_tab = 0.0                                        # no all-gather happened: nothing to subtract
    rows.append({"world": W, "wall_ms_per_step_incl_standin": dt * 1e3, "standin_table_ms": t_tab * 1e3,
                 "rank0_ms_per_step": (dt - t_tab) * 1e3, "mean_unique_samples": n_unq, "mode": mode,
                 "path": {"single": "single-process step (one library call)",
                          "replicated": "every rank runs the single-GPU step (one library call; 32-byte proof every 64 steps)",
                          "sharded": "sharded step (forward of my rows, all-gather, E_loc of my rows, two all-reduces)"}[mode]})
    print(rows[-1], flush=True)
n_params = int(sum(p.numel() for p in wf.model.parameters()))
res = {"what": "training step (published network), rank 0's share per world size, measured on ONE GPU; collectives NOT issued",
       "shard_min_rows": opt.shard_min_rows, "shard_min_table": opt.shard_min_table, "n_params": n_params,
       "molecule": os.path.basename(mol_f), "steps": steps, "unmeasured_on_hardware": True, "per_world": rows}
print(json.dumps(res))
if out_f:
    with open(out_f, "w") as f:
        json.dump(res, f, indent=1)
