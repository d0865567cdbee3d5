"""Host logic of the VMC step on CPU (kernels replaced by the oracle-backed test stand-in):
one _SGD_step against the reference's recorded step, the adaptive sampling loop, checkpoints, and
the multi-process path (gloo, world_size 2) against the single-process result."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, golden
from naqs_amd import packing
from test_nade import ELECTRONS, make_wf

ADAM = [{'lr': 1e-3, 'betas': (0.9, 0.99), 'weight_decay': 0, 'eps': 1e-15, 'amsgrad': False}, {'lr': 1e-2}]


def make_opt(mol, tmp, monkeypatch=None, seed=3, **kw):
    import oracle_backend
    from naqs_amd.optimizer import PartialSamplingOptimizer
    if monkeypatch is not None:
        oracle_backend.install(monkeypatch)
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z)
    N, na, nb = ELECTRONS[mol]
    ham = packing.load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz"))
    args = dict(n_samples=2000, n_samples_max=1e12, n_unq_samples_min=10, n_unq_samples_max=1e5, log_exact_energy=False,
                wavefunction=wf, qubit_hamiltonian=ham, pre_compute_H=False, n_electrons=na + nb, n_alpha_electrons=na,
                n_beta_electrons=nb, normalise_psi=True, grad_clip_factor=None, optimizer=torch.optim.Adam,
                optimizer_args=[dict(a) for a in ADAM], save_loc=str(tmp), pauli_hamiltonian_dtype=np.float64, seed=seed)
    args.update(kw)
    return z, hil, wf, PartialSamplingOptimizer(**args)


@pytest.mark.parametrize("mol", ["LiH", "H2O"])
def test_sgd_step_matches_reference_step(mol, tmp_path, monkeypatch):
    """Same samples, same weights, same Adam: energy, variance and the parameters AFTER the step
    must match what the reference recorded (tests/golden/make_golden.py, energy.py:273-377)."""
    z, hil, wf, opt = make_opt(mol, tmp_path, monkeypatch)
    states = torch.tensor(z["samp_states"])
    counts = torch.tensor(z["samp_counts"])
    keys = hil.state2idx(states).squeeze(-1)
    E, var = opt._SGD_step(states, keys, None, sample_weights=counts.double() / counts.sum().double())
    assert abs(E - float(z["sgd_E"])) < 2e-5 * max(1, abs(E))            # reference rounds E_loc to float32
    assert abs(var - float(z["sgd_Var"])) < 1e-3 * max(1, abs(var))
    assert abs(opt.last_loss.item() - float(z["sgd_loss"])) < 1e-4 * max(1, abs(float(z["sgd_loss"])))
    for name, p in wf.model.named_parameters():
        after = z["sd_after:" + name]
        assert np.max(np.abs(p.detach().numpy() - after)) < 2e-5, name    # lr = 1e-3 first Adam step


def test_run_loop_decreases_energy_and_logs(tmp_path, monkeypatch, capsys):
    from naqs_amd.optimizer import LogKey
    z, hil, wf, opt = make_opt("LiH", tmp_path, monkeypatch, n_samples=20000)
    opt.run(n_epochs=30, save_freq=None, save_final=True, output_freq=10)
    e = [x[1] for x in opt.log[LogKey.E_LOC]]
    assert len(e) == 30 and opt.n_steps == 30 and np.mean(e[-5:]) < np.mean(e[:5]) - 0.05
    assert os.path.exists(tmp_path / "opt_0steps.pth") and os.path.exists(tmp_path / "energy_optimizer.pth")
    assert "Epoch 10" in capsys.readouterr().out
    ck = torch.load(tmp_path / "energy_optimizer.pth", weights_only=False)
    assert set(ck) == {'optimizer:state_dict', 'run_time', 'n_steps', 'n_epochs', 'log', 'sampled_idxs',
                       'wavefunction:fname', 'hamiltonian_fname'}
    # resume
    z2, hil2, wf2, opt2 = make_opt("LiH", tmp_path, None)
    opt2.load()
    assert opt2.n_steps == 30 and len(opt2.log[LogKey.E_LOC]) == 30
    s = torch.tensor(z["eval_states"][:8])
    with torch.no_grad():
        assert torch.allclose(wf.log_psi(s), wf2.log_psi(s))
    opt.save_log(quiet=True)
    import pandas as pd
    df = pd.read_pickle(tmp_path / "log.pkl")
    assert "Iteration" in df.columns and len(df) == 30


def test_adaptive_sample_count(tmp_path, monkeypatch, capsys):
    """get_samples (energy.py:936-971): too few unique samples -> x10; too many -> /10."""
    z, hil, wf, opt = make_opt("LiH", tmp_path, monkeypatch, n_samples=10, n_unq_samples_min=50, n_unq_samples_max=1000)
    states, counts, probs = opt.get_samples()
    assert opt.n_samples > 10 and len(states) >= 50
    assert "increasing batch size" in capsys.readouterr().out
    z, hil, wf, opt = make_opt("LiH", tmp_path, None, n_samples=10 ** 7, n_unq_samples_min=5, n_unq_samples_max=60)
    states, counts, probs = opt.get_samples()
    assert opt.n_samples < 10 ** 7 and len(states) <= 60


def test_shard_bounds_tile():
    from naqs_amd.optimizer import shard_bounds
    for n in (0, 1, 7, 10000, 50001):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            S = -(-n // world)                      # equal padded shards (what the all-gather of the psi table wants)
            assert all(e - s == S for s, e in b if e < n) and all(e - s <= S for s, e in b)


def _worker(rank, world, port, tmp, out, shard_min_rows=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_backend
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle_backend.install(__import__("naqs_amd.optimizer").optimizer)
    z, hil, wf, opt = make_opt("LiH", os.path.join(tmp, f"r{rank}"), None, seed=11, n_samples=5000)
    if shard_min_rows is not None:
        opt.shard_min_rows = shard_min_rows          # 0: every distributed step shards (the branches under test)
    res = []
    for _ in range(3):
        states, counts, probs = opt.get_samples()
        keys = hil.state2idx(states).squeeze(-1)
        res.append(opt._SGD_step(states, keys, None, sample_weights=counts.double() / counts.sum().double()))
    params = torch.cat([p.detach().reshape(-1) for p in wf.model.parameters()])
    if rank == 0:
        torch.save({"res": res, "params": params, "modes": [m for _, m in opt.dist_mode_log]}, out)
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    assert all(torch.equal(g, gathered[0]) for g in gathered), "ranks diverged"
    dist.destroy_process_group()


def test_two_process_step_equals_single_process(tmp_path):
    """world_size 2 over gloo: replicated sampler, row-sharded E_loc + loss, all-reduce of the energy
    accumulators and of the gradient -> same energies / parameters as one process (up to f32 sum order)."""
    import torch.multiprocessing as mp
    import socket
    outs = []
    for world in (1, 2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = str(tmp_path / f"w{world}.pt")
        mp.spawn(_worker, args=(world, port, str(tmp_path / f"w{world}"), out), nprocs=world, join=True)
        outs.append(torch.load(out, weights_only=False))
    for (e1, v1), (e2, v2) in zip(outs[0]["res"], outs[1]["res"]):
        assert abs(e1 - e2) < 1e-6 * max(1, abs(e1)) and abs(v1 - v2) < 1e-5 * max(1, abs(v1))
    assert torch.max(torch.abs(outs[0]["params"] - outs[1]["params"])).item() < 2e-5
    assert outs[1]["modes"] == ["sharded"]


def test_two_process_step_below_the_break_even_is_replicated(tmp_path):
    """The multi-GPU policy: LiH's table (a few dozen rows) is far below `shard_min_rows` rows per rank, so each of the two
    ranks runs the whole single-process step — identical energies and parameters, bit for bit, and no gradient all-reduce."""
    import torch.multiprocessing as mp
    import socket
    outs = []
    for world in (1, 2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = str(tmp_path / f"p{world}.pt")
        mp.spawn(_worker, args=(world, port, str(tmp_path / f"p{world}"), out, None), nprocs=world, join=True)
        outs.append(torch.load(out, weights_only=False))
    assert outs[0]["res"] == outs[1]["res"] and torch.equal(outs[0]["params"], outs[1]["params"])
    assert outs[0]["modes"] == outs[1]["modes"] == ["replicated"]      # (world 1 runs inside a process group of one here)


def test_one_call_loop_adapts_the_sample_count_like_get_samples(tmp_path, monkeypatch, capsys):
    """`_onecall_step` re-implements get_samples' adaptive rules (energy.py:936-971) around `FusedLogPsi.vmc_step`, which
    abandons a step when get_samples would have re-sampled.  Scripted sampler outcomes (unique counts / tree overflows as a
    function of n_samples) drive both: same sequence of sample counts, same messages, same final n_samples."""
    from naqs_amd.nade import MaxBatchSizeExceededError

    def outcome(n_samples, script):
        """-> number of unique samples, or None for a tree overflow"""
        return script(n_samples)

    scripts = [
        lambda n: min(n // 7, 4000),                                    # grows: too few -> x10 ... until enough
        lambda n: None if n > 10 ** 6 else n // 3,                      # overflows above 1e6 -> /10
        lambda n: 3 if n < 10 ** 9 else 50,                             # stays too small up to n_samples_max
        lambda n: None if n >= 10 ** 5 else 2,                          # overflow above, too few below: the rules must not oscillate
        lambda n: 2000,                                                 # fine at once
    ]
    for start in (2000, 10 ** 7):
        for script in scripts:
            logs = []
            for mode in ("get_samples", "onecall"):
                z, hil, wf, opt = make_opt("LiH", tmp_path / mode, monkeypatch, n_samples=start, n_samples_max=10 ** 9,
                                           n_unq_samples_min=100, n_unq_samples_max=3000)
                calls = []
                if mode == "get_samples":
                    def fake_sample(num_samples, **kw):
                        calls.append(int(num_samples))
                        m = outcome(int(num_samples), script)
                        if m is None or m > opt.n_unq_samples_max:
                            raise MaxBatchSizeExceededError
                        t = torch.zeros(m, dtype=torch.int64)
                        return torch.zeros((m, 1)), t, t.float(), t, t.double()
                    monkeypatch.setattr(opt.wavefunction, "sample", fake_sample)
                    opt.use_fused = False
                    opt.get_samples(lazy=True)
                else:
                    class Fake:
                        def vmc_step(self, ham, n_samples, seed, max_unique, m_lo, m_hi, adam=None, keys_out=None):
                            calls.append(int(n_samples))
                            m = outcome(int(n_samples), script)
                            if m is None or m > max_unique:
                                return False, 0, True, None
                            if m < m_lo or m > m_hi:
                                return False, m, False, None
                            t = torch.zeros(m, dtype=torch.int64)
                            return True, m, False, (t, t, t.float(), t.double(), t.float(), t.double(), t.double(), t.float(), t.double())
                    monkeypatch.setattr(opt.wavefunction, "fused", lambda need_phase=True: Fake())
                    monkeypatch.setattr(opt.wavefunction, "fused_repacked", lambda: None)
                    opt.track_sampled_idxs = False
                    opt._onecall_step()
                out = [l for l in capsys.readouterr().out.splitlines() if "unique samples generated" in l or "MaxBatch" in l]
                logs.append((calls, out, opt.n_samples))
            assert logs[0] == logs[1], (start, logs)
            assert len(logs[0][0]) >= 1


def _switch_worker(rank, world, port, tmp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_backend
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle_backend.install(__import__("naqs_amd.optimizer").optimizer)
    z, hil, wf, opt = make_opt("LiH", os.path.join(tmp, f"r{rank}"), None, seed=11, n_samples=5000)
    # a policy under which LiH's table is "big enough" — the switch to the sharded step may then only happen on a count
    # the ranks have proven to share: right after a replica proof (every 3rd step here), never in between
    opt.shard_min_rows, opt.shard_min_table, opt.replica_proof_every = 1, 1, 3
    opt.run(5, output_freq=10 ** 6)
    modes_up = list(opt.dist_mode_log)
    # and back: the table is now "too small"; leaving the sharded step needs the sharded proof to be clean (it is)
    opt.shard_min_rows, opt.shard_min_table = 10 ** 9, 10 ** 9
    opt.run(2, output_freq=10 ** 6)
    params = torch.cat([p.detach().reshape(-1) for p in wf.model.parameters()])
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    torch.save({"modes_up": modes_up, "modes": list(opt.dist_mode_log), "same": all(torch.equal(g, gathered[0]) for g in gathered),
                "n_steps": opt.n_steps}, f"{out}.{rank}")
    dist.destroy_process_group()


def test_ranks_switch_between_replicated_and_sharded_steps_on_an_agreed_count(tmp_path):
    """A rank that enters the sharded step's all-gather while the other still replicates would hang the job.  The
    replicated -> sharded switch is therefore taken only on the all-reduced count of a replica proof (step 3 here, although
    the table was big enough from step 1 on), and both ranks log the same switches at the same steps."""
    import torch.multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sw")
    mp.spawn(_switch_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    r0, r1 = (torch.load(f"{out}.{r}", weights_only=False) for r in range(2))
    assert r0["modes_up"] == r1["modes_up"] == [(0, "replicated"), (3, "sharded")], (r0, r1)
    assert r0["modes"] == r1["modes"] == [(0, "replicated"), (3, "sharded"), (5, "replicated")], (r0, r1)
    assert r0["same"] and r1["same"] and r0["n_steps"] == r1["n_steps"] == 7


def _library_loop_worker(rank, world, port, tmp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_backend
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle_backend.install(__import__("naqs_amd.optimizer").optimizer)
    z, hil, wf, opt = make_opt("LiH", os.path.join(tmp, f"r{rank}"), None, seed=11, n_samples=5000)
    opt.shard_min_rows, opt.shard_min_table, opt.replica_proof_every = 1, 1, 4
    # the replicated ranks take the loop-in-the-library path (naqs_vmc_run on the GPU): here a stand-in that runs the same steps
    # through the per-step calls and books them the way `_library_run` does, so that what is under test is the loop around it
    chunks, proofs = [], []

    def library_run(n_steps):
        chunks.append((opt.n_steps, int(n_steps)))
        for _ in range(int(n_steps)):
            states, counts, probs = opt.get_samples(lazy=True)
            ev = opt._SGD_step(states, opt._sample_keys, None, sample_weights=opt._sample_weights, lazy=True)
            opt.n_steps += 1
            opt.n_epochs += 1
            opt._pending_log.append((opt.n_steps, ev, len(opt._sample_weights), opt.run_time))
        opt._last_M = len(opt._sample_weights)
        return counts, opt._sample_weights, int(n_steps)

    proof = opt._replica_proof

    def replica_proof(keys):
        if opt.n_steps % opt.replica_proof_every == 0:
            proofs.append(opt.n_steps)
        return proof(keys)

    opt._library_run, opt._replica_proof = library_run, replica_proof
    opt._can_onecall = lambda: opt._dist_mode != "sharded"
    opt._can_shard_onecall = lambda: False
    opt._can_run_in_library = lambda: True
    opt.run(10, output_freq=10 ** 6)
    params = torch.cat([p.detach().reshape(-1) for p in wf.model.parameters()])
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    torch.save({"modes": list(opt.dist_mode_log), "chunks": chunks, "proofs": proofs, "n_steps": opt.n_steps,
                "same": all(torch.equal(g, gathered[0]) for g in gathered)}, f"{out}.{rank}")
    dist.destroy_process_group()


def test_library_loop_of_replicated_ranks_stops_for_the_replica_proof(tmp_path):
    """Round-5 advice: replicated ranks run their steps through the loop in the library, chunk by chunk; a chunk must end where
    the ranks owe each other the same-table proof (every `replica_proof_every` steps), the proof must run there, and the
    switch to the sharded step — only ever taken on a proof's agreed count — must stay reachable."""
    import torch.multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "lib")
    mp.spawn(_library_loop_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    r0, r1 = (torch.load(f"{out}.{r}", weights_only=False) for r in range(2))
    for r in (r0, r1):
        assert r["chunks"] == [(0, 1), (1, 3)], r            # the first epoch's own line, then up to the proof at step 4
        assert r["proofs"] == [4] and r["modes"] == [(0, "replicated"), (4, "sharded")], r
        assert r["same"] and r["n_steps"] == 10


def test_step_form_is_decided_again_on_every_run_and_optimizer_reset(tmp_path, monkeypatch):
    """The cached decision (one library call per step or the pieces) must not outlive a run(): use_fused, grad_clip_factor,
    normalize_grads or the optimiser itself may have changed in between (a plain torch optimiser cannot take the one-call
    step at all)."""
    z, hil, wf, opt = make_opt("LiH", tmp_path, monkeypatch, n_samples=2000)
    asked = []
    monkeypatch.setattr(opt, "_can_onecall", lambda: asked.append(1) or False)
    opt.run(2, output_freq=10 ** 6)
    assert len(asked) == 1 and opt._onecall_cached == 0
    opt.run(1, output_freq=10 ** 6)
    assert len(asked) == 2
    opt._onecall_cached = 1
    opt.reset_optimizer()
    assert opt._onecall_cached is None
