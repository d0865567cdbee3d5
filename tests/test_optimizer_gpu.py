"""The VMC step on the GPU (real kernels): one step against the reference's recorded step, a short
H2O run towards FCI, and the diagnostics (exact energy over the whole space, sampled-subspace H)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_opt_gpu(mol, tmp, **kw):
    from naqs_amd import packing
    from naqs_amd.optimizer import PartialSamplingOptimizer
    from test_nade import ELECTRONS, make_wf
    from test_optimizer import ADAM
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z, device="cuda")
    N, na, nb = ELECTRONS[mol]
    ham = packing.load_packed(os.path.join(GOLDEN, f"ham_{mol}.npz"))
    args = dict(n_samples=100000, n_samples_max=1e12, n_unq_samples_min=10, n_unq_samples_max=1e5, log_exact_energy=False,
                wavefunction=wf, qubit_hamiltonian=ham, pre_compute_H=False, n_electrons=na + nb, n_alpha_electrons=na,
                n_beta_electrons=nb, normalise_psi=True, grad_clip_factor=None, optimizer=torch.optim.Adam,
                optimizer_args=[dict(a) for a in ADAM], save_loc=str(tmp), pauli_hamiltonian_dtype=np.float64, seed=5)
    args.update(kw)
    return z, hil, wf, PartialSamplingOptimizer(**args)


@pytest.mark.parametrize("mol", ["LiH", "H2O", "N2"])
def test_sgd_step_matches_reference_step_on_device(mol, tmp_path):
    z, hil, wf, opt = make_opt_gpu(mol, tmp_path)
    states = torch.tensor(z["samp_states"], device="cuda")
    counts = torch.tensor(z["samp_counts"], device="cuda")
    keys = hil.state2idx(states).squeeze(-1)
    # E_loc through the reference-style entry point (psi re/im float32) == the reference's complex128 E_loc
    lp = torch.tensor(z["samp_log_psi"], device="cuda")
    psi = torch.stack([lp[:, 0].exp() * lp[:, 1].cos(), lp[:, 0].exp() * lp[:, 1].sin()], -1)
    e = opt.calculate_local_energy(keys, psi=psi, ret_complex=True)
    want = z["sgd_eloc_c128"]
    assert np.max(np.abs(e - want) / np.maximum(1, np.abs(want))) < 2e-5      # psi recomputed in float32 on device
    E, var = opt._SGD_step(states, keys, None, sample_weights=counts.double() / counts.sum().double())
    assert abs(E - float(z["sgd_E"])) < 2e-5 * max(1, abs(E))
    assert abs(var - float(z["sgd_Var"])) < 1e-3 * max(1, abs(var))
    for name, p in wf.model.named_parameters():
        assert np.max(np.abs(p.detach().cpu().numpy() - z["sd_after:" + name])) < 2e-5, name


def test_short_h2o_run_approaches_fci_and_diagnostics(tmp_path):
    from naqs_amd.optimizer import LogKey
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    z, hil, wf, opt = make_opt_gpu("H2O", tmp_path, optimizer_args=[{'lr': 5e-3, 'betas': (0.9, 0.99), 'eps': 1e-15},
                                                                     {'lr': 1e-2}], n_samples=1000000)
    opt.run(n_epochs=150, save_freq=None, save_final=False, output_freq=50)
    e = [x[1] for x in opt.log[LogKey.E_LOC]]
    fci = kat["fci"]["H2O"]
    assert np.mean(e[-10:]) < np.mean(e[:10]) - 1.0 and np.mean(e[-10:]) > fci - 1e-3     # variational, improving
    exact = opt.calculate_energy(normalise_psi=True)
    assert fci - 1e-6 < exact < np.mean(e[:10])
    val, _, n_unq = opt.solve_H(n_samps=100000)
    assert fci - 1e-8 <= val <= exact + 1e-6 and n_unq > 10


def test_get_h_matches_oracle(tmp_path):
    from oracle import oracle
    z, hil, wf, opt = make_opt_gpu("LiH", tmp_path)
    keys = z["eval_keys"][:60]
    H = opt.pauli_hamiltonian.get_H(keys.astype(np.int64)).toarray()
    assert np.max(np.abs(H - H.T)) < 1e-13
    h = golden("ham_LiH.npz")
    psi = np.random.RandomState(0).normal(size=60) + 0j
    want = oracle.eloc_matrix_free(h["xy"], h["yz"], h["coeff"], keys, psi)
    got = np.conj(H @ psi / psi)
    assert np.max(np.abs(got - want)) < 1e-10


def test_flat_adam_is_torch_adam(tmp_path):
    """FlatAdam (one HIP launch on the flattened parameters) follows torch.optim.Adam step for step with the reference's
    hyper-parameters, and the two exchange state_dicts."""
    import copy
    from naqs_amd.flat_adam import FlatAdam
    from test_nade import make_wf
    from test_optimizer import ADAM
    z = golden("nade_LiH.npz")
    _, wf_a = make_wf("LiH", z, device="cuda")
    _, wf_b = make_wf("LiH", z, device="cuda")
    flat = wf_a.flatten_parameters()
    pa, pb = list(wf_a.model.parameters()), list(wf_b.model.parameters())
    assert wf_a._views_of(flat, pa)
    args = {k: v for k, v in ADAM[0].items()}
    opt_a = FlatAdam([dict(args, params=pa), {'lr': 1e-2, 'params': []}], flat)
    opt_b = torch.optim.Adam([dict(args, params=pb), {'lr': 1e-2, 'params': []}])
    gen = torch.Generator(device="cuda").manual_seed(0)

    def one_step(opts_params, scale):
        grads = [torch.randn(p.shape, device="cuda", generator=gen) * scale for p in pa]
        for opt, params in opts_params:
            for p, g in zip(params, grads):
                p.grad = g.clone()
            opt.step()
            opt.zero_grad()

    for it in range(20):
        one_step([(opt_a, pa), (opt_b, pb)], 10.0 ** (-it % 7))
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-7), float((a - b).abs().max())
    # state_dict interchange, both directions, then keep stepping together
    sd_a, sd_b = copy.deepcopy(opt_a.state_dict()), copy.deepcopy(opt_b.state_dict())
    assert set(sd_a["param_groups"][0]) == set(sd_b["param_groups"][0])
    assert set(sd_a["state"][0]) == set(sd_b["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    opt_b.load_state_dict(sd_a)
    opt_a.load_state_dict(sd_b)
    for it in range(5):
        one_step([(opt_a, pa), (opt_b, pb)], 1e-3)
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=5e-5, atol=5e-7), float((a - b).abs().max())
    assert int(opt_a.state_dict()["state"][0]["step"]) == 25 and int(opt_b.state_dict()["state"][3]["step"]) == 25


def test_optimizer_uses_flat_adam_and_checkpoint_round_trip(tmp_path):
    from naqs_amd.flat_adam import FlatAdam
    z, hil, wf, opt = make_opt_gpu("LiH", tmp_path)
    assert isinstance(opt.optimizer, FlatAdam)
    opt.run(3, output_freq=1000)
    opt.save()
    w = [p.detach().clone() for p in wf.model.parameters()]
    m = opt.optimizer._m.clone()
    z2, hil2, wf2, opt2 = make_opt_gpu("LiH", tmp_path)
    opt2.load()
    assert all(torch.equal(a, b) for a, b in zip(w, wf2.model.parameters()))
    assert torch.equal(m, opt2.optimizer._m) and opt2.optimizer._t == 3 and opt2.n_steps == 3
    assert wf2._views_of(wf2._flat_params, list(wf2.model.parameters()))      # still flattened after load_state_dict


def test_flat_adam_follows_param_group_edits_after_load_state_dict():
    """load_state_dict replaces the param_group dicts; the learning rate the step uses must be the one of the CURRENT
    dict (the -c resume path, then experiments/_base.py's `g['lr'] = 5e-4` schedule and any LR scheduler)."""
    import copy
    from naqs_amd.flat_adam import FlatAdam
    from test_nade import make_wf
    from test_optimizer import ADAM
    z = golden("nade_LiH.npz")
    _, wf = make_wf("LiH", z, device="cuda")
    flat = wf.flatten_parameters()
    params = list(wf.model.parameters())
    opt = FlatAdam([dict(ADAM[0], params=params), {'lr': 1e-2, 'params': []}], flat)

    def step():
        before = flat.clone()
        for p in params:
            p.grad = torch.ones_like(p)
        opt.step()
        return (flat - before).abs().max().item()

    d0 = step()                                          # first Adam step with g = 1: |delta| = lr
    assert abs(d0 - 1e-3) < 1e-6
    sd = copy.deepcopy(opt.state_dict())
    sd["param_groups"][0]["lr"] = 4e-3                   # a checkpoint written with another learning rate
    opt.load_state_dict(sd)
    assert abs(step() - 4e-3) < 1e-5                     # (constant gradient: m_hat / sqrt(v_hat) stays 1)
    for g in opt.param_groups:
        g['lr'] = 5e-4                                   # the default schedule's second half
    assert abs(step() - 5e-4) < 1e-6
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.1)
    step()
    sched.step()
    assert abs(step() - 5e-5) < 1e-7


def _run_workers(tmp_path, backend, world, port, shard_min_rows=0, tag="", extra_env=None):
    """Start ``world`` fresh child processes of tests/dist_step_worker.py on cuda:0 and collect what they wrote.
    shard_min_rows = 0: every distributed step shards (the branches under test); None: the library's own policy."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NAQS_SHARD_MIN_ROWS", None)
    if shard_min_rows is not None:
        env["NAQS_SHARD_MIN_ROWS"] = str(shard_min_rows)
    env.update(extra_env or {})
    outs = [str(tmp_path / f"{backend}{tag}_{world}_{r}.pt") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "dist_step_worker.py"), backend, str(r), str(world),
                               str(port), outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for pr in procs:
        try:
            logs.append(pr.communicate(timeout=900)[0])
        except subprocess.TimeoutExpired:
            pr.kill()
            logs.append(pr.communicate()[0] + "\n[timeout]")
    assert all(pr.returncode == 0 for pr in procs), "\n".join(log[-2000:] for log in logs)
    return [torch.load(o, weights_only=False) for o in outs]


def test_distributed_step_on_one_gpu(tmp_path):
    """The sharded step on the HIP path against the single-process answer (30 optimiser steps on H2O):
    * nccl, world 1 — RCCL itself carries the accumulator and flat-gradient all-reduces;
    * gloo, world 2 — two processes on the one GPU (RCCL refuses two ranks per device): the world > 1 branches of
      ``_SGD_step`` really execute on the device — table log psi through the inference kernel between the training
      forward and backward, row shards [0, M/2) and [M/2, M), both all-reduces, the shard-consistency proof."""
    single = _run_workers(tmp_path, "none", 1, 29576)[0]
    nccl1 = _run_workers(tmp_path, "nccl", 1, 29577)[0]
    gloo2 = _run_workers(tmp_path, "gloo", 2, 29578)
    e0 = np.array(single["energies"])
    assert len(e0) == 30
    assert np.allclose(nccl1["energies"], e0, rtol=0, atol=1e-6)
    assert torch.equal(gloo2[0]["params"], gloo2[1]["params"]), "ranks diverged"
    # two shards sum their float32 gradients in a different order than one process; the optimisation amplifies that
    # rounding noise step by step (measured: 1e-14 at step 1, < 1e-6 up to step 24, 4e-4 at step 30), so the first 20
    # steps are held to the same energies and the rest to the same trajectory
    d = np.abs(np.array(gloo2[0]["energies"]) - e0)
    assert d[:20].max() < 1e-6 and d.max() < 5e-3, d
    assert torch.max(torch.abs(gloo2[0]["params"] - single["params"])).item() < 2e-2


def test_sharded_step_as_four_library_calls_equals_the_call_by_call_path(tmp_path):
    """The row-sharded step runs as four library calls with the three collectives between them (naqs_vmc_shard_sample_forward
    | all-gather | naqs_eloc_gathered | all-reduce | naqs_net_train_backward_vmc | all-reduce | naqs_vmc_shard_update); round
    3's call-by-call path (NAQS_TRAIN_ONECALL=0) launches the same kernels on the same data with the interpreter in between:
    gloo world 2 on one GPU, every energy and every parameter identical."""
    a = _run_workers(tmp_path, "gloo", 2, 29590, tag="_four")
    b = _run_workers(tmp_path, "gloo", 2, 29591, tag="_cbc", extra_env={"NAQS_TRAIN_ONECALL": "0"})
    assert a[0]["energies"] == b[0]["energies"] and a[1]["energies"] == b[1]["energies"]
    assert torch.equal(a[0]["params"], b[0]["params"]) and torch.equal(a[0]["params"], a[1]["params"])
    assert [m for _, m in a[0]["dist_modes"]] == ["sharded"]


def test_small_tables_are_replicated_not_sharded(tmp_path):
    """The multi-GPU policy (`shard_min_rows`): H2O's tables have a few hundred rows — far below the break-even
    of the sharded step — so with the default policy every rank runs the identical single-GPU step: gloo world 2 on one GPU
    reproduces the single-process trajectory BIT FOR BIT (same launches, same seeds, no collective on the data path), the
    ranks prove to each other that they hold the same table, and the log says which mode ran."""
    single = _run_workers(tmp_path, "none", 1, 29586)[0]
    gloo2 = _run_workers(tmp_path, "gloo", 2, 29587, shard_min_rows=None, tag="_policy")
    assert gloo2[0]["energies"] == single["energies"] and gloo2[1]["energies"] == single["energies"]
    assert torch.equal(gloo2[0]["params"], single["params"]) and torch.equal(gloo2[1]["params"], single["params"])
    assert [m for _, m in gloo2[0]["dist_modes"]] == ["replicated"] and single["dist_modes"] == []


def test_replicated_ranks_in_the_library_loop_prove_their_tables_and_reach_the_sharded_step(tmp_path):
    """Round-5 advice: replicated ranks run `naqs_vmc_run` chunks; a chunk has to end at the replica proof (every 8 steps
    here), the proof has to run, and the replicated -> sharded switch — only taken on a proof's agreed count — has to stay
    reachable under the library loop.  gloo world 2 on one GPU, a policy under which H2O's table counts as big enough: the
    first 8 steps are the single-process trajectory bit for bit, then both ranks shard at step 8 and stay together."""
    single = _run_workers(tmp_path, "none", 1, 29592)[0]
    env = {"NAQS_SHARD_MIN_TABLE": "2", "NAQS_REPLICA_PROOF_EVERY": "8"}
    gloo2 = _run_workers(tmp_path, "gloo", 2, 29593, shard_min_rows=1, tag="_libloop", extra_env=env)
    for r in gloo2:
        assert r["dist_modes"] == [(0, "replicated"), (8, "sharded")], r["dist_modes"]
        assert r["energies"][:8] == single["energies"][:8]
    assert torch.equal(gloo2[0]["params"], gloo2[1]["params"]), "ranks diverged"
    d = np.abs(np.array(gloo2[0]["energies"]) - np.array(single["energies"]))
    assert d[:20].max() < 1e-6 and d.max() < 5e-3, d


def test_fused_kernels_follow_parameter_changes(tmp_path):
    """load_state_dict / in-place edits / checkpoint loads must reach the packed weights of the HIP kernels (their
    copy is refreshed from the tensors' version counters), also once the parameters are views of the flat buffer."""
    from test_nade import make_wf
    z = golden("nade_LiH.npz")
    hil, wf = make_wf("LiH", z, device="cuda")
    keys = torch.as_tensor(np.sort(hil._all_keys()), device="cuda")
    wf.flatten_parameters()
    a = wf.fused().log_psi(keys).clone()
    sd = {k: v.clone() for k, v in wf.model.state_dict().items()}
    with torch.no_grad():
        for p in wf.model.parameters():
            p.mul_(1.01)
    b = wf.fused().log_psi(keys).clone()
    assert not torch.allclose(a, b)
    wf.model.load_state_dict(sd)
    assert torch.equal(wf.fused().log_psi(keys), a)
    fname = str(tmp_path / "wf.pth")
    wf.save(fname, quiet=True)
    with torch.no_grad():
        for p in wf.model.parameters():
            p.add_(0.01)
    wf.load(fname)
    assert torch.equal(wf.fused().log_psi(keys), a)
    ref = wf.log_psi(hil.idx2state(keys)).reshape(-1, 2)
    assert torch.max(torch.abs(ref - a)).item() < 5e-5


@pytest.mark.parametrize("mol,kw", [("N2", {}), ("LiH", dict(n_samples=100, n_unq_samples_min=20)),
                                     ("H2O", dict(n_samples=1000000, n_unq_samples_max=120, n_unq_samples_min=5))])
def test_one_call_training_step_equals_the_step_by_step_loop(mol, kw, tmp_path, monkeypatch, capsys):
    """``naqs_vmc_step`` (sampling ... Adam ... re-pack in one library call) against get_samples + _SGD_step: the same
    launches in the same order with the same seeds, so energies, sample counts and parameters agree bit for bit — also
    through the adaptive sample count (too few unique samples: x10; tree overflow: /10) and an LR scheduler."""
    from naqs_amd.optimizer import LogKey
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NAQS_TRAIN_ONECALL", mode)
        z, hil, wf, opt = make_opt_gpu(mol, tmp_path / mode, scheduler=torch.optim.lr_scheduler.StepLR,
                                       scheduler_args=dict(step_size=7, gamma=0.5), **kw)
        assert opt._can_onecall() == (mode == "1")
        opt.sampled_ring_elems = 1          # (the smallest tracking buffer: two slots, folded into the Counter every other step)
        opt.run(n_epochs=25, save_freq=None, save_final=False, output_freq=10)
        out = capsys.readouterr().out
        runs[mode] = dict(e=np.array(opt.log[LogKey.E_LOC]), v=np.array(opt.log[LogKey.E_LOC_VAR]),
                          n=np.array(opt.log[LogKey.N_UNIQUE_SAMP]), p=wf.flatten_parameters().clone(), ns=opt.n_samples,
                          t=opt.optimizer._t, lr=opt.optimizer.param_groups[0]['lr'], idx=dict(opt.sampled_idxs),
                          msgs=[l for l in out.splitlines() if "unique samples generated" in l or "MaxBatch" in l],
                          sd=opt.optimizer.state_dict()['state'][0]['exp_avg'].clone(), loss=float(opt.last_loss))
    a, b = runs["1"], runs["0"]
    assert np.array_equal(a["e"], b["e"]) and np.array_equal(a["v"], b["v"]) and np.array_equal(a["n"], b["n"])
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["sd"], b["sd"])
    assert (a["ns"], a["t"], a["lr"], a["idx"], a["msgs"], a["loss"]) == (b["ns"], b["t"], b["lr"], b["idx"], b["msgs"], b["loss"])
    assert a["t"] == 25 and np.isfinite(a["e"]).all()
    if kw:
        assert a["msgs"], "the adaptive sample count was meant to act in this case"


@pytest.mark.parametrize("mol,kw", [("N2", {}), ("LiH", dict(n_samples=100, n_unq_samples_min=20)),
                                     ("H2O", dict(n_samples=1000000, n_unq_samples_max=120, n_unq_samples_min=5)),
                                     ("N2", dict(defer=True))])
def test_training_loop_in_the_library_equals_the_step_by_step_loop(mol, kw, tmp_path, monkeypatch, capsys):
    """``naqs_vmc_run`` — the loop of PartialSamplingOptimizer.run (energy.py:975-1008) with get_samples' adaptive sample
    count (energy.py:936-971) in C, chunks of steps between the loop's own events (first line, output lines, checkpoints,
    folds of the tracking buffer) — against one ``naqs_vmc_step`` per step: same seeds, same launches, so energies, sample
    counts, messages, sampled-state counts and parameters agree bit for bit."""
    from naqs_amd.optimizer import LogKey
    runs = {}
    kw = dict(kw)
    if kw.pop("defer", False):
        # the phase MLP's half of each step on a second stream beside the next step's sampler (off by default: slower on this
        # pool) — the same kernels on the same operands, joined before naqs_vmc_run returns
        monkeypatch.setenv("NAQS_DEFER_PHASE", "1")
    for mode in ("1", "0"):
        monkeypatch.setenv("NAQS_TRAIN_RUN", mode)
        z, hil, wf, opt = make_opt_gpu(mol, tmp_path / mode, **kw)
        assert opt._can_onecall() and opt._can_run_in_library() == (mode == "1")
        opt.sampled_ring_elems = 1          # (the smallest tracking buffer: the library run stops to let it be folded)
        opt.run(n_epochs=25, save_freq=7, save_final=False, output_freq=10)
        opt.run(n_epochs=6, save_freq=None, save_final=False, output_freq=4)
        out = capsys.readouterr().out
        runs[mode] = dict(e=np.array(opt.log[LogKey.E_LOC]), v=np.array(opt.log[LogKey.E_LOC_VAR]),
                          n=np.array(opt.log[LogKey.N_UNIQUE_SAMP]), p=wf.flatten_parameters().clone(), ns=opt.n_samples,
                          t=opt.optimizer._t, idx=dict(opt.sampled_idxs), calls=wf._sample_calls,
                          msgs=[l for l in out.splitlines() if "unique samples generated" in l or "MaxBatch" in l],
                          lines=[l.split(": <E>")[0] for l in out.splitlines() if l.startswith("Epoch ")],
                          sd=opt.optimizer.state_dict()['state'][0]['exp_avg'].clone(), loss=float(opt.last_loss),
                          saved=sorted(f for f in os.listdir(tmp_path / mode) if f.startswith("opt_")),
                          keys=opt._sample_keys.clone(), w=opt._sample_weights.clone())
    a, b = runs["1"], runs["0"]
    assert np.array_equal(a["e"], b["e"]) and np.array_equal(a["v"], b["v"]) and np.array_equal(a["n"], b["n"])
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["sd"], b["sd"])
    assert torch.equal(a["keys"], b["keys"]) and torch.equal(a["w"], b["w"])
    for k in ("ns", "t", "idx", "calls", "msgs", "lines", "loss", "saved"):
        assert a[k] == b[k], k
    assert a["t"] == 31 and np.isfinite(a["e"]).all() and len(a["lines"]) == 4      # epochs 1, 10, 20 and 28 (= 25 + 3: every 4th of the second run)
    if kw:
        assert a["msgs"], "the adaptive sample count was meant to act in this case"
    # whatever was deferred inside the run has been joined: parameters read right after it are the updated ones
    assert torch.equal(a["p"][-4:], b["p"][-4:])


@pytest.mark.parametrize("mol,kw", [("N2", {}), ("H2O", {}),
                                    # tree overflows and "too few samples" between accepted draws: abandoned draws whose forward
                                    # pass (and hosted finish job) went ahead for nothing
                                    ("N2", dict(n_samples=1000000, n_unq_samples_max=400, n_unq_samples_min=5))])
def test_forward_launched_ahead_of_M_changes_nothing(mol, kw, tmp_path, monkeypatch, capsys):
    """``naqs_vmc_step`` queues the training forward behind the sampler's launches BEFORE the host knows the number of unique
    samples (the kernel reads M on the device; the launch covers the last accepted M plus an eighth, in the kernel form that M
    gets) and launches it again the ordinary way when the real M does not fit or gets another form.  Ahead, not ahead
    (NAQS_SPEC_FORWARD=0) and ahead-but-never-fitting (NAQS_DEBUG_SPEC_SHRINK=1: the launch covers half the last M, every
    step falls back) are the same kernel on the same rows: energies, sample counts and parameters agree bit for bit.
    The reference's loop is synchronous (energy.py:975-998); this is the drop-in's way of hiding its one host round trip."""
    import ctypes
    from naqs_amd import _lib
    from naqs_amd.optimizer import LogKey
    runs = {}
    for mode, env in (("ahead", {"NAQS_SPEC_FORWARD": "1"}), ("behind", {"NAQS_SPEC_FORWARD": "0"}),
                      ("misfit", {"NAQS_SPEC_FORWARD": "1", "NAQS_DEBUG_SPEC_SHRINK": "1"})):
        monkeypatch.delenv("NAQS_DEBUG_SPEC_SHRINK", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        z, hil, wf, opt = make_opt_gpu(mol, tmp_path / mode, **kw)
        assert opt._can_onecall()
        opt.run(n_epochs=30, save_freq=None, save_final=False, output_freq=10)
        out = capsys.readouterr().out
        if kw:
            assert "MaxBatchSizeExceededError" in out, "the adaptive sample count was meant to act in this case"
        c = (ctypes.c_int64 * 2)()
        _lib.check(_lib.load_library().naqs_net_spec_counts(wf._fused._h, c), "naqs_net_spec_counts")
        runs[mode] = dict(e=np.array(opt.log[LogKey.E_LOC]), v=np.array(opt.log[LogKey.E_LOC_VAR]),
                          n=np.array(opt.log[LogKey.N_UNIQUE_SAMP]), p=wf.flatten_parameters().clone(), counts=(c[0], c[1]),
                          ns=opt.n_samples)
    a = runs["ahead"]
    for other in ("behind", "misfit"):
        b = runs[other]
        assert np.array_equal(a["e"], b["e"]) and np.array_equal(a["v"], b["v"]) and np.array_equal(a["n"], b["n"]), other
        assert torch.equal(a["p"], b["p"]) and a["ns"] == b["ns"], other
    assert np.isfinite(a["e"]).all()
    if kw:
        return
    if mol == "H2O":      # (this fixture's phase MLP is not the published 512 x 512 shape: no wave-specialised kernel, nothing goes ahead)
        assert a["counts"] == (0, 0) and runs["misfit"]["counts"] == (0, 0)
        return
    # all but the first step go ahead; most stand even in these first steps from a random start, where M moves fastest
    assert a["counts"][0] >= 25 and a["counts"][1] >= a["counts"][0] // 2, a["counts"]
    assert runs["behind"]["counts"] == (0, 0)
    m = runs["misfit"]["counts"]
    assert m[0] >= 25 and m[1] < a["counts"][1] and m[1] <= m[0] // 4, (m, a["counts"])      # (stands only where M halved)


def test_training_run_is_reproducible_at_large_tables(tmp_path, monkeypatch):
    """Two identically seeded Li2O runs (tables of 10^3 .. 3 x 10^4 unique samples in the first steps: sampler launches with
    more workgroups than are resident at once) give the same energies and sample counts bit for bit.  A two-level sampler
    launch once wrote its output over its own input half of the ping-pong arrays; only this kind of run saw it."""
    from naqs_amd import packing
    from naqs_amd.hilbert import Encoding, Hilbert
    from naqs_amd.optimizer import LogKey, PartialSamplingOptimizer
    from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
    ham = packing.load_packed(os.path.join(GOLDEN, "ham_Li2O.npz"))
    na, nb = int(ham.n_alpha), int(ham.n_beta)
    logs = []
    for rep, multi3_max in enumerate(("2048", "2048", "16384")):
        monkeypatch.setenv("NAQS_SAMPLE_MULTI3_MAX", multi3_max)       # (16384: also the cut with up to 4 096 workgroups per launch)
        torch.manual_seed(1)
        hil = Hilbert.get(int(ham.n_qubits), na, nb, encoding=Encoding.SIGNED)
        wf = NAQSComplex_NADE_orbitals(hil, device="cuda", qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512],
                                       use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=False,
                                       n_alpha_electrons=na, n_beta_electrons=nb)
        opt = PartialSamplingOptimizer(n_samples=1000000, n_samples_max=1e12, n_unq_samples_min=1000, n_unq_samples_max=1e5,
                                       wavefunction=wf, qubit_hamiltonian=ham, pre_compute_H=False, n_electrons=na + nb,
                                       n_alpha_electrons=na, n_beta_electrons=nb, optimizer=torch.optim.Adam,
                                       optimizer_args=[{'lr': 1e-3, 'betas': (0.9, 0.99), 'eps': 1e-15}, {'lr': 1e-2}],
                                       save_loc=str(tmp_path / str(rep)), seed=1, grad_clip_factor=None, log_exact_energy=False,
                                       pauli_hamiltonian_dtype=np.float64, normalise_psi=True)
        opt.run(n_epochs=60, save_freq=None, save_final=False, output_freq=10 ** 9)
        logs.append((np.array(opt.log[LogKey.E_LOC]), np.array(opt.log[LogKey.N_UNIQUE_SAMP])))
    assert logs[0][1][:, 1].max() > 20000
    for e, n in logs[1:]:
        assert np.array_equal(n, logs[0][1]) and np.array_equal(e, logs[0][0])


def test_pending_phase_repack_is_finished_by_whoever_comes_next(tmp_path, monkeypatch):
    """naqs_vmc_step leaves the re-pack of the updated parameters pending: the next sampler call's first launch hosts it
    (workgroups beside the one that samples; round 6, NAQS_PACK_OVERLAP=2: the amplitude blocks' share too, that launch's first
    workgroup packing the fragments of its own four pairs itself — round 4, =1: the phase layers' share only, the amplitude jobs
    as a launch at the end of the step), any other reader starts what it reads first, a new re-pack supersedes it.  Each exit
    against NAQS_PACK_OVERLAP=0 (everything in order), bit for bit: log psi right after training steps (reader first), the
    amplitude blocks alone through the autograd pair naqs_net_logamp / naqs_net_amp_backward (reader of the amplitude share
    only), after a sampler call (hosted), a sampler call in the VALU form of the block MLPs (which cannot host the amplitude
    share), and after loading other parameters (superseded); and the training trajectory itself."""
    from naqs_amd.fused import _LogAmp
    from naqs_amd.hamiltonian import keys_to_device
    from naqs_amd.optimizer import LogKey
    res = {}
    for mode in ("2", "1", "0"):
        monkeypatch.setenv("NAQS_PACK_OVERLAP", mode)
        z, hil, wf, opt = make_opt_gpu("N2", tmp_path / mode)
        assert opt._can_onecall()
        keys = keys_to_device(z["eval_keys"].astype(np.int64), "cuda")
        fused = wf.fused()
        opt.run(n_epochs=3, save_freq=None, save_final=False, output_freq=10 ** 9)
        a = fused.log_psi(keys).clone()                                  # reader first: the pending jobs run in order on its stream
        opt.run(n_epochs=2, save_freq=None, save_final=False, output_freq=10 ** 9)
        amp_params = [q.detach().clone().requires_grad_(True) for q in fused._amp_params]
        la = _LogAmp.apply(fused, keys, *amp_params)                     # reader of the amplitude blocks alone (forward) ...
        opt.run(n_epochs=1, save_freq=None, save_final=False, output_freq=10 ** 9)
        la2 = _LogAmp.apply(fused, keys, *amp_params)
        opt.run(n_epochs=1, save_freq=None, save_final=False, output_freq=10 ** 9)   # (pending again when the backward comes)
        la2.sum().backward()                                             # ... and backward (gradient at the parameters of NOW)
        ga = torch.cat([q.grad.reshape(-1) for q in amp_params]).clone()
        g = torch.Generator(device="cuda").manual_seed(5)
        states, counts, probs, lp = wf.sample(100000, generator=g)       # a sampler call outside the step hosts them
        b = fused.log_psi(keys).clone()
        opt.run(n_epochs=2, save_freq=None, save_final=False, output_freq=10 ** 9)
        monkeypatch.setenv("NAQS_SAMPLE_MFMA", "0")                      # VALU form: reads the rows, cannot pack its own fragments
        g = torch.Generator(device="cuda").manual_seed(6)
        states_v, counts_v, probs_v, lp_v = wf.sample(100000, generator=g)
        monkeypatch.delenv("NAQS_SAMPLE_MFMA")
        opt.run(n_epochs=2, save_freq=None, save_final=False, output_freq=10 ** 9)
        p_now = wf.flatten_parameters().clone()
        with torch.no_grad():
            for p in wf.model.parameters():
                p.mul_(0.5)
        fused.refresh()                                                  # a new re-pack supersedes the pending one
        c = fused.log_psi(keys).clone()
        res[mode] = dict(a=a, b=b, c=c, e=np.array(opt.log[LogKey.E_LOC]), p=p_now, counts=counts.clone(), lp=lp.detach().clone(),
                         la=la.detach().clone(), la2=la2.detach().clone(), ga=ga, counts_v=counts_v.clone(), lp_v=lp_v.detach().clone())
    for m in ("2", "1"):
        x, y = res[m], res["0"]
        assert np.array_equal(x["e"], y["e"]) and torch.equal(x["p"], y["p"]), m
        for k in ("a", "b", "c", "counts", "lp", "la", "la2", "ga", "counts_v", "lp_v"):
            assert torch.equal(x[k], y[k]), (m, k)
        assert not torch.equal(x["a"], x["b"]) and not torch.equal(x["la"], x["la2"])     # (the parameters did move between the read-outs)
