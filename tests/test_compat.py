"""Small drop-in behaviours pinned by vectors recorded from the reference (tests/golden/make_golden.py
``gen_compat_lih``): gradient clipping (energy.py:383-395), the opt-in full-sample ordering quirk Q1
(hamiltonian.py:100-105), a checkpoint written by the reference's own ``save()`` (energy.py:409-443), and the
multi-process guards (same seed on every rank; ranks that sample different tables are detected)."""
import os
import shutil
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, golden
from naqs_amd import packing
from test_nade import make_wf

ADAM = [{'lr': 1e-3, 'betas': (0.9, 0.99), 'weight_decay': 0, 'eps': 1e-15, 'amsgrad': False}, {'lr': 1e-2}]


def _opt(tmp, monkeypatch, **kw):
    import oracle_backend
    from naqs_amd.optimizer import PartialSamplingOptimizer
    if monkeypatch is not None:
        oracle_backend.install(monkeypatch)
    z = golden("compat_LiH.npz")
    cfg = {"cfg_n_hid": 64, "cfg_n_hid_phase": 32, "cfg_n_layer_phase": 2}
    zz = _Fixture(z, cfg)
    hil, wf = make_wf("LiH", zz)
    ham = packing.load_packed(os.path.join(GOLDEN, "ham_LiH.npz"))
    args = dict(n_samples=2000, n_samples_max=1e12, n_unq_samples_min=10, n_unq_samples_max=1e5, log_exact_energy=False,
                wavefunction=wf, qubit_hamiltonian=ham, pre_compute_H=False, n_electrons=4, n_alpha_electrons=2,
                n_beta_electrons=2, normalise_psi=True, grad_clip_factor=None, optimizer=torch.optim.Adam,
                optimizer_args=[dict(a) for a in ADAM], save_loc=str(tmp), pauli_hamiltonian_dtype=np.float64, seed=3)
    args.update(kw)
    return z, hil, wf, PartialSamplingOptimizer(**args)


class _Fixture:
    """compat_LiH.npz + the network shape it was recorded with, in the form ``make_wf`` reads."""

    def __init__(self, z, cfg):
        self._z, self._cfg = z, cfg
        self.files = list(z.files) + list(cfg)

    def __getitem__(self, k):
        return self._cfg[k] if k in self._cfg else self._z[k]


def test_grad_clip_matches_reference(tmp_path, monkeypatch):
    """grad_clip_factor = 0.5: step 0 is not clipped (empty memory -> limit 1e3), steps 1 and 2 are (limit = 0.5 x
    mean of the remembered norms, which themselves are min(limit, norm)): parameters after every step."""
    z, hil, wf, opt = _opt(tmp_path, monkeypatch, grad_clip_factor=0.5)
    assert float(z["clip_factor"]) == 0.5
    states, counts = torch.tensor(z["clip_states"]), torch.tensor(z["clip_counts"])
    keys = hil.state2idx(states).squeeze(-1)
    for step in range(3):
        E, _ = opt._SGD_step(states, keys, None, sample_weights=counts.double() / counts.sum().double())
        assert abs(E - float(z[f"clip_E{step}"])) < 2e-5
        for name, p in wf.model.named_parameters():
            assert np.max(np.abs(p.detach().numpy() - z[f"clip_sd{step}:" + name])) < 3e-5, (step, name)
    hist, n = opt._grad_norms[0]
    norms = z["clip_grad_norms"]
    want = [norms[0], 0.5 * norms[0], 0.5 * np.mean([norms[0], 0.5 * norms[0]])]
    assert n == 3 and np.allclose(hist[:3].numpy(), want, rtol=1e-3)
    # an un-clipped run diverges from the clipped one (the clip really bit)
    z2, hil2, wf2, opt2 = _opt(tmp_path, None, grad_clip_factor=None)
    for step in range(2):
        opt2._SGD_step(states, keys, None, sample_weights=counts.double() / counts.sum().double())
    d = max(np.max(np.abs(p.detach().numpy() - z["clip_sd1:" + name])) for name, p in wf2.model.named_parameters())
    assert d > 1e-4


def test_reference_constructor_default_clip_is_accepted(tmp_path, monkeypatch):
    import inspect
    from naqs_amd.optimizer import OptimizerBase
    assert inspect.signature(OptimizerBase.__init__).parameters["grad_clip_factor"].default == 3     # energy.py:63
    z, hil, wf, opt = _opt(tmp_path, monkeypatch, grad_clip_factor=3)
    states, counts = torch.tensor(z["clip_states"]), torch.tensor(z["clip_counts"])
    opt._SGD_step(states, hil.state2idx(states).squeeze(-1), None, sample_weights=counts.double() / counts.sum().double())
    assert opt._grad_norms[0][1] == 1


def test_full_sample_order_quirk_is_opt_in(tmp_path, monkeypatch):
    """All 225 LiH states in one batch: the reference pairs rows in restricted-basis order with psi in sample order
    (Q1).  Default = the correct E_loc (differs from the reference by Hartrees); opt-in flag = the reference's values."""
    z, hil, wf, opt = _opt(tmp_path, monkeypatch)
    keys = torch.from_numpy(z["q1_keys"].astype(np.int64))
    assert np.array_equal(hil.get_subspace(ret_states=False, ret_idxs=True).numpy().astype(np.uint64),
                          z["q1_restricted_order_keys"])
    psi = torch.tensor(z["q1_psi_f32"])
    ref = z["q1_eloc_c128"]
    e = opt.calculate_local_energy(keys, psi=psi, ret_complex=True)
    assert np.max(np.abs(e - ref)) > 0.5                                     # the quirk is not mirrored by default
    opt.bug_compat_full_sample_order = True
    e = opt.calculate_local_energy(keys, psi=psi, ret_complex=True)
    assert np.max(np.abs(e - ref) / np.maximum(1, np.abs(ref))) < 1e-11
    e_part = opt.calculate_local_energy(keys[:100], psi=psi[:100], ret_complex=True)   # partial batches: untouched
    opt.bug_compat_full_sample_order = False
    assert np.array_equal(e_part, opt.calculate_local_energy(keys[:100], psi=psi[:100], ret_complex=True))


def test_loads_checkpoint_written_by_the_reference(tmp_path, monkeypatch):
    """ckpt_LiH/ was written by the reference's OptimizerBase.save() after the three clipped steps: parameters, Adam
    moments, step counters and the log (keyed by the reference's LogKey enum) must arrive; then training resumes."""
    from naqs_amd.optimizer import LogKey
    z, hil, wf, opt = _opt(tmp_path, monkeypatch)
    shutil.copytree(os.path.join(GOLDEN, "ckpt_LiH"), tmp_path / "ck")
    opt.save_loc = str(tmp_path / "ck")
    opt.load()
    for name, p in wf.model.named_parameters():
        assert np.array_equal(p.detach().numpy(), z["clip_sd2:" + name]), name
    assert (opt.n_steps, opt.n_epochs, opt.run_time) == (3, 3, 1.25)
    assert [s for s, _ in opt.log[LogKey.E_LOC]] == [1, 2, 3]
    assert abs(opt.log[LogKey.E_LOC][2][1] - float(z["clip_E2"])) < 1e-12
    st = opt.optimizer.state_dict()["state"]
    assert len(st) == len(list(wf.model.parameters())) and all(int(v["step"]) == 3 for v in st.values())
    # (block 0's first-layer weight multiplies a constant-zero input, nade.py:509-511: its moments stay zero)
    assert sum(float(v["exp_avg_sq"].abs().sum()) > 0 for v in st.values()) == len(st) - 1
    opt.run(2, output_freq=10 ** 6)
    assert opt.n_steps == 5 and len(opt.log[LogKey.E_LOC]) == 5


# ---- multi-process guards (gloo, world size 2) --------------------------------------------------------------------
def _seed_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from experiments._base import _agree_on_seed
    drawn = _agree_on_seed(-1)                       # "-s -1": every rank must end up with rank 0's draw
    same = _agree_on_seed(123)
    try:
        _agree_on_seed(100 + rank)                   # explicit but different seeds: refused on the odd rank
        clash = False
    except RuntimeError:
        clash = True
    torch.save({"drawn": drawn, "same": same, "clash": clash}, f"{out}.{rank}")
    dist.destroy_process_group()


def _mismatch_worker(rank, world, port, tmp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_backend
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle_backend.install(__import__("naqs_amd.optimizer").optimizer)
    from test_optimizer import make_opt
    z, hil, wf, opt = make_opt("LiH", os.path.join(tmp, f"r{rank}"), None, seed=11 + rank, n_samples=300)   # different generators
    opt.shard_min_rows = 0                           # the sharded step (its proof rides in the accumulator all-reduce)
    states, counts, probs = opt.get_samples()
    keys = hil.state2idx(states).squeeze(-1)
    try:
        opt._SGD_step(states, keys, None, sample_weights=counts.double() / counts.sum().double())
        raised = False
    except RuntimeError as e:
        raised = "different tables" in str(e)
    # the replicated step proves the same thing with its own small all-reduce every `replica_proof_every` steps
    opt._shard_mismatch = None
    opt.shard_min_rows, opt.shard_min_table, opt.replica_proof_every = 10 ** 9, 10 ** 9, 1
    try:
        opt.run(2, output_freq=10 ** 6)
        raised_replicated = False
    except RuntimeError as e:
        raised_replicated = "different tables" in str(e)
    torch.save({"raised": raised, "raised_replicated": raised_replicated, "M": len(keys)}, f"{out}.{rank}")
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_ranks_agree_on_the_seed(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "seed")
    mp.spawn(_seed_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = (torch.load(f"{out}.{r}", weights_only=False) for r in range(2))
    assert r0["drawn"] == r1["drawn"] and r0["same"] == r1["same"] == 123
    assert not r0["clash"] and r1["clash"]


def test_ranks_with_different_tables_are_detected(tmp_path):
    """Two ranks whose samplers are seeded differently shard different tables; the all-reduce sizes still agree, so
    nothing hangs — the step must notice (sample count / key checksum folded into the accumulator all-reduce)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "mm")
    mp.spawn(_mismatch_worker, args=(2, _free_port(), str(tmp_path), out), nprocs=2, join=True)
    r0, r1 = (torch.load(f"{out}.{r}", weights_only=False) for r in range(2))
    assert r0["raised"] and r1["raised"], (r0, r1)
    assert r0["raised_replicated"] and r1["raised_replicated"], (r0, r1)
