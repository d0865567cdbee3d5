"""Experiment harness: the counterpart of the reference's experiments/_base.py (flags :441-554,
assembly :32-390) for the MI355X path.  Every command-line flag of the reference is accepted with
the same spelling (single-dash long options), default and meaning; flags that select features
outside the hot path (look-up-table blocks, combined amp/phase blocks, phase symmetry, cached
Hamiltonian files, excitation limits, pre-training) are parsed and rejected with a clear message
when switched on.

Multi-GPU: launch under ``torch.distributed.run`` — each rank binds to ``LOCAL_RANK``; the unique
samples are sharded over the ranks (naqs_amd.optimizer).  ``--farm`` instead gives every rank its own
molecule from a comma-separated ``-m`` list (the N2 bond-dissociation sweep, one geometry per GPU,
like experiments/bash/naqs/N2_energy_surface.sh pins one run per CUDA_VISIBLE_DEVICES).
"""
import argparse
import os
import threading
import time

import numpy as np
import torch

from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.nade import InputEncoding, NadeMasking, SoftmaxLogProbAmps
from naqs_amd.optimizer import LogKey, PartialSamplingOptimizer
from naqs_amd.system import load_molecule, mk_dir, set_global_seed
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals

_EXP_BASE_NAME = "data/naqs"

# (flags, keyword of get_parser that supplies the default, argparse kwargs)
_VALUE_FLAGS = [
    (("-m", "--molecule"), "molecule", dict(help="The molecule folder")),
    (("-hf", "--hamiltonian_fname"), "hamiltonian_fname", dict(help="The qubit hamiltonian pkl file location.")),
    (("-o", "--out"), "out", dict(help="The output folder")),
    (("-n", "--number"), "number", dict(type=int, help="The number of experimental runs")),
    (("-qo", "--qubit_ordering"), "qubit_ordering", dict(type=int, help="Qubit ordering (+/-1)")),
    (("-l", "--load"), "pretrained_model_loc", dict(help="The (optional) location of a pre-trained model to load.")),
    (("-n_samps",), "n_samps", dict(type=int, help="The (initial) number of samples per batch")),
    (("-n_samps_max",), "n_samps_max", dict(type=int, help="The maximum of samples per batch")),
    (("-n_unq_samps_max",), "n_unq_samps_max", dict(type=int, help="The maximum number of unique samples per batch")),
    (("-n_unq_samps_min",), "n_unq_samps_min", dict(type=int, help="The minimum number of unique samples per batch")),
    (("-lr",), "lr", dict(type=float, help="The learning rate.")),
    (("-lr_lut",), "lr_lut", dict(type=float, help="The lut learning rate.")),
    (("-n_train",), "n_train", dict(type=int, help="The number of training epochs.")),
    (("-n_pretrain",), "n_pretrain", dict(type=int, help="The number of pre-training epochs.")),
    (("-n_lut",), "n_lut", dict(type=int, help="The number of luts.")),
    (("-n_hid",), "n_hid", dict(type=int, help="The number of hidden units per layer.")),
    (("-n_layer",), "n_layer", dict(type=int, help="The number of layers.")),
    (("-n_hid_phase",), "n_hid_phase", dict(type=int, help="Hidden units per layer of the phase network (-1: as amplitude).")),
    (("-n_layer_phase",), "n_layer_phase", dict(type=int, help="Layers of the phase network (-1: as amplitude).")),
    (("-output_freq",), "output_freq", dict(type=int, help="The logging frequency (in epochs).")),
    (("-save_freq",), "save_freq", dict(type=int, help="The saving frequency (in epochs).")),
    (("-n_excitations_max",), "n_excitations_max", dict(type=int, help="Maximum number of excitations.")),
    (("-s", "--seed"), "seed", dict(type=int, help="Training seed.")),
]
# (flags, dest, default expression on the get_parser keywords, help)
_SWITCHES = [
    (("-c", "--cont"), "cont", lambda k: k["cont"], "Continue previous training run if possible."),
    (("-r", "--resetOpt"), "resetOpt", lambda k: k["reset_opt"], "Reset the parameter optimizer."),
    (("-weight_by_psi",), "weight_by_psi", lambda k: k["reweight_samples_by_psi"], "Reweight samples by |psi|^2."),
    (("-no_mask_psi",), "no_mask_psi", lambda k: k["no_mask_psi"], "Do not mask the wavefunction to the restricted space."),
    (("-full_mask_psi",), "full_mask_psi", lambda k: k["full_mask_psi"], "Mask every conditional to the restricted space."),
    (("-loadH",), "loadH", lambda k: k["load_hamiltonian"], "Load the Hamiltonian from file."),
    (("-overwriteH",), "overwriteH", lambda k: k["overwrite_hamiltonian"], "Save the Hamiltonian to a file."),
    (("-presolveH",), "presolveH", lambda k: k["presolve_hamiltonian"], "Pre-solve the full Hamiltonian."),
    (("-comb_amp_phase",), "comb_amp_phase", lambda k: k["comb_amp_phase"], "Combine amplitude and phase blocks."),
    (("-no_amp_sym",), "no_amp_sym", lambda k: not k["use_amp_spin_sym"], "Neglect amplitude exchange symmetry."),
    (("-phase_sym",), "phase_sym", lambda k: k["use_phase_spin_sym"], "Apply phase exchange symmetry."),
    (("-single_phase",), "single_phase", lambda k: not k["aggregate_phase"], "Use only a single phase block."),
    (("-no_restrictedH",), "no_restrictedH", lambda k: not k["restrict_H"], "Do not restrict the ansatz to physical states."),
    (("-v", "--verbose"), "verbose", lambda k: k["verbose"], "Verbose logging."),
]
_DEFAULTS = dict(molecule="molecules/H2", hamiltonian_fname=None, out=None, number=1, qubit_ordering=-1, lr=-1,
                 lr_lut=1e-2, n_samps=1e6, n_samps_max=1e12, n_unq_samps_min=50000, n_unq_samps_max=1e5,
                 reweight_samples_by_psi=False, no_mask_psi=False, full_mask_psi=False, n_train=5000, n_pretrain=0,
                 n_lut=0, n_hid=32, n_layer=1, n_hid_phase=-1, n_layer_phase=-1, output_freq=25, save_freq=-1,
                 load_hamiltonian=False, overwrite_hamiltonian=False, presolve_hamiltonian=False,
                 pretrained_model_loc=None, cont=False, n_excitations_max=-1, comb_amp_phase=False,
                 use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=True, restrict_H=True,
                 reset_opt=False, verbose=False, seed=-1)


def get_parser(**overrides):
    unknown = set(overrides) - set(_DEFAULTS)
    if unknown:
        raise TypeError(f"unknown defaults: {sorted(unknown)}")
    k = dict(_DEFAULTS, **overrides)
    p = argparse.ArgumentParser(description="Run experimental script.", allow_abbrev=True)
    for flags, key, kw in _VALUE_FLAGS:
        default = k[key]
        if kw.get("type") is int and isinstance(default, float):
            default = int(default)
        p.add_argument(*flags, nargs="?", default=default, **kw)
    for flags, dest, default, help_ in _SWITCHES:
        p.add_argument(*flags, dest=dest, default=bool(default(k)), action="store_true", help=help_)
    p.add_argument("--farm", action="store_true",
                   help="one molecule of a comma-separated -m list per rank (no communication)")
    p.add_argument("--per-gpu", dest="per_gpu", type=int, default=1,
                   help="farm mode: independent runs sharing one GPU (rank r runs on device r // per_gpu)")
    p.add_argument("--seeds", dest="seeds", type=str, default=None,
                   help="farm mode: comma-separated seeds; the jobs are every (molecule, seed) pair (the reference's "
                        "five-seeds-per-molecule protocol, batch_train.sh:11-15)")
    p.add_argument("--gpus", dest="farm_gpus", type=int, default=0,
                   help="farm mode without a launcher: GPUs to use (default: all visible); per_gpu x gpus jobs run at a time")
    return p


def _samp_str(n):
    return (f"{int(n)}" if n < 1e3 else f"{int(n / 1e3)}k" if n < 1e6 else f"{int(n / 1e6)}M" if n < 1e9
            else f"{int(n / 1e9)}B")


def _mol_name(path):
    return os.path.splitext(os.path.split(os.path.normpath(path))[-1])[0]


def _setup_distributed(farm, per_gpu=1):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if torch.cuda.is_available():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        # farm mode: `per_gpu` independent runs share a device (late-training steps launch ~80 workgroups on a 256-CU chip);
        # the self-launching farm names the device outright
        dev = int(os.environ["NAQS_FARM_DEVICE"]) if "NAQS_FARM_DEVICE" in os.environ else (local // max(1, per_gpu) if farm else local)
        torch.cuda.set_device(dev)
    if world > 1 and not farm:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    return rank, world


def _agree_on_seed(seed):
    """Sharded multi-GPU runs replicate the network and the sampler on every rank, so every rank must seed
    identically: with ``-s -1`` (draw a seed) rank 0 draws it and broadcasts it; an explicit seed is checked to be
    the same everywhere.  Farm mode and single-process runs are untouched."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seed
    import random
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = int(seed) if seed >= 0 else random.randint(0, 2 ** 32)
    t = torch.tensor([mine], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0)
    agreed = int(t.item())
    if seed >= 0 and agreed != int(seed):
        raise RuntimeError(f"rank {dist.get_rank()}: seed {seed} differs from rank 0's {agreed}; sharded runs need "
                           "one seed for all ranks")
    return agreed


def _broadcast_parameters(wavefunction):
    """Belt and braces after seeding: every rank starts from rank 0's parameters."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        with torch.no_grad():
            for p in wavefunction.model.parameters():
                dist.broadcast(p.data, src=0)
        wavefunction.parameters_changed()


def open_shell_amp_spin_sym(n_alpha, n_beta, use_amp_spin_sym):
    """The reference's rule (experiments/_base.py:110-114, restrict_to_ms=True): m_s = |n_alpha - n_beta| // 2; only m_s != 0
    switches the amplitude spin symmetry off (and says so).  Integer division: a doublet has m_s == 0 and keeps the flag."""
    m_s = abs(int(n_alpha) - int(n_beta)) // 2
    if m_s != 0:
        print("S!=0 and we are restricting ourselves to ms=S --> turning off use_amp_spin_sym as this is not helpful.")
        return False
    return use_amp_spin_sym


def _run(molecule_fname, hamiltonian_fname, exp_name, num_experiments, pretrained_model_loc, continue_experiment,
         reset_optimizer, qubit_ordering, masking, lr, lr_lut, n_samps, n_samps_max, n_unq_samps_min, n_unq_samps_max,
         reweight_samples_by_psi, n_train, n_pretrain, output_freq, save_freq, n_lut, n_hid, n_layer, n_hid_phase,
         n_layer_phase, n_excitations_max, comb_amp_phase, use_amp_spin_sym, use_phase_spin_sym, aggregate_phase,
         use_restrictedH, loadH, presolveH, overwrite_pauli_hamiltonian, verbose, seed, device=None):
    # (-phase_sym runs on the HIP kernels since round 5, -comb_amp_phase as PyTorch modules on the device — no published script
    # uses either; -n_pretrain is
    # OptimizerBase.pre_flatten; -weight_by_psi is accepted and, as in the reference, has no effect on this optimiser:
    # PartialSamplingOptimizer forces reweight_samples_by_psi = False, energy.py:744)
    rejected = [name for name, on in (("-n_lut", n_lut), ("-loadH", loadH), ("-overwriteH", overwrite_pauli_hamiltonian),
                                      ("-n_excitations_max", n_excitations_max is not None)) if on]
    if rejected:
        raise NotImplementedError("options outside the MI355X hot path: " + ", ".join(rejected))
    # Everything that draws from the process-wide random generators (seeding, parameter initialisation, the optimiser's own
    # generator) happens under one lock: the farm's `--per-gpu k` runs k jobs as threads of one process, and a run must be the
    # same run whether or not it has neighbours.  The training loop itself only uses the optimiser's generator.
    _SETUP_LOCK.acquire()
    locked = [True]
    try:
        return _run_locked(locked, molecule_fname, hamiltonian_fname, exp_name, num_experiments, pretrained_model_loc,
                           continue_experiment, reset_optimizer, qubit_ordering, masking, lr, lr_lut, n_samps, n_samps_max,
                           n_unq_samps_min, n_unq_samps_max, n_train, n_pretrain, output_freq, save_freq, n_lut, n_hid, n_layer,
                           n_hid_phase, n_layer_phase, comb_amp_phase, use_amp_spin_sym, use_phase_spin_sym, aggregate_phase,
                           use_restrictedH, presolveH, verbose, seed, device)
    finally:
        if locked[0]:
            _SETUP_LOCK.release()


_SETUP_LOCK = threading.RLock()


def _run_locked(locked, molecule_fname, hamiltonian_fname, exp_name, num_experiments, pretrained_model_loc, continue_experiment,
                reset_optimizer, qubit_ordering, masking, lr, lr_lut, n_samps, n_samps_max, n_unq_samps_min, n_unq_samps_max,
                n_train, n_pretrain, output_freq, save_freq, n_lut, n_hid, n_layer, n_hid_phase, n_layer_phase, comb_amp_phase,
                use_amp_spin_sym, use_phase_spin_sym, aggregate_phase, use_restrictedH, presolveH, verbose, seed, device):
    seed = set_global_seed(_agree_on_seed(seed))
    molecule, qubit_hamiltonian = load_molecule(molecule_fname, hamiltonian_fname=hamiltonian_fname, verbose=True)
    N = molecule.n_qubits
    results = []
    for i in range(num_experiments):
        if not locked[0]:
            _SETUP_LOCK.acquire()
            locked[0] = True
        print(f"\nRunning experiment {i + 1}/{num_experiments}")
        exp_name_i = exp_name + (f"_{i}" if num_experiments > 1 else "")
        n_alpha, n_beta = molecule.get_n_alpha_electrons(), molecule.get_n_beta_electrons()
        # open shell: the reference restricts to m_s = S (restrict_to_ms=True is not a command-line option,
        # experiments/_base.py:72, :101-123) — the same (n_alpha, n_beta)-restricted space with n_alpha != n_beta — and
        # switches the amplitude spin symmetry off when m_s = |n_alpha - n_beta| // 2 is non-zero (:110-114).  A doublet
        # (|n_alpha - n_beta| == 1: m_s == 0 in the reference's integer arithmetic) keeps the caller's use_amp_spin_sym,
        # exactly like the reference — same ansatz, interchangeable checkpoints.
        use_amp_spin_sym = open_shell_amp_spin_sym(n_alpha, n_beta, use_amp_spin_sym)
        print("\n--- Initialising Hilbert ---\n")
        hilbert = Hilbert.get(N=N, N_alpha=n_alpha, N_beta=n_beta, encoding=Encoding.SIGNED, make_basis=True,
                              verbose=verbose)
        print(f"Initialised Hilbert space with N={hilbert.N}, and {hilbert.size} physically valid configurations.")
        if n_hid_phase == -1:
            n_hid_phase = n_hid
        if n_layer_phase == -1:
            n_layer_phase = n_layer
        print("\n--- Initialising NAQSComplex ---\n")
        wf_args = dict(qubit_ordering=qubit_ordering, masking=masking, num_lut=n_lut, input_encoding=InputEncoding.BINARY,
                       amp_hidden_size=[n_hid] * n_layer, amp_bias=True,
                       phase_hidden_size=[n_hid_phase] * n_layer_phase, phase_bias=True,
                       combined_amp_phase_blocks=comb_amp_phase, use_amp_spin_sym=use_amp_spin_sym,
                       use_phase_spin_sym=use_phase_spin_sym, aggregate_phase=aggregate_phase,
                       amp_activation=SoftmaxLogProbAmps, phase_activation=None, device=device)
        if use_restrictedH:
            wf_args.update(n_alpha_electrons=n_alpha, n_beta_electrons=n_beta)
        wavefunction = NAQSComplex_NADE_orbitals(hilbert, **wf_args)
        if pretrained_model_loc is not None:
            wavefunction.load(pretrained_model_loc)
        _broadcast_parameters(wavefunction)
        print("\n---Preparing Optimizer---\n")
        use_default_lr_schedule = lr < 0
        if use_default_lr_schedule:
            lr = 1e-3
        opt = PartialSamplingOptimizer(
            n_samples=n_samps, n_samples_max=n_samps_max, n_unq_samples_min=n_unq_samps_min,
            n_unq_samples_max=n_unq_samps_max, log_exact_energy=bool(presolveH and hilbert.N < 28),
            wavefunction=wavefunction, qubit_hamiltonian=qubit_hamiltonian, pre_compute_H=presolveH,
            n_electrons=molecule.n_electrons, n_alpha_electrons=n_alpha, n_beta_electrons=n_beta,
            n_fixed_electrons=None, n_excitations_max=None, reweight_samples_by_psi=False, normalise_psi=True,
            normalize_grads=False, grad_clip_factor=None, grad_clip_memory_length=50, optimizer=torch.optim.Adam,
            optimizer_args=[{'lr': lr, 'betas': (0.9, 0.99), 'weight_decay': 0, 'eps': 1e-15, 'amsgrad': False},
                            {'lr': lr_lut}],
            save_loc=exp_name_i, pauli_hamiltonian_dtype=np.float64, verbose=verbose, seed=seed + i)
        print("\n---System summary---\n")
        print(f"Size of restricted subspace : {hilbert.size}.")
        print("Qubit ordering in model :", wavefunction.qubit2model_permutation)
        print("")
        print(wavefunction.model)
        wavefunction.count_parameters()
        if continue_experiment:
            opt.load()
        else:
            if n_pretrain:
                print('\n----------Pre-training NAQS.----------\n')
            opt.pre_flatten(n_pretrain, n_samps, optimizer_args={'lr': 1e-3}, output_freq=output_freq, use_sampling=False,
                            max_batch_size=550000, flatten_phase=False)           # experiments/_base.py:284-289
            if n_pretrain:
                # pre-training runs through autograd on every rank by itself (randperm batches, backward kernels that may not
                # be deterministic): the replicated step assumes bit-identical parameters, so rank 0's are sent again
                _broadcast_parameters(wavefunction)
            opt.save()
        if reset_optimizer:
            opt.reset_optimizer()
        print("\n----------Training NAQS----------\n")
        _SETUP_LOCK.release()
        locked[0] = False
        t0 = time.time()
        if not use_default_lr_schedule:
            opt.run(n_epochs=n_train, save_freq=save_freq, save_final=True, output_freq=output_freq)
        else:                                   # _base.py:303-320: 1e-3 for the first half, 5e-4 for the second
            print("Using default lr schedule...lr --> 1e-3\n")
            opt.run(n_epochs=n_train // 2, save_freq=save_freq, save_final=True, output_freq=output_freq)
            print("\nlr --> 5e-4\n")
            for g in opt.optimizer.param_groups:
                g['lr'] = 5e-4
            opt.run(n_epochs=n_train // 2, save_freq=save_freq, save_final=True, output_freq=output_freq)
        train_time = time.time() - t0
        eig_val, _, n_unq = opt.solve_H(n_samps=opt.n_samples, ret_n_samps=True)
        results.append(_summarise(opt, molecule, exp_name_i, eig_val, n_unq, train_time))
    return results


def _summarise(opt, molecule, exp_name, eig_val, n_unq, train_time):
    """summary.txt next to the checkpoints (the reference's _base.py:330-390, same quantities)."""
    e = np.array([x[1] for x in opt.log[LogKey.E_LOC]])
    window = min(50, len(e))
    final = float(e[-window:].mean()) if len(e) else float("nan")
    fci = molecule.fci_energy
    lines = [f"molecule : {molecule.name}", f"n_steps : {opt.n_steps}", f"training time (s) : {train_time:.1f}",
             f"final <E_loc> (mean of last {window}) : {final:.8f}",
             f"min <E_loc> : {float(e.min()) if len(e) else float('nan'):.8f}",
             f"sampled-subspace diagonalisation ({n_unq} states) : {eig_val:.8f}",
             f"HF : {molecule.hf_energy}", f"CCSD : {molecule.ccsd_energy}", f"FCI : {fci}"]
    if fci is not None:
        lines.append(f"error to FCI (mHa) : {(final - fci) * 1e3:.4f}")
        lines.append(f"subspace-diag error to FCI (mHa) : {(eig_val - fci) * 1e3:.4f}")
    mk_dir(opt.save_loc, quiet=True)
    with open(os.path.join(opt.save_loc, "summary.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    opt.save_log(quiet=True)
    print("\n".join(lines))
    return dict(final=final, fci=fci, eig=eig_val, n_unq=n_unq, time=train_time)


def farm_jobs(molecules, seeds, seed):
    """The farm's job list: every (molecule, seed) pair, molecule-major; without --seeds one job per molecule with -s."""
    mols = molecules.split(",")
    seed_list = [int(x) for x in seeds.split(",")] if seeds else [seed]
    return [(m, sd) for m in mols for sd in seed_list]


def _farm_launch(args, argv):
    """`--farm` without a launcher (no WORLD_SIZE in the environment): this process — which never touches a GPU — starts ONE
    child per GPU (`python -m experiments.run ...` with the device and its share of the job list in the environment); a child
    runs its jobs through `per_gpu` THREADS, each with its own HIP stream (`_farm_threads`).  Threads of one process, not
    processes: two processes sharing an MI355X run 1.8x the runs per hour, four or more stall for seconds at a time on this
    pool (profiles/r04_replicas_per_gpu.txt), while streams of one process overlap like the benchmark's two batches do."""
    import subprocess
    import sys
    jobs = farm_jobs(args.molecule, args.seeds, args.seed)
    n_gpus = args.farm_gpus if args.farm_gpus > 0 else max(1, torch.cuda.device_count())
    n_gpus = min(n_gpus, len(jobs))
    argv = list(sys.argv[1:] if argv is None else argv)
    t0 = time.time()
    procs = []
    for g in range(n_gpus):
        mine = ",".join(str(j) for j in range(g, len(jobs), n_gpus))
        env = dict(os.environ, WORLD_SIZE=str(n_gpus), RANK=str(g), LOCAL_RANK=str(g), NAQS_FARM_DEVICE=str(g), NAQS_FARM_JOBS=mine)
        if args.per_gpu > 1:
            # (NAQS_DEFER_PHASE=1 would give every replica a second stream of its own: a hardware queue too many — see below)
            env["NAQS_DEFER_PHASE"] = "0"
            # the runs' sampler calls take turns on the device (include/naqs_hip.h: naqs_net_share_device): a
            # look-back launch is only certain to end while it is the one such launch in flight
            env.setdefault("NAQS_SHARED_GPU", "1")
        if args.per_gpu > 2:
            # more than two ACTIVE hardware queues are time-sliced in multi-millisecond quanta on this pool (k = 4 threads on
            # four queues: 3 000-step runs take 40-130 s instead of 9): let the runtime map the threads' streams onto two
            env.setdefault("GPU_MAX_HW_QUEUES", "2")
        procs.append(subprocess.Popen([sys.executable, "-m", "experiments.run"] + argv, env=env))
    rcs = [pr.wait() for pr in procs]
    print(f"farm: {len(jobs)} runs on {n_gpus} GPU(s) x {max(1, args.per_gpu)} per GPU in {time.time() - t0:.1f} s")
    if any(rcs):
        raise RuntimeError(f"farm: worker return codes {rcs}")
    return list(range(len(jobs)))


def _farm_threads(args, job_ids):
    """This process's share of the farm's jobs on `per_gpu` threads: a thread takes the next job, enters its own HIP stream
    and runs it start to finish (`_run` serialises the seeded set-up phases; the training loops overlap — the library call
    that is a VMC step releases the interpreter lock)."""
    from concurrent.futures import ThreadPoolExecutor
    if args.number != 1 and args.per_gpu > 1:
        raise NotImplementedError("--per-gpu > 1 with -n > 1: later experiments of a job continue the process-wide random stream")
    jobs = farm_jobs(args.molecule, args.seeds, args.seed)
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None

    def attempt(mol, seed):
        if dev is None:
            return _run_job(args, mol, seed)
        torch.cuda.set_device(dev)                             # (the current device is per thread)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            res = _run_job(args, mol, seed)
            torch.cuda.current_stream().synchronize()
            return res

    def one(j):
        # A run is seeded start to finish, so one that fails because a device-side wait ran out of its budget (NAQS_ERR_HIP,
        # "... wait timed out ...": csrc/naqs_poll.hpp) is simply started again — same numbers.  With two runs per GPU their
        # launches' waiting workgroups could hold each other's slots until both budgets expire (the samplers now take
        # turns, NAQS_SHARED_GPU=1, which closes the one exposure found; the restart stays as the net under it); anything else,
        # and a third failure, is raised.
        from naqs_amd._lib import NaqsError
        mol, seed = jobs[j]
        for tries_left in (2, 1, 0):
            try:
                return attempt(mol, seed)
            except NaqsError as exc:
                if tries_left == 0 or "wait timed out" not in str(exc):
                    raise
                print(f"farm: job {j} ({mol}, seed {seed}) lost a launch to a device-side wait that gave up ({exc}); "
                      f"the run is seeded — starting it again", flush=True)

    import sys
    old = sys.getswitchinterval()
    if args.per_gpu > 1:
        # a thread coming back from the library (which releases the interpreter lock) would otherwise wait up to the default
        # 5 ms for a neighbour's bytecode to yield — a step is 0.2 ms
        sys.setswitchinterval(float(os.environ.get("NAQS_FARM_SWITCH_INTERVAL", "2e-5")))
    try:
        with ThreadPoolExecutor(max(1, args.per_gpu)) as ex:
            out = list(ex.map(one, job_ids))
    finally:
        sys.setswitchinterval(old)
    return [r for res in out for r in res]


def run_from_parser(parser, argv=None):
    args = parser.parse_args(argv)
    if args.no_mask_psi and args.full_mask_psi:
        raise Exception("Invalid option combination: at most one of -no_mask_psi and -full_mask_psi can be specified.")
    if args.farm and "WORLD_SIZE" not in os.environ and (args.per_gpu > 1 or args.seeds or args.farm_gpus > 0):
        return _farm_launch(args, argv)
    rank, world = _setup_distributed(args.farm, args.per_gpu)
    if args.farm and "NAQS_FARM_JOBS" in os.environ:           # a child of the self-launching farm: my share, on threads
        return _farm_threads(args, [int(j) for j in os.environ["NAQS_FARM_JOBS"].split(",") if j != ""])
    molecule_fname, seed = args.molecule, args.seed
    if args.farm:
        jobs = farm_jobs(molecule_fname, args.seeds, args.seed)
        if rank >= len(jobs):
            print(f"rank {rank}: no molecule assigned")
            return []
        molecule_fname, seed = jobs[rank]
    return _run_job(args, molecule_fname, seed)


def _run_job(args, molecule_fname, seed):
    exp_name = args.out
    if exp_name is None:
        exp_name = os.path.join(_EXP_BASE_NAME, _mol_name(molecule_fname))
        exp_name += f"_{_samp_str(args.n_samps)}_samps"
    elif args.farm:
        exp_name = os.path.join(exp_name, _mol_name(molecule_fname) + (f"_s{seed}" if args.seeds else ""))
    for on, suffix in ((args.no_amp_sym, "_noAmpSym"), (args.phase_sym, "_phaseSym"), (args.no_restrictedH, "_no_restrictedH"),
                       (args.no_mask_psi, "_no_mask_psi"), (args.full_mask_psi, "_full_mask_psi")):
        if on:
            exp_name += suffix
    masking = NadeMasking.NONE if args.no_mask_psi else (NadeMasking.FULL if args.full_mask_psi else NadeMasking.PARTIAL)
    print(f"Running experimental script: {__file__}\nResults will be saved to: {exp_name}/\n\nscript options:")
    for key, val in sorted(vars(args).items()):
        print(f"\t{key} : {val}")
    print("")
    return _run(molecule_fname=molecule_fname, hamiltonian_fname=args.hamiltonian_fname, exp_name=exp_name,
                num_experiments=args.number, pretrained_model_loc=args.load, continue_experiment=args.cont,
                reset_optimizer=args.resetOpt, qubit_ordering=args.qubit_ordering, masking=masking, lr=args.lr,
                lr_lut=args.lr_lut, n_samps=args.n_samps, n_samps_max=args.n_samps_max,
                n_unq_samps_min=args.n_unq_samps_min, n_unq_samps_max=args.n_unq_samps_max,
                reweight_samples_by_psi=args.weight_by_psi, n_train=args.n_train, n_pretrain=args.n_pretrain,
                output_freq=args.output_freq, save_freq=None if args.save_freq < 0 else args.save_freq, n_lut=args.n_lut,
                n_hid=args.n_hid, n_layer=args.n_layer, n_hid_phase=args.n_hid_phase, n_layer_phase=args.n_layer_phase,
                n_excitations_max=None if args.n_excitations_max < 0 else args.n_excitations_max,
                comb_amp_phase=args.comb_amp_phase, use_amp_spin_sym=not args.no_amp_sym,
                use_phase_spin_sym=args.phase_sym, aggregate_phase=not args.single_phase,
                use_restrictedH=not args.no_restrictedH, loadH=args.loadH, presolveH=args.presolveH,
                overwrite_pauli_hamiltonian=args.overwriteH, verbose=args.verbose, seed=seed)


def run(*args, **kwargs):
    argv = kwargs.pop("argv", None)
    return run_from_parser(get_parser(*args, **kwargs), argv)
