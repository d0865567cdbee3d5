"""VMC energy optimisation: the counterpart of the reference's ``OptimizerBase`` /
``PartialSamplingOptimizer`` (src/optimizer/energy.py:43-538, :731-1056) on top of the MI355X
local-energy path.

Same constructor keywords (experiments/_base.py:209-246), method names and log / checkpoint
formats; what changes is where the work happens:

  * ``calculate_local_energy`` is one call into ``libnaqs_hip.so`` (matrix-free, f64, on the GPU)
    instead of update_H + get_H + sparse_dense_mv on the host (energy.py:219-263);
  * samples, log psi, weights, E_loc and the VMC loss stay on the device for the whole step
    (the reference moves every network output to the CPU, nade.py:194);
  * multi-GPU (``torch.distributed`` initialised, backend nccl = RCCL on ROCm): every rank draws the
    same unique-sample table (same generator seed), evaluates E_loc and back-propagates the loss for
    its contiguous shard of rows, and the only collectives are an all-reduce of the four energy
    accumulators and one of the flat gradient buffer.  The reference has no distributed code.

Reference quirk NOT mirrored: the full-sample reordering bug (hamiltonian.py:100-105) —
E_loc here is always in sample order.  Energies are reported from float64 E_loc (the reference
rounds E_loc to float32 first, energy.py:260-261).
"""
import math
import os
import time
from collections import Counter
from enum import Enum

import numpy as np
import torch

from .hamiltonian import PauliHamiltonian, keys_to_device
from .nade import MaxBatchSizeExceededError


class LogKey(Enum):                       # src/optimizer/utils.py:9-17
    E = "Energy"
    E_LOC = "Local energy"
    E_LOC_VAR = "Local energy variance"
    N_UNIQUE_SAMP = "Number of unique samples"
    TIME = "Time"

    def __str__(self):
        return self.value


class _CheckpointPickle:
    """``pickle_module`` for ``torch.load``: a checkpoint written by the reference pickles its log keys as
    ``src.optimizer.utils.LogKey`` (energy.py:431, utils.py:9-17); they are mapped to the ``LogKey`` of this
    module so that such files load without the reference on the import path."""
    import pickle as _p
    __name__ = "pickle"
    load, loads, dump, dumps = _p.load, _p.loads, _p.dump, _p.dumps
    Pickler, PickleError, UnpicklingError = _p.Pickler, _p.PickleError, _p.UnpicklingError
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = _p.HIGHEST_PROTOCOL, _p.DEFAULT_PROTOCOL

    class Unpickler(_p.Unpickler):
        def find_class(self, module, name):
            if module == "src.optimizer.utils" and name == "LogKey":
                return LogKey
            return super().find_class(module, name)


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def shard_bounds(n, rank, world):
    """Contiguous row shards of equal padded size S = ceil(n / world): rank r owns [min(n, r S), min(n, (r + 1) S)).  Equal
    padded shards are what an all-gather wants — the ranks' (log|psi|, phase) shards, each padded to S rows, land
    back to back as the table of all n rows (``bench.py --shard rows`` uses the same split)."""
    S = -(-n // world) if world > 0 else n
    b = min(n, rank * S)
    return b, min(n, b + S)


def vmc_loss(log_psi, e_loc, weights, e_mean):
    """2 * Re sum_i w_i * log psi_i * (E_loc_i - <E>)  with log psi = (log|psi|, phase) and complex
    E_loc = (Re, Im): energy.py:328-329 with complex.py:49-58 written out."""
    ec = e_loc - e_mean
    return 2.0 * (weights * (log_psi[:, 0] * ec[:, 0] - log_psi[:, 1] * ec[:, 1])).sum()


_GC_FROZEN = False


def _freeze_garbage_collector():
    """Once per process, before the first training loop: move everything alive (torch's ~10^6 module-level objects, the
    network, the optimiser) to the collector's permanent generation.  A VMC step here is ~0.4 ms of host work that
    allocates a few hundred containers, and every full collection walks all tracked objects: with the reference's default
    ansatz (80 parameter tensors) the loop spent half its wall time in the collector (0.83 -> 0.43 ms per N2 step,
    tools/train_loop_profile.py with NAQS_GC=freeze).  Nothing is leaked that was not going to live as long as the
    process anyway; NAQS_GC_FREEZE=0 turns it off."""
    global _GC_FROZEN
    if _GC_FROZEN or os.environ.get("NAQS_GC_FREEZE", "1") == "0":
        return
    import gc
    gc.collect()
    gc.freeze()
    _GC_FROZEN = True


class OptimizerBase:
    def __init__(self, wavefunction, qubit_hamiltonian, pre_compute_H=True, n_electrons=None,
                 n_alpha_electrons=None, n_beta_electrons=None, n_fixed_electrons=None, n_excitations_max=None,
                 reweight_samples_by_psi=False, normalise_psi=False, normalize_grads=False, grad_clip_factor=3,
                 grad_clip_memory_length=50, optimizer=torch.optim.Adam, optimizer_args={'lr': 1e-3},
                 scheduler=None, scheduler_args=None, save_loc='./', pauli_hamiltonian_fname=None,
                 overwrite_pauli_hamiltonian=False, pauli_hamiltonian_dtype=np.float32, verbose=False, seed=None,
                 bug_compat_full_sample_order=False):
        self.grad_clip_factor, self.grad_clip_memory_length = grad_clip_factor, int(grad_clip_memory_length)
        # opt-in reproduction of the reference's full-sample quirk (hamiltonian.py:100-105): when a batch
        # holds EVERY state of the restricted space, get_H returns H in restricted-basis order while psi stays in
        # sample order.  Off by default (the mathematically correct E_loc); on only to retrace reference runs on
        # small molecules whose sampler saturates the space (LiH, H2O).
        self.bug_compat_full_sample_order = bool(bug_compat_full_sample_order)
        if n_fixed_electrons not in (None, 0) or n_excitations_max is not None:
            raise NotImplementedError("frozen-core / excitation-limited runs are out of scope")
        self.wavefunction = wavefunction
        self.hilbert = wavefunction.hilbert
        self.qubit_hamiltonian = qubit_hamiltonian
        self.device = wavefunction.device
        self.reweight_samples_by_psi = reweight_samples_by_psi
        self.normalise_psi = normalise_psi
        self.n_electrons, self.n_alpha_electrons, self.n_beta_electrons = n_electrons, n_alpha_electrons, n_beta_electrons
        self.n_fixed_electrons, self.n_excitations_max = n_fixed_electrons, n_excitations_max
        # energy.py:93-97: what every get_subspace call of the optimiser carries (the restricted space validates them)
        self.subspace_args = {"N_up": n_electrons, "N_alpha": n_alpha_electrons, "N_beta": n_beta_electrons,
                              "N_occ": n_fixed_electrons, "N_exc_max": n_excitations_max}
        self.optimizer_callable, self.optimizer_args = optimizer, optimizer_args
        self.scheduler_callable, self.scheduler_args = scheduler, scheduler_args
        self.normalize_grads = normalize_grads
        self.save_loc = save_loc
        self.verbose = verbose
        self.pauli_hamiltonian_fname = pauli_hamiltonian_fname
        self.overwrite_pauli_hamiltonian = False          # no matrix cache to write (matrix-free)
        self.pauli_hamiltonian = PauliHamiltonian.get(self.hilbert, qubit_hamiltonian, verbose=verbose,
                                                      dtype=pauli_hamiltonian_dtype, device=self.device)
        self.generator = torch.Generator(device=self.device)
        if seed is not None:
            self.generator.manual_seed(int(seed))
        else:
            self.generator.manual_seed(torch.initial_seed() % (2 ** 63))
        self._sampled_idxs, self._sampled_pending = Counter(), []
        self.track_sampled_idxs = True
        self.solve_H_max_states = 10000       # energy.py:773-776
        self.use_fused = True                # HIP sampler / training kernels / FlatAdam when the network supports them
        self._loss_terms = self._last_loss = None
        self._shard_mismatch = None
        # multi-GPU policy: a small table's step is a chain of launch-bound kernels — sharding it adds launches and
        # three collectives and saves nothing — so below `shard_min_table` unique samples (or `shard_min_rows` rows per rank)
        # every rank runs the IDENTICAL single-GPU step (same seed, same parameters, deterministic kernels: bit-identical
        # updates, no collectives) and the ranks only prove every `replica_proof_every` steps that they still hold the same
        # table.  Measured break-even of the four-call sharded step (tools/scaling_model.py, one GPU, collectives assumed at
        # 80 us): N2 at 1.1 k samples loses at every world size (0.36 vs 0.21 ms), Li2O at 20 k samples wins down to 2.5 k rows
        # per rank (0.49 vs 0.81 ms at 8 ranks).  shard_min_rows = 0 forces sharding (tests).
        self.shard_min_rows = int(os.environ.get("NAQS_SHARD_MIN_ROWS", "1024"))
        self.shard_min_table = int(os.environ.get("NAQS_SHARD_MIN_TABLE", "8192"))
        self.replica_proof_every = int(os.environ.get("NAQS_REPLICA_PROOF_EVERY", "64"))
        self._dist_mode, self._last_M, self.dist_mode_log = None, None, []
        self.reset_log()
        self.reset_optimizer()

    # how often each basis state has appeared as a unique sample (reference: a Counter updated every step)
    def _track_sampled(self, keys):
        """energy.py:300 (a Counter updated every step), deferred: the step's keys stay on the device and are folded into the
        Counter when it is looked at, or once 2^25 of them (268 MB) are waiting.  Folding is a data-dependent-size operation
        (torch.unique), i.e. a host synchronisation plus a sort: every 256 steps it was ~0.05 ms per step of a 0.28 ms step."""
        if keys.numel() and keys.untyped_storage().nbytes() > 32 * keys.numel():
            keys = keys.clone()                  # (a view would pin its whole max_unique-sized buffer until the next fold)
        self._sampled_pending.append(keys)
        self._sampled_pending_n = getattr(self, "_sampled_pending_n", 0) + int(keys.shape[0])
        if self._sampled_pending_n >= (1 << 25) or len(self._sampled_pending) >= (1 << 16):
            self._flush_sampled_idxs()

    def _sampled_ring_slot(self, cap):
        """The one-call step lets the sampler write its keys straight into the tracking buffer: -> a view of `cap` elements
        at the buffer's write offset (folding the buffer first when the slot would not fit)."""
        ring = getattr(self, "_sampled_ring", None)
        if ring is None or ring.numel() < 2 * cap:
            self._flush_sampled_idxs()
            # sized from the problem: 64 steps' worth of the largest table the sampler may return (51 MB at the published
            # n_unq_samples_max = 1e5), never more than 2^25 keys (268 MB), never less than two slots.  ``_sample_keys`` is
            # a view into this buffer: valid until the next step's sampler runs (a fold resets the write offset).
            elems = getattr(self, "sampled_ring_elems", None)
            elems = min(1 << 25, 64 * cap) if elems is None else int(elems)
            ring = self._sampled_ring = torch.empty(max(elems, 2 * cap), dtype=torch.int64, device=self.device)
            self._sampled_ring_off = 0
        if self._sampled_ring_off + cap > ring.numel():
            self._flush_sampled_idxs()
        return ring[self._sampled_ring_off:self._sampled_ring_off + cap]

    def _flush_sampled_idxs(self):
        self._sampled_pending_n = 0
        off = getattr(self, "_sampled_ring_off", 0)
        if off:
            self._sampled_pending.append(self._sampled_ring[:off])
            self._sampled_ring_off = 0
        if self._sampled_pending:
            k, c = torch.unique(torch.cat(self._sampled_pending), return_counts=True)
            self._sampled_pending = []
            self._sampled_idxs.update(dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist())))

    @property
    def sampled_idxs(self):
        self._flush_sampled_idxs()
        return self._sampled_idxs

    @sampled_idxs.setter
    def sampled_idxs(self, value):
        self._sampled_idxs, self._sampled_pending = Counter(value), []
        self._sampled_ring_off = self._sampled_pending_n = 0

    @property
    def last_loss(self):
        """The loss of the last ``_SGD_step`` (evaluated on demand when the step ran on the graph-free path)."""
        if self._last_loss is None and self._loss_terms is not None:
            g, lp = self._loss_terms
            self._last_loss = (g * lp).sum()
        return self._last_loss

    # ---- bookkeeping (energy.py:141-187) ----
    def _flush_log(self):
        """Move the steps whose <E>, Var are still device scalars into ``self.log`` (one transfer for all of them)."""
        self._check_shards()
        if not self._pending_log:
            return
        vals = torch.stack([p[1] for p in self._pending_log]).cpu().numpy()
        for (step, _, n_unq, t), (e, var) in zip(self._pending_log, vals):
            self.log[LogKey.E_LOC].append((step, float(e)))
            self.log[LogKey.E_LOC_VAR].append((step, float(var)))
            self.log[LogKey.N_UNIQUE_SAMP].append((step, n_unq))
            self.log[LogKey.TIME].append((step, t))
        self._pending_log = []

    def _check_shards(self):
        """Raise if some step's ranks did not shard one and the same sample table (different seeds / parameters)."""
        bad = getattr(self, "_shard_mismatch", None)
        if bad is not None and bool(bad.item()):
            raise RuntimeError("distributed VMC step: the ranks sampled different tables (sample count or key checksum "
                               "differs across ranks) — seed every rank identically and broadcast the parameters")

    # ---- multi-GPU policy ----
    def _active_dist(self):
        """torch.distributed when THIS step shards its table over the ranks, else None (single process, or a step every rank
        replicates: ``_choose_dist_mode``)."""
        return _dist() if getattr(self, "_dist_mode", None) == "sharded" else None

    def _choose_dist_mode(self, trust_local=False):
        """'single' | 'replicated' | 'sharded' for the next step, from the unique-sample count of the LAST step — the count
        moves slowly, and the decision has to be made before sampling because the replicated step is one library call that
        includes the sampler.  The first step replicates.

        Every rank must take the same decision (a rank that enters the sharded step's all-gather while another stays
        replicated hangs the job), so a switch is only taken on a count the ranks have PROVEN to share:
          * replicated -> sharded only right after a replica proof (every `replica_proof_every` steps, the same step number on
            every rank), on the all-reduced M and parameter checksum of that proof — read on the host here, once per proof;
          * sharded -> replicated only after `_check_shards()` has seen the last sharded steps' proof (same M, same keys).
        ``trust_local`` (a direct `_SGD_step` call outside `run()`: the caller hands every rank the table) decides from the
        table at hand."""
        dist = _dist()
        if dist is None:
            mode = "single"
        else:
            world = dist.get_world_size()
            M = self._last_M if self._last_M is not None else 0
            big_enough = self.shard_min_rows <= 0 or (M >= self.shard_min_table and M // world >= self.shard_min_rows)
            mode = "sharded" if big_enough else "replicated"
            if trust_local:
                pass
            elif self._dist_mode == "replicated" and self.replica_proof_every > 0 and self.shard_min_rows > 0:
                proof = getattr(self, "_proof_at", None)
                if proof is None or proof[0] != self.n_steps:
                    mode = "replicated"                       # no agreed count for this step: stay (same rule on every rank)
                else:
                    self._check_shards()                      # raises when the proof failed (tables or parameters differ)
                    M = int(round(float(proof[1][0].item()) / world))
                    mode = "sharded" if (M >= self.shard_min_table and M // world >= self.shard_min_rows) else "replicated"
            elif self._dist_mode == "sharded" and mode != "sharded":
                self._check_shards()
        if mode != self._dist_mode:
            if dist is not None:
                self.dist_mode_log.append((self.n_steps, mode))
                if dist.get_rank() == 0:
                    what = ("row-sharded step (all-gather of log psi shards, all-reduce of accumulators and gradient)"
                            if mode == "sharded" else "every rank runs the identical single-GPU step (no collectives)")
                    print(f"\tdistributed step at epoch {self.n_epochs}: {self._last_M} unique samples over "
                          f"{dist.get_world_size()} ranks (shard_min_table={self.shard_min_table}, "
                          f"shard_min_rows={self.shard_min_rows}) --> {what}")
            self._dist_mode = mode
            self._onecall_cached = None
        return mode

    def _replica_proof(self, keys):
        """Replicated steps: every `replica_proof_every` steps one 48-byte all-reduce of (M, M^2, c, c^2, p, p^2), c = 20 low
        bits of the key sum, p = 20 low bits of the sum of the parameters' bit patterns — W * sum x^2 == (sum x)^2 iff all
        ranks hold the same x (exact integers in float64).  A mismatch is reported at the next log flush, like the sharded
        step's proof, and before any switch to the sharded step (`_choose_dist_mode`)."""
        dist = _dist()
        if dist is None or self.replica_proof_every <= 0 or self.n_steps % self.replica_proof_every != 0:
            return
        world = dist.get_world_size()
        c = (keys.sum() & 0xFFFFF).double()
        m = torch.tensor(float(keys.shape[0]), dtype=torch.float64, device=self.device)
        flat = getattr(self.wavefunction, "_flat_params", None)
        if flat is None:
            flat = torch.cat([p.detach().reshape(-1) for p in self.wavefunction.param_list()])
        p = (flat.detach().view(torch.int32).sum(dtype=torch.int64) & 0xFFFFF).double()
        ext = torch.stack([m, m * m, c, c * c, p, p * p])
        dist.all_reduce(ext)
        bad = ~((world * ext[1] == ext[0] * ext[0]) & (world * ext[3] == ext[2] * ext[2]) & (world * ext[5] == ext[4] * ext[4]))
        self._shard_mismatch = bad if self._shard_mismatch is None else self._shard_mismatch | bad
        self._proof_at = (self.n_steps, ext)

    def reset_log(self):
        self._pending_log = []
        self.log = {LogKey.E: [], LogKey.E_LOC: [], LogKey.E_LOC_VAR: [], LogKey.N_UNIQUE_SAMP: [], LogKey.TIME: []}
        self.n_steps = self.n_epochs = 0
        self.run_time = 0

    def reset_optimizer(self, cond_idx=None):
        self._onecall_cached = None            # (a plain torch optimiser cannot take the one-call step)
        if isinstance(self.optimizer_args, dict):
            self.optimizer = self.optimizer_callable(self.wavefunction.conditional_parameters(cond_idx),
                                                     **self.optimizer_args)
        else:       # list of per-group arguments: group 0 = network, group 1 = look-up tables (energy.py:167-172)
            groups = []
            for idx, a in enumerate(self.optimizer_args):
                a = dict(a)
                a['params'] = self.wavefunction.parameters(idx)
                groups.append(a)
            self.optimizer = None
            if (self.optimizer_callable is torch.optim.Adam and self.use_fused and cond_idx is None
                    and not any(g.get('amsgrad') for g in groups) and self.wavefunction.fused() is not None):
                # same rule and state_dict as torch.optim.Adam, one launch per step on the flattened parameters
                from .flat_adam import FlatAdam
                flat = self.wavefunction.flatten_parameters()
                self.wavefunction.parameters_changed()
                self.optimizer = FlatAdam(groups, flat)
            if self.optimizer is None:
                self.optimizer = self.optimizer_callable(groups)
        self.scheduler = (self.scheduler_callable(self.optimizer, **self.scheduler_args)
                          if self.scheduler_callable is not None else None)
        # clipping memory (energy.py:187): the last `grad_clip_memory_length` (clipped) gradient norms per parameter
        # group, kept on the device so that clipping costs no host synchronisation
        self._grad_norms = [[torch.zeros(max(1, self.grad_clip_memory_length), dtype=torch.float64, device=self.device), 0]
                            for _ in self.optimizer.param_groups]

    @torch.no_grad()
    def _clip_grads(self):
        """energy.py:383-395 with torch_utils.clip_grad_norm_ (network/torch_utils.py:24-53): per parameter group the
        2-norm of all gradients is limited to grad_clip_factor x the mean of the remembered norms (1e3 while the
        memory is empty) and min(limit, norm) is remembered."""
        if self.grad_clip_factor is None:
            return
        for mem, group in zip(self._grad_norms, self.optimizer.param_groups):
            hist, n = mem
            L = hist.numel()
            max_norm = (self.grad_clip_factor * hist[:min(n, L)].mean() if n > 0
                        else torch.tensor(1e3, dtype=torch.float64, device=self.device))
            grads = [p.grad for p in group['params'] if p.grad is not None]
            if grads:
                first = grads[0]
                owner = first._base if first._base is not None else first
                flat = None
                if owner.dim() == 1 and owner.numel() == sum(g.numel() for g in grads) and all(
                        g._base is owner or g is owner for g in grads):
                    flat = owner                                  # the gradients are views of one flat buffer
                norm = (flat.norm(2) if flat is not None
                        else torch.norm(torch.stack([torch.norm(g.detach(), 2) for g in grads]), 2)).double()
                coef = (max_norm / (norm + 1e-6)).clamp(max=1.0).to(first.dtype)
                if flat is not None:
                    flat.mul_(coef)
                else:
                    for g in grads:
                        g.mul_(coef)
            else:
                norm = torch.zeros((), dtype=torch.float64, device=self.device)
            hist[n % L] = torch.minimum(max_norm, norm)
            mem[1] = n + 1

    def _eloc_keys(self, keys):
        """The key table handed to the E_loc kernel: the sample keys — or, with ``bug_compat_full_sample_order`` and
        a batch that covers the whole restricted space, the keys in restricted-basis order.  Row i then belongs to
        state rho_i but is paired with psi of sample i: exactly what the reference computes in that case,
        conj(sum_j H[rho_i, rho_j] psi(s_j) / psi(s_i))  (hamiltonian.py:100-105 feeding energy.py:248)."""
        if self.bug_compat_full_sample_order and keys.shape[0] == self.hilbert.size:
            if getattr(self, "_restricted_order_keys", None) is None:
                self._restricted_order_keys = keys_to_device(
                    self.hilbert.get_subspace(ret_states=False, ret_idxs=True, **self.subspace_args), self.device)
            return self._restricted_order_keys
        return keys

    # ---- the hot path ----
    @torch.no_grad()
    def calculate_local_energy(self, states_idx, psi=None, set_unsampled_states_to_zero=True, ret_complex=False,
                               log_psi=None, row_begin=0, n_rows=None):
        """E_loc of the sampled states (energy.py:219-263).  ``psi``: [M, 2] (Re, Im) like the
        reference, or pass ``log_psi`` [M, 2] = (log|psi|, phase) directly (better conditioned).
        Un-sampled connected states contribute zero — the only mode the reference implements.
        Returns a float64 device tensor [n_rows, 2], or a complex128 numpy array with ret_complex."""
        if not set_unsampled_states_to_zero:
            raise NotImplementedError()
        keys = keys_to_device(states_idx, self.device)
        if psi is None and log_psi is None:
            log_psi = self.wavefunction.log_psi(self.hilbert.idx2state(keys))
        keys = self._eloc_keys(keys)
        if log_psi is not None:
            e = self.pauli_hamiltonian.local_energy(keys, log_psi.detach().to(self.device), kind="log_psi",
                                                    row_begin=row_begin, n_rows=n_rows)
        else:
            e = self.pauli_hamiltonian.local_energy(keys, psi.detach().to(self.device), kind="psi",
                                                    row_begin=row_begin, n_rows=n_rows)
        if ret_complex:
            v = e.cpu().numpy()
            return v[:, 0] + 1j * v[:, 1]
        return e

    @torch.no_grad()
    def calculate_energy(self, normalise_psi=None):
        """<psi|H|psi> over the whole restricted space (energy.py:189-217); small spaces only."""
        fused = self.wavefunction.fused(need_phase=True) if self.use_fused else None
        if fused is not None:               # keys in, one library call for log psi + E_loc
            keys = keys_to_device(self.hilbert.get_subspace(ret_states=False, ret_idxs=True, **self.subspace_args), self.device)
            lp, e = fused.log_psi_and_local_energy(self.pauli_hamiltonian, keys)
        else:
            states, idx = self.hilbert.get_subspace(ret_states=True, ret_idxs=True, **self.subspace_args)
            keys = keys_to_device(idx, self.device)
            with torch.no_grad():
                lp = self.wavefunction.log_psi(states.to(self.device)).reshape(-1, 2)
            e = self.pauli_hamiltonian.local_energy(keys, lp, kind="log_psi")
        p = (2.0 * lp[:, 0].double()).exp()
        sums = self.pauli_hamiltonian.reduce(p, e)
        energy = sums[0] / (sums[3] if normalise_psi else 1.0)
        return float(energy.item())

    def _SGD_step(self, states, states_idx, log_psi=None, sample_weights=None, log_psi_eval=None,
                  regularisation_loss=None, n_samps=None, e_loc_clip_factor=None, lazy=False):
        """One VMC step for the sampled states (energy.py:273-377): E_loc (no grad) -> loss
        2 Re sum w log psi (E_loc - <E>) -> backward -> optimiser step -> (<E>, Var)."""
        if not getattr(self, "_in_run", False):           # called outside run(): the policy decides from the table at hand
            self._last_M = int(keys_to_device(states_idx, self.device).shape[0])
            self._choose_dist_mode(trust_local=True)
        dist = self._active_dist()
        world, rank = (dist.get_world_size(), dist.get_rank()) if dist else (1, 0)
        keys = keys_to_device(states_idx, self.device)
        M = keys.shape[0]
        if self.track_sampled_idxs:          # energy.py:300; folded into the Counter in batches (no per-step sync)
            self._track_sampled(keys)
        # shard of rows this rank owns (the whole table when single-process)
        b, e_ = shard_bounds(M, rank, world)
        saved = None
        if log_psi is not None:
            # reference-style call: log psi of the whole table, carrying gradients
            lp_all, lp_mine = log_psi.reshape(-1, 2), log_psi.reshape(-1, 2)[b:e_]
        else:
            # gradients only for the owned rows; the table needed for the psi look-ups is evaluated
            # without autograd (the rows of other ranks are theirs to differentiate)
            fused = self.wavefunction.fused(need_phase=True) if self.use_fused else None
            pre = None
            quirk = self._eloc_keys(keys) is not keys
            if (fused is not None and regularisation_loss is None and not self.normalize_grads and world == 1
                    and fused.train_mode == "hip" and sample_weights is not None and not quirk
                    and os.environ.get("NAQS_TRAIN_FUSED_ELOC", "1") == "1"):
                # single GPU: forward (activations kept) + E_loc + weighted sums in one library call — or already done by
                # the sampler's call for exactly these keys and weights (get_samples, training loop)
                pf = getattr(self, "_prefused", None)
                self._prefused = None
                if pf is not None and pf[0] is states_idx and sample_weights is self._sample_weights:
                    pre = pf[1]
                else:
                    pre = fused.forward_saved_with_local_energy(self.pauli_hamiltonian, keys, sample_weights.reshape(-1))
                lp_mine, saved = pre[0], pre[1]
            elif fused is not None and regularisation_loss is None and not self.normalize_grads:
                # HIP amplitude forward/backward + explicit chain rule of the phase MLP: no autograd graph at all
                lp_mine, saved = fused.forward_saved(keys[b:e_])
            elif fused is not None and not fused.aggregate:     # same kernels behind a torch.autograd.Function
                lp_mine = fused.log_psi_train(keys[b:e_])
            else:
                lp_mine = self.wavefunction.log_psi(states[b:e_]).reshape(-1, 2)
            if world == 1:
                lp_all = lp_mine
            else:
                # the table every rank's E_loc rows look psi_j up in: ONE all-gather of the ranks' own (log|psi|, phase)
                # shards — no rank evaluates the network on rows it does not own (what `bench.py --shard rows`
                # measures).  The contribution has a size every rank computes WITHOUT looking at its table (rows for the
                # largest table the sampler may return, n_unq_samples_max): ranks whose samplers diverged then still meet in
                # a well-formed collective and the key-checksum proof below reports them, instead of a size mismatch
                # wedging the communicator.  The bytes are cheap (100 KB per rank at the published settings; latency-bound).
                S = -(-M // world)
                S_pad = max(S, -(-int(self.n_unq_samples_max) // world)) if getattr(self, "n_unq_samples_max", None) else S
                gb = getattr(self, "_gather_bufs", None)
                if gb is None or gb[0].shape[0] != S_pad or gb[1].shape[0] != S_pad * world:
                    gb = self._gather_bufs = (torch.zeros((S_pad, 2), dtype=torch.float32, device=self.device),
                                             torch.empty((S_pad * world, 2), dtype=torch.float32, device=self.device))
                mine, table = gb                       # rows past my shard keep old values: they are never looked at (>= M)
                mine[:e_ - b] = lp_mine.detach()
                dist.all_gather_into_tensor(table, mine)
                lp_all = table.view(world, S_pad, 2)[:, :S].reshape(-1, 2)[:M]
        if sample_weights is None:
            if not self.reweight_samples_by_psi:
                raise NotImplementedError("Re-weighting by the number of samples is not yet implemented.")
            sample_weights = lp_all.detach()[..., 0].exp().pow(2)
            if self.normalise_psi:
                sample_weights = sample_weights / sample_weights.sum()
        w = sample_weights.reshape(-1).to(self.device, torch.float64)

        # E_loc of the owned rows + (sum w Re, sum w Im, sum w Re^2, sum w) in one launch
        if log_psi is None and pre is not None:
            e_loc, sums = pre[2], pre[3]
        else:
            e_loc, sums = self.pauli_hamiltonian.local_energy(self._eloc_keys(keys), lp_all.detach(), kind="log_psi", row_begin=b,
                                                              n_rows=e_ - b, weights=w[b:e_])
        shard_ok = None
        if dist:
            # one collective for the accumulators AND a proof that every rank sharded the same table: with
            # (M, M^2, c, c^2) appended (c = 20 low bits of the key sum), W * sum x^2 == (sum x)^2 holds iff all
            # ranks contributed the same x (Cauchy-Schwarz; everything is an exact integer in float64)
            if self.device.type == "cuda":
                # (one launch of the library instead of eight tiny torch kernels and a host-to-device copy of M: at
                # M ~ 10^3 the sharded step is bound by its launches)
                from . import _lib
                from .fused import _stream_ptr
                ext = torch.empty(8, dtype=torch.float64, device=self.device)
                _lib.check(_lib.load_library().naqs_shard_proof(M, keys.data_ptr(), sums.data_ptr(), ext.data_ptr(),
                                                                 _stream_ptr(self.device)), "naqs_shard_proof")
            else:
                c = (keys.sum() & 0xFFFFF).double()
                m = torch.tensor(float(M), dtype=torch.float64, device=self.device)
                ext = torch.cat([sums, torch.stack([m, m * m, c, c * c])])
            dist.all_reduce(ext)
            sums = ext[:4]
            shard_ok = (world * ext[5] == ext[4] * ext[4]) & (world * ext[7] == ext[6] * ext[6])
            self._shard_mismatch = (~shard_ok if self._shard_mismatch is None else self._shard_mismatch | ~shard_ok)
        self.optimizer.zero_grad()
        ev = None
        if saved is not None:
            # d loss / d log psi of vmc_loss, written out: (2 w Re(E_loc - <E>), -2 w Im(E_loc - <E>)); the same launch
            # leaves (<E>, Var) of energy.py:372-375 on the device
            g, ev = fused.backward_from_local_energy(saved, e_loc, w[b:e_].contiguous(), sums)
            self._loss_terms, self._last_loss = (g, lp_mine), None
        else:
            e_mean = torch.stack([sums[0], sums[1]])                    # (sum w E_loc), like energy.py:328 (w not renormalised)
            loss = vmc_loss(lp_mine, e_loc.to(lp_mine.dtype), w[b:e_].to(lp_mine.dtype), e_mean.to(lp_mine.dtype))
            if self.normalize_grads:
                loss = loss / loss.detach().abs()
            if regularisation_loss is not None:
                loss = loss + regularisation_loss
            loss.backward()
            self._loss_terms, self._last_loss = None, loss.detach()
        if dist:
            params = [p for g in self.optimizer.param_groups for p in g['params'] if p.grad is not None]
            gflat = getattr(fused, "_grad_flat", None) if saved is not None else None
            if gflat is not None and params and params[0].grad.data_ptr() == gflat.data_ptr():
                dist.all_reduce(gflat)                                  # the gradients ARE one flat buffer: in place
            else:
                flat = torch.cat([p.grad.reshape(-1) for p in params])
                dist.all_reduce(flat)                                   # shards SUM to the full-batch gradient
                off = 0
                for p in params:
                    p.grad.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()
        self._clip_grads()
        self.optimizer.step()
        self.wavefunction.parameters_changed()
        if self.use_fused and saved is not None and os.environ.get("NAQS_TRAIN_EARLY_REFRESH", "1") == "1":
            # re-pack the kernels' weight layouts NOW, behind the optimiser launch, instead of at the start of the next
            # sampling call: the host gets there tens of microseconds later and the GPU would wait for it
            self.wavefunction.fused(need_phase=True)
        self.optimizer.zero_grad()
        if self.scheduler is not None:
            self.scheduler.step()

        with torch.no_grad():                                           # energy.py:367-377
            if ev is None:
                energy = sums[0] / sums[3]
                ev = torch.stack([energy, sums[2] / sums[3] - energy * energy])
            if shard_ok is not None:        # ranks that sampled different tables must not report a plausible energy
                ev = torch.where(shard_ok, ev, torch.full_like(ev, float("nan")))
        if lazy:        # device scalars: the caller reads them later, so the host can queue the next step meanwhile
            return ev
        self._check_shards()
        energy, variance = ev.tolist()
        return float(energy), float(variance)

    # ---- checkpoints / logs: same keys as the reference (energy.py:400-538) ----
    def _fmt(self, fname):
        if os.path.splitext(fname)[-1] != '.pth':
            fname += '.pth'
        if not os.path.isabs(fname):
            fname = os.path.join(self.save_loc, fname)
        return fname

    def save(self, fname="energy_optimizer", quiet=False):
        self._flush_log()
        dist = _dist()
        if dist and dist.get_rank() != 0:
            return
        fname = self._fmt(fname)
        d = os.path.dirname(fname)
        if d:
            os.makedirs(d, exist_ok=True)
        wf_fname = self.wavefunction.save(os.path.splitext(fname)[0] + '_naqs', quiet)
        torch.save({'optimizer:state_dict': self.optimizer.state_dict(), 'run_time': self.run_time,
                    'n_steps': self.n_steps, 'n_epochs': self.n_epochs, 'log': self.log,
                    'sampled_idxs': self.sampled_idxs, 'wavefunction:fname': wf_fname,
                    'hamiltonian_fname': self.pauli_hamiltonian_fname}, fname)
        if not quiet:
            print(f"Saving checkpoint {fname}...done.")

    def load(self, fname="energy_optimizer", quiet=False):
        fname = self._fmt(fname)
        ck = torch.load(fname, map_location=self.device, weights_only=False, pickle_module=_CheckpointPickle)
        # the wavefunction file: where the checkpoint says (energy.py:466-469), else next to the checkpoint under
        # the name save() gives it (a checkpoint directory that was moved or copied)
        wf_fname = ck['wavefunction:fname']
        if not os.path.exists(wf_fname):
            sibling = os.path.splitext(fname)[0] + '_naqs.pth'
            if os.path.exists(sibling):
                wf_fname = sibling
        try:
            self.wavefunction.load(wf_fname)
        except Exception:
            print(f"\twavefunction not found (expected at {ck['wavefunction:fname']})")
        try:
            self.optimizer.load_state_dict(ck['optimizer:state_dict'])
        except Exception:
            print("\tOptimizer could not be loaded.")
        self.log, self.n_steps, self.n_epochs = ck['log'], ck['n_steps'], ck['n_epochs']
        self.run_time, self.sampled_idxs = ck['run_time'], ck['sampled_idxs']
        if not quiet:
            print(f"Loading checkpoint {fname}...done.")

    def save_log(self, fname="log", quiet=False):
        self._flush_log()
        import pandas as pd
        fname = os.path.splitext(os.path.join(self.save_loc, fname))[0] + ".pkl"
        os.makedirs(os.path.dirname(fname) or ".", exist_ok=True)
        df = None
        for key, value in self.log.items():
            dk = pd.DataFrame(value, columns=["Iteration", key])
            df = dk if df is None else pd.merge(df, dk, how="outer", on="Iteration")
        if df is not None:
            df.sort_values("Iteration").reset_index(drop=True).to_pickle(fname)
            if not quiet:
                print("Log saved to", fname)


class PartialSamplingOptimizer(OptimizerBase):
    """Optimise with partial sampling of the Hilbert space (energy.py:731-1056)."""

    def __init__(self, n_samples, n_samples_max=1e9, n_unq_samples_min=1000, n_unq_samples_max=1e6,
                 log_exact_energy=True, **kwargs):
        kwargs['reweight_samples_by_psi'] = False
        super().__init__(**kwargs)
        self.log_exact_energy = log_exact_energy
        self.n_samples = int(n_samples)
        self.n_samples_max = int(n_samples_max)
        self.n_unq_samples_min = int(n_unq_samples_min)
        self.n_unq_samples_max = int(n_unq_samples_max)

    def get_n_samples(self):
        return self.n_samples

    def pre_flatten(self, n_epochs, n_samps=1e6, flatten_phase=False, norm_reg_weight=0, use_sampling=True, max_batch_size=-1,
                    optimizer=torch.optim.Adam, optimizer_args={'lr': 5e-3}, output_freq=50):
        """Pre-train the amplitudes towards the uniform superposition over the restricted space (energy.py:840-904; what
        ``-n_pretrain n`` runs, experiments/_base.py:284-289: ``use_sampling=False, max_batch_size=550000``): n supervised
        epochs, mean-squared error of log|psi| against log(1 / sqrt(size)), Adam on all parameters.  PyTorch modules with
        autograd on the network's device — a set-up phase over the enumerated space, not the hot path.  (The reference's
        ``use_sampling=True`` branch returns nothing from its epoch function and cannot run; it is refused here.)"""
        import math
        import torch.nn.functional as F
        print("Pre-flattening NAQS amplitudes", end="...")
        if not n_epochs:                                      # (the reference enumerates the space even for zero epochs)
            print("done.")
            return
        if use_sampling:
            raise NotImplementedError("pre_flatten(use_sampling=True): the reference's own branch fails (energy.py:880-887 returns None)")
        wf = self.wavefunction
        states = self.hilbert.get_subspace(ret_states=True, ret_idxs=False, **self.subspace_args).to(self.device)   # energy.py:859
        opt = optimizer(wf.parameters(), **optimizer_args)
        log_amp_target = math.log(1 / math.sqrt(len(states)))
        if max_batch_size < 0:
            max_batch_size = len(states)
        n_batches = (len(states) - 1) // max_batch_size + 1
        print(f"using {n_batches} batch(es) of size of at most {max_batch_size}.")
        t0 = time.time()
        for i in range(1, n_epochs + 1):
            for idx_batch in torch.randperm(len(states)).chunk(n_batches):
                opt.zero_grad()
                log_psi = wf.log_psi(states[idx_batch.to(states.device)])
                target = ([log_amp_target, 0] if flatten_phase else [log_amp_target]) * len(idx_batch)
                if not flatten_phase:
                    log_psi = log_psi[..., 0]
                loss = F.mse_loss(log_psi.reshape(-1), torch.tensor(target, dtype=log_psi.dtype, device=log_psi.device))
                loss.backward()
                opt.step()
            if (i % output_freq == 0) or (i == 1):
                tpe = (time.time() - t0) / (1 if i == 1 else output_freq)
                print(f"\t Epoch {i} : loss = {loss.item():.5e}, |psi|^2 = {log_psi[..., 0].exp().norm().detach().item() if flatten_phase else log_psi.exp().norm().detach().item():.3f}, epoch time={tpe:.2f}s")
                t0 = time.time()
        opt.zero_grad()
        for p_ in wf.param_list():
            p_.grad = None
        wf.parameters_changed()                               # the fused kernels' packed weights follow
        print("done.")

    def _fused_step_conditions(self):
        """What the single-GPU fused branch of _SGD_step and every one-call form of the step (single process or sharded) have
        in common — everything except who owns the rows: the fused HIP path with nothing between forward and E_loc, and the
        A/B switches NAQS_TRAIN_FUSED_ELOC / NAQS_TRAIN_PREFUSE."""
        if not self.use_fused or self.normalize_grads or self.bug_compat_full_sample_order:
            return False
        if os.environ.get("NAQS_TRAIN_FUSED_ELOC", "1") != "1" or os.environ.get("NAQS_TRAIN_PREFUSE", "1") != "1":
            return False
        fused = self.wavefunction.fused(need_phase=True)
        return fused is not None and fused.train_mode == "hip"

    def _can_prefuse(self):
        """The conditions under which _SGD_step takes its single-GPU fused branch (forward + E_loc in one call) — known
        before sampling, so that the sampler's call can include them."""
        return self._active_dist() is None and self._fused_step_conditions()

    def _onecall_update_conditions(self):
        """The update half of a one-call step: FlatAdam on the network's own flat parameter vector with nothing between the
        backward pass and the update (no gradient clipping: experiments/_base.py:224 runs with grad_clip_factor=None)."""
        from .flat_adam import FlatAdam
        if os.environ.get("NAQS_TRAIN_ONECALL", "1") != "1":
            return False
        if self.grad_clip_factor is not None or not isinstance(self.optimizer, FlatAdam):
            return False
        wf = self.wavefunction
        flat = getattr(wf, "_flat_params", None)
        return (flat is not None and flat.data_ptr() == self.optimizer._flat.data_ptr() and wf._views_of(flat, wf.param_list())
                and all(p.grad is None for p in wf.param_list()))

    def _can_onecall(self):
        """The conditions under which a whole training step is ONE library call (``FusedLogPsi.vmc_step``): the single-GPU
        fused branch of _SGD_step, followed by FlatAdam with nothing in between."""
        return self._can_prefuse() and self._onecall_update_conditions()

    def _onecall_step(self):
        """get_samples (adaptive sample count, energy.py:936-971) + _SGD_step (energy.py:273-377) through
        ``FusedLogPsi.vmc_step``: the library abandons a step right after sampling when get_samples would have re-sampled
        (tree overflow, or too few unique samples while the count may still grow), and this loop then adapts n_samples with
        the reference's rules and messages.  -> (counts, weights, ev) of the step that was taken."""
        wf = self.wavefunction
        fused = wf.fused(need_phase=True)
        last_action = 0
        while True:
            free = (self.n_samples != self.n_unq_samples_min) and (self.n_samples != self.n_samples_max)
            m_lo = self.n_unq_samples_min if (free and last_action >= 0) else 0
            seed = wf._next_sample_seed(self.generator)
            slot = self._sampled_ring_slot(int(self.n_unq_samples_max)) if self.track_sampled_idxs else None
            taken, n_unq, overflow, out = fused.vmc_step(self.pauli_hamiltonian, self.n_samples, seed, self.n_unq_samples_max,
                                                          m_lo, self.n_unq_samples_max, adam=self.optimizer, keys_out=slot)
            if taken:
                break
            action = 0
            if overflow:
                print("MaxBatchSizeExceededError")
                n_unq, action = self.n_unq_samples_max + 1, -1
            if free or overflow:
                if n_unq < self.n_unq_samples_min and last_action >= 0:
                    action = 1
                    self.n_samples = int(min(self.n_samples * 10, self.n_samples_max))
                    print(f"\t...{n_unq} unique samples generated --> increasing batch size to "
                          f"{self.n_samples / 1e6:.1f}M at epoch {self.n_epochs}.")
                elif n_unq > self.n_unq_samples_max and last_action <= 0:
                    action = -1
                    self.n_samples = int(max(self.n_samples / 10, self.n_unq_samples_min))
                    print(f"\t...{n_unq} unique samples generated --> decreasing batch size to "
                          f"{self.n_samples / 1e6:.1f}M at epoch {self.n_epochs}.")
            if action == 0:
                raise RuntimeError(f"VMC step abandoned without a reason to re-sample (M={n_unq})")
            last_action = action
        keys, counts, probs, weights, lp, e_loc, sums, g, ev = out
        wf.fused_repacked()
        self._sample_keys, self._sample_weights, self._prefused = keys, weights, None
        if self.track_sampled_idxs:
            self._sampled_ring_off += int(keys.shape[0])       # the keys are in the tracking buffer already (energy.py:300)
        self._loss_terms, self._last_loss = (g, lp), None
        if self.scheduler is not None:
            self.scheduler.step()
        return counts, weights, ev

    def _can_shard_onecall(self):
        """The sharded step as four library calls (``FusedLogPsi.shard_*``): the one-call step's conditions (one shared
        predicate, A/B switches included) with a process group in place of the single process."""
        if self._active_dist() is None or not self._fused_step_conditions() or not self._onecall_update_conditions():
            return False
        return not self.wavefunction.fused(need_phase=True).aggregate

    def _sharded_onecall_step(self):
        """One row-sharded VMC step = four library calls and three collectives (include/naqs_hip.h): sampler +
        forward of my rows | all-gather | E_loc of my rows + sums + proof | all-reduce | backward | all-reduce | Adam + re-pack.
        The adaptive sample count is ``_onecall_step``'s.  -> (counts, weights, ev)."""
        dist = self._active_dist()
        world, rank = dist.get_world_size(), dist.get_rank()
        wf = self.wavefunction
        fused = wf.fused(need_phase=True)
        last_action = 0
        while True:
            free = (self.n_samples != self.n_unq_samples_min) and (self.n_samples != self.n_samples_max)
            m_lo = self.n_unq_samples_min if (free and last_action >= 0) else 0
            seed = wf._next_sample_seed(self.generator)
            slot = self._sampled_ring_slot(int(self.n_unq_samples_max)) if self.track_sampled_idxs else None
            taken, n_unq, overflow, sb = fused.shard_sample_forward(self.n_samples, seed, self.n_unq_samples_max, m_lo,
                                                                    self.n_unq_samples_max, rank, world, keys_out=slot)
            if taken:
                break
            action = 0
            if overflow:
                print("MaxBatchSizeExceededError")
                n_unq, action = self.n_unq_samples_max + 1, -1
            if free or overflow:
                if n_unq < self.n_unq_samples_min and last_action >= 0:
                    action = 1
                    self.n_samples = int(min(self.n_samples * 10, self.n_samples_max))
                    print(f"\t...{n_unq} unique samples generated --> increasing batch size to "
                          f"{self.n_samples / 1e6:.1f}M at epoch {self.n_epochs}.")
                elif n_unq > self.n_unq_samples_max and last_action <= 0:
                    action = -1
                    self.n_samples = int(max(self.n_samples / 10, self.n_unq_samples_min))
                    print(f"\t...{n_unq} unique samples generated --> decreasing batch size to "
                          f"{self.n_samples / 1e6:.1f}M at epoch {self.n_epochs}.")
            if action == 0:
                raise RuntimeError(f"VMC step abandoned without a reason to re-sample (M={n_unq})")
            last_action = action
        M = n_unq
        prof = os.environ.get("NAQS_SHARD_PROFILE") == "1"             # developer aid: synchronised per-stage wall times
        if prof:
            torch.cuda.synchronize(); t_ = [time.perf_counter()]
            def mark():
                torch.cuda.synchronize(); t_.append(time.perf_counter())
        else:
            def mark():
                pass
        dist.all_gather_into_tensor(sb["table"], sb["mine"])           # equal padded shards: [world][S_pad][2]
        mark()
        b, e = fused.shard_eloc(self.pauli_hamiltonian, sb, M, rank, world)
        mark()
        ext = sb["ext"]
        dist.all_reduce(ext)
        shard_ok = (world * ext[5] == ext[4] * ext[4]) & (world * ext[7] == ext[6] * ext[6])
        self._shard_mismatch = (~shard_ok if self._shard_mismatch is None else self._shard_mismatch | ~shard_ok)
        mark()
        fused.shard_backward(sb, b, e)
        mark()
        dist.all_reduce(fused._grad_flat)                              # shards SUM to the full-batch gradient
        fused.shard_update(self.optimizer)
        mark()
        if prof:
            acc = self.__dict__.setdefault("_shard_prof", [0.0] * 6)
            for i_ in range(5):
                acc[i_] += t_[i_ + 1] - t_[i_]
            acc[5] += 1
            if acc[5] % 20 == 0:
                print("[shard profile] gather %.0f us | eloc+proof %.0f | reduce+check %.0f | backward %.0f | reduce+update %.0f (rows %d of %d)"
                      % tuple([a_ / acc[5] * 1e6 for a_ in acc[:5]] + [e - b, M]), flush=True)
        wf.fused_repacked()
        keys = sb["keys_used"][:M]
        self._sample_keys, self._sample_weights, self._prefused = keys, sb["weights"][:M], None
        if self.track_sampled_idxs:
            self._sampled_ring_off += M
        self._loss_terms, self._last_loss = (sb["g"][:e - b], sb["mine"][:e - b]), None
        if self.scheduler is not None:
            self.scheduler.step()
        ev = torch.where(shard_ok, sb["ev"], torch.full_like(sb["ev"], float("nan")))
        return sb["counts"][:M], sb["weights"][:M], ev

    def get_samples(self, last_action=0, lazy=False):
        """Adaptive sample count (energy.py:936-971): x10 while too few unique samples, /10 when too many
        or when the unique-prefix tree exceeds ``n_unq_samples_max``.  -> (states, counts, probs); the keys and the
        weights counts / sum(counts) of the same draw are left in ``self._sample_keys`` / ``self._sample_weights``.
        ``lazy`` (the training loop): states come back as ``LazyStates`` (built from the keys only if looked at)."""
        action = 0
        if self.use_fused:
            # one full re-pack of the parameters now (sampler, forward and backward all read it) instead of an
            # amplitude-only one here and a full one before the forward pass
            self.wavefunction.fused(need_phase=True)
        self._prefused = None
        try:
            if lazy and self._can_prefuse():
                # training loop, single GPU, HIP forward/backward: sampling, forward and E_loc in one library call — what
                # _SGD_step would compute first for these keys is already on its way when the host learns M
                states, counts, probs, self._sample_keys, self._sample_weights, pre = self.wavefunction.sample_with_local_energy(
                    self.pauli_hamiltonian, self.n_samples, self.n_unq_samples_max, generator=self.generator)
                self._prefused = (self._sample_keys, pre)
            else:
                states, counts, probs, self._sample_keys, self._sample_weights = self.wavefunction.sample(
                    self.n_samples, ret_log_psi=False, max_batch_size=self.n_unq_samples_max, generator=self.generator,
                    ret_keys=True, lazy_states=lazy, ret_weights=True)
            n_unq, completed = len(states), True
        except MaxBatchSizeExceededError:
            print("MaxBatchSizeExceededError")
            n_unq, completed, action = self.n_unq_samples_max + 1, False, -1
        if ((self.n_samples != self.n_unq_samples_min) and (self.n_samples != self.n_samples_max)) or not completed:
            if n_unq < self.n_unq_samples_min and last_action >= 0:
                action = 1
                self.n_samples = int(min(self.n_samples * 10, self.n_samples_max))
                print(f"\t...{n_unq} unique samples generated --> increasing batch size to "
                      f"{self.n_samples / 1e6:.1f}M at epoch {self.n_epochs}.")
            elif n_unq > self.n_unq_samples_max and last_action <= 0:
                action = -1
                self.n_samples = int(max(self.n_samples / 10, self.n_unq_samples_min))
                print(f"\t...{n_unq} unique samples generated --> decreasing batch size to "
                      f"{self.n_samples / 1e6:.1f}M at epoch {self.n_epochs}.")
        if action != 0:
            return self.get_samples(action, lazy=lazy)
        return states, counts, probs

    def solve_H(self, n_samps=None, ret_n_samps=True):
        """Lowest eigenpair of H restricted to the sampled states (energy.py:762-786)."""
        import scipy.sparse.linalg as spla
        if n_samps is None:
            n_samps = self.get_n_samples()
        with torch.no_grad():
            states, counts, probs = self.wavefunction.sample(n_samps, ret_log_psi=False, generator=self.generator)
        n_unq = len(states)
        limit = self.solve_H_max_states
        if n_unq > limit:
            print(f"Limiting number of sampled states from {n_unq} is to most likely {limit}.")
            states = states[torch.argsort(counts)[-limit:]]
        keys = self.hilbert.state2idx(states).squeeze(-1)
        if self.device.type == "cuda" and len(keys) >= 3:
            # matrix-free Lanczos on the device: no sub-matrix is formed, so the reference's 10 000-state cap
            # (there for the cost of the CSR slice) is only kept as the default of solve_H_max_states
            val, vec = self.pauli_hamiltonian.lowest_eigenpair(keys_to_device(keys, self.device))
            vec = vec.cpu().numpy()
            vec = vec * np.sign(vec[0])
            return (float(val), vec[0], n_unq) if ret_n_samps else (float(val), vec[0])
        H = self.pauli_hamiltonian.get_H(keys)
        if H.shape[0] < 3:
            w, v = np.linalg.eigh(H.toarray())
            val, vec = w[0], v[:, 0]
        else:
            w, v = spla.eigsh(H.astype(np.float64), k=1, which='SA')
            val, vec = w[0], v[:, 0]
        vec = vec * np.sign(vec[0])
        return (float(val), vec[0], n_unq) if ret_n_samps else (float(val), vec[0])

    def run(self, n_epochs, save_freq=None, save_final=False, reset_log=False, reset_optimizer=False,
            output_freq=50):
        if reset_log:
            self.reset_log()
        if reset_optimizer:
            self.reset_optimizer()
        run_time_at_last_log, steps_at_last_log = self.run_time, self.n_steps
        # the step form (one call / sharded one-call / pieces) is decided once per run() and on a distributed-mode switch:
        # use_fused, grad_clip_factor, normalize_grads, the optimiser ... may all have changed since the last run()
        self._onecall_cached = None
        _freeze_garbage_collector()
        print("Training NAQS energy.  Samples will be weighted by their frequency.")
        if self.n_steps == 0:
            self.save(os.path.join(self.save_loc, f"opt_{self.n_steps}steps"), quiet=False)
        self._in_run = True
        try:
            self._run_epochs(n_epochs, save_freq, output_freq, run_time_at_last_log, steps_at_last_log)
        finally:
            self._in_run = False
        self._flush_log()
        if save_final:
            self.save(quiet=False)

    def _can_run_in_library(self):
        """The loop itself in C (``naqs_vmc_run``): the one-call step without an LR scheduler stepping in between
        (NAQS_TRAIN_RUN=0: one library call per step, as before)."""
        return self.scheduler is None and os.environ.get("NAQS_TRAIN_RUN", "1") == "1"

    def _library_run(self, n_steps):
        """Up to ``n_steps`` steps of `run` through ``FusedLogPsi.vmc_run`` — sampling with the adaptive sample count
        (energy.py:936-971), E_loc, backward, Adam, re-pack per step, nothing returning to the interpreter in between.  Books
        the steps (log entries, counters, tracking buffer) exactly as the per-step loop does and prints the reference's
        adaptation messages.  -> (counts, weights, steps taken) of the LAST step; fewer steps than asked when the tracking
        buffer has to be folded first."""
        wf = self.wavefunction
        fused = wf.fused(need_phase=True)
        cap = int(self.n_unq_samples_max)
        ring, off = None, 0
        if self.track_sampled_idxs:
            self._sampled_ring_slot(cap)                   # (creates the buffer / folds it when the next slot would not fit)
            ring, off = self._sampled_ring, self._sampled_ring_off
        base = int(self.generator.initial_seed()) if self.generator is not None else int(torch.initial_seed())
        t0, run_time0, epoch0 = time.time(), self.run_time, self.n_epochs
        res = fused.vmc_run(self.pauli_hamiltonian, n_steps, self.optimizer, self.n_samples, int(self.n_samples_max),
                            int(self.n_unq_samples_min), cap, base, wf._sample_calls, ring=ring, ring_off=off)
        for step, n_unq, overflow, action, n_samples in res["events"]:
            if overflow:
                print("MaxBatchSizeExceededError")
            if action > 0:
                print(f"\t...{n_unq} unique samples generated --> increasing batch size to "
                      f"{n_samples / 1e6:.1f}M at epoch {epoch0 + step}.")
            elif action < 0:
                print(f"\t...{n_unq} unique samples generated --> decreasing batch size to "
                      f"{n_samples / 1e6:.1f}M at epoch {epoch0 + step}.")
        self.n_samples = res["n_samples"]
        wf._sample_calls = res["sample_calls"]
        if self.track_sampled_idxs:
            self._sampled_ring_off = res["ring_off"]
        done = res["steps"]
        for i in range(done):
            self.n_steps += 1
            self.n_epochs += 1
            self._pending_log.append((self.n_steps, res["ev"][i], res["M"][i], run_time0 + res["t"][i]))
        self.run_time = run_time0 + (time.time() - t0)
        if done:
            wf.fused_repacked()
            self._last_M = res["M"][-1]
            self._sample_keys, self._sample_weights, self._prefused = res["keys"], res["weights"], None
            self._loss_terms, self._last_loss = (res["g"], res["log_psi"]), None
        # failures are raised only now: the `done` steps that finished are applied updates, and the counters above (Adam's step,
        # the sampler's call number, the tracking ring, the log) have to describe the parameters as they are
        if res["error"] is not None:
            raise res["error"]
        if res["stop_reason"] == 3:
            raise RuntimeError(f"VMC step abandoned without a reason to re-sample (M={res['events'][-1][1] if res['events'] else '?'})")
        return res["counts"], res["weights"], done

    def _run_epochs(self, n_epochs, save_freq, output_freq, run_time_at_last_log, steps_at_last_log):
        remaining = n_epochs
        while remaining > 0:
            t0 = time.time()
            self._choose_dist_mode()                 # multi-GPU: replicate or shard this step (resets the cache on a switch)
            onecall = getattr(self, "_onecall_cached", None)
            if onecall is None:
                onecall = self._onecall_cached = (2 if self._can_shard_onecall() else (1 if self._can_onecall() else 0))
            if onecall == 1 and self._can_run_in_library():
                # as many steps as fit before the loop has something to do itself: the first epoch's line, an output line,
                # a checkpoint
                chunk = min(remaining, 1 if self.n_epochs == 0 else output_freq - self.n_epochs % output_freq)
                if save_freq is not None and save_freq > 0:
                    chunk = min(chunk, save_freq - self.n_epochs % save_freq)
                if self._dist_mode == "replicated" and self.replica_proof_every > 0:
                    # several ranks, every one running the identical single-GPU loop: a chunk ends where the ranks owe each
                    # other the same-table proof (and where a switch to the sharded step may be decided on its agreed count)
                    chunk = min(chunk, self.replica_proof_every - self.n_steps % self.replica_proof_every)
                counts, weights, done = self._library_run(chunk)
                if done and self._dist_mode == "replicated":
                    self._replica_proof(self._sample_keys)       # (gates on n_steps % replica_proof_every itself)
                remaining -= done
                if done == 0:
                    continue                         # (the tracking buffer was full: folded on the way back in)
            else:
                if onecall == 2:
                    counts, weights, ev = self._sharded_onecall_step()
                elif onecall:
                    counts, weights, ev = self._onecall_step()
                else:
                    states, counts, probs = self.get_samples(lazy=True)
                    weights = self._sample_weights                                            # counts / sum(counts), energy.py:993
                    keys = self._sample_keys                                                  # = hilbert.state2idx(states)
                    # <E>, Var stay on the device until they are printed or saved: reading them here would drain the queue
                    # every step, and the ~25 launches of the next sampling call would be issued to an idle GPU
                    ev = self._SGD_step(states, keys, None, sample_weights=weights, lazy=True)
                self.n_steps += 1
                self._last_M = len(weights)
                if self._dist_mode == "replicated":
                    self._replica_proof(self._sample_keys)
                self.run_time += time.time() - t0
                self._pending_log.append((self.n_steps, ev, len(weights), self.run_time))
                self.n_epochs += 1
                remaining -= 1
            if (self.n_epochs % output_freq == 0) or (self.n_epochs == 1):
                self._flush_log()
                var = self.log[LogKey.E_LOC_VAR][-1][1]
                energy = self.calculate_energy(normalise_psi=True) if self.log_exact_energy else None
                self.log[LogKey.E].append((self.n_steps, energy))
                recent = [x[1] for x in self.log[LogKey.E_LOC][-min(output_freq, self.n_epochs):]]
                tpe = (self.run_time - run_time_at_last_log) / output_freq
                run_time_at_last_log = self.run_time
                n_steps_in = self.n_steps - steps_at_last_log
                steps_at_last_log = self.n_steps
                n = counts.sum().item()
                samp = f"{n:.1f}" if n < 1e3 else (f"{n / 1e3:.1f}k" if n < 1e6 else
                                                   (f"{n / 1e6:.1f}M" if n < 1e9 else f"{n / 1e9:.1f}B"))
                e_str = "N/A" if energy is None else f"{energy:.5f}"
                print(f"Epoch {self.n_epochs} ({n_steps_in} SGD steps with {samp} samples ({len(weights)} unq.) : "
                      f"<E>={e_str}, <E_loc>={np.mean(recent):.5f} +\\- {np.std(recent):.5f}, "
                      f"var(<E_loc>)={var:.5f}, epoch time={tpe:.2f}s, total time={self.run_time:.1f}s")
            if save_freq is not None and self.n_epochs % save_freq == 0:
                self.save(os.path.join(self.save_loc, f"opt_{self.n_steps}steps"), quiet=True)
