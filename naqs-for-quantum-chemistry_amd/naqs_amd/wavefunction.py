"""Ansatz API: the counterpart of the reference's ``NAQSComplex_NADE_orbitals``
(src/naqs/wavefunction.py:18-521) with the network resident on the GPU.

Same constructor keywords, method names and return conventions (``sample`` ->
``[states int8 [M, N] in qubit order, counts int64 [M], probs float32 [M], log_psi float32 [M, 2]]``
with ``log_psi`` carrying gradients; ``log_psi`` / ``psi`` / ``amplitude`` / ``phase`` /
``parameters(group_idx)`` / ``conditional_parameters`` / ``save`` / ``load``).  Unlike the reference
(``out_device="cpu"``, wavefunction.py:27, nade.py:194) tensors stay on ``self.device`` unless the
caller asks for numpy output.
"""
import os

import numpy as np
import torch
from torch import nn

from .hilbert import Encoding
from .nade import (InputEncoding, MaxBatchSizeExceededError, NadeMasking, OrbitalNADE,  # noqa: F401
                   SoftmaxLogProbAmps)


class LazyStates:
    """The occupation strings of a batch of sampled keys, materialised on demand.  The training loop only needs the keys
    (the fused kernels read occupations straight from the key bits); building the int8 [M, N] tensor the reference's
    ``sample`` returns costs half a dozen launches per step, so it is deferred until somebody looks at it."""

    def __init__(self, hilbert, keys):
        self._hilbert, self._keys, self._states = hilbert, keys, None

    def __len__(self):
        return int(self._keys.shape[0])

    @property
    def shape(self):
        return (len(self), self._hilbert.N)

    def tensor(self):
        if self._states is None:
            self._states = self._hilbert.idx2state(self._keys)
        return self._states

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return self._hilbert.idx2state(self._keys[idx])
        return self.tensor()[idx]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):         # torch.*(lazy) sees the tensor
        args = tuple(a.tensor() if isinstance(a, LazyStates) else a for a in args)
        return func(*args, **(kwargs or {}))

    def __getattr__(self, name):                                             # .to(), .numpy(), .dim() ...
        return getattr(self.tensor(), name)


class NAQSComplex_NADE_orbitals:
    _cplx_dtype = np.complex64

    def __init__(self, hilbert, N_up=None, N_alpha=None, N_beta=None, qubit_ordering=-1, num_lut=0,
                 input_encoding=InputEncoding.BINARY, n_electrons=None, n_alpha_electrons=None,
                 n_beta_electrons=None, masking=NadeMasking.PARTIAL,
                 amp_hidden_size=(), amp_hidden_activation=nn.ReLU, amp_bias=True,
                 phase_hidden_size=(), phase_hidden_activation=nn.ReLU, phase_bias=True,
                 combined_amp_phase_blocks=False, use_amp_spin_sym=True, use_phase_spin_sym=True,
                 aggregate_phase=True, amp_batch_norm=False, phase_batch_norm=False, batch_norm_momentum=1,
                 amp_activation=SoftmaxLogProbAmps, phase_activation=None, device=None, out_device=None):
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        self.out_device = self.device if out_device is None else torch.device(out_device)
        self.hilbert = hilbert
        self.encoding = hilbert.encoding
        if hilbert.N_occ != 0:
            raise NotImplementedError("frozen (always-occupied) qubits are out of scope")
        self._N_model, self._N_fixed = hilbert.N, 0
        N = hilbert.N
        # qubit <-> model permutations (wavefunction.py:56-83, :369-383)
        if qubit_ordering == 1:
            self.permute_qubits = False
            q2m = np.arange(N)
            self.state2model_permutation_shell = np.arange(N // 2)
        elif qubit_ordering == -1:
            self.permute_qubits = True
            q2m = np.stack([np.arange(N - 2, -1, -2), np.arange(N - 1, -1, -2)], 1).reshape(-1)
            self.state2model_permutation_shell = np.arange(N // 2 - 1, -1, -1)
        else:
            self.permute_qubits = True
            if isinstance(qubit_ordering, int) and qubit_ordering == 0:
                q2m = np.random.permutation(N)
            else:
                q2m = np.array(qubit_ordering)
                assert len(q2m) == N and len(set(q2m.tolist())) == N, "custom ordering must list each qubit once"
            self.state2model_permutation_shell = q2m[1::2] // 2
        self.qubit2model_permutation = q2m
        self.model2qubit_permutation = np.argsort(q2m)
        self.model2state_permutation_shell = np.argsort(self.state2model_permutation_shell)

        self.model = OrbitalNADE(
            num_qubits=N, num_lut=num_lut, input_encoding=input_encoding, n_electrons=n_electrons,
            n_alpha_electrons=n_alpha_electrons, n_beta_electrons=n_beta_electrons, masking=masking,
            amp_hidden_size=list(amp_hidden_size), amp_hidden_activation=amp_hidden_activation, amp_bias=amp_bias,
            phase_hidden_size=list(phase_hidden_size), phase_hidden_activation=phase_hidden_activation,
            phase_bias=phase_bias, combined_amp_phase_blocks=combined_amp_phase_blocks,
            use_amp_spin_sym=use_amp_spin_sym, use_phase_spin_sym=use_phase_spin_sym,
            aggregate_phase=aggregate_phase, amp_batch_norm=amp_batch_norm, phase_batch_norm=phase_batch_norm,
            batch_norm_momentum=batch_norm_momentum, amp_activation=amp_activation,
            phase_activation=phase_activation, device=self.device)
        self._q2m = torch.as_tensor(self.qubit2model_permutation, device=self.device)
        self._m2q = torch.as_tensor(self.model2qubit_permutation, device=self.device)
        self._m2s_shell = torch.as_tensor(self.model2state_permutation_shell, device=self.device)
        self.model.train()
        self.model.predict()
        self._fused, self._fused_version, self._fused_amp_version = None, None, None
        self._param_epoch = 0
        self._sample_calls = 0
        self._param_list = None
        self._flat_params = None

    # ---- mode helpers (wavefunction.py:90-100)
    def train_model(self):
        self.model.train()

    def eval_model(self):
        self.model.eval()

    def sample_model(self):
        self.model.sample()

    def predict_model(self):
        self.model.predict()

    # ---- evaluation (wavefunction.py:167-215, :397-414, :466-481)
    def state2shell(self, s):
        shp = s.shape
        s = s.to(self.device)
        return ((s.reshape(shp[0], shp[-1] // 2, 2) > 0).long() * torch.tensor([1, 2], device=self.device)).sum(-1)

    def _evaluate_model(self, s):
        self.model.predict()
        x = s.to(self.device)[..., self._q2m].float()
        return self.model(x)[:, self._m2s_shell]

    def _evaluate_log_psi(self, s, gather_state=True):
        out = self._evaluate_model(s)                                  # [B, N/2 (state shell order), 4, 2]
        if gather_state:
            sel = self.state2shell(s).view(out.shape[0], -1, 1, 1).expand(-1, -1, 1, 2)
            out = out.gather(-2, sel)
        return out

    def log_psi(self, s, ret_complex=False, combine_conditionals=True):
        if s.dim() < 2:
            s = s.unsqueeze(0)
        log_psi = self._evaluate_log_psi(s, gather_state=True)
        if combine_conditionals:
            log_psi = log_psi.sum(axis=1)
        log_psi = log_psi.squeeze()
        if ret_complex:
            v = log_psi.detach().cpu().numpy()
            return (v[..., 0] + 1j * v[..., 1]).astype(self._cplx_dtype)
        return log_psi

    def psi(self, s, ret_complex=False, combine_conditionals=True):
        lp = self.log_psi(s, ret_complex=False, combine_conditionals=combine_conditionals)
        psi = torch.stack([lp[..., 0].exp() * lp[..., 1].cos(), lp[..., 0].exp() * lp[..., 1].sin()], -1)
        if ret_complex:
            v = psi.detach().cpu().numpy()
            return (v[..., 0] + 1j * v[..., 1]).astype(self._cplx_dtype)
        return psi

    def amplitude(self, s, combine_conditionals=True):
        log_amps = self.log_psi(s, combine_conditionals=False)[..., 0]
        if combine_conditionals:
            log_amps = log_amps.sum(axis=-1).squeeze()
        if log_amps.dim() == 0:
            log_amps = log_amps.unsqueeze(0)
        return log_amps.exp()

    def phase(self, s, combine_conditionals=True):
        phases = self.log_psi(s, combine_conditionals=False)[..., 1]
        if combine_conditionals:
            phases = phases.sum(axis=-1).squeeze()
        return phases

    # ---- sampling (wavefunction.py:488-521)
    # ---- fused HIP kernels for this network (naqs_amd.fused), kept in step with the parameters
    def fused(self, need_phase=True):
        """The ``FusedLogPsi`` handle of this network (created on first use; ``None`` when the architecture is
        outside the fused family or the network is not on a HIP device).  Its packed copy of the weights is
        refreshed whenever a parameter has been modified in place since the last call (tensor version counters:
        optimiser steps, ``load_state_dict``); ``need_phase=False`` (sampling, training forward/backward) re-packs
        only the amplitude blocks."""
        if self._fused is None:
            if self.device.type != "cuda":
                return None
            from .fused import FusedLogPsi
            try:
                self._fused = FusedLogPsi(self)
            except NotImplementedError as why:
                self._fused = False
                # loud, once: the caller asked for an architecture outside the fused family
                print(f"[naqs_amd] fused HIP network kernels not available for this ansatz ({why}): sampling, log psi and "
                      f"back-propagation run as PyTorch modules on {self.device}; E_loc stays on the HIP kernels.")
            self._fused_version = self._fused_amp_version = self._param_version()
        if self._fused is False:
            return None
        v = self._param_version()
        if need_phase and v != self._fused_version:
            self._fused.refresh()
            self._fused_version = self._fused_amp_version = v
        elif not need_phase and v != self._fused_amp_version:
            self._fused.refresh(amp_only=True)
            self._fused_amp_version = v
            self._fused_version = None
        return self._fused

    def param_list(self):
        """The network's parameters in state_dict order, cached: ``nn.Module.parameters()`` walks the module tree on
        every call (~0.4 ms for this network), which the per-step bookkeeping below would pay several times."""
        if self._param_list is None:
            self._param_list = list(self.model.parameters())
        return self._param_list

    def flatten_parameters(self):
        """Make every network parameter a view into one flat float32 buffer (state_dict order) and return it.  The
        fused kernels then read the parameters without gathering them, and ``FlatAdam`` updates them in one launch.
        ``load_state_dict`` / optimiser steps write through the views, so the buffer is always current."""
        params = self.param_list()
        flat = getattr(self, "_flat_params", None)
        if flat is not None and self._views_of(flat, params):
            return flat
        flat = torch.cat([p.detach().reshape(-1) for p in params]).to(torch.float32).contiguous()
        off = 0
        for p in params:
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self._flat_params = flat
        return flat

    @staticmethod
    def _views_of(flat, params):
        off, size = 0, flat.element_size()
        for p in params:
            if p.data_ptr() != flat.data_ptr() + off * size or not p.is_contiguous():
                return False
            off += p.numel()
        return off == flat.numel()

    def _param_version(self):
        # (a Parameter whose .data was pointed into the flat buffer keeps its OWN version counter, so load_state_dict /
        # in-place edits of a parameter show up here and not on the flat tensor: every parameter is looked at)
        return (self._param_epoch,) + tuple((p.data_ptr(), p._version) for p in self.param_list())

    def parameters_changed(self):
        """Tell the fused kernels that the parameters were modified by something that does not bump the tensors'
        version counters (the multi-tensor fused optimisers write through raw pointers)."""
        self._param_epoch += 1

    def fused_repacked(self):
        """The library has updated the parameters in place AND re-packed its weight layouts from them
        (``FusedLogPsi.vmc_step``): move on to the new parameter version without packing again."""
        self._param_epoch += 1
        self._fused_version = self._fused_amp_version = self._param_version()

    def _next_sample_seed(self, generator=None):
        """One 64-bit seed per sampling call: splitmix64 of the generator's seed and a call counter."""
        base = int(generator.initial_seed()) if generator is not None else int(torch.initial_seed())
        self._sample_calls += 1
        x = (base + 0x9E3779B97F4A7C15 * self._sample_calls) & 0xFFFFFFFFFFFFFFFF
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return x ^ (x >> 31)

    def sample_with_local_energy(self, ham, num_samples, max_batch_size, generator=None):
        """The training loop's ``sample(..., ret_keys, lazy_states, ret_weights)`` followed by the forward pass (activations
        kept for the backward) and the local energies of the sampled table, as ONE library call on the fused HIP path
        (``FusedLogPsi.sample_forward_local_energy``).  -> (states, counts, probs, keys, weights, pre) with ``pre`` what
        ``forward_saved_with_local_energy`` would have returned for these keys."""
        fused = self.fused(need_phase=True)
        seed = self._next_sample_seed(generator)
        keys, counts, probs, weights, pre = fused.sample_forward_local_energy(ham, int(num_samples), seed, int(max_batch_size))
        return LazyStates(self.hilbert, keys), counts, probs, keys, weights, pre

    def sample(self, num_samples=1, ret_probs=True, ret_log_psi=True, ret_norm_reg=False, eval_mode=False,
               max_batch_size=None, generator=None, use_fused=None, ret_keys=False, lazy_states=False, ret_weights=False):
        """wavefunction.py:488-521.  On a HIP device the draw is ``naqs_net_sample`` (one library call for the
        whole tree); ``use_fused=False`` selects the PyTorch formulation of the same sampler
        (``OrbitalNADE._forward_sample``, ``torch.binomial`` on the device), which is also what runs for
        architectures outside the fused family.  ``ret_keys`` appends the int64 keys of the states, ``ret_weights`` the
        float64 weights counts / sum(counts) (energy.py:993; from the sampler's own last launch on the fused path);
        ``lazy_states`` returns the states as a ``LazyStates`` that builds the int8 tensor only when looked at."""
        if ret_norm_reg:
            raise NotImplementedError("ret_norm_reg")
        fused = self.fused(need_phase=False) if use_fused in (None, True) else None
        if use_fused and fused is None:
            raise NotImplementedError("fused sampler: network not on a HIP device or architecture not supported")
        keys = None
        if fused is not None:
            # one 64-bit seed per call, derived on the host (splitmix64 of the generator's seed and a call counter):
            # drawing it from a device generator would cost a queue-draining read-back before the sampler is queued
            seed = self._next_sample_seed(generator)
            if max_batch_size is not None:
                cap = int(max_batch_size)
            else:       # live prefixes are bounded by the physical space unless no conditional is masked
                bound = 4 ** (self.hilbert.N // 2)
                if self.model.masking is not NadeMasking.NONE and self.model.use_restricted_hilbert:
                    bound = min(bound, self.hilbert.size)
                cap = int(min(bound, 2 ** 22))
            if ret_weights:
                keys, counts, probs, weights = fused.sample(int(num_samples), seed, cap, with_weights=True)
            else:
                keys, counts, probs = fused.sample(int(num_samples), seed, cap)
            states = LazyStates(self.hilbert, keys) if lazy_states else self.hilbert.idx2state(keys)
        else:
            was_training = self.model.training
            self.model.eval() if eval_mode else self.model.train()
            self.model.sample()
            try:
                states_m, counts, probs = self.model(num_samples, ret_output=False, max_batch_size=max_batch_size,
                                                     generator=generator)
            finally:
                self.model.predict()
                self.model.train(was_training)
            states = states_m[:, self._m2q].to(self.hilbert.get_state_dtype("torch"))
        out = [states, counts]
        if ret_probs:
            out.append(probs)
        if ret_log_psi:
            # differentiable log psi of the unique samples: one batched teacher-forced pass
            lp = self.log_psi(states)
            out.append(lp.reshape(-1, 2))
        if ret_keys:
            out.append(keys if keys is not None else self.hilbert.state2idx(states).squeeze(-1).to(torch.int64))
        if ret_weights:
            out.append(weights if keys is not None else counts.double() / counts.sum().double())
        return out

    # ---- parameters (wavefunction.py:416-451)
    def parameters(self, group_idx=None):
        if group_idx is None or group_idx == -1 or group_idx == 0:
            return list(self.model.parameters())
        if group_idx == 1:
            return []                                   # look-up-table blocks: none on this path
        raise NotImplementedError()

    def conditional_parameters(self, cond_idx=None):
        if cond_idx is None:
            return list(self.model.parameters())
        params = list(self.model.amp_layers[cond_idx].parameters())
        if self.model.aggregate_phase:
            params += list(self.model.phase_layers[cond_idx].parameters())
        elif cond_idx == self._N_model // 2 - 1:
            params += list(self.model.phase_layers[0].parameters())
        return params

    def count_parameters(self, print_verbose=True):
        total = 0
        for name, p in self.model.named_parameters():
            if p.requires_grad:
                if print_verbose:
                    print(f"{name} : {p.numel()}")
                total += p.numel()
        print(f"\n--> Total trainable params: {total}")
        return total

    # ---- checkpoints, same keys as the reference (wavefunction.py:240-262)
    def save(self, fname, quiet=False):
        d = os.path.dirname(fname)
        if d:
            os.makedirs(d, exist_ok=True)
        ckpt = {"model:state_dict": {k: v.cpu() for k, v in self.model.state_dict().items()},
                "wavefunction:permute_qubits": self.permute_qubits,
                "wavefunction:qubit2model_permutation": self.qubit2model_permutation,
                "wavefunction:model2qubit_permutation": self.model2qubit_permutation}
        if os.path.splitext(fname)[-1] != ".pth":
            fname += ".pth"
        torch.save(ckpt, fname)
        if not quiet:
            print("Saved NAQSComplex wavefunction model to {}.".format(fname))
        return fname

    def load(self, fname):
        ckpt = torch.load(fname, map_location=self.device, weights_only=False)
        self.model.load_state_dict(ckpt["model:state_dict"])
        self.permute_qubits = ckpt["wavefunction:permute_qubits"]
        self.qubit2model_permutation = np.asarray(ckpt["wavefunction:qubit2model_permutation"])
        self.model2qubit_permutation = np.asarray(ckpt["wavefunction:model2qubit_permutation"])
        self._q2m = torch.as_tensor(self.qubit2model_permutation, device=self.device)
        self._m2q = torch.as_tensor(self.model2qubit_permutation, device=self.device)
        # the fused kernels hold a packed copy of the weights and the qubit permutation: rebuild on next use
        if self._fused not in (None, False):
            self._fused.close()
        self._fused = None
        self.parameters_changed()
        print("Loaded NAQSComplex wavefunction from {}.".format(fname))

    @torch.no_grad()
    def save_psi(self, fname="psi", subspace_args={}, normalise=True):
        basis, basis_idxs = self.hilbert.get_subspace(ret_states=True, ret_idxs=True)
        amps = self.amplitude(basis)
        if normalise:
            amps = amps / amps.pow(2).sum().pow(0.5)
        phases = self.phase(basis)
        order = torch.argsort(amps, descending=True).cpu()
        psi = torch.stack([amps.cpu()[order], phases.cpu()[order]]).transpose(0, 1)
        np.savetxt(f"{fname}.txt", psi.numpy(), fmt="%5e")
        np.savetxt(f"{fname}_basis.txt", basis[order].clamp(min=0).numpy(), fmt="%i")
        np.savetxt(f"{fname}_basis_idxs.txt", basis_idxs[order].numpy(), fmt="%i")
