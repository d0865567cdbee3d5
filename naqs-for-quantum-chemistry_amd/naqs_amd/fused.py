"""Fused HIP evaluation of log psi for a batch of keys (``naqs_net_*`` entry points).

The product path of the metric's "log-psi eval": two kernels instead of ~200 eager launches.
Supported architecture families:
  * what the reference's published runs use (batch_train.sh:14): one hidden layer per amplitude block (a multiple of 16,
    <= 128 units), a single phase block (``aggregate_phase=False``) of 1..8 hidden layers <= 512 wide;
  * the reference's default ansatz (experiments/run.py:11-31): the same amplitude blocks and one phase block per orbital
    pair (``aggregate_phase=True``), each with one hidden layer (multiple of 16, <= 128 units);
both with or without the phase spin symmetry (``-phase_sym``, nade.py:281, 507-533, 590-610: spin-ordered inputs of the phase
block(s), 3 outputs, the sign shift — since round 5), <= 16 orbital pairs.  Anything else (combined amplitude-phase blocks,
look-up-table blocks) raises ``NotImplementedError`` — callers then stay on the PyTorch modules (same numbers, more launches),
and ``wavefunction.fused()`` says so on stdout.
"""
import ctypes
import os

import torch

from . import _lib
from .hamiltonian import _stream_ptr


class _LogAmp(torch.autograd.Function):
    """log|psi|(keys) with the amplitude blocks' forward and backward as HIP kernels
    (``naqs_net_logamp`` / ``naqs_net_amp_backward``)."""

    @staticmethod
    def forward(ctx, fused, keys, *amp_params):
        M = keys.shape[0]
        out = torch.empty(M, dtype=torch.float32, device=keys.device)
        st = fused._lib.naqs_net_logamp(fused._h, M, keys.data_ptr(), out.data_ptr(), _stream_ptr(fused.device))
        _lib.check(st, "naqs_net_logamp")
        ctx.fused, ctx.keys = fused, keys
        ctx.shapes = [p.shape for p in amp_params]
        return out

    @staticmethod
    def backward(ctx, g):
        fused, keys = ctx.fused, ctx.keys
        g = g.to(torch.float32).contiguous()
        flat = torch.empty(fused.n_amp_params, dtype=torch.float32, device=keys.device)
        st = fused._lib.naqs_net_amp_backward(fused._h, keys.shape[0], keys.data_ptr(), g.data_ptr(), flat.data_ptr(),
                                              _stream_ptr(fused.device))
        _lib.check(st, "naqs_net_amp_backward")
        grads, off = [], 0
        for shp in ctx.shapes:
            n = shp.numel()
            grads.append(flat[off:off + n].view(shp))
            off += n
        return (None, None, *grads)


def _accumulate(p, grad):
    if p.grad is None:
        p.grad = grad
    else:
        p.grad.add_(grad)


class FusedLogPsi:
    def __init__(self, wavefunction):
        wf, m = wavefunction, wavefunction.model
        if m.device.type != "cuda":
            raise _lib.NaqsError("FusedLogPsi needs the network on a HIP device (no CPU fallback)")
        self.aggregate = bool(m.aggregate_phase)
        self.phase_sym = bool(getattr(m, "use_phase_spin_sym", False))
        if getattr(m, "combined_amp_phase_blocks", False):
            raise NotImplementedError("fused log-psi: combined amplitude-phase blocks (no published script uses them) run as "
                                      "PyTorch modules")
        if len(m.phase_layers) != (m.P if self.aggregate else 1):
            raise NotImplementedError("fused log-psi: unexpected number of phase blocks")
        if len(m.amp_layers[0].linears()) != 2:
            raise NotImplementedError("fused log-psi: amplitude blocks need exactly one hidden layer")
        ha = m.amp_layers[0].linears()[0].out_features
        if ha % 16 or ha > 128:
            raise NotImplementedError(f"fused log-psi: amplitude hidden width {ha} (a multiple of 16, <= 128, is supported)")
        if m.P > _lib.NET_MAX_PAIRS or m.P < 2:
            raise NotImplementedError("fused log-psi: 2..16 orbital pairs")
        phase_lin = m.phase_layers[0].linears()
        hidden = [lin.out_features for lin in phase_lin[:-1]]
        if self.aggregate:
            if len(hidden) != 1 or hidden[0] % 16 or hidden[0] > 128:
                raise NotImplementedError("fused log-psi: aggregate_phase=True with other than one phase hidden layer per "
                                          f"block of a multiple of 16, <= 128 units (got {hidden})")
        elif not 1 <= len(hidden) <= _lib.NET_MAX_PHASE_LAYERS or max(hidden) > 512:
            raise NotImplementedError("fused log-psi: 1..8 phase hidden layers of width <= 512")
        self._lib = _lib.load_library()
        self.wf = wf
        self.device = m.device
        cfg = _lib.NetConfig()
        cfg.n_qubits = m.N
        cfg.n_alpha = m.n_alpha_up if m.use_restricted_hilbert else -1
        cfg.n_beta = m.n_beta_up if m.use_restricted_hilbert else -1
        cfg.masking = m.masking.value
        cfg.use_amp_spin_sym = int(m.use_amp_spin_sym)
        cfg.amp_hidden = m.amp_layers[0].linears()[0].out_features
        cfg.n_phase_hidden = len(hidden)
        for i, h in enumerate(hidden):
            cfg.phase_hidden[i] = h
        for i, q in enumerate(wf.qubit2model_permutation):
            cfg.qubit2model[i] = int(q)
        cfg.aggregate_phase = int(self.aggregate)
        cfg.use_phase_spin_sym = int(self.phase_sym)          # -phase_sym with one phase block (nade.py:281, 507-533, 590-610)
        self._h = ctypes.c_void_p(None)
        st = self._lib.naqs_net_create(ctypes.byref(cfg), self.device.index or 0, ctypes.byref(self._h))
        _lib.check(st, "naqs_net_create")
        n = ctypes.c_int64(0)
        _lib.check(self._lib.naqs_net_param_count(self._h, ctypes.byref(n)), "naqs_net_param_count")
        self.n_params = n.value
        self._samp = None
        self._grad_flat, self._grad_views = None, None
        self.train_mode = os.environ.get("NAQS_TRAIN_MODE", "hip")     # "hip" | "blas" (phase MLP through torch/rocBLAS)
        if self.aggregate or self.phase_sym:
            self.train_mode = "hip"                                     # (the per-pair phase blocks / spin-ordered inputs have no BLAS formulation here)
        assert self.n_params == sum(p.numel() for p in m.parameters()), "parameter layout mismatch"
        _lib.check(self._lib.naqs_net_amp_param_count(self._h, ctypes.byref(n)), "naqs_net_amp_param_count")
        self.n_amp_params = n.value
        self._amp_params = [p for blk in m.amp_layers for p in blk.parameters()]
        assert self.n_amp_params == sum(p.numel() for p in self._amp_params), "amplitude parameter layout mismatch"
        # key bits feeding the phase block (alpha then beta occupations of model pairs 0..P-2) and the last pair's outcome
        q2m = [int(q) for q in wf.qubit2model_permutation]
        P = m.P
        self._phase_shifts = torch.tensor([q2m[2 * k] for k in range(P - 1)] + [q2m[2 * k + 1] for k in range(P - 1)],
                                          dtype=torch.int64, device=self.device)
        self._last_a, self._last_b = q2m[2 * (P - 1)], q2m[2 * (P - 1) + 1]
        self.refresh()

    def refresh(self, amp_only=False):
        """Re-pack the current network parameters (after every optimiser step).  ``amp_only`` packs just the
        amplitude blocks — all that sampling and the training forward/backward need; the phase layers are then
        stale for ``log_psi`` / ``log_psi_and_local_energy`` until a full refresh."""
        all_params = self.wf.param_list()
        flat = getattr(self.wf, "_flat_params", None)
        if flat is None or not self.wf._views_of(flat, all_params):       # parameters not (or no longer) flattened: gather
            params = self._amp_params if amp_only else all_params
            flat = torch.cat([p.detach().reshape(-1) for p in params]).to(torch.float32).contiguous()
        fn = self._lib.naqs_net_set_amp_weights if amp_only else self._lib.naqs_net_set_weights
        _lib.check(fn(self._h, flat.data_ptr(), flat.numel(), _stream_ptr(self.device)), "naqs_net_set_weights")
        self._flat = flat          # keep alive until the async copy has been consumed

    def log_psi_train(self, keys):
        """Differentiable log psi [M, 2] of int64 device keys: amplitude blocks through the HIP forward/backward
        pair, the phase MLP (three Linear layers) through PyTorch/rocBLAS.  Same function of the parameters as
        ``wavefunction.log_psi(states)``."""
        m = self.wf.model
        if self.aggregate or self.phase_sym:
            raise NotImplementedError("log_psi_train (autograd.Function form) for aggregate_phase / phase-symmetric networks: use "
                                      "forward_saved / backward_saved")
        keys = keys.contiguous()
        log_amp = _LogAmp.apply(self, keys, *self._amp_params)
        x = ((keys.unsqueeze(-1) >> self._phase_shifts) & 1).to(torch.float32).mul_(2.0).sub_(1.0)
        out = m.phase_layers[0](x)                                                   # nade.py:563-569
        occ = ((keys >> self._last_a) & 1) + 2 * ((keys >> self._last_b) & 1)
        phase = out.gather(1, occ.unsqueeze(1)).squeeze(1)
        return torch.stack([log_amp, phase], -1)

    # ---- training step without the autograd engine --------------------------------------------------------
    @torch.no_grad()
    def forward_saved(self, keys):
        """log psi [M, 2] of int64 device keys, evaluated without recording an autograd graph, plus the token
        ``backward_saved`` needs.  ``mode="hip"`` (default): ``naqs_net_train_forward`` — the inference kernels, the
        phase activations stay in the handle.  ``mode="blas"``: HIP amplitude kernels + one addmm/relu per phase layer."""
        keys = keys.contiguous()
        M = keys.shape[0]
        if self.train_mode == "hip":
            log_psi = torch.empty((M, 2), dtype=torch.float32, device=self.device)
            st = self._lib.naqs_net_train_forward(self._h, M, keys.data_ptr(), log_psi.data_ptr(), _stream_ptr(self.device))
            _lib.check(st, "naqs_net_train_forward")
            return log_psi, (keys, None, None)
        m = self.wf.model
        log_amp = torch.empty(M, dtype=torch.float32, device=self.device)
        st = self._lib.naqs_net_logamp(self._h, M, keys.data_ptr(), log_amp.data_ptr(), _stream_ptr(self.device))
        _lib.check(st, "naqs_net_logamp")
        lin = m.phase_layers[0].linears()
        x = torch.empty((M, lin[0].in_features), dtype=torch.float32, device=self.device)
        occ = torch.empty((M, 1), dtype=torch.int64, device=self.device)
        st = self._lib.naqs_net_phase_inputs(self._h, M, keys.data_ptr(), x.data_ptr(), occ.data_ptr(),
                                             _stream_ptr(self.device))
        _lib.check(st, "naqs_net_phase_inputs")
        acts = [x]
        for layer in lin[:-1]:
            acts.append(torch.addmm(layer.bias, acts[-1], layer.weight.t()).relu_())
        out = torch.addmm(lin[-1].bias, acts[-1], lin[-1].weight.t())
        log_psi = torch.stack([log_amp, out.gather(1, occ).squeeze(1)], -1)
        return log_psi, (keys, acts, occ)

    @torch.no_grad()
    def forward_saved_with_local_energy(self, ham, keys, weights):
        """``forward_saved`` (HIP mode) and the local energies of the same table in one library call
        (``naqs_net_train_forward_eloc``): -> (log psi [M, 2] f32, saved token, E_loc [M, 2] f64, sums [4] f64)."""
        keys = keys.contiguous()
        M = keys.shape[0]
        log_psi = torch.empty((M, 2), dtype=torch.float32, device=self.device)
        eloc = torch.empty((M, 2), dtype=torch.float64, device=self.device)
        sums = torch.empty(4, dtype=torch.float64, device=self.device)
        w = weights.to(device=self.device, dtype=torch.float64).contiguous()
        st = self._lib.naqs_net_train_forward_eloc(self._h, ham._h, M, keys.data_ptr(), w.data_ptr(), log_psi.data_ptr(),
                                                   eloc.data_ptr(), sums.data_ptr(), _stream_ptr(self.device))
        _lib.check(st, "naqs_net_train_forward_eloc")
        return log_psi, (keys, None, None), eloc, sums

    def vmc_loss_grad(self, e_loc, weights, sums, with_energy=False):
        """g [M, 2] float32 = d loss / d (log|psi|, phase) of the VMC loss (``naqs_vmc_loss_grad``); with
        ``with_energy`` also the device pair (<E>, Var) of energy.py:372-375 from the same launch
        (``naqs_vmc_loss_grad_ev``) -> (g, ev float64 [2])."""
        M = e_loc.shape[0]
        g = torch.empty((M, 2), dtype=torch.float32, device=self.device)
        if with_energy:
            ev = torch.empty(2, dtype=torch.float64, device=self.device)
            st = self._lib.naqs_vmc_loss_grad_ev(M, e_loc.data_ptr(), weights.data_ptr(), sums.data_ptr(), g.data_ptr(),
                                                 ev.data_ptr(), _stream_ptr(self.device))
            _lib.check(st, "naqs_vmc_loss_grad_ev")
            return g, ev
        st = self._lib.naqs_vmc_loss_grad(M, e_loc.data_ptr(), weights.data_ptr(), sums.data_ptr(), g.data_ptr(),
                                          _stream_ptr(self.device))
        _lib.check(st, "naqs_vmc_loss_grad")
        return g

    @torch.no_grad()
    def backward_from_local_energy(self, saved, e_loc, weights, sums):
        """``vmc_loss_grad(..., with_energy=True)`` + ``backward_saved`` as ONE library call
        (``naqs_net_train_backward_vmc``): the loss gradient, its amplitude column and the output layer's delta come out of
        one launch instead of three.  -> (g [M, 2] f32, ev f64 [2]); the parameters' ``.grad`` are set like
        ``backward_saved`` does.  Needs the token of ``forward_saved*`` in HIP mode and parameters without gradients."""
        keys = saved[0]
        M = keys.shape[0]
        params = self.wf.param_list()
        if saved[1] is not None or not all(p.grad is None for p in params):
            g, ev = self.vmc_loss_grad(e_loc, weights, sums, with_energy=True)
            self.backward_saved(saved, g)
            return g, ev
        if self._grad_flat is None:
            self._grad_flat = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
            self._grad_views, off = [], 0
            for p in params:
                n = p.numel()
                self._grad_views.append(self._grad_flat[off:off + n].view(p.shape))
                off += n
        g = torch.empty((M, 2), dtype=torch.float32, device=self.device)
        ev = torch.empty(2, dtype=torch.float64, device=self.device)
        st = self._lib.naqs_net_train_backward_vmc(self._h, M, keys.data_ptr(), e_loc.data_ptr(), weights.data_ptr(), sums.data_ptr(),
                                                   g.data_ptr(), ev.data_ptr(), self._grad_flat.data_ptr(), _stream_ptr(self.device))
        _lib.check(st, "naqs_net_train_backward_vmc")
        for p, gv in zip(params, self._grad_views):
            p.grad = gv
        return g, ev

    @torch.no_grad()
    def backward_saved(self, saved, g):
        """Accumulate d/d theta sum_i (g[i, 0] log|psi_i| + g[i, 1] phase_i) into the ``.grad`` of every network
        parameter: ``naqs_net_amp_backward`` for the amplitude blocks, the phase MLP's chain rule as plain GEMMs
        (what autograd would run for three Linear layers, without building or walking a graph)."""
        keys, acts, occ = saved
        m = self.wf.model
        g = g.to(torch.float32)
        if acts is None:                     # naqs_net_train_backward: every gradient in one flat buffer
            g = g.contiguous()
            params = self.wf.param_list()
            fresh = all(p.grad is None for p in params)
            if fresh:
                # the usual case (zero_grad before every step): the kernels write straight into a persistent flat
                # buffer whose per-parameter views were made once
                if self._grad_flat is None:
                    self._grad_flat = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
                    self._grad_views, off = [], 0
                    for p in params:
                        n = p.numel()
                        self._grad_views.append(self._grad_flat[off:off + n].view(p.shape))
                        off += n
                flat = self._grad_flat
            else:
                flat = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
            st = self._lib.naqs_net_train_backward(self._h, keys.shape[0], keys.data_ptr(), g.data_ptr(), flat.data_ptr(),
                                                   _stream_ptr(self.device))
            _lib.check(st, "naqs_net_train_backward")
            if fresh:
                for p, gv in zip(params, self._grad_views):
                    p.grad = gv
            else:
                off = 0
                for p in params:
                    n = p.numel()
                    _accumulate(p, flat[off:off + n].view(p.shape))
                    off += n
            return
        flat = torch.empty(self.n_amp_params, dtype=torch.float32, device=self.device)
        ga = g[:, 0].contiguous()
        st = self._lib.naqs_net_amp_backward(self._h, keys.shape[0], keys.data_ptr(), ga.data_ptr(), flat.data_ptr(),
                                              _stream_ptr(self.device))
        _lib.check(st, "naqs_net_amp_backward")
        off = 0
        for p in self._amp_params:
            n = p.numel()
            _accumulate(p, flat[off:off + n].view(p.shape))
            off += n
        lin = m.phase_layers[0].linears()
        delta = torch.zeros((keys.shape[0], lin[-1].out_features), dtype=torch.float32, device=self.device)
        delta.scatter_(1, occ, g[:, 1:2])
        for l in range(len(lin) - 1, -1, -1):
            _accumulate(lin[l].weight, delta.t() @ acts[l])
            _accumulate(lin[l].bias, delta.sum(0))
            if l > 0:
                delta = (delta @ lin[l].weight).mul_(acts[l] > 0)

    def log_psi(self, keys, out=None):
        """keys: int64 device tensor [M] (uint64 bit patterns, qubit order) -> float32 [M, 2]."""
        M = keys.shape[0]
        if out is None:
            out = torch.empty((M, 2), dtype=torch.float32, device=self.device)
        st = self._lib.naqs_net_logpsi(self._h, M, keys.contiguous().data_ptr(), out.data_ptr(),
                                       _stream_ptr(self.device))
        _lib.check(st, "naqs_net_logpsi")
        return out

    def log_psi_and_local_energy(self, ham, keys, weights=None, log_psi_out=None, eloc_out=None, sums_out=None):
        """keys -> (log psi float32 [M, 2], E_loc float64 [M, 2][, sums float64 [4]]) in one library call
        (``naqs_logpsi_eloc``): the E_loc stage is fed by the log-psi kernels, no preparation kernel."""
        M = keys.shape[0]
        if log_psi_out is None:
            log_psi_out = torch.empty((M, 2), dtype=torch.float32, device=self.device)
        if eloc_out is None:
            eloc_out = torch.empty((M, 2), dtype=torch.float64, device=self.device)
        w_ptr = s_ptr = None
        if weights is not None:
            weights = weights.to(device=self.device, dtype=torch.float64).contiguous()
            if sums_out is None:
                sums_out = torch.empty(4, dtype=torch.float64, device=self.device)
            w_ptr, s_ptr = weights.data_ptr(), sums_out.data_ptr()
        st = self._lib.naqs_logpsi_eloc(self._h, ham._h, M, keys.contiguous().data_ptr(), w_ptr,
                                        log_psi_out.data_ptr(), eloc_out.data_ptr(), s_ptr, _stream_ptr(self.device))
        _lib.check(st, "naqs_logpsi_eloc")
        return (log_psi_out, eloc_out, sums_out) if weights is not None else (log_psi_out, eloc_out)

    def sample(self, n_samples, seed, max_unique, with_weights=False):
        """Draw ``n_samples`` from |psi|^2 on the device (``naqs_net_sample``): -> (keys int64 [M] in qubit order,
        counts int64 [M], probs float32 [M]), M unique bit-strings in (prefix, outcome) order; ``with_weights`` appends
        the float64 weights counts / sum(counts) written by the sampler's last launch (``naqs_net_sample_weighted``).
        Raises ``MaxBatchSizeExceededError`` when more than ``max_unique`` prefixes are alive at some level
        (nade.py:710-712).  One host synchronisation (to learn M); the results are views of buffers allocated for
        this call (copies only when they would pin a much larger buffer)."""
        from .nade import MaxBatchSizeExceededError
        cap = int(max_unique)
        keys = torch.empty(cap, dtype=torch.int64, device=self.device)
        counts = torch.empty(cap, dtype=torch.int64, device=self.device)
        probs = torch.empty(cap, dtype=torch.float32, device=self.device)
        if self._samp is None:
            self._samp = torch.empty(2, dtype=torch.int64, device=self.device)
        info = self._samp
        if with_weights:
            weights = torch.empty(cap, dtype=torch.float64, device=self.device)
            st = self._lib.naqs_net_sample_weighted(self._h, int(n_samples), int(seed) & (2 ** 64 - 1), cap, keys.data_ptr(),
                                                    counts.data_ptr(), probs.data_ptr(), weights.data_ptr(), info.data_ptr(),
                                                    _stream_ptr(self.device))
        else:
            st = self._lib.naqs_net_sample(self._h, int(n_samples), int(seed) & (2 ** 64 - 1), cap, keys.data_ptr(),
                                           counts.data_ptr(), probs.data_ptr(), info.data_ptr(), _stream_ptr(self.device))
        _lib.check(st, "naqs_net_sample")
        m, overflow = info.tolist()
        if overflow:
            # (a look-back wait that ran out also ends the call through the overflow flag: that is an error, not a big tree)
            _lib.check(self._lib.naqs_net_check(self._h), "naqs_net_sample")
            raise MaxBatchSizeExceededError
        out = (keys[:m], counts[:m], probs[:m]) + ((weights[:m],) if with_weights else ())
        if 4 * m < cap and cap > (1 << 16):
            # a view keeps its whole cap-sized buffer alive (28 B x cap: 117 MB at the evaluation path's cap of 2^22) for as
            # long as the caller holds the result; the training path (cap = n_unq_samples_max) keeps its views
            out = tuple(t.clone() for t in out)
        return out

    @torch.no_grad()
    def sample_forward_local_energy(self, ham, n_samples, seed, max_unique):
        """``sample(with_weights=True)`` and ``forward_saved_with_local_energy`` of the sampled table in ONE library call
        (``naqs_vmc_sample_forward_eloc``): the host synchronisation that learns M happens inside the library, directly
        followed by the forward / E_loc launches, instead of a return to the interpreter in between (the GPU idles for
        that long).  -> (keys, counts, probs, weights, (log psi, saved token, E_loc, sums))."""
        from .nade import MaxBatchSizeExceededError
        cap = int(max_unique)
        dev = self.device
        keys = torch.empty(cap, dtype=torch.int64, device=dev)
        counts = torch.empty(cap, dtype=torch.int64, device=dev)
        probs = torch.empty(cap, dtype=torch.float32, device=dev)
        weights = torch.empty(cap, dtype=torch.float64, device=dev)
        log_psi = torch.empty((cap, 2), dtype=torch.float32, device=dev)
        eloc = torch.empty((cap, 2), dtype=torch.float64, device=dev)
        sums = torch.empty(4, dtype=torch.float64, device=dev)
        if self._samp is None:
            self._samp = torch.empty(2, dtype=torch.int64, device=dev)
        info_host = (ctypes.c_int64 * 2)(0, 0)
        st = self._lib.naqs_vmc_sample_forward_eloc(self._h, ham._h, int(n_samples), int(seed) & (2 ** 64 - 1), cap, keys.data_ptr(),
                                                    counts.data_ptr(), probs.data_ptr(), weights.data_ptr(), log_psi.data_ptr(),
                                                    eloc.data_ptr(), sums.data_ptr(), self._samp.data_ptr(), info_host,
                                                    _stream_ptr(dev))
        _lib.check(st, "naqs_vmc_sample_forward_eloc")
        m, overflow = int(info_host[0]), int(info_host[1])
        if overflow:
            raise MaxBatchSizeExceededError
        k = keys[:m]
        return k, counts[:m], probs[:m], weights[:m], (log_psi[:m], (k, None, None), eloc[:m], sums)

    @torch.no_grad()
    def vmc_step(self, ham, n_samples, seed, max_unique, m_lo, m_hi, adam=None, keys_out=None):
        """One whole VMC training step as ONE library call (``naqs_vmc_step``): sampling, the host's look at (M, overflow),
        forward + E_loc, loss gradient + backward, Adam on the flat parameter vector and the re-pack of the kernels' weight
        layouts — the interpreter is not on the GPU's critical path between the first and the last launch of the step.
        ``adam``: a ``FlatAdam`` whose flat vector is this network's (its step counter is advanced here), or None to stop
        after the backward pass (gradients in ``self._grad_flat``).  The step is abandoned after sampling when the tree
        overflowed or M is outside [m_lo, m_hi].  -> (taken, M, overflow, (keys, counts, probs, weights, log psi, E_loc,
        sums, g, ev)) — the tensors are views of length M (None when not taken)."""
        cap = int(max_unique)
        dev = self.device
        # keys_out: where the sampler writes the keys (int64 [>= cap], contiguous; e.g. the caller's tracking buffer)
        keys = keys_out if keys_out is not None else torch.empty(cap, dtype=torch.int64, device=dev)
        if keys.dtype != torch.int64 or keys.numel() < cap or not keys.is_contiguous():
            raise ValueError("vmc_step: keys_out must be a contiguous int64 tensor of at least max_unique elements")
        # the step's cap-sized outputs live in per-handle buffers (allocated once per cap: eight allocator calls per step were
        # ~15 us of host time on a 0.2 ms step); they are valid until the next step of this handle — what the optimiser keeps
        # across steps are the two device scalars (<E>, Var), which therefore get fresh storage every step
        ob = getattr(self, "_onecall_bufs", None)
        if ob is None or ob[0] != cap:
            ob = self._onecall_bufs = (cap, torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.float32, device=dev),
                                       torch.empty(cap, dtype=torch.float64, device=dev), torch.empty((cap, 2), dtype=torch.float32, device=dev),
                                       torch.empty((cap, 2), dtype=torch.float64, device=dev), torch.empty((cap, 2), dtype=torch.float32, device=dev))
        _, counts, probs, weights, log_psi, eloc, g = ob
        small = torch.empty(6, dtype=torch.float64, device=dev)
        sums, ev = small[:4], small[4:]
        if self._grad_flat is None:
            self._grad_flat = torch.empty(self.n_params, dtype=torch.float32, device=dev)
            self._grad_views, off = [], 0
            for p in self.wf.param_list():
                n = p.numel()
                self._grad_views.append(self._grad_flat[off:off + n].view(p.shape))
                off += n
        info = (ctypes.c_int64 * 3)(0, 0, 0)
        if adam is not None:
            grp = next(g_ for g_ in adam.param_groups if g_['params'])
            flat, m1, m2, t = adam._flat, adam._m, adam._v, adam._t + 1
            if flat.numel() != self.n_params:
                raise ValueError("vmc_step: the optimiser's flat vector is not this network's")
            hyper = (float(grp['lr']), float(grp['betas'][0]), float(grp['betas'][1]), float(grp['eps']), float(grp['weight_decay']))
            ptrs = (flat.data_ptr(), m1.data_ptr(), m2.data_ptr())
        else:
            hyper, ptrs, t = (0.0, 0.0, 0.0, 0.0, 0.0), (None, None, None), 0
        st = self._lib.naqs_vmc_step(self._h, ham._h, int(n_samples), int(seed) & (2 ** 64 - 1), cap, int(m_lo), int(m_hi),
                                     keys.data_ptr(), counts.data_ptr(), probs.data_ptr(), weights.data_ptr(), log_psi.data_ptr(),
                                     eloc.data_ptr(), sums.data_ptr(), g.data_ptr(), ev.data_ptr(), self._grad_flat.data_ptr(),
                                     ptrs[0], ptrs[1], ptrs[2], *hyper, t, info, _stream_ptr(dev))
        _lib.check(st, "naqs_vmc_step")
        m, overflow, taken = int(info[0]), bool(info[1]), bool(info[2])
        if not taken:
            return False, m, overflow, None
        if adam is not None:
            if not adam.state:
                adam._bind_state()
            adam._t = t
            adam._opt_called = True          # (what torch's LR schedulers look at to see that a step preceded theirs)
            self._flat = adam._flat
        return True, m, False, (keys[:m], counts[:m], probs[:m], weights[:m], log_psi[:m], eloc[:m], sums, g[:m], ev)

    @torch.no_grad()
    def vmc_run(self, ham, n_steps, adam, n_samples, n_samples_max, n_unq_min, n_unq_max, seed_base, sample_calls, ring=None, ring_off=0):
        """``n_steps`` training steps in ONE library call (``naqs_vmc_run``): the loop of ``PartialSamplingOptimizer.run``
        with get_samples' adaptive sample count in C.  ``ring``: the optimiser's tracking buffer (int64; the keys of step i go to
        ``ring[off:off + M]``) or None.  -> dict(steps, stop_reason, error [the NaqsError of a failed call, for the caller to raise
        once it has booked the ``steps`` that did finish; None otherwise], events [(step, n_unique, overflow, action, n_samples)],
        n_samples, sample_calls, ring_off, M [steps], ns [steps], t [steps], ev [steps, 2], sums [steps, 4], and the last step's
        table views keys / counts / probs / weights / log_psi / eloc / g)."""
        dev, cap = self.device, int(n_unq_max)
        ob = getattr(self, "_onecall_bufs", None)
        if ob is None or ob[0] != cap:
            ob = self._onecall_bufs = (cap, torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.float32, device=dev),
                                       torch.empty(cap, dtype=torch.float64, device=dev), torch.empty((cap, 2), dtype=torch.float32, device=dev),
                                       torch.empty((cap, 2), dtype=torch.float64, device=dev), torch.empty((cap, 2), dtype=torch.float32, device=dev))
        _, counts, probs, weights, log_psi, eloc, g = ob
        if ring is None:
            kb = getattr(self, "_run_keys", None)
            if kb is None or kb.numel() < cap:
                kb = self._run_keys = torch.empty(cap, dtype=torch.int64, device=dev)
            keys_buf, ring_elems = kb, 0
        else:
            if ring.dtype != torch.int64 or not ring.is_contiguous() or ring.numel() < cap:
                raise ValueError("vmc_run: the tracking buffer must be a contiguous int64 tensor of at least max_unique elements")
            keys_buf, ring_elems = ring, ring.numel()
        if self._grad_flat is None:
            self._grad_flat = torch.empty(self.n_params, dtype=torch.float32, device=dev)
            self._grad_views, off = [], 0
            for p in self.wf.param_list():
                n = p.numel()
                self._grad_views.append(self._grad_flat[off:off + n].view(p.shape))
                off += n
        n = int(n_steps)
        ev = torch.empty((max(n, 1), 2), dtype=torch.float64, device=dev)
        sums = torch.empty((max(n, 1), 4), dtype=torch.float64, device=dev)
        m_log, ns_log, t_log = (ctypes.c_int64 * max(n, 1))(), (ctypes.c_int64 * max(n, 1))(), (ctypes.c_double * max(n, 1))()
        ev_cap = 64 + 4 * n
        events = (_lib.VmcEvent * ev_cap)()
        grp = next(g_ for g_ in adam.param_groups if g_['params'])
        if adam._flat.numel() != self.n_params:
            raise ValueError("vmc_run: the optimiser's flat vector is not this network's")
        a = _lib.VmcRunArgs()
        a.n_samples, a.n_samples_max = int(n_samples), int(n_samples_max)
        a.n_unq_samples_min, a.n_unq_samples_max = int(n_unq_min), cap
        a.seed_base, a.sample_calls = int(seed_base) & (2 ** 64 - 1), int(sample_calls)
        a.param_dev, a.exp_avg_dev, a.exp_avg_sq_dev = adam._flat.data_ptr(), adam._m.data_ptr(), adam._v.data_ptr()
        a.grad_dev = self._grad_flat.data_ptr()
        a.lr, a.beta1, a.beta2 = float(grp['lr']), float(grp['betas'][0]), float(grp['betas'][1])
        a.eps, a.weight_decay, a.adam_step = float(grp['eps']), float(grp['weight_decay']), int(adam._t)
        a.keys_dev, a.ring_elems, a.ring_off = keys_buf.data_ptr(), int(ring_elems), int(ring_off)
        a.counts_dev, a.probs_dev, a.weights_dev = counts.data_ptr(), probs.data_ptr(), weights.data_ptr()
        a.logpsi_dev, a.eloc_dev, a.g_dev = log_psi.data_ptr(), eloc.data_ptr(), g.data_ptr()
        a.ev_log_dev, a.sums_log_dev = ev.data_ptr(), sums.data_ptr()
        a.m_log_host, a.ns_log_host, a.t_log_host = m_log, ns_log, t_log
        a.events, a.events_cap = events, ev_cap
        st = self._lib.naqs_vmc_run(self._h, ham._h, n, ctypes.byref(a), _stream_ptr(dev))
        # A failure part-way through (a bounded device wait expiring, a HIP error at step k) leaves k finished steps behind:
        # k Adam updates are applied and the counters in `a` have advanced.  They are copied back below like a good run's, and
        # the error travels in the result ("error") so that the caller books the finished steps BEFORE it raises.
        error = None
        try:
            _lib.check(st, "naqs_vmc_run")
        except _lib.NaqsError as exc:
            error = exc
        done = int(a.steps_done)
        if done:
            if not adam.state:
                adam._bind_state()
            adam._t = int(a.adam_step)
            adam._opt_called = True
            self._flat = adam._flat
        m_last = int(m_log[done - 1]) if done else 0
        k0 = int(a.last_keys_off)
        return dict(steps=done, stop_reason=int(a.stop_reason), error=error,
                    events=[(int(e.step), int(e.n_unique), bool(e.overflow), int(e.action), int(e.n_samples)) for e in events[:int(a.n_events)]],
                    n_samples=int(a.n_samples), sample_calls=int(a.sample_calls), ring_off=int(a.ring_off),
                    M=[int(x) for x in m_log[:done]], ns=[int(x) for x in ns_log[:done]], t=[float(x) for x in t_log[:done]],
                    ev=ev, sums=sums, keys=keys_buf[k0:k0 + m_last], counts=counts[:m_last], probs=probs[:m_last],
                    weights=weights[:m_last], log_psi=log_psi[:m_last], eloc=eloc[:m_last], g=g[:m_last])

    # ---- the row-sharded step (world > 1): four library calls, the three collectives between them (include/naqs_hip.h) ----
    def _step_buffers(self, cap, world, keys_out=None):
        """Per-handle buffers of the sharded step, allocated once per (cap, world): the step's outputs are views of them and
        stay valid until the next step (the one-call step of a single process allocates per step; here the gather and
        reduce buffers have to be persistent anyway)."""
        key = (int(cap), int(world))
        sb = getattr(self, "_shard_bufs", None)
        if sb is None or sb["key"] != key:
            dev = self.device
            S_pad = -(-int(cap) // int(world))
            sb = self._shard_bufs = dict(
                key=key, S_pad=S_pad,
                keys=torch.empty(cap, dtype=torch.int64, device=dev), counts=torch.empty(cap, dtype=torch.int64, device=dev),
                probs=torch.empty(cap, dtype=torch.float32, device=dev), weights=torch.empty(cap, dtype=torch.float64, device=dev),
                mine=torch.zeros((S_pad, 2), dtype=torch.float32, device=dev),
                table=torch.empty((S_pad * int(world), 2), dtype=torch.float32, device=dev),
                eloc=torch.empty((S_pad, 2), dtype=torch.float64, device=dev), g=torch.empty((S_pad, 2), dtype=torch.float32, device=dev),
                ext=torch.empty(8, dtype=torch.float64, device=dev), ev=torch.empty(2, dtype=torch.float64, device=dev))
        if self._grad_flat is None:
            self._grad_flat = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
            self._grad_views, off = [], 0
            for p in self.wf.param_list():
                n = p.numel()
                self._grad_views.append(self._grad_flat[off:off + n].view(p.shape))
                off += n
        return sb

    @torch.no_grad()
    def shard_sample_forward(self, n_samples, seed, max_unique, m_lo, m_hi, rank, world, keys_out=None):
        """Call 1 (``naqs_vmc_shard_sample_forward``): the sampler, the host's look at (M, overflow), and the training forward
        of this rank's rows, whose (log|psi|, phase) land at the front of the gather contribution.
        -> (taken, M, overflow, buffers)."""
        cap = int(max_unique)
        sb = self._step_buffers(cap, world)
        keys = keys_out if keys_out is not None else sb["keys"]
        if keys.dtype != torch.int64 or keys.numel() < cap or not keys.is_contiguous():
            raise ValueError("shard_sample_forward: keys_out must be a contiguous int64 tensor of at least max_unique elements")
        info = (ctypes.c_int64 * 3)(0, 0, 0)
        st = self._lib.naqs_vmc_shard_sample_forward(self._h, int(n_samples), int(seed) & (2 ** 64 - 1), cap, int(m_lo), int(m_hi),
                                                     int(rank), int(world), keys.data_ptr(), sb["counts"].data_ptr(),
                                                     sb["probs"].data_ptr(), sb["weights"].data_ptr(), sb["mine"].data_ptr(), info,
                                                     _stream_ptr(self.device))
        _lib.check(st, "naqs_vmc_shard_sample_forward")
        sb["keys_used"] = keys
        return bool(info[2]), int(info[0]), bool(info[1]), sb

    @torch.no_grad()
    def shard_eloc(self, ham, sb, M, rank, world):
        """Call 2 (``naqs_eloc_gathered``) on the all-gathered table: E_loc of my rows, weighted sums and the same-table proof
        -> ext8 (to be all-reduced).  -> (b, e): my row range."""
        S = -(-M // world)
        b = min(M, rank * S)
        e = min(M, b + S)
        st = self._lib.naqs_eloc_gathered(ham._h, M, sb["keys_used"].data_ptr(), sb["table"].data_ptr(), S, sb["S_pad"], b, e - b,
                                          sb["weights"][b:].data_ptr() if e > b else None, sb["eloc"].data_ptr(), sb["ext"].data_ptr(),
                                          _stream_ptr(self.device))
        _lib.check(st, "naqs_eloc_gathered")
        return b, e

    @torch.no_grad()
    def shard_backward(self, sb, b, e):
        """Call 3 (``naqs_net_train_backward_vmc``) for my rows with the all-reduced sums -> flat gradient of my shard."""
        st = self._lib.naqs_net_train_backward_vmc(self._h, e - b, sb["keys_used"][b:].data_ptr() if e > b else None,
                                                   sb["eloc"].data_ptr(), sb["weights"][b:].data_ptr() if e > b else None,
                                                   sb["ext"].data_ptr(), sb["g"].data_ptr(), sb["ev"].data_ptr(),
                                                   self._grad_flat.data_ptr(), _stream_ptr(self.device))
        _lib.check(st, "naqs_net_train_backward_vmc")

    @torch.no_grad()
    def shard_update(self, adam):
        """Call 4 (``naqs_vmc_shard_update``): Adam on the flat parameter vector with the all-reduced gradient + re-pack."""
        grp = next(g_ for g_ in adam.param_groups if g_['params'])
        t = adam._t + 1
        st = self._lib.naqs_vmc_shard_update(self._h, self._grad_flat.data_ptr(), adam._flat.data_ptr(), adam._m.data_ptr(),
                                             adam._v.data_ptr(), float(grp['lr']), float(grp['betas'][0]), float(grp['betas'][1]),
                                             float(grp['eps']), float(grp['weight_decay']), t, _stream_ptr(self.device))
        _lib.check(st, "naqs_vmc_shard_update")
        if not adam.state:
            adam._bind_state()
        adam._t = t
        adam._opt_called = True
        self._flat = adam._flat

    def prof_enable(self, n, stride=1):
        _lib.check(self._lib.naqs_net_prof_enable(self._h, int(n)), "naqs_net_prof_enable")
        _lib.check(self._lib.naqs_net_prof_stride(self._h, int(stride)), "naqs_net_prof_stride")

    def prof_read(self):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        _lib.check(self._lib.naqs_net_prof_read(self._h, ctypes.byref(ms), ctypes.byref(n)), "naqs_net_prof_read")
        return ms.value, n.value

    def last_kernel(self):
        """Name (with template arguments) of the kernel the most recent call launched (measurement aid)."""
        buf = ctypes.create_string_buffer(128)
        _lib.check(self._lib.naqs_net_last_kernel(self._h, buf, 128), "naqs_net_last_kernel")
        return buf.value.decode()

    def share_device(self, on=None):
        """Two runs per GPU (the farm's `--per-gpu 2`): this handle's sampler calls take turns with those of the device's other
        sharing handles (`naqs_net_share_device`; default NAQS_SHARED_GPU when the handle was created).  `on=None` only reads.
        -> sampler calls so far that had to wait for another handle's turn to end."""
        turns = ctypes.c_int64(0)
        _lib.check(self._lib.naqs_net_share_device(self._h, -1 if on is None else int(bool(on)), ctypes.byref(turns)),
                   "naqs_net_share_device")
        return turns.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.naqs_net_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
