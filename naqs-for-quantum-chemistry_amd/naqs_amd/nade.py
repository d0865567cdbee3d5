"""Orbital-pair NADE amplitude/phase network and its unique-prefix tree sampler, fully on device.

Own implementation of the semantics of the reference's
``ComplexAutoregressiveMachine1D_OrbitalNade`` (src/naqs/network/nade.py:157-777) for the
configurations the run path uses (experiments/_base.py:150-187, batch_train.sh:14):
MLP blocks (no look-up-table blocks, no batch-norm), binary input encoding, separate amplitude
and phase blocks, optional amplitude spin-exchange symmetry, no phase symmetry,
``SoftmaxLogProbAmps`` amplitudes, raw phases.

What is different from the reference by design (MI355X-first):
  * teacher-forced evaluation runs ALL N/2 conditional blocks as two batched GEMMs over a padded
    input tensor instead of a Python loop of N/2 small Linear stacks with a device->host copy per
    block (nade.py:738-770, :552-574);
  * the tree sampler keeps prefixes, counts and probabilities on the GPU; the multinomial split of
    the counts is a binomial chain drawn with ``torch.binomial`` on device instead of numpy on
    the host (nade.py:20-37, :692), and it runs without autograd — log psi of the unique samples
    (with gradients) is one batched teacher-forced pass afterwards, numerically the same
    quantity the reference accumulates along the tree (nade.py:714-723);
  * parameter names / shapes of ``state_dict()`` are identical to the reference's, so
    checkpoints interchange.
"""
import math
from enum import Enum

import torch
from torch import nn
from torch.nn import functional as F


class NadeMasking(Enum):           # src/naqs/network/base.py:20-23
    NONE = 0
    PARTIAL = 1
    FULL = 2


class InputEncoding(Enum):         # src/naqs/network/base.py:16-18
    BINARY = 0
    INTEGER = 1


class AmplitudeEncoding(Enum):     # src/naqs/network/base.py:10-14
    AMP = 0
    LOG_AMP = 1
    PROB = 2
    LOG_PROB = 3


class MaxBatchSizeExceededError(Exception):      # nade.py:39-40
    pass


class SoftmaxLogProbAmps(nn.Module):
    """0.5 * log_softmax(2x) with masked entries at -inf (src/naqs/network/activations.py:40-46)."""
    amplitude_encoding = AmplitudeEncoding.LOG_AMP

    def forward(self, x, mask=None, dim=-1):
        x = 2 * x
        if mask is not None:
            x = x.masked_fill(~mask.bool(), float("-inf"))
        return 0.5 * F.log_softmax(x, dim=dim)


class OrbitalBlock(nn.Module):
    """MLP of one conditional; same module tree as the reference so state_dict keys match
    (``layers.<l>.0.weight``; nade.py:72-115)."""

    def __init__(self, num_in=2, n_hid=(), num_out=4, hidden_activation=nn.ReLU, bias=True):
        super().__init__()
        dims = [num_in] + list(n_hid) + [num_out]
        layers = []
        for i, (n_in, n_out) in enumerate(zip(dims, dims[1:])):
            mods = [nn.Linear(n_in, n_out, bias=bias)]
            if hidden_activation is not None and i < len(dims) - 2:
                mods.append(hidden_activation())
            layers.append(nn.Sequential(*mods))
        self.layers = nn.Sequential(*layers)
        self.num_in, self.num_out = num_in, num_out

    def forward(self, x):
        return self.layers(x)

    def linears(self):
        return [seq[0] for seq in self.layers]


# (alpha, beta) occupation of the 4 outcomes of one orbital pair, in block order |00>,|10>,|01>,|11>
_BLOCK_ALPHA = (0, 1, 0, 1)
_BLOCK_BETA = (0, 0, 1, 1)
# symmetrisation gather, indexed by x_order (nade.py:585): 0 = inputs were swapped, 1 = equal, 2 = as given
_IDX2SORT = ((0, 3, 4, 2), (0, 1, 1, 2), (0, 4, 3, 2))


class OrbitalNADE(nn.Module):
    def __init__(self, num_qubits, n_alpha_electrons=None, n_beta_electrons=None,
                 masking=NadeMasking.PARTIAL, input_encoding=InputEncoding.BINARY, num_lut=0,
                 amp_hidden_size=(), amp_hidden_activation=nn.ReLU, amp_bias=True,
                 phase_hidden_size=(), phase_hidden_activation=nn.ReLU, phase_bias=True,
                 combined_amp_phase_blocks=False, use_amp_spin_sym=True, use_phase_spin_sym=False,
                 aggregate_phase=True, amp_batch_norm=False, phase_batch_norm=False, batch_norm_momentum=1,
                 amp_activation=SoftmaxLogProbAmps, phase_activation=None, n_electrons=None,
                 device=None, out_device=None):
        super().__init__()
        unsupported = []
        if num_lut:
            unsupported.append("num_lut > 0")
        if input_encoding is not InputEncoding.BINARY:
            unsupported.append("InputEncoding.INTEGER")
        if amp_batch_norm or phase_batch_norm:
            unsupported.append("batch norm")
        if amp_activation is not SoftmaxLogProbAmps or phase_activation is not None:
            unsupported.append("non-default output activations")
        if unsupported:
            raise NotImplementedError("not on the MI355X path (unused by the reference's run path): "
                                      + ", ".join(unsupported))
        if num_qubits % 2:
            raise ValueError("Symmetric NADE requires an even number of qubits.")
        self.N = self.num_qubits = int(num_qubits)
        self.P = self.N // 2
        self.masking = masking
        self.use_amp_spin_sym = bool(use_amp_spin_sym)
        self.use_phase_spin_sym = bool(use_phase_spin_sym)
        self.aggregate_phase = bool(aggregate_phase)
        # -comb_amp_phase (nade.py:257-262): one block per pair emits the amplitude AND the phase outputs, and the phase
        # symmetry setting follows the amplitude's
        self.combined_amp_phase_blocks = bool(combined_amp_phase_blocks)
        if self.combined_amp_phase_blocks:
            print("\tUsing combined amplitude and phase blocks:\n\t\t--> defaulting to amp network params for these blocks.")
            if self.use_amp_spin_sym != self.use_phase_spin_sym:
                print("\t\t--> Warning: must use same spin-sym settings for both amplitude and phase when combining them into a single block.")
                print(f"\t\t\t--> setting self.use_amp_spin_sym=self.use_phase_spin_sym={self.use_amp_spin_sym}")
                self.use_phase_spin_sym = self.use_amp_spin_sym
        self.amplitude_encoding = AmplitudeEncoding.LOG_AMP
        self.use_restricted_hilbert = n_alpha_electrons is not None and n_beta_electrons is not None
        if self.use_restricted_hilbert:
            self.n_alpha_up, self.n_beta_up = int(n_alpha_electrons), int(n_beta_electrons)
            self.n_alpha_down = math.ceil(self.N / 2) - self.n_alpha_up
            self.n_beta_down = self.N // 2 - self.n_beta_up
            self._min_n_set = min(self.n_alpha_up, self.n_beta_up, self.n_alpha_down, self.n_beta_down)
        else:
            self._min_n_set = 0
        self._n_out_amp = 5 if self.use_amp_spin_sym else 4
        # with the phase symmetry three outputs: |00>, |01> == |10>, |11> (nade.py:281, 593-595)
        self._n_out_phase = 3 if self.use_phase_spin_sym else 4

        amp, phase = [], []
        for n in range(self.P):
            n_in = max(1, 2 * n)
            with_phase = self.aggregate_phase or n == self.P - 1
            n_amp_out = self._n_out_amp + (self._n_out_phase if (with_phase and self.combined_amp_phase_blocks) else 0)
            amp.append(OrbitalBlock(n_in, amp_hidden_size, n_amp_out, amp_hidden_activation, amp_bias))
            if with_phase and not self.combined_amp_phase_blocks:
                phase.append(OrbitalBlock(n_in, phase_hidden_size, self._n_out_phase, phase_hidden_activation,
                                          phase_bias))
        self.amp_layers = nn.ModuleList(amp)
        self.phase_layers = nn.ModuleList(phase)
        self.amplitude_activation = SoftmaxLogProbAmps()
        self.phase_activation = None
        self.sampling = False
        self.register_buffer("_idx2sort", torch.tensor(_IDX2SORT, dtype=torch.long), persistent=False)
        self.register_buffer("_blk_alpha", torch.tensor(_BLOCK_ALPHA, dtype=torch.bool), persistent=False)
        self.register_buffer("_blk_beta", torch.tensor(_BLOCK_BETA, dtype=torch.bool), persistent=False)
        self.register_buffer("_blockidx2spin", torch.tensor([[-1., -1.], [1., -1.], [-1., 1.], [1., 1.]]),
                             persistent=False)
        if device is not None:
            self.to(device)

    # mode switches kept for API compatibility (nade.py:378-391)
    def sample(self, mode=True):
        self.sampling = mode

    def predict(self):
        self.sampling = False

    def clear_cache(self):
        pass

    @property
    def device(self):
        return next(self.parameters()).device

    # ------------------------------------------------------------------ shared pieces
    def _mask_from_counts(self, n, up_a, up_b):
        """Electron-budget mask of block n given the alpha/beta electrons already placed
        (nade.py:417-474): outcome (a, b) allowed iff alpha/beta can still take that value.
        up_a/up_b: integer tensors [...]; returns bool [..., 4] (all True when un-restricted or when
        n < max(min_n_set, 1), like the reference)."""
        shape = up_a.shape + (4,)
        if not self.use_restricted_hilbert or n < max(self._min_n_set, 1):
            return torch.ones(shape, dtype=torch.bool, device=up_a.device)
        a_up_ok = (up_a < self.n_alpha_up).unsqueeze(-1)
        a_dn_ok = ((n - up_a) < self.n_alpha_down).unsqueeze(-1)
        b_up_ok = (up_b < self.n_beta_up).unsqueeze(-1)
        b_dn_ok = ((n - up_b) < self.n_beta_down).unsqueeze(-1)
        ok_a = torch.where(self._blk_alpha, a_up_ok, a_dn_ok)
        ok_b = torch.where(self._blk_beta, b_up_ok, b_dn_ok)
        return ok_a & ok_b

    def _softmax_mask_active(self, n, masking):
        masking = self.masking if masking is None else masking
        return not (masking is NadeMasking.NONE or (masking is NadeMasking.PARTIAL and n == self.P - 1))

    def _symmetrise(self, amp, x_order):
        if self.use_amp_spin_sym:       # nade.py:576-594
            first = amp[..., [0, 1, 1, 2]]
            second = amp.gather(-1, self._idx2sort[x_order])
            return (first + second) / 2
        return amp[..., :4]

    # ------------------------------------------------------------------ teacher-forced evaluation
    def _stacked_first_layer(self, blocks, width):
        """Pad the first Linear of every block to a common [first half | second half] input layout."""
        Ws, bs = [], []
        for n, blk in enumerate(blocks):
            lin = blk.linears()[0]
            W = lin.weight
            if n == 0:
                # block 0 sees a constant-zero input (nade.py:509-511): its single weight column sits on
                # an always-zero input slot, so it gets a zero gradient exactly like in the reference
                Wp = F.pad(W[:, :1], (0, 2 * width - 1))
            else:
                Wp = torch.cat([F.pad(W[:, :n], (0, width - n)), F.pad(W[:, n:2 * n], (0, width - n))], dim=1)
            Ws.append(Wp)
            bs.append(lin.bias if lin.bias is not None else W.new_zeros(W.shape[0]))
        return torch.stack(Ws), torch.stack(bs)

    def _forward_predict(self, x, masking=None):
        """x: [B, N] occupations (+-1) in model order -> conditional log-amplitudes / phases
        [B, N/2, 4, 2] (nade.py:738-770)."""
        if x.dim() < 2:
            x = x.unsqueeze(0)
        x = x.to(self.device, torch.float32)
        B, P = x.shape[0], self.P
        a, b = x[:, 0::2], x[:, 1::2]                        # alpha / beta occupations per block
        bits_a, bits_b = (a > 0).long(), (b > 0).long()
        # exclusive prefix sums: electrons placed and spin-string index of the first n orbital pairs
        up_a = torch.cumsum(bits_a, 1) - bits_a
        up_b = torch.cumsum(bits_b, 1) - bits_b
        pw = (1 << torch.arange(P, device=x.device)).long()
        idx_a = torch.cumsum(bits_a * pw, 1) - bits_a * pw
        idx_b = torch.cumsum(bits_b * pw, 1) - bits_b * pw
        W = max(P - 1, 1)
        k = torch.arange(W, device=x.device)
        n_idx = torch.arange(P, device=x.device)
        visible = (k.unsqueeze(0) < n_idx.unsqueeze(1)).to(x.dtype)                  # [P, W]: input k visible to block n
        a_in = a[:, :W].unsqueeze(1) * visible                                        # [B, P, W]
        b_in = b[:, :W].unsqueeze(1) * visible
        if self.use_amp_spin_sym:
            swap = (idx_a > idx_b).unsqueeze(-1)                                      # nade.py:519-530
            first = torch.where(swap, b_in, a_in)
            second = torch.where(swap, a_in, b_in)
            x_order = torch.where(idx_a > idx_b, 0, torch.where(idx_a == idx_b, 1, 2))
        else:
            first, second, x_order = a_in, b_in, None
        h = torch.cat([first, second], -1)                                            # [B, P, 2W]

        ordered_in = h
        if self.use_phase_spin_sym and not self.use_amp_spin_sym:                     # the phase blocks' ordered inputs (nade.py:507-533)
            swap_p = (idx_a > idx_b).unsqueeze(-1)
            ordered_in = torch.cat([torch.where(swap_p, b_in, a_in), torch.where(swap_p, a_in, b_in)], -1)
        ragged = self.combined_amp_phase_blocks and not self.aggregate_phase          # the last block is wider than the others
        if ragged:
            outs = []
            for n, blk in enumerate(self.amp_layers):                                 # plain per-block evaluation
                xin = h[:, n, :1] * 0 if n == 0 else torch.cat([h[:, n, :n], h[:, n, W:W + n]], -1)
                outs.append(blk(xin))
            h = torch.stack([o[:, :self._n_out_amp] for o in outs], 1)
            comb_phase = torch.cat([x.new_zeros((B, P - 1, self._n_out_phase)), outs[-1][:, self._n_out_amp:].unsqueeze(1)], 1)
        else:
            W1, b1 = self._stacked_first_layer(self.amp_layers, W)
            h = torch.einsum("bnk,nhk->bnh", h, W1) + b1
            n_lin = len(self.amp_layers[0].linears())
            for l in range(1, n_lin):
                h = torch.relu(h)
                Wl = torch.stack([blk.linears()[l].weight for blk in self.amp_layers])
                bl = torch.stack([blk.linears()[l].bias for blk in self.amp_layers])
                h = torch.einsum("bnk,nhk->bnh", h, Wl) + bl
            comb_phase = h[..., self._n_out_amp:] if self.combined_amp_phase_blocks else None
            h = h[..., :self._n_out_amp]
        amp = self._symmetrise(h, x_order)                                            # [B, P, 4]

        masks = torch.stack([self._mask_from_counts(n, up_a[:, n], up_b[:, n]) if self._softmax_mask_active(n, masking)
                             else torch.ones((B, 4), dtype=torch.bool, device=x.device) for n in range(P)], 1)
        log_amp = self.amplitude_activation(amp, masks)

        if self.combined_amp_phase_blocks:
            phase = comb_phase
        elif self.aggregate_phase:
            ph_in = ordered_in if self.use_phase_spin_sym else torch.cat([a_in, b_in], -1)
            W1, b1 = self._stacked_first_layer(self.phase_layers, W)
            g = torch.einsum("bnk,nhk->bnh", ph_in, W1) + b1
            for l in range(1, len(self.phase_layers[0].linears())):
                g = torch.relu(g)
                Wl = torch.stack([blk.linears()[l].weight for blk in self.phase_layers])
                bl = torch.stack([blk.linears()[l].bias for blk in self.phase_layers])
                g = torch.einsum("bnk,nhk->bnh", g, Wl) + bl
            phase = g
        else:
            if P > 1:
                if self.use_phase_spin_sym:                                           # the last block's spin-ordered input
                    ph_in = torch.cat([ordered_in[:, P - 1, :P - 1], ordered_in[:, P - 1, W:W + P - 1]], -1)
                else:
                    ph_in = torch.cat([a[:, :P - 1], b[:, :P - 1]], -1)
            else:
                ph_in = x.new_zeros((B, 1))
            last = self.phase_layers[0](ph_in)                                        # nade.py:563-569
            phase = torch.cat([x.new_zeros((B, P - 1, self._n_out_phase)), last.unsqueeze(1)], 1)
        if self.use_phase_spin_sym:
            phase = phase[..., [0, 1, 1, 2]]                                          # nade.py:593-595
            # spin-exchanged partner configurations differ by a sign per |01> pair (nade.py:597-610), applied to the LAST
            # block's outputs (:758-759): + pi (N_01 mod 2) where idx(alpha string) < idx(beta string)
            tot_a, tot_b = (bits_a * pw).sum(1), (bits_b * pw).sum(1)
            n01 = ((a <= 0) & (b > 0)).sum(1)
            shift = torch.where(tot_a < tot_b, math.pi * (n01 % 2).to(phase.dtype), torch.zeros_like(phase[:, 0, 0]))
            phase = torch.cat([phase[:, :P - 1], phase[:, P - 1:] + shift.view(-1, 1, 1)], 1)
        return torch.stack([log_amp, phase], -1)

    # ------------------------------------------------------------------ one conditional (sampler)
    def _block_log_amp(self, n, a, b, masking=None):
        """Block n for U prefixes.  a, b: [U, n] (+-1) -> (log-amplitudes [U, 4], physical mask [U, 4])."""
        U = a.shape[0]
        if n == 0:
            x_in = a.new_zeros((U, 1))
            x_order = torch.ones(U, dtype=torch.long, device=a.device)
            up_a = up_b = torch.zeros(U, dtype=torch.long, device=a.device)
        else:
            bits_a, bits_b = (a > 0).long(), (b > 0).long()
            up_a, up_b = bits_a.sum(1), bits_b.sum(1)
            if self.use_amp_spin_sym:
                pw = (1 << torch.arange(n, device=a.device)).long()
                ia, ib = (bits_a * pw).sum(1), (bits_b * pw).sum(1)
                swap = (ia > ib).unsqueeze(-1)
                x_in = torch.cat([torch.where(swap, b, a), torch.where(swap, a, b)], -1)
                x_order = torch.where(ia > ib, 0, torch.where(ia == ib, 1, 2))
            else:
                x_in, x_order = torch.cat([a, b], -1), None
        amp = self._symmetrise(self.amp_layers[n](x_in)[..., :self._n_out_amp], x_order)
        phys = self._mask_from_counts(n, up_a, up_b)
        log_amp = self.amplitude_activation(amp, phys if self._softmax_mask_active(n, masking) else None)
        return log_amp, phys

    @torch.no_grad()
    def _forward_sample(self, batch_size, ret_output=True, masking=None, max_batch_size=None, generator=None):
        """Draw ``batch_size`` samples as a tree of unique prefixes with counts (nade.py:632-736).

        Returns [states float [U, N] (+-1, model order), counts int64 [U], probs float32 [U]];
        with ret_output the teacher-forced log psi [U, 2] is appended (no gradients here — call
        ``forward`` on the returned states for the differentiable value)."""
        dev = self.device
        states = torch.zeros((1, 0), device=dev)
        counts = torch.tensor([int(batch_size)], dtype=torch.float64, device=dev)
        probs = torch.ones(1, dtype=torch.float32, device=dev)
        for n in range(self.P):
            a, b = states[:, 0::2], states[:, 1::2]
            log_amp, phys = self._block_log_amp(n, a, b, masking)
            p = log_amp.exp().pow(2)                                       # float32, nade.py:673
            next_probs = probs.unsqueeze(1) * p
            p64 = p.double()
            p64 = p64 / p64.sum(-1, keepdim=True)                          # nade.py:682-683
            new_counts = _multinomial_counts(counts, p64, generator)       # [U, 4]
            new_counts = new_counts * phys.to(new_counts.dtype)            # throw away unphysical samples, :695
            keep = new_counts > 0
            parent, child = torch.nonzero(keep, as_tuple=True)             # row-major: parents in order, children 0..3
            states = torch.cat([states[parent], self._blockidx2spin[child]], 1)
            counts = new_counts[keep]
            probs = next_probs[keep]
            if max_batch_size is not None and states.shape[0] > max_batch_size:
                raise MaxBatchSizeExceededError
        ret = [states, counts.round().long(), probs]
        if ret_output:
            cond = self._forward_predict(states, masking)
            occ = ((states[:, 0::2] > 0).long() + 2 * (states[:, 1::2] > 0).long())
            sel = cond.gather(2, occ.view(-1, self.P, 1, 1).expand(-1, -1, 1, 2)).squeeze(2)
            ret.append(sel.sum(1))
        return ret

    def forward(self, x, *args, **kwargs):
        if self.sampling:
            return self._forward_sample(x, *args, **kwargs)
        return self._forward_predict(x, *args, **kwargs)


def _multinomial_counts(counts, p, generator=None):
    """Split counts[u] over 4 outcomes with probabilities p[u] (float64): the conditional-binomial
    chain of the reference's ``multinomial_arr`` (nade.py:20-37), drawn on device."""
    remaining = counts.clone()
    out = torch.zeros_like(p)
    cs = p.cumsum(-1)
    condp = torch.where(cs > 0, p / cs, torch.zeros_like(p)).clamp_(0.0, 1.0)
    for j in range(p.shape[-1] - 1, 0, -1):
        draw = torch.binomial(remaining, condp[..., j], generator=generator)
        out[..., j] = draw
        remaining = remaining - draw
    out[..., 0] = remaining
    return out
