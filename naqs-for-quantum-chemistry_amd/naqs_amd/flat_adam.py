"""Adam on a flat parameter vector: one HIP launch per step (``naqs_adam_step``).

``torch.optim.Adam`` walks the parameter list (26 tensors for the published network); even its fused
multi-tensor variant costs ~0.2 ms of host time and two ~40 us kernels per step, which is a sixth of the
whole VMC step once everything else is a handful of launches.  Here the network's parameters are views into
one flat buffer (``NAQSComplex_NADE_orbitals.flatten_parameters``) and so are the moment estimates, so the
update is one elementwise kernel.  The update rule, the ``param_groups`` keys and the ``state_dict`` layout
(per parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``) are ``torch.optim.Adam``'s, so checkpoints written
by either optimiser load into the other (reference: experiments/_base.py:228, energy.py:409-538)."""
import torch

from . import _lib
from .hamiltonian import _stream_ptr


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **unused):
        if amsgrad:
            raise NotImplementedError("FlatAdam: amsgrad")
        # the same param_group keys as this torch version's Adam, so state_dicts are interchangeable
        defaults = dict(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]).defaults)
        defaults.update(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False)
        super().__init__(params, defaults)
        self._lib = _lib.load_library()
        self._flat = flat_param
        self._m = torch.zeros_like(flat_param)
        self._v = torch.zeros_like(flat_param)
        self._t = 0
        base, size = flat_param.data_ptr(), flat_param.element_size()
        self._span = {}
        for group in self.param_groups:
            for p in group['params']:
                off = (p.data_ptr() - base) // size
                if not (0 <= off and off + p.numel() <= flat_param.numel() and p.is_contiguous()):
                    raise ValueError("FlatAdam: every parameter must be a contiguous view of flat_param")
                self._span[p] = (off, p.numel())
        live = [g for g in self.param_groups if g['params']]
        if len(live) != 1 or sum(n for _, n in self._span.values()) != flat_param.numel():
            raise ValueError("FlatAdam: one non-empty parameter group covering flat_param exactly")
        self._order = sorted(self._span, key=lambda p: self._span[p][0])

    def _bind_state(self):
        for p, (off, n) in self._span.items():
            # one `step` tensor per parameter like torch.optim.Adam (it increments each of them separately);
            # they are brought up to date when the state is read (state_dict), not on every step
            self.state[p] = {'step': torch.tensor(float(self._t)), 'exp_avg': self._m[off:off + n].view(p.shape),
                             'exp_avg_sq': self._v[off:off + n].view(p.shape)}

    def _sync_steps(self):
        for st in self.state.values():
            st['step'].fill_(float(self._t))

    def state_dict(self):
        if not self.state:
            self._bind_state()
        self._sync_steps()
        return super().state_dict()

    def _flat_grad(self):
        """The gradients as one vector: zero-copy when they already are consecutive views of one buffer
        (``FusedLogPsi.backward_saved`` in HIP mode), otherwise gathered."""
        first = self._order[0].grad
        if first is not None:
            base, size, ok = first.data_ptr(), first.element_size(), True
            for p in self._order:
                g = p.grad
                if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != base + self._span[p][0] * size:
                    ok = False
                    break
            if ok and self._span[self._order[0]][0] == 0:
                owner = first._base if first._base is not None else first
                if owner.numel() >= self._flat.numel() and owner.data_ptr() == base:
                    return owner.reshape(-1)[:self._flat.numel()]
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32)
                          for p in self._order])

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if not self.state:
            self._bind_state()
        g = self._flat_grad()
        # looked up per step: load_state_dict() replaces the param_group dicts, and callers / LR schedulers edit
        # ``param_groups[i]['lr']`` of whatever dict is current (experiments/_base.py: the 1e-3 -> 5e-4 schedule)
        grp = next(g_ for g_ in self.param_groups if g_['params'])
        self._t += 1
        st = self._lib.naqs_adam_step(self._flat.numel(), self._flat.data_ptr(), g.data_ptr(), self._m.data_ptr(),
                                      self._v.data_ptr(), float(grp['lr']), float(grp['betas'][0]), float(grp['betas'][1]),
                                      float(grp['eps']), float(grp['weight_decay']), self._t, _stream_ptr(self._flat.device))
        _lib.check(st, "naqs_adam_step")
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        t = 0
        for p, (off, n) in self._span.items():
            st = self.state.get(p)
            if st:
                self._m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                self._v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                t = max(t, int(float(st['step'])))
        self._t = t
        self._bind_state()
