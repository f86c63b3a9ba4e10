"""Adam over one flat arena (ref: SISR/models/__init__.py:299-308 `optim.Adam(params, lr, betas)`, stepped by
standard_update :481-489).

torch's Adam walks ~2 400 parameter tensors of a QRCAN with seven multi-tensor kernels chunked into ~190 launches and
~14 ms of host work per step.  Here parameters, gradients and both moment estimates live in four flat fp32 arenas laid
out alike; every nn.Parameter / .grad / state tensor is a view into them, so the update is ONE launch of
`sisr_adam_flat` (csrc/misc.hip, torch's operation order) and nothing is gathered:
  * conv weight gradients are written into the gradient arena by the weight-gradient kernels themselves
    (ops.GRAD_SINK), the small remaining gradients are copied in by one multi-tensor copy;
  * a data-parallel GradReducer all-reduces slices of the same arena (parallel.py), no bucket copies.
It IS a torch.optim.Adam: param_groups, state and state_dict() keep torch's schema ({'step', 'exp_avg',
'exp_avg_sq'} per parameter that has received a gradient), so checkpoints interchange with the reference, and LR
schedulers drive it through param_groups[0]['lr'].  Parameters that get no gradient in a step are skipped exactly as
torch skips them (their range is left out of the launch).  CUDA (HIP) parameters only; on CPU the handlers fall back to
constructing a plain torch Adam, which is never stepped (the networks refuse CPU tensors).
"""
import math

import torch
from torch import optim

from . import hip

ALIGN = 4  # floats: every parameter starts on a 16-byte boundary


class FlatAdam(optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, late=()):
        """late: parameters whose gradients only arrive at the very end of a backward pass (the meta-attention layers, whose
        gates are all computed -- and differentiated -- by one launch before the first block).  They are laid out BEHIND
        the others in the arenas, so that a data-parallel reducer's buckets (contiguous runs of the gradient arena) of conv
        weights complete, and can be all-reduced, while backward is still running.  The arena order is internal:
        param_groups, state and state_dict() keep the registration order."""
        params = list(params)
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        if len(self.param_groups) != 1:
            raise NotImplementedError("FlatAdam keeps one parameter group (the reference never uses more)")
        ps = self.param_groups[0]['params']
        if not ps or not all(p.is_cuda and p.dtype == torch.float32 for p in ps):
            raise RuntimeError("FlatAdam needs fp32 parameters on a HIP device")
        dev = ps[0].device
        self.offsets, off = {}, 0
        late_ids = {id(p) for p in late}
        self.arena_order = [p for p in ps if id(p) not in late_ids] + [p for p in ps if id(p) in late_ids]
        for p in self.arena_order:
            self.offsets[p] = off
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.flat_p = torch.zeros(off, device=dev)
        self.flat_g = torch.zeros(off, device=dev)
        self.flat_m = torch.zeros(off, device=dev)
        self.flat_v = torch.zeros(off, device=dev)
        with torch.no_grad():
            for p in ps:
                view = self._view(self.flat_p, p)
                view.copy_(p.data)
                p.data = view
        self.grad_views = {p: self._view(self.flat_g, p) for p in ps}
        from . import ops
        for p, view in self.grad_views.items():  # gradients the kernels can produce straight into the arena
            ops.GRAD_SINK[p.data_ptr()] = view
        self._t = 0          # steps taken
        self._missed = {}    # parameter -> number of those steps it had no gradient in (torch counts per parameter)
        self._plan = None    # (frozenset of skipped parameters, [(offset, length, step count)])

    def _view(self, flat, p):
        o = self.offsets[p]
        return flat[o:o + p.numel()].view_as(p)

    # -- state in torch's schema, created when a parameter first receives a gradient (as torch's _init_group does)
    def _ensure_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = torch.tensor(0.0, dtype=torch.float32)
            st['exp_avg'] = self._view(self.flat_m, p)
            st['exp_avg_sq'] = self._view(self.flat_v, p)
            self._missed[p] = self._t  # it has missed every step so far

    def _segments(self, skipped):
        skip = frozenset(id(p) for p in skipped)  # identity: `in` on tensors would compare element-wise
        key = (skip, tuple(self._missed.values()) if any(self._missed.values()) else ())
        if self._plan is not None and self._plan[0] == key:
            return self._plan[1]
        segs = []
        for p in self.arena_order:
            if id(p) in skip:
                continue
            o, n = self.offsets[p], (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            missed = self._missed.get(p, 0)
            if segs and segs[-1][0] + segs[-1][1] == o and segs[-1][2] == missed:
                segs[-1][1] += n
            else:
                segs.append([o, n, missed])
        self._plan = (key, segs)
        return segs

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FlatAdam.step takes no closure")
        group = self.param_groups[0]
        lr, (b1, b2), eps = group['lr'], group['betas'], group['eps']
        if group.get('weight_decay', 0) or group.get('amsgrad') or group.get('maximize'):
            raise NotImplementedError("FlatAdam implements plain Adam (no weight decay / amsgrad / maximize)")
        skipped, src, dst = [], [], []
        for p in group['params']:
            g = p.grad
            if g is None:
                skipped.append(p)
                continue
            if len(self.state[p]) == 0:
                self._ensure_state(p)
            view = self.grad_views[p]
            if g.data_ptr() != view.data_ptr():
                src.append(g)
                dst.append(view)
        if dst:
            torch._foreach_copy_(dst, src)
        self._t += 1
        for p in skipped:
            if p in self._missed:
                self._missed[p] += 1
        L, st = hip.lib(), hip.stream()
        for o, n, missed in self._segments(skipped):
            t = self._t - missed
            bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
            hip.check(L.sisr_adam_flat(self.flat_p.data_ptr() + 4 * o, self.flat_g.data_ptr() + 4 * o,
                                       self.flat_m.data_ptr() + 4 * o, self.flat_v.data_ptr() + 4 * o, n, b2, 1 - b1,
                                       1 - b2, eps, lr / bc1, math.sqrt(bc2), 1.0, st), "sisr_adam_flat")

    # -- torch-compatible (de)serialisation
    def state_dict(self):
        for p, st in self.state.items():
            if len(st):
                st['step'] = torch.tensor(float(self._t - self._missed.get(p, 0)), dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)  # torch casts / copies the loaded tensors next to the parameters
        steps = []
        for p in self.param_groups[0]['params']:
            st = self.state.get(p)
            if not st:
                continue
            for key, flat in (('exp_avg', self.flat_m), ('exp_avg_sq', self.flat_v)):
                view = self._view(flat, p)
                view.copy_(st[key])
                st[key] = view
            steps.append((p, int(float(st['step']))))
        self._t = max([s for _, s in steps], default=0)
        self._missed = {p: self._t - s for p, s in steps}
        self._plan = None
