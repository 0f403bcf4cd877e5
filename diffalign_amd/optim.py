"""AdamW with amsgrad (the reference's optimiser: utils/setup.py:356, lr / weight_decay from configs/train/default.yaml) as ONE
fused HIP pass over a flat fp32 parameter buffer (C-ABI da_adamw).  The parameters and their gradients are re-pointed at views
of two flat buffers, so that gathering the gradients is one multi-tensor copy, the data-parallel all-reduce is one RCCL call over
the whole 32.7 MB gradient (distributed.allreduce_flat_gradient) and the update is one kernel launch.

A `torch.optim.Optimizer`: torch's learning-rate schedulers drive `param_groups[0]['lr']` (utils/setup.py:357), and
`state_dict()` / `load_state_dict()` speak torch.optim.AdamW's format, so the `optimizer_state_dict` of a reference checkpoint
(training/helpers.py:99-110, utils/setup.py:331-346) resumes here and a checkpoint written here resumes there.

Parameters with requires_grad=False (the reference builds AdamW over the diffusion object's parameters, utils/setup.py:356, which with
use_ema include the frozen EMA copy) stay in the parameter group -- the indices of state_dict()['param_groups'][0]['params'] are
positions in the list the optimiser was built from, as in torch -- and are left out of the flat buffers only.

One divergence from torch, documented: the kernel's bias correction uses ONE update count for every parameter.  torch counts per
parameter, so a parameter that receives its first gradient k steps late is corrected with step - k there and with step here.  The
saved 'step' of such a parameter is its own count (step minus the step it joined at), as torch would have saved it; on load the
largest count becomes the shared one.  In this model every trained parameter has a gradient from the first step on."""
from __future__ import annotations

import torch

from . import _lib as L


def _pad(k: int) -> int:
    return (k + 63) // 64 * 64                                   # every parameter starts on a 256-byte boundary of the flat buffers


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-12, amsgrad=True, device_step=False):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        assert len(self.param_groups) == 1, 'one parameter group (the reference has one)'
        self.all_params = self.param_groups[0]['params']          # torch's index space (frozen parameters included)
        self.group_index = [j for j, p in enumerate(self.all_params) if p.requires_grad]   # flat slot i -> index in the group
        self.params = [self.all_params[j] for j in self.group_index]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params), 'fp32 parameters on a HIP device'
        dev = self.params[0].device
        n = sum(_pad(p.numel()) for p in self.params)
        self.flat_param = torch.zeros(n, device=dev)
        self.flat_grad = torch.zeros(n, device=dev)
        o = 0
        for p in self.params:
            k = p.numel()
            self.flat_param[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + k].view_as(p)
            o += _pad(k)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.max_exp_avg_sq = torch.zeros(n, device=dev) if amsgrad else None
        self.steps = 0
        # device_step: the update count lives on the device (da_adamw_devstep), so that step() can be captured in a HIP graph
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev) if device_step else None
        # one byte per 64-element slot of the flat buffers: 1 = the slot's parameter has a gradient in this step.  Parameters without
        # one are skipped by the kernel as torch.optim skips them (no weight decay, moments frozen); rewritten only when the set changes
        self.slot_active = torch.zeros(n // 64, dtype=torch.uint8, device=dev)
        self._active_key = None
        self._written = set()      # indices of parameters whose gradient slot holds this step's gradient
        self._ever = set()         # indices of parameters that have ever been updated with a gradient (torch keeps state for these only)
        self._joined = {}          # slot -> value of `steps` when the parameter had its first gradient (its own count = steps - that)

    # the hyper-parameters live in the parameter group, where schedulers and load_state_dict change them
    lr = property(lambda self: self.param_groups[0]['lr'])
    betas = property(lambda self: self.param_groups[0]['betas'])
    eps = property(lambda self: self.param_groups[0]['eps'])
    weight_decay = property(lambda self: self.param_groups[0]['weight_decay'])

    def zero_grad(self, set_to_none: bool = True):
        """`.grad = None` for every parameter (no kernel): backward() then hands over its own gradient tensors, which `gather` copies
        into the flat buffer with one multi-tensor launch -- accumulating into pre-bound views would cost one add kernel per parameter
        (~300 launches of a host-bound step).  Slots of parameters the objective never reaches stay at their initial zero."""
        for p in self.params:
            p.grad = None

    def gather(self):
        """Copy the gradients backward() left on the parameters into the flat buffer and re-point `.grad` at its views."""
        o, src, dst = 0, [], []
        for i, p in enumerate(self.params):
            k = p.numel()
            view = self.flat_grad[o:o + k].view_as(p)
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                src.append(p.grad); dst.append(view)
            if p.grad is not None:
                p.grad = view
                self._written.add(i)
                if i not in self._ever:
                    self._ever.add(i)
                    self._joined[i] = self.steps
            elif i in self._written:                             # had a gradient in an earlier step, none now: no stale values
                view.zero_()
                self._written.discard(i)
            o += _pad(k)
        if src:
            torch._foreach_copy_(dst, src)
        key = frozenset(self._written)
        if key != self._active_key:                              # (first step, or the objective changed: not inside a captured step)
            mask = torch.zeros(self.slot_active.numel(), dtype=torch.uint8)
            o = 0
            for i, p in enumerate(self.params):
                if i in key:
                    mask[o // 64:(o + _pad(p.numel())) // 64] = 1
                o += _pad(p.numel())
            self.slot_active.copy_(mask)
            self._active_key = key

    def step(self, closure=None, grad_scale: float = 1.0):
        """One fused AdamW pass over the flat buffers; call after `gather` (and after the gradient all-reduce)."""
        assert closure is None, 'closures are not supported'
        self.steps += 1
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd = float(g['lr']), g['betas'], float(g['eps']), float(g['weight_decay'])
        if self.step_dev is not None:
            L.check(L.lib().da_adamw_devstep(L.ptr(self.flat_param), L.ptr(self.flat_grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                             L.ptr(self.max_exp_avg_sq), self.flat_param.numel(), lr, b1, b2, eps, wd, L.ptr(self.step_dev),
                                             float(grad_scale), L.ptr(self.slot_active), L.stream()), 'da_adamw_devstep')
            L.weights_epoch += 1
            return
        L.check(L.lib().da_adamw(L.ptr(self.flat_param), L.ptr(self.flat_grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                 L.ptr(self.max_exp_avg_sq), self.flat_param.numel(), lr, b1, b2, eps, wd, self.steps, float(grad_scale),
                                 L.ptr(self.slot_active), L.stream()), 'da_adamw')
        L.weights_epoch += 1            # the fused inference path re-packs its bf16 copies on its next call

    # -- checkpoints in torch.optim.AdamW's format ---------------------------------------------------------------------------
    def _slots(self):
        o = 0
        for i, p in enumerate(self.params):
            yield i, p, o
            o += _pad(p.numel())

    def state_dict(self):
        state = {}
        for i, p, o in self._slots():
            if i not in self._ever:
                continue                                         # torch has no entry for a parameter that never had a gradient
            k = p.numel()
            cut = lambda buf: buf[o:o + k].view_as(p).clone()
            j = self.group_index[i]
            state[j] = {'step': torch.tensor(float(self.steps - self._joined.get(i, 0))), 'exp_avg': cut(self.exp_avg), 'exp_avg_sq': cut(self.exp_avg_sq)}
            if self.max_exp_avg_sq is not None:
                state[j]['max_exp_avg_sq'] = cut(self.max_exp_avg_sq)
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        group['params'] = list(range(len(self.all_params)))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, state_dict):
        """torch.optim.AdamW's format.  The saved parameter list must be this optimiser's list (same length, same order), or, for a
        reference checkpoint written with use_ema, a superset of it whose extra entries come last (the frozen EMA copy, which has no
        state): parameters are matched by position."""
        groups = state_dict['param_groups']
        assert len(groups) == 1 and len(groups[0]['params']) >= len(self.all_params), 'optimizer state of another parameter list'
        saved_ids = list(groups[0]['params'])
        extra = set(saved_ids[len(self.all_params):])
        assert not (extra & set(state_dict['state'])), 'optimizer state of another parameter list (state for parameters this optimizer does not have)'
        for k in ('lr', 'betas', 'eps', 'weight_decay'):
            if k in groups[0]:
                self.param_groups[0][k] = groups[0][k]
        for k in ('initial_lr',):                                # what a scheduler left in the group
            if k in groups[0]:
                self.param_groups[0][k] = groups[0][k]
        if bool(groups[0].get('amsgrad', self.max_exp_avg_sq is not None)) != (self.max_exp_avg_sq is not None):
            raise ValueError('amsgrad differs between the checkpoint and this optimizer')
        for buf in (self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq):
            if buf is not None:
                buf.zero_()
        slot_of = {j: i for i, j in enumerate(self.group_index)}
        own = {}                                                 # slot -> that parameter's own update count
        for pid, st in state_dict['state'].items():
            j = saved_ids.index(pid)
            assert j in slot_of, f'optimizer state for parameter {j}, which does not require a gradient here'
            own[slot_of[j]] = int(float(st['step']))
        steps = max(own.values(), default=0)
        self._ever = set(own)
        self._joined = {i: steps - c for i, c in own.items()}
        for i, p, o in self._slots():
            if i not in self._ever:
                continue
            st = state_dict['state'][saved_ids[self.group_index[i]]]
            k = p.numel()
            self.exp_avg[o:o + k].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + k].copy_(st['exp_avg_sq'].reshape(-1))
            if self.max_exp_avg_sq is not None:
                self.max_exp_avg_sq[o:o + k].copy_(st['max_exp_avg_sq'].reshape(-1))
        self.steps = steps
        if self.step_dev is not None:
            self.step_dev.fill_(steps)
