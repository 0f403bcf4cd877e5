"""AdamW with amsgrad (the reference's optimiser: utils/setup.py:356, lr / weight_decay from configs/train/default.yaml) as ONE
fused HIP pass over a flat fp32 parameter buffer (C-ABI da_adamw).  The parameters and their gradients are re-pointed at views
of two flat buffers, so that gathering the gradients is one multi-tensor copy, the data-parallel all-reduce is one RCCL call over
the whole 32.7 MB gradient (distributed.allreduce_flat_gradient) and the update is one kernel launch."""
from __future__ import annotations

import torch

from . import _lib as L


class FlatAdamW:
    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-12, amsgrad=True, device_step=False):
        self.params = [p for p in params if p.requires_grad]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params), 'fp32 parameters on a HIP device'
        dev = self.params[0].device
        pad = lambda k: (k + 63) // 64 * 64                      # every parameter starts on a 256-byte boundary of the flat buffers
        n = sum(pad(p.numel()) for p in self.params)
        self.flat_param = torch.zeros(n, device=dev)
        self.flat_grad = torch.zeros(n, device=dev)
        o = 0
        for p in self.params:
            k = p.numel()
            self.flat_param[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + k].view_as(p)
            o += pad(k)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.max_exp_avg_sq = torch.zeros(n, device=dev) if amsgrad else None
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.steps = 0
        # device_step: the update count lives on the device (da_adamw_devstep), so that step() can be captured in a HIP graph
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev) if device_step else None
        self._written = set()      # indices of parameters whose gradient slot has ever been filled

    def zero_grad(self):
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
            elif i in self._written:                             # had a gradient in an earlier step, none now: no stale values
                view.zero_()
                self._written.discard(i)
            o += (k + 63) // 64 * 64
        if src:
            torch._foreach_copy_(dst, src)

    def step(self, grad_scale: float = 1.0):
        """One fused AdamW pass over the flat buffers; call after `gather` (and after the gradient all-reduce)."""
        self.steps += 1
        if self.step_dev is not None:
            L.check(L.lib().da_adamw_devstep(L.ptr(self.flat_param), L.ptr(self.flat_grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                             L.ptr(self.max_exp_avg_sq), self.flat_param.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                             self.weight_decay, L.ptr(self.step_dev), float(grad_scale), L.stream()), 'da_adamw_devstep')
            L.weights_epoch += 1
            return
        L.check(L.lib().da_adamw(L.ptr(self.flat_param), L.ptr(self.flat_grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                 L.ptr(self.max_exp_avg_sq), self.flat_param.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                 self.weight_decay, self.steps, float(grad_scale), L.stream()), 'da_adamw')
        L.weights_epoch += 1            # the fused inference path re-packs its bf16 copies on its next call
