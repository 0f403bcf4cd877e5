"""Exponential moving average of the denoiser's weights (diffalign/neuralnet/ema_pytorch.py:18-181, used by
diffusion_abstract.py:94-97, 333-339 and training/helpers.py:143-144 when cfg.neuralnet.use_ema).

Same constructor arguments, `update()` schedule, decay warm-up and state-dict keys (`online_model.*`, `ema_model.*`, `initted`,
`step`) as the reference's class, so its checkpoints' `ema_state_dict` load here.  What differs is how an update runs: the reference
walks the parameters in Python (one `lerp_` launch and several string checks per tensor, and two `.item()` syncs per call); here the
tensors are sorted into a lerp list and a copy list once, the step counter lives on the host, and an update is one multi-tensor
`torch._foreach_lerp_` (plus one `_foreach_copy_` when names are excluded)."""
from __future__ import annotations

import copy

import torch
from torch import nn

from . import _lib as L


class EMA(nn.Module):
    def __init__(self, model, ema_model=None, beta=0.9999, update_after_step=100, update_every=10, inv_gamma=1.0, power=2 / 3,
                 min_value=0.0, param_or_buffer_names_no_ema=set(), ignore_names=set(), ignore_startswith_names=set(),
                 include_online_model=True):
        super().__init__()
        self.beta = beta
        self.include_online_model = include_online_model
        if include_online_model:
            self.online_model = model
        else:
            self.online_model = [model]                                   # kept out of the module tree (ema_pytorch.py:60-63)
        self.ema_model = ema_model
        if self.ema_model is None:
            packed = getattr(model, '_packed', None)
            if packed is not None:
                model._packed = None                                      # the packed inference weights are a cache, not state
            self.ema_model = copy.deepcopy(model)
            if hasattr(self.ema_model, '_plist'):
                self.ema_model._plist = None                              # likewise: rebuilt from the copy's own parameters
            if packed is not None:
                model._packed = packed
        self.ema_model.requires_grad_(False)
        ok = (torch.float, torch.float16)
        self.parameter_names = {n for n, p in self.ema_model.named_parameters() if p.dtype in ok}
        self.buffer_names = {n for n, b in self.ema_model.named_buffers() if b.dtype in ok}
        self.update_every = update_every
        self.update_after_step = update_after_step
        self.inv_gamma = inv_gamma
        self.power = power
        self.min_value = min_value
        assert isinstance(param_or_buffer_names_no_ema, (set, list))
        self.param_or_buffer_names_no_ema = param_or_buffer_names_no_ema
        self.ignore_names = ignore_names
        self.ignore_startswith_names = ignore_startswith_names
        self.register_buffer('initted', torch.Tensor([False]))
        self.register_buffer('step', torch.tensor([0]))
        self._host_step = 0                                               # mirrors of the two buffers: no device sync per update
        self._host_initted = False
        self._lists = None

    @property
    def model(self):
        return self.online_model if self.include_online_model else self.online_model[0]

    def restore_ema_model_device(self):
        self.ema_model.to(self.initted.device)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if prefix + 'step' in state_dict:
            self._host_step = int(state_dict[prefix + 'step'].reshape(-1)[0])
        if prefix + 'initted' in state_dict:
            self._host_initted = bool(state_dict[prefix + 'initted'].reshape(-1)[0])
        self._lists = None

    # -- which tensors take part (ema_pytorch.py:105-122, 154-178), resolved once
    def _pairs(self):
        if self._lists is None:
            cur = dict(self.model.named_parameters()); cur.update(dict(self.model.named_buffers()))
            ema = dict(self.ema_model.named_parameters()); ema.update(dict(self.ema_model.named_buffers()))
            names = [n for n, _ in self.ema_model.named_parameters() if n in self.parameter_names]
            names += [n for n, _ in self.ema_model.named_buffers() if n in self.buffer_names]
            everything = [(ema[n], cur[n]) for n in names]
            lerp, copy_ = [], []
            for n in names:
                if n in self.ignore_names or any(n.startswith(p) for p in self.ignore_startswith_names):
                    continue
                (copy_ if n in self.param_or_buffer_names_no_ema else lerp).append((ema[n], cur[n]))
            self._lists = (everything, lerp, copy_)
        return self._lists

    @staticmethod
    def _data(pairs):
        """The tensors behind the (EMA, online) pairs, read now: an optimiser may have re-bound a parameter's storage since the last call
        (optim.FlatAdamW moves every parameter into one flat buffer)."""
        return [e.data for e, _ in pairs], [c.data for _, c in pairs]

    @torch.no_grad()
    def copy_params_from_model_to_ema(self):
        everything = self._pairs()[0]
        if everything:
            torch._foreach_copy_(*self._data(everything))
        L.weights_epoch += 1

    def get_current_decay(self):
        epoch = max(self._host_step - self.update_after_step - 1, 0.)
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        if epoch <= 0:
            return 0.
        return min(max(value, self.min_value), self.beta)

    def update(self):
        step = self._host_step
        self._host_step += 1
        self.step += 1
        if (step % self.update_every) != 0:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not self._host_initted:
            self.copy_params_from_model_to_ema()
            self._host_initted = True
            self.initted.fill_(1.0)
        self.update_moving_average(self.ema_model, self.model)

    @torch.no_grad()
    def update_moving_average(self, ma_model=None, current_model=None):
        decay = self.get_current_decay()
        _, lerp, copy_ = self._pairs()
        if lerp:
            torch._foreach_lerp_(*self._data(lerp), 1. - decay)
        if copy_:
            torch._foreach_copy_(*self._data(copy_))
        L.weights_epoch += 1                                              # the EMA module's packed inference weights are stale

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)
