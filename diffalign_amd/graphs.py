"""One reverse-diffusion step captured in a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm).

A step is ~110 kernel launches issued from Python through ctypes; replaying them as one graph removes the host launch
gaps (about 3 ms of a 22 ms step at B=64, n=300).  Everything that changes between steps lives in static device
buffers that are overwritten before each replay: the state z_t (updated inside the graph), the step index y and the
six gathered transition tables."""
from __future__ import annotations

import torch

from . import helpers
from .placeholder import PlaceHolder


class GraphedReverseStep:
    def __init__(self, model, dense_data: PlaceHolder, z_t: PlaceHolder, pos_encodings, node_masks, temps=(1.0, 1.0),
                 warmup: int = 2):
        self.model, self.data, self.masks, self.temps, self.pos = model, dense_data, node_masks, temps, pos_encodings
        dev = z_t.X.device
        B, n, kx = z_t.X.shape
        ke = z_t.E.shape[-1]
        self.B = B
        self.z = z_t.get_new_object(y=torch.zeros(B, 1, device=dev))
        self.tabs = [torch.zeros(B, kx, kx, device=dev) for _ in range(3)] + [torch.zeros(B, ke, ke, device=dev) for _ in range(3)]
        self.qX = torch.empty(B, n, kx, device=dev); self.qE = torch.empty(B, n, n, ke, device=dev)
        self.tm = model.transition_model_eval
        self.step_size = model.T // model.cfg.diffusion.diffusion_steps_eval
        self._set_t(1)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        keepX, keepE = self.z.X.clone(), self.z.E.clone()
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(s)
        self.z.X.copy_(keepX); self.z.E.copy_(keepE)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self._body()
        self.z.X.copy_(keepX); self.z.E.copy_(keepE)

    def _set_t(self, t: int):
        t_arr = torch.full((self.B, 1), t, dtype=torch.long, device=self.z.X.device)
        self.z.y.fill_(float(t * self.step_size))
        Qt, Qsb, Qtb = self.tm.get_Qt(t_arr), self.tm.get_Qt_bar(t_arr - 1), self.tm.get_Qt_bar(t_arr)
        for dst, src in zip(self.tabs, (Qt.X, Qsb.X, Qtb.X, Qt.E, Qsb.E, Qtb.E)):
            dst.copy_(src)

    def _body(self):
        pred = self.model.forward(self.z, pos_encodings=self.pos)
        self.qX.exponential_(1.0); self.qE.exponential_(1.0)
        ph = lambda X, E: PlaceHolder(X=X, E=E, y=self.z.y)
        _, _, zsX, zsE = helpers.posterior_sample(pred.X, pred.E, self.z, ph(self.tabs[0], self.tabs[3]), ph(self.tabs[1], self.tabs[4]),
                                                  ph(self.tabs[2], self.tabs[5]), temperature=self.temps, noise=(self.qX, self.qE),
                                                  orig=self.data, node_masks=self.masks)
        self.z.X.copy_(zsX); self.z.E.copy_(zsE)

    def step(self, t: int) -> PlaceHolder:
        """z_t <- sample of p(z_{t-1} | z_t); returns the (static) state holder."""
        self._set_t(t)
        self.graph.replay()
        return self.z


class GraphedTrainStep:
    """One whole training step -- apply_noise, apply_mask, extra features, the differentiable fp32 forward, the objective, backward(),
    the gradient gather, (the data-parallel all-reduce) and the fused AdamW -- captured once as a HIP graph and replayed per batch.
    An eager step issues ~2,400 launches from Python and autograd and is host-bound at the reference's batch size;
    the replay costs one launch.  What changes between steps lives in static device buffers: the batch (copied in before a replay),
    the positional encodings (they depend on the condition only: computed outside, eagerly) and the optimiser's step counter
    (optim.FlatAdamW(device_step=True)).  Random draws (t, the noise, dropout masks) come from torch's graph-safe device generator.
    One graph per batch shape.

    Building the object does not train: the `warmup` eager steps (they settle allocator pools, workspaces and lazily built tables)
    and the capture pass run on the example batch, and the weights, both moments, the device step counter and the host's `steps` are
    put back to what they were afterwards (torch's generator state is consumed).  After N calls opt.steps == int(opt.step_dev)
    == the count before construction + N."""

    def __init__(self, model, optimizer, example: PlaceHolder, warmup: int = 3, group=None):
        assert optimizer.step_dev is not None, 'GraphedTrainStep needs optim.FlatAdamW(..., device_step=True)'
        assert model.model.training, 'call model.train() first'
        self.model, self.opt, self.group = model, optimizer, group
        self.data = example.get_new_object()                 # static copies (get_new_object clones every field)
        self.pos = model.get_pos_encodings(self.data)
        if self.pos is not None:
            self.pos = self.pos.clone()
        opt = optimizer
        bufs = [b for b in (opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.max_exp_avg_sq, opt.step_dev) if b is not None]
        keep = [b.clone() for b in bufs]
        keep_host = (opt.steps, set(opt._ever), dict(opt._joined))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):                          # real steps on the example batch; undone below
                model.train_one_step(self.data, optimizer, group=group, pos_encodings=self.pos, update_ema=False)
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = model.train_one_step(self.data, optimizer, group=group, pos_encodings=self.pos, update_ema=False)
        for b, k in zip(bufs, keep):                         # the capture pass enqueued nothing; the warm-up steps are taken back
            b.copy_(k)
        opt.steps = keep_host[0]                             # (the capture pass ran FlatAdamW.step() in Python: its count too)
        if keep_host[0] > 0 or keep_host[1]:
            opt._ever, opt._joined = keep_host[1], keep_host[2]
        else:
            opt._joined = {i: 0 for i in opt._ever}          # a fresh optimiser: every trained parameter joins at step 0
        from . import _lib as L_
        L_.weights_epoch += 1

    def __call__(self, data: PlaceHolder) -> torch.Tensor:
        """One optimisation step on `data` (same shapes as the example).  Returns the (static) loss tensor of this step."""
        from . import _lib as L
        for name in ('X', 'E', 'y', 'node_mask', 'atom_map_numbers', 'mol_assignments'):
            dst, src = getattr(self.data, name), getattr(data, name)
            if dst is not None and src is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        if self.pos is not None and data is not self.data:
            self.pos.copy_(self.model.get_pos_encodings(self.data))
        self.graph.replay()
        self.opt.steps += 1
        L.weights_epoch += 1                                 # the bf16 inference path re-packs on its next call
        if getattr(self.model, 'ema', None) is not None:     # outside the graph: the decay schedule is host state
            self.model.ema.update()
        return self.loss
