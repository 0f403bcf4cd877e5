"""HIP graphs (torch.cuda.CUDAGraph = hipGraph on ROCm) of the two loops whose steps are launch-bound at small sizes: the reverse
chain of small graphs (GraphedChain: one graph per step number, replayed for every chain of the same shape) and the training step
(GraphedTrainStep: the whole step, one graph per batch shape)."""
from __future__ import annotations

import os
from typing import Optional
from collections import OrderedDict

import torch

from . import helpers
from .placeholder import PlaceHolder

# Reverse chains of SMALL graphs are host-bound: a step is ~130 kernel launches behind ~2 ms of Python / ctypes, while the GPU needs
# 2.6 ms at B = 100, n = 72 and less below.  GraphedChain replays every step as a HIP graph.  'auto': for batches of at most
# CHAIN_GRAPH_MAX_PAIRS pairs (B n^2) -- above, the edge kernels hide the host (B = 64, n = 300: 5.8 M pairs).  '0' / '1' force it.
CHAIN_GRAPHS = os.environ.get('DA_CHAIN_GRAPHS', 'auto')
CHAIN_GRAPH_MAX_PAIRS = 1_600_000
CHAIN_GRAPH_CACHE = 3            # shapes kept per model (each holds its graphs' private memory pool)


_CAPTURE_READY = set()


def _prepare_capture(dev) -> None:
    """torch creates the generator's graph-safe seed / offset tensors at the first capture of a process and updates them in place at
    every later one: if that first capture happens under torch.inference_mode (the inference API) they are inference tensors, and the
    next capture outside it (a sampler under no_grad) fails with 'Inplace update to inference tensor'.  One trivial capture outside
    inference mode, once per device, before any of ours."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key in _CAPTURE_READY or torch.cuda.is_current_stream_capturing():
        return
    with torch.inference_mode(False):
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            x = torch.zeros(8, device=dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                x += 1
        torch.cuda.current_stream(dev).wait_stream(s)
    _CAPTURE_READY.add(key)


def chain_graphs_wanted(B: int, n: int) -> bool:
    if CHAIN_GRAPHS == '0' or torch.cuda.is_current_stream_capturing():
        return False
    return CHAIN_GRAPHS == '1' or B * n * n <= CHAIN_GRAPH_MAX_PAIRS


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
        _prepare_capture(example.X.device)
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

    def __call__(self, data: PlaceHolder, next_data: Optional[PlaceHolder] = None) -> torch.Tensor:
        """One optimisation step on `data` (same shapes as the example).  Returns the (static) loss tensor of this step.
        `next_data`: the batch of the NEXT call, when the caller already has it (a prefetching data loader): its positional encodings
        -- a batched eigendecomposition of small matrices that depends on the batch alone and cannot be captured (the library routine
        synchronises) -- are then computed on a side stream beside this step's replay instead of in front of the next one
        (3 of 25.5 ms at B = 16, n = 140)."""
        from . import _lib as L
        for name in ('X', 'E', 'y', 'node_mask', 'atom_map_numbers', 'mol_assignments'):
            dst, src = getattr(self.data, name), getattr(data, name)
            if dst is not None and src is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        if self.pos is not None and data is not self.data:
            ahead = self.__dict__.pop('_ahead', None)
            if ahead is not None and ahead[0] is data:           # computed beside the previous replay
                torch.cuda.current_stream().wait_event(ahead[2])
                self.pos.copy_(ahead[1])
                ahead[1].record_stream(torch.cuda.current_stream())
            else:
                self.pos.copy_(self.model.get_pos_encodings(self.data))
        ready = torch.cuda.current_stream().record_event()      # next_data's tensors are ordered behind what the caller enqueued so far
        self.graph.replay()
        if self.pos is not None and next_data is not None and next_data is not self.data:
            side = self.__dict__.setdefault('_pos_stream', torch.cuda.Stream())
            with torch.cuda.stream(side):
                side.wait_event(ready)                           # (not for the replay: the encodings read next_data only)
                pos_next = self.model.get_pos_encodings(next_data)
                done = side.record_event()
            self._ahead = (next_data, pos_next, done)
        self.opt.steps += 1
        L.weights_epoch += 1                                 # the bf16 inference path re-packs on its next call
        if getattr(self.model, 'ema', None) is not None:     # outside the graph: the decay schedule is host state
            self.model.ema.update()
        return self.loss


class GraphedChain:
    """The reverse chain of SamplingMixin.sample_one_batch for one batch shape, every step a HIP graph of the SAME work the eager loop
    enqueues: the fused reverse step with its three-stream arrangement (engine._layers_node_on_main: the side streams join the
    capture through their events), the extra features with the eigensolver's fixed-block cache, the draw from per-chain counter
    noise, apply_mask and inpainting.  One graph per step number t (the step's transition tables, its y input and its noise call index
    are constants of that graph), captured the first time the step runs and replayed for every later chain of the same shape; what a
    chain brings -- condition, masks, positional encodings, seeds, inpainting lists, start state -- is copied into static buffers.
    Same kernels, same arguments: the samples equal the eager loop's bit for bit (tests/test_gpu_sampling.py)."""

    @staticmethod
    def get(model, data: PlaceHolder, masks, pos, temps, fix_nodes, fix_edges, with_eig_cache: bool, first_call: int):
        B, n, kx = data.X.shape
        key = (B, n, kx, data.E.shape[-1], int(model.cfg.diffusion.diffusion_steps_eval), tuple(temps),
               None if pos is None else tuple(pos.shape), None if fix_nodes is None else int(fix_nodes.shape[0]),
               None if fix_edges is None else int(fix_edges.shape[0]), bool(with_eig_cache), int(first_call),
               data.atom_map_numbers is not None, bool(model.cfg.diffusion.diffuse_edges), bool(model.cfg.diffusion.diffuse_nodes))
        cache = model.__dict__.setdefault('_chain_graphs', OrderedDict())
        pm = model._net()._pack(data.X.device)
        if key in cache and cache[key].pm is not pm:
            del cache[key]                                   # the weights moved: the packed bf16 copies the graphs point at are stale
        if key not in cache:
            while len(cache) >= CHAIN_GRAPH_CACHE:
                cache.popitem(last=False)
            cache[key] = GraphedChain(model, data, masks, pos, temps, fix_nodes, fix_edges, with_eig_cache, first_call)
        cache.move_to_end(key)
        return cache[key]

    def __init__(self, model, data, masks, pos, temps, fix_nodes, fix_edges, with_eig_cache, first_call):
        from . import _lib as L
        from .extra_features import EigCache
        dev = data.X.device
        _prepare_capture(dev)
        B, n, kx = data.X.shape
        self.model, self.temps, self.first_call, self.B, self.n = model, tuple(temps), int(first_call), B, n
        self.pm = model._net()._pack(dev)                    # the packed weights the graphs point at (kept alive; identity = freshness)
        self.T_eval = int(model.cfg.diffusion.diffusion_steps_eval)
        cl = lambda t: None if t is None else t.clone()
        self.data = data.get_new_object()                                        # static copies (every tensor field cloned)
        self.masks = (masks[0].clone(), masks[1].clone())
        self.pos = cl(pos)
        self.fix_nodes, self.fix_edges = cl(fix_nodes), cl(fix_edges)
        self.seeds = torch.zeros(B, dtype=torch.int64, device=dev)
        self.pad_noise = torch.zeros(B, n, device=dev)
        # the chain's state alternates between two holders: step t reads zs[t & 1] and the reverse step writes zs[(t - 1) & 1] directly
        # (two copies of the state per step before: two launches and two kernel boundaries of the 30 a step of a small batch has)
        self.zs = [self.data.get_new_object(y=torch.zeros(B, 1, device=dev)) for _ in range(2)]
        step = model.T // self.T_eval
        self.all_t = torch.arange(self.T_eval + 1, device=dev)[:, None, None].expand(-1, B, 1).contiguous()
        self.all_y = self.all_t.float() * step
        tm = model.transition_model_eval
        rep = lambda Q: Q[:, None].expand(-1, B, -1, -1).contiguous()
        self.tabs = (rep(tm.Qts_x), rep(tm.Qts_e), rep(tm.Qt_bars_x), rep(tm.Qt_bars_e))   # this object's own: never re-laid-out
        # query rows the denoiser's last layer computes (graph.free_rows of the loaded chain): the list and its length live in static
        # buffers and the kernels read the length on the device, so the captured steps serve every chain; the estimate that fixes the
        # launch geometry at capture time is the first chain's count
        self.rows = torch.zeros(B * n, dtype=torch.int32, device=dev)
        self.n_rows = torch.zeros(1, dtype=torch.int32, device=dev)
        self.rows_hint = None
        # per-row tile masks of the last two layers (graph.row_tile_masks), static like the list
        self.tiles_prev = torch.zeros(B * n, dtype=torch.int32, device=dev) if n <= 1024 else None
        self.tiles_last = torch.zeros(B * n, dtype=torch.int32, device=dev) if n <= 1024 else None
        self.eig = None
        if with_eig_cache:                                   # sized for any fixed block of this shape; filled per chain by load()
            self.eig = EigCache(torch.zeros(B, dtype=torch.int32, device=dev), n)
        self.graphs = {}
        self.pool = torch.cuda.graph_pool_handle()
        self.warm = False

    def load(self, data, masks, pos, fix_nodes, fix_edges, seeds, z_start, eig_cache, pad_noise, t_start=None):
        """A new chain of this shape: its inputs go into the static buffers, the eigensolver's cache starts empty.
        `t_start`: the step the chain starts at (default T_eval): the start state goes into that step's input holder."""
        for name in ('X', 'E', 'y', 'node_mask', 'atom_map_numbers', 'mol_assignments'):
            dst, src = getattr(self.data, name), getattr(data, name)
            if dst is not None:
                dst.copy_(src)
        self.masks[0].copy_(masks[0]); self.masks[1].copy_(masks[1])
        if self.pos is not None:
            self.pos.copy_(pos)
        if self.fix_nodes is not None:
            self.fix_nodes.copy_(fix_nodes)
        if self.fix_edges is not None:
            self.fix_edges.copy_(fix_edges)
        self.seeds.copy_(seeds)
        from . import graph
        rows = graph.free_rows(masks)
        self.rows[:rows.numel()].copy_(rows)
        self.n_rows.fill_(int(rows.numel()))
        if self.rows_hint is None:
            self.rows_hint = int(rows.numel())
        if self.tiles_prev is not None:
            tp, tl = graph.row_tile_masks(masks)
            self.tiles_prev.copy_(tp); self.tiles_last.copy_(tl)
        if pad_noise is not None:                            # drawn per chain only when the extra features' eigensolver runs
            self.pad_noise.copy_(pad_noise)
        ts = self.T_eval if t_start is None else int(t_start)
        z0, z1 = self.zs[ts & 1], self.zs[(ts - 1) & 1]
        z0.X.copy_(z_start.X); z0.E.copy_(z_start.E)
        # the other holder starts as the condition: the pairs that are not free to diffuse have the same value at every step
        # (apply_mask), so no step writes them again (reverse_step out_holds_condition: 73 % of the benchmark's pairs)
        z1.E.copy_(data.E)
        for z in self.zs:
            for name in ('node_mask', 'atom_map_numbers', 'mol_assignments'):
                if getattr(z, name) is not None:
                    getattr(z, name).copy_(getattr(data, name))
        if self.eig is not None:
            self.eig.fixed_from.copy_(eig_cache.fixed_from)
            self.eig.state.zero_()

    def _tables(self, t):
        qx, qe, bx, be = self.tabs
        P = lambda X, E: PlaceHolder(X=X, E=E, y=None)
        return P(qx[t], qe[t]), P(bx[t - 1], be[t - 1]), P(bx[t], be[t])

    def _body(self, t):
        from . import graph
        m, z, zo = self.model, self.zs[t & 1], self.zs[(t - 1) & 1]
        z.eig_cache, z.pad_noise = self.eig, (self.pad_noise if self.eig is not None else None)
        z.y = self.all_y[t]
        m.reverse_step(z, self.all_t[t], pos_encodings=self.pos, transition_model=m.transition_model_eval, temperature=self.temps,
                       orig=self.data, node_masks=self.masks, t_int=t, chain_seeds=self.seeds,
                       call=self.first_call + (self.T_eval - t), tables=self._tables(t), out=(zo.X, zo.E), out_holds_condition=True,
                       last_rows=graph.LastLayersPlan(self.rows, self.n_rows, self.rows_hint, self.tiles_prev, self.tiles_last) if self.rows_hint else None)
        graph.apply_inpainting(zo, self.data, self.fix_nodes, self.fix_edges)
        if not m.cfg.diffusion.diffuse_edges:
            zo.E.copy_(self.data.E)
        if not m.cfg.diffusion.diffuse_nodes:
            zo.X.copy_(self.data.X)

    def run(self):
        """All T_eval steps on the loaded chain; returns the static state holder the last step wrote (clone what must outlive the next chain)."""
        return self.run_steps(self.T_eval, 1)

    def run_steps(self, t_from: int, t_to: int):
        """Steps t_from, t_from - 1, ..., t_to on the loaded chain (whose state sits in the input holder of step t_from)."""
        with torch.no_grad():
            for t in range(t_from, t_to - 1, -1):
                g = self.graphs.get(t)
                if g is None:
                    if not self.warm:
                        # one eager step on a side stream first (lazy set-up that must not happen under capture: weight packing,
                        # function attributes, allocator pools); the state it changes is put back
                        keep = None if self.eig is None else self.eig.state.clone()     # (the step's input holder is only read)
                        s = torch.cuda.Stream()
                        s.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(s):
                            self._body(t)
                        torch.cuda.current_stream().wait_stream(s)
                        if self.eig is not None:
                            self.eig.state.copy_(keep)
                        self.warm = True
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=self.pool):
                        self._body(t)
                    self.graphs[t] = g
                g.replay()
        return self.zs[(t_to - 1) & 1]
