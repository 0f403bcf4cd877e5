"""Multi-GPU sampling: one process per GPU (torch.distributed; backend 'nccl' = RCCL on ROCm, 'gloo' in CPU tests).

The unit of work -- one reverse-diffusion chain of one sample of one condition -- is independent, so the flattened
(condition, sample) index space is cut into contiguous shards, one per rank, and NO collective runs inside the
100-step loop.  One final gather of the collapsed uint8 results (n + n*n bytes per chain) brings them to rank 0.
The reference does the same coarsely with SLURM array jobs over condition ranges (sampling/helpers.py:10-27,
slurm/slurm_sample.py:65-73) and duplicates every condition n_samples times (utils/graph.py:145-152)."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, end) of `total` units for `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def chain_seed(seed: int, global_index: int) -> int:
    """Noise stream of a chain depends on (seed, global chain index) only, never on the rank layout."""
    return (seed * 0x9E3779B1 + global_index * 0x85EBCA77) & 0x7FFFFFFF


def collective_device(group=None) -> torch.device:
    """Device the group's collectives run on: the current HIP device under 'nccl' (= RCCL), the host under 'gloo'."""
    if dist.is_initialized() and 'nccl' in str(dist.get_backend(group)):
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def gather_collapsed(X: torch.Tensor, E: torch.Tensor, group=None):
    """The final gather of sharded sampling for equal shards: every rank's collapsed uint8 results (m, n) / (m, n, n) ->
    lists of `world` tensors on every rank (one all_gather each; weak-scaling bench)."""
    world = dist.get_world_size(group)
    Xs = [torch.empty_like(X) for _ in range(world)]
    Es = [torch.empty_like(E) for _ in range(world)]
    dist.all_gather(Xs, X.contiguous(), group=group)
    dist.all_gather(Es, E.contiguous(), group=group)
    return Xs, Es


def sample_sharded(n_conditions: int, n_samples: int, n_nodes: int,
                   sample_fn: Callable[[List[Tuple[int, int]]], Tuple[torch.Tensor, torch.Tensor]],
                   group=None, dst: int = 0, device=None) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """sample_fn(units) -> (X_idx uint8 (len(units), n), E_idx uint8 (len(units), n, n)) for units = [(cond, sample)].
    Returns on rank `dst` the results for all n_conditions*n_samples chains in (condition-major) order, else None.
    `device`: where the gathered tensors live (default: the group's collective device); a rank whose shard is empty
    (fewer chains than ranks) contributes empty tensors on that same device."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    total = n_conditions * n_samples
    s, e = shard_range(total, rank, world)
    units = [(g // n_samples, g % n_samples) for g in range(s, e)]
    dev = torch.device(device) if device is not None else collective_device(group)
    if units:
        X, E = sample_fn(units)
        X, E = X.to(device=dev, dtype=torch.uint8).contiguous(), E.to(device=dev, dtype=torch.uint8).contiguous()
    else:
        X = torch.empty(0, n_nodes, dtype=torch.uint8, device=dev); E = torch.empty(0, n_nodes, n_nodes, dtype=torch.uint8, device=dev)
    if world == 1:
        return X, E
    # pad every shard to the largest shard so one all_gather moves everything
    m = (total + world - 1) // world
    pad = lambda t: torch.cat([t, t.new_zeros((m - t.shape[0],) + tuple(t.shape[1:]))], 0) if t.shape[0] < m else t
    Xp, Ep = pad(X), pad(E)
    Xs = [torch.empty_like(Xp) for _ in range(world)]
    Es = [torch.empty_like(Ep) for _ in range(world)]
    dist.all_gather(Xs, Xp, group=group)
    dist.all_gather(Es, Ep, group=group)
    if rank != dst:
        return None
    outX, outE = [], []
    for r in range(world):
        rs, re_ = shard_range(total, r, world)
        outX.append(Xs[r][:re_ - rs]); outE.append(Es[r][:re_ - rs])
    return torch.cat(outX, 0), torch.cat(outE, 0)


def sample_conditions_sharded(conditions, n_samples: int, sample_for_condition: Callable, seed: int = 0, inpaint_node_idx=None,
                              inpaint_edge_idx=None, group=None, dst: int = 0, device=None):
    """BASELINE configs[2]: `n_samples` reverse-diffusion chains for each of len(conditions) conditions, the flattened (condition,
    sample) index space cut over the ranks (shard_range), no collective inside the chains, one final gather.

    conditions: list of single-graph dense conditions (PlaceHolder-like, X (1,n_c,kx) ...), known to every rank.
    sample_for_condition(dense_data=, n_samples=k, inpaint_node_idx=, inpaint_edge_idx=, chain_seeds=) -> object with collapsed
    .X (k,n_c) and .E (k,n_c,n_c): DiscreteDenoisingDiffusionRxn.sample_for_condition (diffusion/diffusion_rxn.py:107-128).
    A rank calls it once per condition that has chains in its shard, with k = the number of those chains and chain_seeds =
    chain_seed(seed, global chain index) -- so the samples do not depend on the number of ranks.
    Returns on rank `dst` (X uint8 (C*n_samples, n_max), E uint8 (C*n_samples, n_max, n_max)) in condition-major order, graphs
    smaller than n_max = the largest condition zero-padded; None on the other ranks."""
    n_max = max(int(c.X.shape[1]) for c in conditions) if conditions else 0

    def run(units):
        outX, outE = [], []
        k0 = 0
        while k0 < len(units):
            c = units[k0][0]
            k1 = k0
            while k1 < len(units) and units[k1][0] == c:
                k1 += 1
            cond = conditions[c]
            seeds = torch.tensor([chain_seed(seed, c * n_samples + s_) for _, s_ in units[k0:k1]], dtype=torch.int64)
            res = sample_for_condition(dense_data=cond.get_new_object(), n_samples=k1 - k0,
                                       inpaint_node_idx=[inpaint_node_idx[c]] if inpaint_node_idx is not None else None,
                                       inpaint_edge_idx=[inpaint_edge_idx[c]] if inpaint_edge_idx is not None else None, chain_seeds=seeds)
            X, E = res.X.to(torch.uint8), res.E.to(torch.uint8)
            pad = n_max - X.shape[1]
            if pad:
                X = torch.nn.functional.pad(X, (0, pad)); E = torch.nn.functional.pad(E, (0, pad, 0, pad))
            outX.append(X); outE.append(E)
            k0 = k1
        return torch.cat(outX, 0), torch.cat(outE, 0)

    return sample_sharded(len(conditions), n_samples, n_max, run, group=group, dst=dst, device=device)


def allreduce_flat_gradient(flat_grad: torch.Tensor, group=None) -> int:
    """Data-parallel training (BASELINE configs[3]): SUM all-reduce of the flat fp32 gradient buffer of optim.FlatAdamW -- the whole
    32.7 MB align_absorbing gradient as ONE ring all-reduce (xGMI is per-link bound: few large transfers) -- in place; returns the
    world size (the optimiser divides by it: da_adamw's grad_scale).  Parameters the objective never reaches (mlp_out_y,
    self_attn.y_y, skip_scaling_3) keep zero entries on every rank, so no unused-parameter handling is needed."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 1
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return dist.get_world_size(group)


def allreduce_gradients(grads: Dict[str, torch.Tensor], group=None, bucket_bytes: int = 32 << 20) -> Dict[str, torch.Tensor]:
    """Data-parallel training: average the ranks' gradients in place and return them.  The tensors are packed
    in sorted-name order into flat fp32 buckets of <= bucket_bytes (the whole align_absorbing gradient is 32.7 MB: one or
    two large ring all-reduces, which is what xGMI's per-link bound favours over many small ones), every bucket's
    all-reduce is issued asynchronously before the first is waited for.  Every rank must pass the same names and shapes."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return grads
    world = dist.get_world_size(group)
    names = sorted(grads)
    buckets, cur, cur_bytes = [], [], 0
    for k in names:
        nb = grads[k].numel() * 4
        if cur and cur_bytes + nb > bucket_bytes:
            buckets.append(cur); cur, cur_bytes = [], 0
        cur.append(k); cur_bytes += nb
    if cur:
        buckets.append(cur)
    flats, works = [], []
    for ks in buckets:
        flat = torch.cat([grads[k].detach().reshape(-1).float() for k in ks])
        flats.append(flat)
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for ks, flat, w in zip(buckets, flats, works):
        w.wait()
        flat /= world
        o = 0
        for k in ks:
            m = grads[k].numel()
            grads[k].copy_(flat[o:o + m].reshape(grads[k].shape)); o += m
    return grads
