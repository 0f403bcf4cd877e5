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


def unit_cost(n: int) -> float:
    """Relative cost of one chain of a condition with n nodes: the edge kernels are proportional to n^2 pairs, the node / y
    chains and the launch boundaries to n and to a constant (measured steps: 2.65 ms at B = 100, n = 72; 12.7 ms at B = 64, n = 300)."""
    return float(n * n + 48 * n + 2000)


def balanced_cuts(costs: List[float], world: int) -> List[Tuple[int, int]]:
    """Contiguous cuts [start, end) of a cost sequence into `world` shards whose cost sums are as equal as contiguous cuts allow:
    cut r is the prefix whose cost is nearest to (r + 1) / world of the total (monotone; empty shards when there are fewer units
    than ranks).  Every rank computes the same cuts from the same sequence: no communication."""
    n = len(costs)
    pre = [0.0]
    for c in costs:
        pre.append(pre[-1] + c)
    cuts, lo = [], 0
    for r in range(world):
        if r == world - 1:
            hi = n
        else:
            target = pre[-1] * (r + 1) / world
            hi = lo
            while hi < n and abs(pre[hi + 1] - target) <= abs(pre[hi] - target):
                hi += 1
        cuts.append((lo, hi))
        lo = hi
    return cuts


def plan_condition_shards(sizes: List[int], n_samples: int, world: int, balance: bool = True):
    """The (condition, sample) units of every rank.  balance=False: condition-major order cut into equal counts (shard_range).
    balance=True: units ordered by graph size (largest first; ties by condition, sample) and cut by cost (unit_cost: ~ n^2), so a
    rank that holds the large conditions holds fewer chains, and conditions of equal size sit next to each other (they run as one
    batch).  Returns a list of `world` lists of (condition, sample)."""
    units = [(c, s_) for c in range(len(sizes)) for s_ in range(n_samples)]
    if not balance:
        return [units[slice(*shard_range(len(units), r, world))] for r in range(world)]
    units.sort(key=lambda u: (-sizes[u[0]], u[0], u[1]))
    cuts = balanced_cuts([unit_cost(sizes[c]) for c, _ in units], world)
    return [units[a:b] for a, b in cuts]


def _stack_conditions(conditions, counts):
    """One dense batch with condition c repeated counts[c] times, in the order of `counts` (a list of (c, k)); every condition
    has the same number of nodes.  Works on any condition object whose tensor attributes are batch-first with batch 1."""
    first = conditions[counts[0][0]]
    out = first.get_new_object()
    for name, v in vars(first).items():
        if torch.is_tensor(v):
            setattr(out, name, torch.cat([getattr(conditions[c], name).expand(k, *getattr(conditions[c], name).shape[1:]) for c, k in counts], 0).contiguous())
    return out


# default bound on the pairs (B * n^2) of one sample_one_batch call: twice the benchmark batch (64 graphs of 300 nodes)
MAX_PAIRS_PER_BATCH = 2 * 64 * 300 * 300
MAX_CHAINS_PER_BATCH = 1024


def sample_conditions_sharded(conditions, n_samples: int, sample_for_condition: Callable, seed: int = 0, inpaint_node_idx=None,
                              inpaint_edge_idx=None, group=None, dst: int = 0, device=None, balance: bool = True,
                              max_pairs: Optional[int] = None, force_collective: bool = False):
    """BASELINE configs[2]: `n_samples` reverse-diffusion chains for each of len(conditions) conditions, the (condition, sample)
    units cut over the ranks (plan_condition_shards: by cost ~ n^2 when `balance`), no collective inside the chains, one final gather.

    conditions: list of single-graph dense conditions (PlaceHolder-like, X (1,n_c,kx) ...), known to every rank.
    sample_for_condition(dense_data=, n_samples=1, inpaint_node_idx=, inpaint_edge_idx=, chain_seeds=) -> object with collapsed
    .X (k,n) and .E (k,n,n): DiscreteDenoisingDiffusionRxn.sample_for_condition (diffusion/diffusion_rxn.py:107-128).
    A rank runs its chains in batches of EQUAL graph size -- several conditions of n nodes in one sample_one_batch call, each
    already repeated once per chain (a graph is never padded to a larger n: the denoiser's softmax has no key mask and its pooled
    statistics count padding, neuralnet/layers.py:254,15-47) -- of at most max_pairs pairs; chain_seeds = chain_seed(seed, global
    chain index), so a chain's draws depend neither on the number of ranks nor on its batch.
    Returns on rank `dst` (X uint8 (C*n_samples, n_max), E uint8 (C*n_samples, n_max, n_max)) in condition-major order, graphs
    smaller than n_max = the largest condition zero-padded; None on the other ranks.  force_collective: run the gather's all_gather
    even in a group of one rank (exercises the RCCL path on a single GPU)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sizes = [int(c.X.shape[1]) for c in conditions]
    n_max = max(sizes) if sizes else 0
    total = len(conditions) * n_samples
    plan = plan_condition_shards(sizes, n_samples, world, balance)
    mine = plan[rank]
    dev = torch.device(device) if device is not None else collective_device(group)
    cap = int(max_pairs) if max_pairs else MAX_PAIRS_PER_BATCH
    outX, outE = [], []
    k0 = 0
    while k0 < len(mine):
        n = sizes[mine[k0][0]]
        bmax = max(1, min(MAX_CHAINS_PER_BATCH, cap // max(1, n * n)))
        k1 = k0
        while k1 < len(mine) and k1 - k0 < bmax and sizes[mine[k1][0]] == n:
            k1 += 1
        batch = mine[k0:k1]
        counts = []
        for c, _ in batch:
            if counts and counts[-1][0] == c:
                counts[-1][1] += 1
            else:
                counts.append([c, 1])
        seeds = torch.tensor([chain_seed(seed, c * n_samples + s_) for c, s_ in batch], dtype=torch.int64)
        res = sample_for_condition(dense_data=_stack_conditions(conditions, counts), n_samples=1,
                                   inpaint_node_idx=[inpaint_node_idx[c] for c, _ in batch] if inpaint_node_idx is not None else None,
                                   inpaint_edge_idx=[inpaint_edge_idx[c] for c, _ in batch] if inpaint_edge_idx is not None else None,
                                   chain_seeds=seeds)
        X, E = res.X.to(device=dev, dtype=torch.uint8), res.E.to(device=dev, dtype=torch.uint8)
        pad = n_max - X.shape[1]
        if pad:
            X = torch.nn.functional.pad(X, (0, pad)); E = torch.nn.functional.pad(E, (0, pad, 0, pad))
        outX.append(X); outE.append(E)
        k0 = k1
    if outX:
        X, E = torch.cat(outX, 0).contiguous(), torch.cat(outE, 0).contiguous()
    else:
        X = torch.empty(0, n_max, dtype=torch.uint8, device=dev); E = torch.empty(0, n_max, n_max, dtype=torch.uint8, device=dev)
    if world == 1 and not (force_collective and dist.is_initialized()):
        Xs, Es = [X], [E]
    else:
        # pad every shard to the largest shard so one all_gather moves everything
        m = max(len(p) for p in plan)
        pad = lambda t: torch.cat([t, t.new_zeros((m - t.shape[0],) + tuple(t.shape[1:]))], 0) if t.shape[0] < m else t
        Xp, Ep = pad(X), pad(E)
        Xs = [torch.empty_like(Xp) for _ in range(world)]
        Es = [torch.empty_like(Ep) for _ in range(world)]
        dist.all_gather(Xs, Xp, group=group)
        dist.all_gather(Es, Ep, group=group)
        if rank != dst:
            return None
    # back to condition-major order: shard r's k-th chain sits at row r * m + k of the concatenated (padded) shards
    m = max((len(p) for p in plan), default=0)
    where = torch.empty(total, dtype=torch.int64)
    for r in range(world):
        for k, (c, s_) in enumerate(plan[r]):
            where[c * n_samples + s_] = r * m + k
    where = where.to(Xs[0].device)
    return torch.cat(list(Xs), 0).index_select(0, where), torch.cat(list(Es), 0).index_select(0, where)


def allreduce_flat_gradient(flat_grad: torch.Tensor, group=None, force: bool = False) -> int:
    """Data-parallel training (BASELINE configs[3]): SUM all-reduce of the flat fp32 gradient buffer of optim.FlatAdamW -- the whole
    32.7 MB align_absorbing gradient as ONE ring all-reduce (xGMI is per-link bound: few large transfers) -- in place; returns the
    world size (the optimiser divides by it: da_adamw's grad_scale).  Parameters the objective never reaches (mlp_out_y,
    self_attn.y_y, skip_scaling_3) keep zero entries on every rank, so no unused-parameter handling is needed.  The all-reduce is
    issued after backward, not overlapped with it: at 32.7 MB it is ~0.4 ms of a 30-45 ms step.  force: run the collective even in a
    group of one rank (exercises the RCCL path on a single GPU)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
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
