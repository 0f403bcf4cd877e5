"""CPU, world_size 2 over gloo: the N>1 sharding path (unit partition, per-chain seeds, final gather)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from diffalign_amd.distributed import shard_range
    for total in (0, 1, 7, 100, 301):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in cuts]
            assert max(sizes) - min(sizes) <= 1


def _fake_sampler(n):
    from diffalign_amd.distributed import chain_seed

    def fn(units):
        X = torch.zeros(len(units), n, dtype=torch.uint8); E = torch.zeros(len(units), n, n, dtype=torch.uint8)
        for k, (c, s) in enumerate(units):
            g = torch.Generator().manual_seed(chain_seed(5, c * 1000 + s))
            X[k] = torch.randint(0, 29, (n,), generator=g, dtype=torch.uint8)
            e = torch.randint(0, 7, (n, n), generator=g, dtype=torch.uint8)
            E[k] = torch.triu(e, 1) + torch.triu(e, 1).T
        return X, E
    return fn


def _worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from diffalign_amd.distributed import sample_sharded
    res = sample_sharded(3, 5, 6, _fake_sampler(6))
    if rank == 0:
        torch.save(res, out)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_sampling_equals_single_rank(tmp_path):
    from diffalign_amd.distributed import sample_sharded
    ref = sample_sharded(3, 5, 6, _fake_sampler(6))           # world 1
    out = str(tmp_path / 'res.pt')
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    X, E = torch.load(out)
    assert torch.equal(X, ref[0]) and torch.equal(E, ref[1])


class _Cond:
    """Minimal dense condition for the sharded driver (CPU)."""
    def __init__(self, X, E):
        self.X, self.E = X, E

    def get_new_object(self):
        return _Cond(self.X.clone(), self.E.clone())


def _stub_sample_for_condition(dense_data, n_samples, inpaint_node_idx, inpaint_edge_idx, chain_seeds):
    """Stands in for DiscreteDenoisingDiffusionRxn.sample_for_condition on the CPU: 'samples' that are a deterministic function
    of the condition and of each chain's own counter-based noise (oracle.counter_exp_noise) -- as the real chains are.  The driver
    hands over a batch of equal-size conditions, each row already one chain (n_samples = 1), with per-row inpainting lists."""
    from types import SimpleNamespace
    from oracle import oracle as O
    k, n = dense_data.X.shape[:2]
    assert n_samples == 1 and chain_seeds.shape == (k,)
    qX, qE = O.counter_exp_noise(chain_seeds, 0, n, 29, 7)
    X = (1.0 / qX).argmax(-1) + dense_data.X.argmax(-1)                      # (k, n)
    E = (1.0 / qE).argmax(-1)
    E = torch.triu(E, 1); E = E + E.transpose(1, 2)
    if inpaint_node_idx is not None:
        assert len(inpaint_node_idx) == k
        for r, idx in enumerate(inpaint_node_idx):
            X[r, idx] = 99
    return SimpleNamespace(X=X, E=E)


def _conditions():
    g = torch.Generator().manual_seed(3)
    sizes = [6, 9, 4, 6, 9]
    return [_Cond(torch.nn.functional.one_hot(torch.randint(0, 29, (1, n), generator=g), 29).float(), torch.zeros(1, n, n, 7)) for n in sizes]


INPAINT = [[0], [1, 2], [], [3], []]


def _cond_worker(rank, world, port, out, n_cond, n_samples, balance):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from diffalign_amd.distributed import sample_conditions_sharded
    res = sample_conditions_sharded(_conditions()[:n_cond], n_samples, _stub_sample_for_condition, seed=11,
                                    inpaint_node_idx=INPAINT[:n_cond], balance=balance, max_pairs=4 * 81)
    if rank == 0:
        torch.save(res, out)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_condition_sampling_equals_single_rank(tmp_path):
    """BASELINE configs[2] driver: (condition, sample) units cut over 2 ranks mid-condition (equal counts, or by cost ~ n^2 with the
    large conditions first), equal-size conditions run as one batch, per-chain seeds, ragged condition sizes padded for the gather
    -> the same samples as one rank in condition-major order whatever the partition; and fewer chains than ranks (an empty shard)."""
    from diffalign_amd.distributed import sample_conditions_sharded
    sizes = [6, 9, 4, 6, 9]
    k = 0
    for n_cond, n_samples in [(5, 3), (3, 5), (1, 1)]:
        ref = sample_conditions_sharded(_conditions()[:n_cond], n_samples, _stub_sample_for_condition, seed=11,
                                        inpaint_node_idx=INPAINT[:n_cond], balance=False)
        assert ref[0].shape == (n_cond * n_samples, max(sizes[:n_cond])) and ref[0].dtype == torch.uint8
        one = sample_conditions_sharded(_conditions()[:n_cond], n_samples, _stub_sample_for_condition, seed=11,
                                        inpaint_node_idx=INPAINT[:n_cond], balance=True)
        assert torch.equal(one[0], ref[0]) and torch.equal(one[1], ref[1])          # one rank, size-ordered batches: same chains
        for balance in (False, True):
            out = str(tmp_path / f'c{k}.pt')
            port = 33500 + (os.getpid() % 2000) + k
            k += 1
            mp.spawn(_cond_worker, args=(2, port, out, n_cond, n_samples, balance), nprocs=2, join=True)
            X, E = torch.load(out)
            assert torch.equal(X, ref[0]) and torch.equal(E, ref[1])
    assert (ref[0][0, 0] == 99)


def test_cost_balanced_partition():
    """plan_condition_shards: every unit exactly once; with skewed sizes the cost-balanced cuts keep the largest shard's cost well
    below what equal-count condition-major cuts give; equal sizes stay adjacent (they run as one batch)."""
    from diffalign_amd.distributed import plan_condition_shards, balanced_cuts, unit_cost
    sizes = [300, 40, 40, 72, 72, 72, 150, 40]
    for world in (1, 2, 3, 8):
        for balance in (False, True):
            plan = plan_condition_shards(sizes, 7, world, balance)
            flat = [u for p in plan for u in p]
            assert sorted(flat) == [(c, s) for c in range(len(sizes)) for s in range(7)] and len(plan) == world
    cost = lambda p: sum(unit_cost(sizes[c]) for c, _ in p)
    bal, eq = plan_condition_shards(sizes, 7, 4, True), plan_condition_shards(sizes, 7, 4, False)
    assert max(map(cost, bal)) < 0.6 * max(map(cost, eq))
    assert max(map(cost, bal)) < 1.35 * sum(map(cost, bal)) / 4
    for p in bal:                                              # sizes are non-increasing inside a shard
        ns = [sizes[c] for c, _ in p]
        assert ns == sorted(ns, reverse=True)
    assert balanced_cuts([], 3) == [(0, 0), (0, 0), (0, 0)]
    assert balanced_cuts([1.0], 2) in ([(0, 1), (1, 1)], [(0, 0), (0, 1)])


def _grad_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from diffalign_amd.distributed import allreduce_gradients
    g = torch.Generator().manual_seed(100 + rank)
    grads = {'b.weight': torch.randn(7, 5, generator=g), 'a.bias': torch.randn(5, generator=g), 'c': torch.randn(1, generator=g),
             'd.weight': torch.randn(33, 9, generator=g)}
    res = allreduce_gradients(grads, bucket_bytes=160)           # several buckets, one tensor larger than a bucket
    assert res is grads
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_is_the_mean(tmp_path):
    from diffalign_amd.distributed import allreduce_gradients
    per_rank = []
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank)
        per_rank.append({'b.weight': torch.randn(7, 5, generator=g), 'a.bias': torch.randn(5, generator=g), 'c': torch.randn(1, generator=g),
                         'd.weight': torch.randn(33, 9, generator=g)})
    single = {k: v.clone() for k, v in per_rank[0].items()}
    assert all(torch.equal(v, per_rank[0][k]) for k, v in allreduce_gradients(single).items())   # world 1: unchanged
    out = str(tmp_path / 'g.pt')
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_grad_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    for k in per_rank[0]:
        assert torch.allclose(res[k], (per_rank[0][k] + per_rank[1][k]) / 2, rtol=1e-6, atol=1e-7), k


def _flat_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from diffalign_amd.distributed import allreduce_flat_gradient
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 3))      # same weights on every rank
    data = torch.Generator().manual_seed(300)
    x, y = torch.randn(8, 6, generator=data), torch.randn(8, 3, generator=data)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]                                  # this rank's half of the batch
    flat = torch.zeros(sum(p.numel() for p in net.parameters()))
    o = 0
    for p in net.parameters():                                                                       # optim.FlatAdamW's layout
        p.grad = flat[o:o + p.numel()].view_as(p); o += p.numel()
    ((net(xs) - ys) ** 2).mean().backward()
    w = allreduce_flat_gradient(flat)
    assert w == world
    if rank == 0:
        torch.save(flat / w, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_gradient_equals_the_whole_batch_gradient(tmp_path):
    """The data-parallel training step (TrainingMixin.train_one_step): SUM all-reduce of the flat gradient buffer, divided by the
    world size inside the optimiser, == the gradient of the mean loss over the concatenated batch (equal shards)."""
    from diffalign_amd.distributed import allreduce_flat_gradient
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 3))
    data = torch.Generator().manual_seed(300)
    x, y = torch.randn(8, 6, generator=data), torch.randn(8, 3, generator=data)
    ((net(x) - y) ** 2).mean().backward()
    want = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert allreduce_flat_gradient(want.clone()) == 1                                                # no process group: world 1, untouched
    out = str(tmp_path / 'f.pt')
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_flat_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-7)
