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
