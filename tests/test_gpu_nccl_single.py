"""The RCCL branch of the multi-GPU path, executed on ONE GPU: a world-size-1 'nccl' process group on cuda:0 (two ranks on one GPU are
not a supported RCCL layout).  The collectives of diffalign_amd/distributed.py run on device tensors, the sharded condition sampler runs
the real (tiny) HIP model under the group, and everything is compared bit for bit with the group-less result.  The N > 1 logic itself
(partition, seeds, gather order) is covered on the CPU by tests/test_distributed_gloo.py (gloo, world size 2)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
D = 'cuda:0'


@pytest.fixture(scope='module')
def nccl_group():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device(D))
    yield dist.group.WORLD
    dist.destroy_process_group()


def test_collectives_on_device_tensors(nccl_group):
    from diffalign_amd.distributed import gather_collapsed, allreduce_flat_gradient, allreduce_gradients, collective_device
    assert collective_device().type == 'cuda'
    g = torch.Generator(device=D).manual_seed(5)
    X = torch.randint(0, 29, (7, 40), device=D, generator=g).to(torch.uint8)
    E = torch.randint(0, 7, (7, 40, 40), device=D, generator=g).to(torch.uint8)
    Xs, Es = gather_collapsed(X, E)                               # the final gather of sharded sampling (bench.py's timed region)
    assert len(Xs) == 1 and torch.equal(Xs[0], X) and torch.equal(Es[0], E) and Xs[0].data_ptr() != X.data_ptr()
    flat = torch.randn(8_177_192, device=D, generator=g)          # the align_absorbing gradient: 32.7 MB, one all-reduce
    want = flat.clone()
    assert allreduce_flat_gradient(flat) == 1 and torch.equal(flat, want)              # shortcut: no collective in a group of one
    assert allreduce_flat_gradient(flat, force=True) == 1 and torch.equal(flat, want)  # RCCL all_reduce(SUM) over one rank
    grads = {'a': torch.randn(33, 9, device=D, generator=g), 'b': torch.randn(5, device=D, generator=g)}
    keep = {k: v.clone() for k, v in grads.items()}
    tt = torch.tensor([1.25], device=D, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)                     # bench.py's max-over-ranks of the step time
    dist.barrier()
    assert float(tt) == 1.25 and all(torch.equal(allreduce_gradients(grads)[k], keep[k]) for k in keep)


def test_sharded_condition_sampling_under_the_group_equals_groupless(nccl_group):
    """sample_conditions_sharded with the real tiny HIP model: (i) under the nccl group with the gather's all_gather forced == without
    any collective; (ii) conditions of EQUAL size run as one batch == one sample_one_batch call per condition, bit for bit (per-chain
    counter noise; no graph is ever padded to a larger n)."""
    from tests.test_gpu_graph_drivers import _tiny_model, _ph
    from diffalign_amd import synthetic as S
    from diffalign_amd.distributed import sample_conditions_sharded
    z, model = _tiny_model()
    model.cfg.neuralnet.pos_encoding_type = 'smiles_pos_enc'
    conds = []
    for seed, n in ((91, 24), (92, 30), (93, 24), (94, 24), (95, 30)):
        b = S.make_reaction_batch(1, n, seed=seed, P=(n - S.NB_DUMMY - 1) // 2)
        conds.append(_ph(b['X'], b['E'], b['node_mask'], b['atom_map_numbers'], b['mol_assignments']))
    kw = dict(seed=3, inpaint_node_idx=[[0], [], [1, 2], [], []], inpaint_edge_idx=[[], [[0, 1]], [], [], []])
    calls = []
    def sfc(**a):
        calls.append(int(a['dense_data'].X.shape[0]))
        return model.sample_for_condition(**a)
    X0, E0 = sample_conditions_sharded(conds, 3, sfc, **kw)                               # batches of equal size, no collective
    assert sorted(calls) == [6, 9] and X0.shape == (15, 30) and X0.dtype == torch.uint8 and X0.device.type == 'cuda'
    X1, E1 = sample_conditions_sharded(conds, 3, model.sample_for_condition, force_collective=True, **kw)   # RCCL all_gather
    assert torch.equal(X0, X1) and torch.equal(E0, E1)
    calls.clear()
    X2, E2 = sample_conditions_sharded(conds, 3, sfc, balance=False, max_pairs=3 * 24 * 24, **kw)   # at most one condition per call
    assert calls == [3, 1, 1, 1, 3, 3, 1, 1, 1]
    assert torch.equal(X0, X2) and torch.equal(E0, E2)
    assert torch.equal(X0[0:3, 0].cpu(), conds[0].X[0, 0].argmax(-1).cpu().expand(3).to(torch.uint8))   # inpainted node kept
    assert not torch.equal(X0[0], X0[1])
