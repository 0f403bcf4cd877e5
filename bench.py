#!/usr/bin/env python3
"""Headline benchmark: reverse-diffusion graph-steps/sec (align_absorbing) on N MI355X GPUs of one node.

One "step" = one reverse step of sample_one_batch for one batch of padded reaction graphs resident in HBM:
denoiser forward (9-layer align_absorbing graph transformer, bf16 MFMA) + fused posterior / categorical draw /
apply_mask.  Workload = BASELINE.json configs[1]: batch 64 per GPU padded to MAX_ATOMS_RXN = 300 (synthetic 'full'
graphs, P = 142, random-init weights: no checkpoint/dataset exists offline).  Conditions x samples are sharded over
ranks with no collective in the data path (weak scaling); value = graphs * steps * ranks / max-over-ranks time.

  python bench.py --gpus N --steps K --warmup W         (N > 1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import math
import os
import sys
import time
from types import SimpleNamespace as ns

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

FULL = dict(n_layers=9, hidden_dims={'dx': 256, 'de': 64, 'dy': 64, 'n_head': 8, 'dim_ffX': 256, 'dim_ffE': 128, 'dim_ffy': 128},
            hidden_mlp_dims={'X': 256, 'E': 128, 'y': 128})
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def make_cfg(dims, T=500, T_eval=100):
    return ns(diffusion=ns(diffusion_steps=T, diffusion_steps_eval=T_eval, transition='absorbing_masknoedge',
                           mask_nodes='product_and_sn', mask_edges='product_and_sn', node_states_to_mask=['SuNo', 'none'],
                           edge_states_to_mask=['mol', 'within', 'across'], diffuse_nodes=True, diffuse_edges=True,
                           temperature_scaling_node=1.0, temperature_scaling_edge=1.0, classifier_free_guidance=False,
                           lambda_train=[5, 0], lambda_test=1, ce_lambda=2, denoiser='neuralnet', diffusion_noise_schedule='cosine'),
              neuralnet=ns(architecture='with_y_atommap_number_pos_enc', n_layers=dims['n_layers'], hidden_dims=dims['hidden_dims'],
                           hidden_mlp_dims=dims['hidden_mlp_dims'], dropout=0.2, improved=False, extra_features=True,
                           pos_encoding_type='laplacian_pos_enc', p_to_r_skip_connection=True, p_to_r_init=1.0,
                           alignment_type='correct', input_alignment=False, increase_y_dim_for_multigpu=True, use_ema=False,
                           pos_emb_permutations=0, num_lap_eig_vectors=20),
              train=ns(loss='ce'), test=ns(with_denoising=True, loss_0_repeat=1))


def make_info():
    from diffalign_amd.synthetic import ATOM_DECODER, BOND_DECODER
    kx = len(ATOM_DECODER)
    # fields of the extra-feature calculators (datasets/supernode_dataset.py:185-240); synthetic valencies / weights
    extra = dict(max_n_nodes=300, valencies=[0] + [(3 * i) % 5 + 1 for i in range(1, kx - 1)] + [0], bond_orders=[0, 1, 2, 3, 0, 0, 0],
                 atom_weights={a: (0.0 if i in (0, kx - 1) else 10.811 + 3.37 * i) for i, a in enumerate(ATOM_DECODER)}, max_weight=390)
    return ns(**extra, atom_decoder=list(ATOM_DECODER), bond_decoder=list(BOND_DECODER),
              input_dims={'X': len(ATOM_DECODER), 'E': len(BOND_DECODER), 'y': 1},
              output_dims={'X': len(ATOM_DECODER), 'E': len(BOND_DECODER), 'y': 0}, nodes_dist=None,
              node_types=torch.ones(len(ATOM_DECODER)), edge_types=torch.ones(len(BOND_DECODER)))


def edge_kernel_source_hash():
    """sha256 over the sources of k_edge_layer (the kernel file and the headers it includes), as stored beside the PMC numbers."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'diffalign_amd', 'csrc')
    for f in ('k_edge_layer.hip', 'mfma_util.h', 'common.h', 'edge_in_layout.h'):
        with open(os.path.join(csrc, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def cpu_baseline(seconds_budget=25.0):
    """The oracle (CPU restatement of the reference, oracle/oracle.py) timed on this host: same network, same step,
    a bounded sample of the same workload (graphs of the benchmark's own size n=300, P=144, one at a time, for about
    `seconds_budget` seconds of CPU work)."""
    from oracle import oracle as O
    from diffalign_amd import synthetic as S
    from diffalign_amd.transformer_model_with_y import GraphTransformerWithYAtomMapPosEmb
    import torch.nn as nn
    cores = min(os.cpu_count() or 1, int(os.environ.get('OMP_NUM_THREADS', '16')))
    torch.set_num_threads(cores)
    B, n = 1, 300
    torch.manual_seed(0)
    KX, KE = len(S.ATOM_DECODER), len(S.BOND_DECODER)
    m = GraphTransformerWithYAtomMapPosEmb(9, {'X': KX + 8, 'E': KE, 'y': 13}, FULL['hidden_mlp_dims'], FULL['hidden_dims'],
                                           {'X': KX, 'E': KE, 'y': 1}, nn.ReLU(), nn.ReLU(), p_to_r_skip_connection=True,
                                           p_to_r_init=1.0, alignment_type='correct')
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    true = S.make_reaction_batch(B, n, seed=1, P=(n - S.NB_DUMMY - 1) // 2)
    zt = S.noised_state(true, t=50, T=100, abs_x=S.IDX_AU, seed=2)
    g = torch.Generator().manual_seed(3)
    X = torch.cat([zt['X'], torch.rand(B, n, 8, generator=g)], -1)
    y = torch.cat([zt['y'], torch.rand(B, 12, generator=g)], -1)
    pos = S.shared_pos_encodings(true, 256, seed=4)
    tabs = O.absorbing_tables(100, KX, KE, 1, S.IDX_AU, 0, [KX - 1, 0], [4, 5, 6])
    mx, me = O.free_masks(true['X'], true['node_mask'], true['atom_map_numbers'], 'product_and_sn', S.IDX_SUNO)
    qX, qE = S.exp_noise((B, n, KX), 5), S.exp_noise((B, n, n, KE), 6)
    steps, t0 = 0, time.time()
    with torch.no_grad():
        while True:
            lX, lE, _ = O.denoiser_forward(sd, X, zt['E'], y, zt['node_mask'], 9, 8, (KX, KE, 1), zt['atom_map_numbers'], pos,
                                           zt['mol_assignments'], True)
            O.reverse_step(lX, lE, zt['X'], zt['E'], zt['node_mask'], tabs, 50, true['X'], true['E'], mx, me, qX, qE)
            steps += 1
            if time.time() - t0 > seconds_budget * 0.6 or steps >= 64:
                break
    dt = time.time() - t0
    return {'value': B * steps / dt, 'unit': 'graph-steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/oracle.py (fp32 torch CPU, {cores} threads), full align_absorbing dims, {steps} reverse step(s) of '
                      f'B={B} graph at n={n} (the graph size of the GPU workload) in {dt:.1f}s'}


def train_rooflines(B, n, dev, bf16, dx=256, de=64):
    """HBM rooflines of the training step's two largest own kernels, timed live with HIP events on tensors of the step's shapes
    (inside the captured step a launch cannot be bracketed): the attention core's backward and the weight gradient of a per-pair
    Linear (e_mul / e_add, 64 -> 256).  Algorithmic bytes = every operand the kernel must read or write once."""
    from diffalign_amd import _lib as L
    g = torch.Generator(device=dev).manual_seed(0)
    dt = torch.bfloat16 if bf16 else torch.float32
    eb = 2 if bf16 else 4
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    nm = torch.ones(B, n, dtype=torch.uint8, device=dev)
    Q, K, V, ye1, ye2 = r(B, n, dx), r(B, n, dx), r(B, n, dx), r(B, dx), r(B, dx)
    E1, E2 = r(B, n, n, dx).to(dt), r(B, n, n, dx).to(dt)
    Z = torch.empty_like(E1); wV = torch.empty_like(Q); sm = torch.empty_like(Q); sl = torch.empty_like(Q)
    lib = L.lib()
    fwd, bwd = (lib.da_attn_core_fwd_bf16, lib.da_attn_core_bwd_bf16) if bf16 else (lib.da_attn_core_fwd, lib.da_attn_core_bwd)
    S = L.stream()
    L.check(fwd(L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(E1), L.ptr(E2), L.ptr(nm), L.ptr(ye1), L.ptr(ye2), 0.177, L.ptr(Z), L.ptr(wV), L.ptr(sm),
                L.ptr(sl), B, n, dx, S), 'attn fwd')
    dZ, dwV = r(B, n, n, dx).to(dt), r(B, n, dx)
    ws = torch.empty(int(lib.da_attn_core_bwd_workspace(B, n, dx)), dtype=torch.uint8, device=dev)
    dQ, dK, dV, dE1, dE2, d1, d2 = (torch.empty_like(t) for t in (Q, Q, Q, E1, E1, ye1, ye1))
    f_bwd = lambda: L.check(bwd(L.ptr(Q), L.ptr(K), L.ptr(V), L.ptr(E1), L.ptr(E2), L.ptr(nm), L.ptr(ye2), 0.177, L.ptr(wV), L.ptr(sm), L.ptr(sl),
                                L.ptr(dZ), L.ptr(dwV), L.ptr(dQ), L.ptr(dK), L.ptr(dV), L.ptr(dE1), L.ptr(dE2), L.ptr(d1), L.ptr(d2), L.ptr(ws),
                                ws.numel(), B, n, dx, S), 'attn bwd')
    R = B * n * n
    x, dy = r(R, de).to(dt), r(R, dx).to(dt)
    dW, db = torch.empty(dx, de, device=dev), torch.empty(dx, device=dev)
    ws2 = torch.empty(int(lib.da_linear_wgrad_workspace(R, de, dx)), dtype=torch.uint8, device=dev)
    f_wg = lambda: L.check(lib.da_linear_wgrad_mixed(L.ptr(x), int(bf16), L.ptr(dy), int(bf16), L.ptr(dW), L.ptr(db), L.ptr(ws2), ws2.numel(), R, de,
                                                     dx, S), 'wgrad')

    def timed(fn, reps=20):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    out = []
    for name, fn, nbytes, what in (
            ('k_attn_bwd4 (+ k_attn_fold)', f_bwd, R * dx * eb * 5, 'E1, E2, dZ read, dE1, dE2 written once per pair-channel'),
            (('k_wgrad_partial_bf16<64,256>' if bf16 else 'k_wgrad_partial<64,256>') + ' (+ k_wgrad_fold)', f_wg, R * (de + dx) * eb, 'X and dY rows read once')):
        ms = timed(fn)
        out.append({'kernel': name, 'bound': 'hbm', 'achieved': nbytes / ms / 1e6, 'peak': 8000.0, 'unit': 'GB/s', 'frac': nbytes / ms / 1e6 / 8000.0,
                    'avg_launch_ms': ms, 'algorithmic_bytes_per_launch': nbytes, 'storage': 'bf16' if bf16 else 'f32', 'what': what, 'traffic': None})
    return out


def train_main(args, dev, dist, world, rank, emit=True, steps=None, warmup=None):
    """BASELINE.json configs[3]: data-parallel training steps of the align_absorbing denoiser (9 layers, 8.18M parameters, dropout
    0.2 as configs/neuralnet/default.yaml) on synthetic USPTO-like graphs; every rank holds its own batch, the only collective is the
    SUM all-reduce of the 32.7 MB flat gradient.  Reported as training graphs/s over all ranks; not the headline metric."""
    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd.optim import FlatAdamW
    from diffalign_amd import synthetic as S
    B, n = args.train_batch, args.train_nodes
    torch.manual_seed(0)
    cfg = make_cfg(FULL, T=500)
    cfg.train.loss = args.train_loss
    model = DiscreteDenoisingDiffusion(cfg, make_info(), device=dev).train()
    model.model.train_precision = args.train_precision
    true = S.make_reaction_batch(B, n, seed=1 + rank, P=None)
    data = PlaceHolder(**{k: v.to(dev) for k, v in true.items()})
    # the whole step as one HIP graph (eager is launch-bound: ~1,900 launches a step); with more than one rank only on request, since the
    # gradient all-reduce would then be captured too
    train_graph = args.graph == 'on' or (args.graph == 'auto' and world == 1)
    opt = FlatAdamW(model.model.parameters(), lr=2e-4, weight_decay=1e-12, amsgrad=True, device_step=train_graph)
    group = None
    step_fn = lambda: model.train_one_step(data, opt, group=group)
    if train_graph:                                    # the whole step replayed as one HIP graph (diffalign_amd/graphs.py)
        from diffalign_amd.graphs import GraphedTrainStep
        gstep = GraphedTrainStep(model, opt, data, group=group)
        step_fn = lambda: gstep(data, next_data=data)     # the next batch is known (a prefetching loader): its encodings overlap this step

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    steps = args.steps if steps is None else steps
    for _ in range(args.warmup if warmup is None else warmup):
        step_fn()
    sync()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(step_fn().clone())
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if not emit:
        return {'train_ms_per_step': dt / steps * 1e3, 'train_workload': f'B={B}, n={n}, loss {args.train_loss}, {args.train_precision} per-pair tensors (fp32 master weights), '
                                                                          f'AdamW(amsgrad); {steps} steps', 'train_graphs_per_s': B * steps / dt}
    if rank == 0:
        print(json.dumps({'metric': 'training graphs/sec per node (align_absorbing, data parallel)', 'value': B * steps * world / dt,
                          'unit': 'graphs/s', 'n_gpus': world, 'steps': steps, 'warmup': args.warmup, 'ms_per_step': dt / steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.train_precision == 'bf16' else 'f32', 'data': 'synthetic',
                          'config': {'workload': f'NOT the headline -- BASELINE.json configs[3]: align_absorbing training step, loss {args.train_loss}, '
                                                 f'batch {B}/GPU padded to n={n} (USPTO-like sizes), dropout 0.2, AdamW(amsgrad), extra features and '
                                                 'Laplacian position encodings computed on the device inside the step' , 'global_batch': B * world,
                                     'parallelism': f'dp{world} (one flat-gradient all-reduce per step)', 'hip_graph': train_graph},
                          'roofline': train_rooflines(B, n, dev, args.train_precision == 'bf16'),
                          'loss_first_last': [float(losses[0]), float(losses[-1])],
                          'peak_hbm_GB': torch.cuda.max_memory_allocated() / 1e9}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def loop_object(args, model, step, z_t, data, masks, dev):
    """What the headline leaves out of a real sampling loop, measured in the same process right after it: the same reverse step with
    the 8 + 12 extra features computed on the device inside it (as sample_one_batch runs it; the headline takes them as inputs, SURVEY
    8d), the feature set and its eigensolver alone, and one training step (configs[3])."""
    from diffalign_amd import extra_features as XF
    out = {}
    saved = model.extra_features_fn
    model.extra_features_fn = None
    z_t.eig_cache = XF.EigCache.from_free_masks(masks[1], data.node_mask)
    evs, orig = [], XF._eigen_into

    def timed_eigen(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record(); evs.append((e0, e1))
        return r

    try:
        with torch.no_grad():
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            out['ms_per_step'] = (time.perf_counter() - t0) / args.steps * 1e3
            out['graph_steps_per_s'] = z_t.X.shape[0] * 1e3 / out['ms_per_step']
            XF._eigen_into = timed_eigen
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model.compute_extra_data(z_t)
            torch.cuda.synchronize()
            out['extra_features_ms'] = (time.perf_counter() - t0) / args.steps * 1e3
            out['eigensolver_ms'] = sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))
            out['eigensolver_share_of_step'] = out['eigensolver_ms'] / out['ms_per_step']
    finally:
        XF._eigen_into = orig
        model.extra_features_fn = saved
        z_t.eig_cache = None
    out['what'] = ('ms_per_step: the headline step with --extra-features device (features + eigensolver inside the step, fixed block kept '
                   'along the chain); extra_features_ms / eigensolver_ms: the whole feature set / k_extra_eig_solve alone, back to back')
    return out


TABLE_PREFETCH = os.environ.get('DA_TABLE_PREFETCH', '1') != '0'   # A/B switch: per-step transition tables as slices of tables laid out once (reverse_step t_int=)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--nodes', type=int, default=300)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--extra-features', choices=['inputs', 'device'], default='inputs',
                    help="'inputs' (default, SURVEY §8d: the 8 + 12 extra features are given); 'device': computed inside the step "
                         '(csrc/k_extra.hip + the batched partial eigensolver of csrc/k_extra_eig.hip), as a real sampling loop does')
    ap.add_argument('--workload', choices=['full', 'uspto_like', 'train'], default='full',
                    help="'full' (default, the headline: every graph has 299 real nodes, BASELINE.json configs[1]); 'uspto_like': product "
                         'sizes P ~ clip(N(26,10), 5, 142) padded to --nodes (SURVEY 8d), i.e. what a real USPTO batch looks like: most '
                         'pairs are padding and take the edge kernel\'s dead-tile path; '
                         "'train': NOT the headline -- BASELINE.json configs[3], one data-parallel training step (fp32 autograd path, ELBO "
                         "loss, flat-gradient all-reduce, fused AdamW) on --train-batch graphs per GPU padded to --train-nodes")
    ap.add_argument('--train-batch', type=int, default=16)
    ap.add_argument('--train-nodes', type=int, default=140)
    ap.add_argument('--train-loss', choices=['ce', 'vb', 'vbce'], default='vb')
    ap.add_argument('--train-precision', choices=['fp32', 'bf16'], default='bf16',
                    help="'bf16' (default): mixed precision for the per-pair tensors, as the reference's fp16 autocast; 'fp32': everything fp32")
    ap.add_argument('--force-dist', action='store_true',
                    help='with --gpus 1: create a world-size-1 nccl (RCCL) process group anyway and take the multi-GPU branch (barrier, final '
                         'gather inside the timed region, MAX all-reduce of the time) -- a single-GPU rehearsal of what every rank of an N-GPU run does')
    ap.add_argument('--no-loop', action='store_true', help="skip the 'loop' object (the same step with the extra features computed on the device, "
                                                           'the eigensolver alone, one training step), measured after the headline')
    ap.add_argument('--graph', choices=['auto', 'on', 'off'], default='auto', nargs='?', const='on',
                    help="replay every reverse step as a HIP graph (diffalign_amd/graphs.py: GraphedChain, what sample_one_batch does for small "
                         "batches).  'auto' (default): as the sampling loop decides, i.e. for batches of at most graphs.CHAIN_GRAPH_MAX_PAIRS pairs -- "
                         "not at the headline size, where the edge kernels hide the host; the training workload: 'on' captures the whole step")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU, BEFORE anything touches the GPU in this
        # process (the ranks are children; this process only waits and hands on their exit code and rank 0's JSON line)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                         f'(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_
        dist = dist_
        # RCCL writes its version banner to stdout when the communicator is created (at init with device_id, or at the first collective):
        # both happen here with stdout pointing at stderr, so that the JSON line stays the only thing this program writes to stdout
        sys.stdout.flush()
        keep_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if 'MASTER_ADDR' in os.environ and 'RANK' in os.environ:
                dist.init_process_group('nccl', device_id=dev)
            else:                                              # --force-dist outside a launcher: a group of this one process
                import socket
                with socket.socket() as sk:
                    sk.bind(('127.0.0.1', 0))
                    port = sk.getsockname()[1]
                dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep_fd, 1)
            os.close(keep_fd)

    from diffalign_amd.diffusion_abstract import DiscreteDenoisingDiffusion
    from diffalign_amd.placeholder import PlaceHolder
    from diffalign_amd import synthetic as S, helpers, graph, engine

    if args.workload == 'train':
        return train_main(args, dev, dist, world, rank)
    B, n = args.batch, args.nodes
    torch.manual_seed(0)                                      # weights: default nn.Linear/LayerNorm init, seed 0
    model = DiscreteDenoisingDiffusion(make_cfg(FULL), make_info(), device=dev).eval()
    P = (n - S.NB_DUMMY - 1) // 2 if args.workload == 'full' else None
    true = S.make_reaction_batch(B, n, seed=1 + rank, P=P)    # this rank's shard of conditions x samples
    data = PlaceHolder(**{k: v.to(dev) for k, v in true.items()})
    pos = S.shared_pos_encodings(true, 256, seed=3).to(dev)
    model.pos_encodings_fn = lambda z: pos
    g = torch.Generator().manual_seed(7)
    xe, ye = torch.rand(B, n, 8, generator=g).to(dev), torch.rand(B, 12, generator=g).to(dev)
    if args.extra_features == 'inputs':
        model.extra_features_fn = lambda z: (xe, ye)           # extra features are inputs
    kx, ke = data.X.shape[-1], data.E.shape[-1]
    T_eval = 100
    zt0 = S.noised_state(true, t=50, T=T_eval, abs_x=S.IDX_AU, seed=11 + rank)
    z_t = data.get_new_object(X=zt0['X'].to(dev), E=zt0['E'].to(dev), y=zt0['y'].to(dev))
    masks = graph.get_node_masks(data, model.dataset_info.atom_decoder, model.dataset_info.bond_decoder, 'product_and_sn',
                                 'product_and_sn')
    tm = model.transition_model_eval
    if args.extra_features == 'device':
        # as sample_one_batch: the fixed side's share of the features' eigensolver is kept along the chain (DA_EIG_CACHE=0: off)
        from diffalign_amd.extra_features import EigCache
        z_t.eig_cache = EigCache.from_free_masks(masks[1], data.node_mask)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    state = {'t': 50}

    # as sample_one_batch: one seed per chain from the generator, step numbers and y inputs laid out once
    seeds = torch.randint(0, 2 ** 62, (B,), device=dev, generator=gen)
    all_t = torch.arange(51, device=dev)[:, None, None].expand(-1, B, 1).contiguous()
    all_y = all_t.float() * 5
    state['call'] = 0
    holders = [(torch.empty(B, n, kx, device=dev), data.E.float().clone()) for _ in range(2)]
    # per chain, as sample_one_batch: the last layer computes the rows of free nodes only (DA_LAST_ROWS=0: A/B switch)
    last_rows = None
    if os.environ.get('DA_LAST_ROWS', '1') == 'list':            # the row list alone, every tile of those rows in full
        last_rows = graph.free_rows(masks)
    elif os.environ.get('DA_LAST_ROWS', '1') != '0':
        last_rows = graph.last_layers_plan(masks)

    def step():
        t = state['t']
        t_arr = all_t[t]
        z_t.y = all_y[t]
        state['call'] += 1
        # what sample_one_batch runs per step: denoiser forward with the edge half of posterior / draw / apply_mask drawn inside its
        # output kernel (per-chain counter noise, seeds from `gen`), the node half behind the node output path
        # (as sample_one_batch: the state alternates between two holders that start as the condition, so the pairs apply_mask keeps fixed
        # are not written again)
        out = holders[1] if z_t.E.data_ptr() == holders[0][1].data_ptr() else holders[0]
        zsX, zsE, _ = model.reverse_step(z_t, t_arr, pos_encodings=pos, transition_model=tm, orig=data, node_masks=masks, chain_seeds=seeds, call=state['call'],
                                        t_int=t if TABLE_PREFETCH else None, out=out, out_holds_condition=True, last_rows=last_rows)
        z_t.X, z_t.E = zsX, zsE
        state['t'] = t - 1 if t > 1 else 50

    from diffalign_amd import graphs as G
    use_graph = args.graph == 'on' or (args.graph == 'auto' and G.chain_graphs_wanted(B, n))
    eager_step = step
    if use_graph:
        # as sample_one_batch does for a batch of this size: every step a HIP graph of the same fused, three-stream work; the steps the
        # timed region will replay are captured (once, untimed) before it starts
        gc = G.GraphedChain.get(model, data, masks, pos, (1.0, 1.0), None, None, args.extra_features == 'device', 1)
        cache = getattr(z_t, 'eig_cache', None)
        gc.T_eval = T_eval
        pad = torch.randn(B, n, device=dev)
        gc.load(data, masks, pos, None, None, seeds, z_t, cache, pad, t_start=50)

        def step():
            t = state['t']
            g = gc.graphs.get(t)
            if g is None:
                gc.run_steps(t, t)
            else:
                g.replay()
            state['t'] = t - 1 if t > 1 else 50

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        if use_graph:
            for _ in range(2):                                # the edge kernel's live timing: two eager steps (a graph has no events inside)
                eager_step()
            engine.PROFILE_EVENTS = []
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            ev_eager = engine.PROFILE_EVENTS
            engine.PROFILE_EVENTS = None
            state['t'] = 50
            for _ in range(args.warmup + args.steps):         # capture the steps the timed region replays
                step()
            state['t'] = 50
        for _ in range(args.warmup):
            step()
        engine.PROFILE_EVENTS = [] if not use_graph else None
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if dist is not None:
            # the one collective of sharded sampling (diffalign_amd/distributed.py): the final gather of the collapsed uint8
            # results, n + n*n bytes per chain, inside the timed region
            from diffalign_amd.distributed import gather_collapsed
            gather_collapsed(z_t.X.argmax(-1).to(torch.uint8), z_t.E.argmax(-1).to(torch.uint8))
        sync()
        dt = time.perf_counter() - t0
    ev = engine.PROFILE_EVENTS if not use_graph else ev_eager
    engine.PROFILE_EVENTS = None
    ev_ms = [(a.elapsed_time(b), share) for a, b, share in ev]
    edge_ms = sum(m for m, _ in ev_ms) / max(1, len(ev_ms))              # mean over ALL launches (what rocprofv3 --stats averages)
    # mean share of a full launch's algorithmic work per launch (the last two layers of a fused step leave out what nobody reads:
    # graph.LastLayersPlan.share_prev / share_last)
    edge_share = sum(sh for _, sh in ev_ms) / max(1, len(ev_ms))
    full_ms = [m for m, sh in ev_ms if sh == 1.0]
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = B * args.steps * world / dt

    # roofline of the dominant kernel (k_edge_layer, one launch per layer): algorithmic bytes = each 64-wide bf16 edge
    # row read once + written once; algorithmic FLOPs = 131,072 per pair
    pairs = B * n * n
    # (the mean launch keeps `edge_share` of that: above)
    bytes_launch = pairs * 64 * 2 * 2 * edge_share
    flops_launch = pairs * 131072.0 * edge_share
    t_s = edge_ms * 1e-3
    tf = flops_launch / t_s / 1e12 if t_s > 0 else 0.0
    gbs = bytes_launch / t_s / 1e9 if t_s > 0 else 0.0
    # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic_bench.sh + tools/traffic_json.py; FETCH_SIZE doubled per the
    # gfx950 note in MI355X_MICROARCH.md).  Valid only for the workload AND the kernel source they were taken on: the file carries the
    # sha256 of the kernel's sources, and a different hash (or workload) gives traffic = null instead of a stale number.
    traffic, traffic_note = None, None
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        with open(os.path.join(here, 'profiles', 'r03', 'edge_layer_traffic.json')) as f:
            tj = json.load(f)
        if tj.get('kernel_source_sha256') != edge_kernel_source_hash():
            traffic_note = 'profiles/r03/edge_layer_traffic.json was measured on a different k_edge_layer source: not reported'
        elif tj['workload'] == {'B': B, 'n': n} and args.workload == 'full' and engine.TABLE_IN == bool(tj.get('table_input_first_layer')):
            nl = len(model.model.tf_layers)
            first = tj['table_input_first_layer']['hbm_bytes_per_launch']
            rest = tj['with_stats_tail' if engine.use_stats_tail(n) else 'bare']['hbm_bytes_per_launch']
            traffic = (first + (nl - 1) * rest) / nl      # mean over the step's launches
    except (OSError, KeyError, ValueError, TypeError):
        pass
    roof = {'kernel': 'k_edge_layer<64,256,128>', 'bound': 'mfma', 'achieved': tf, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': tf / MFMA_BF16_PEAK_TFLOPS, 'traffic': traffic, 'traffic_unit': 'bytes/launch (PMC)',
            'algorithmic_bytes_per_launch': bytes_launch, 'algorithmic_flops_per_launch': flops_launch, 'avg_launch_ms': edge_ms,
            'launches_timed': len(ev), 'mean_share_of_full_launch_work': edge_share,
            'avg_full_launch_ms': (sum(full_ms) / len(full_ms)) if full_ms else None,
            'hbm_algorithmic_GBps': gbs, 'hbm_frac_of_8TBps': gbs / HBM_PEAK_GBS}
    if traffic_note:
        roof['traffic_note'] = traffic_note
    if engine.use_stats_tail(n):
        roof['note'] = ('launches of layers >= 1 include the Etoy statistics tail (da_edge_layer_stats: early-finishing waves stream '
                        'E_in a second time, +0.74 GB and +0.06 to +0.12 ms per launch) that replaces a separate 0.155 ms pass; '
                        'DA_STATS_TAIL=0 times the bare kernel')
    if args.workload != 'full':
        roof['note_padding'] = ('achieved / frac count the algorithmic FLOPs of every padded pair, while the kernel leaves tiles without one real pair '
                                'out of its attention part: on this workload frac is not a utilisation figure (the headline workload has no padding)')
    if rank == 0:
        out = {'metric': 'reverse-diffusion graph-steps/sec per node (align_absorbing)', 'value': value, 'unit': 'graph-steps/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
               'config': {'workload': (f'align_absorbing denoiser forward + posterior/sample/mask, batch {B}/GPU padded to n={n} '
                                       '(BASELINE.json configs[1]), 9 layers, 8.18M params, random-init weights' if args.workload == 'full' else
                                       f'NOT the headline -- USPTO-like product sizes (P ~ N(26,10), {float(true["node_mask"].float().mean()):.0%} real '
                                       f'nodes) padded to n={n}, batch {B}/GPU, same network and step'),
                          'extra_features': 'given as inputs (SURVEY 8d)' if args.extra_features == 'inputs' else 'computed on the device inside the step',
                          'global_batch': B * world, 'n_nodes': n, 'parallelism': f'shard{world} (conditions x samples, no collective)',
                          'hip_graph_per_step': bool(use_graph),
                          'dead_work_elided': ('as sample_one_batch: with the product side fixed by the mask, the last layer computes the query rows of the '
                                               'free nodes only, the last two layers form the edge features somebody reads only, and the fixed '
                                               'pairs of z_s are not rewritten -- z_s is bit-identical (tests/test_gpu_fused_step.py); '
                                               'DA_LAST_ROWS=0 computes everything, =list the row list alone') if last_rows is not None and not use_graph else None},
               'roofline': roof}
        if world == 1 and not args.no_loop and args.extra_features == 'inputs' and not use_graph:
            out['loop'] = loop_object(args, model, step, z_t, data, masks, dev)
            del model, z_t
            torch.cuda.empty_cache()
            try:
                out['loop'].update(train_main(args, dev, None, 1, 0, emit=False, steps=3, warmup=2))
            except Exception as e:                             # the training leg must never take the headline line down with it
                out['loop']['train_error'] = f'{type(e).__name__}: {e}'
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
