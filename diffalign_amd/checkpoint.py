"""Training checkpoints in the reference's layout (training/helpers.py:99-110 assemble_checkpoint_state; utils/setup.py:300-348
load_weights / load_all_state_dicts): one dict with `model_state_dict`, `optimizer_state_dict`, `scheduler_state_dict`,
`scaler_state_dict`, `rng_state` and, with cfg.neuralnet.use_ema, `ema_state_dict`.  `optim.FlatAdamW` writes and reads
torch.optim.AdamW's state format, `ema.EMA` the reference class's keys, so a run can move between the two code bases at a checkpoint.

Differences, by construction: there is no GradScaler (the training forward is fp32; `scaler` may be None and `scaler_state_dict` is
`{}`, which the reference's loader accepts, utils/setup.py:338); nn.DataParallel does not exist here (one process per GPU), so
`module.` prefixes are only ever stripped; files are read with `weights_only=True` (tensors and plain containers, nothing executed)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch


def check_if_dataparallel_dict(state_dict) -> bool:
    """utils/setup.py:300-304 (the LAST key: an EMA dict starts with 'initted')."""
    return 'module' in list(state_dict.keys())[-1]


def dataparallel_dict_to_regular_state_dict(state_dict):
    """utils/setup.py:306-310."""
    return OrderedDict((k.replace('module.', ''), v) for k, v in state_dict.items())


def load_weights(model, model_state_dict, device_count: Optional[int] = 1):
    """utils/setup.py:319-329 for device_count <= 1 semantics on every rank (each process holds the plain module)."""
    if check_if_dataparallel_dict(model_state_dict):
        model_state_dict = dataparallel_dict_to_regular_state_dict(model_state_dict)
    own = model.state_dict()
    # buffers this build does not keep as module state (transition tables are rebuilt from cfg) are ignored; every key this model
    # has must be present
    missing = [k for k in own if k not in model_state_dict]
    if missing:
        raise KeyError(f'checkpoint lacks {len(missing)} keys of the model, e.g. {missing[:3]}')
    model.load_state_dict({k: v for k, v in model_state_dict.items() if k in own})
    return model


def assemble_checkpoint_state(model, optimizer, scheduler=None, scaler=None, cfg=None) -> dict:
    """training/helpers.py:99-110."""
    cfg = cfg if cfg is not None else model.cfg
    states = {'model_state_dict': model.state_dict(),
              'optimizer_state_dict': optimizer.state_dict(),
              'scheduler_state_dict': scheduler.state_dict() if scheduler is not None else {},
              'scaler_state_dict': scaler.state_dict() if scaler is not None else {},
              'rng_state': torch.get_rng_state()}
    if cfg.neuralnet.use_ema:
        states['ema_state_dict'] = model.ema.state_dict()
    return states


def save_checkpoint(path, model, optimizer, scheduler=None, scaler=None, cfg=None):
    torch.save(assemble_checkpoint_state(model, optimizer, scheduler, scaler, cfg), path)


def load_all_state_dicts(cfg, model, optimizer, lr_scheduler, scaler, checkpoint_file, device_count: Optional[int] = 1, restore_rng=False):
    """utils/setup.py:331-348, including the legacy form (a bare model state dict)."""
    checkpoint = torch.load(checkpoint_file, map_location='cpu', weights_only=True)
    if 'model_state_dict' not in checkpoint:
        load_weights(model, checkpoint, device_count=device_count)
        return
    load_weights(model, checkpoint['model_state_dict'], device_count=device_count)
    optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
    if lr_scheduler is not None and checkpoint.get('scheduler_state_dict'):
        lr_scheduler.load_state_dict(checkpoint['scheduler_state_dict'])
    if scaler is not None and checkpoint.get('scaler_state_dict'):
        scaler.load_state_dict(checkpoint['scaler_state_dict'])
    if 'ema_state_dict' in checkpoint and getattr(model, 'ema', None) is not None:
        ema_sd = checkpoint['ema_state_dict']
        if check_if_dataparallel_dict(ema_sd):
            ema_sd = dataparallel_dict_to_regular_state_dict(ema_sd)
        model.ema.load_state_dict(ema_sd)
    if restore_rng and 'rng_state' in checkpoint:
        torch.set_rng_state(checkpoint['rng_state'])
    from . import _lib as L
    L.weights_epoch += 1                                      # packed inference weights are stale
