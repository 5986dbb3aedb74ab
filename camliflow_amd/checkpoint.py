"""Checkpoint save / resume and model-config loading in the reference's formats.

* checkpoints: the dict ``{'last_epoch', 'state_dict', 'best_metrics'}`` that ``train.py:228-247`` writes with
  ``torch.save`` and reads back with a STRICT ``load_state_dict`` -- the cores keep the reference's module and
  parameter names, so its published checkpoints load unchanged.  Like the reference, optimizer / scaler / scheduler
  state is not part of the file (the scheduler is re-stepped from the epoch, train.py:111).
* configs: ``conf/model/*.yaml`` (Hydra group files, ``# @package _global_`` with one top-level ``model:`` key) read
  with plain PyYAML into the attribute tree the model classes consume (``cfgs.n_iters_train``,
  ``cfgs.pwc2d.max_displacement`` ...); ``override`` merges a dict of dotted keys the way ``utils.override_cfgs`` does.
"""
import os
from types import SimpleNamespace

import torch


def save_ckpt(model, filepath, last_epoch, best_metrics=None):
    """train.py:228-238 (rank-0 side): the model may be wrapped (``.module``) as under DistributedDataParallel."""
    module = getattr(model, 'module', model)
    os.makedirs(os.path.dirname(os.path.abspath(filepath)), exist_ok=True)
    torch.save({'last_epoch': int(last_epoch), 'state_dict': module.state_dict(), 'best_metrics': best_metrics}, filepath)
    return filepath


def load_ckpt(model, filepath, resume=True, map_location='cpu', trust_pickle=False):
    """train.py:240-247: strict load; returns (epoch to continue from, best_metrics).  Not resuming leaves the trainer's
    initial state untouched, i.e. (1, None): ``curr_epoch`` starts at 1 (train.py:39) and only a resume overwrites it.

    The file is a plain dict of tensors, ints and a metrics dict, so it is read with ``weights_only=True`` -- published
    checkpoints are untrusted input and a full unpickle executes whatever they contain.  ``trust_pickle=True`` opts into
    the full unpickler for a file whose ``best_metrics`` holds objects outside that allow-list."""
    checkpoint = torch.load(filepath, map_location=map_location, weights_only=not trust_pickle)
    module = getattr(model, 'module', model)
    module.load_state_dict(checkpoint['state_dict'], strict=True)
    if resume:
        return checkpoint['last_epoch'] + 1, checkpoint.get('best_metrics')
    return 1, None


def _to_namespace(node):
    if isinstance(node, dict):
        return SimpleNamespace(**{k: _to_namespace(v) for k, v in node.items()})
    if isinstance(node, list):
        return [_to_namespace(v) for v in node]
    return node


def load_model_config(path, override=None):
    """conf/model/<name>.yaml -> attribute tree of its ``model`` section.  ``override``: {'dotted.key': value}."""
    import yaml
    with open(path) as f:
        tree = yaml.safe_load(f)
    tree = tree.get('model', tree)
    for dotted, value in (override or {}).items():
        node = tree
        keys = dotted.split('.')
        for k in keys[:-1]:
            node = node.setdefault(k, {})
        node[keys[-1]] = value
    return _to_namespace(tree)


def model_from_config(path, override=None):
    """yaml file -> model instance (factory.py:21-35 dispatch on ``model.name``)."""
    from .cores import model_factory
    return model_factory(load_model_config(path, override))
