"""Training losses and running metrics (counterpart of models/losses.py and models/base.py).

Metric bookkeeping difference (by design): the reference issues two 4-byte
all-reduces plus a host ``.item()`` per metric update (base.py:16-38, ~24 per step).  Here updates
stay on the device as [sum, count] pairs; ``get_metrics()`` does ONE packed all-reduce.
"""
import torch
import torch.nn as nn

from . import runtime
from .geometry import batch_indexing, resize_flow2d

_GAMMA_WEIGHTS = {}


def _masked_mean_error(diff, mask, order):
    if order == 'l2-norm':
        err = torch.linalg.norm(diff, dim=1)
    elif order == 'l1':
        err = torch.sum(diff.abs(), dim=1)
    elif order == 'robust':
        err = torch.pow(diff.abs().sum(dim=1) + 0.01, 0.4)
    else:
        raise ValueError(order)
    # mean over the masked elements without boolean indexing (err[mask] costs a device->host sync
    # per loss term); identical value whenever the mask is not empty
    zero = torch.zeros((), dtype=err.dtype, device=err.device)
    return torch.where(mask, err, zero).sum() / mask.sum()


def _sequence_loss(flow_preds, target, cfgs, n_flow_channels):
    """sum_i gamma^(n-1-i) * mean error of prediction i (losses.py:64-119)."""
    n_preds = len(flow_preds)
    if target.shape[1] == n_flow_channels + 1:
        mask = target[:, n_flow_channels] > 0
    else:
        mask = torch.ones_like(target)[:, 0] > 0
    if (cfgs.order == 'l2-norm' and runtime.fused() and target.is_cuda and n_flow_channels in (2, 3)
            and not target.requires_grad and runtime.atomics_ok('sequence_loss')):
        # one kernel per iterate (camli_masked_l2_fwd/bwd) instead of nine pointwise / reduction launches
        from ..csrc import fused
        sums = fused.masked_l2_sums(list(flow_preds), target, n_flow_channels)
        key = (n_preds, float(cfgs.gamma), target.device)
        weights = _GAMMA_WEIGHTS.get(key)
        if weights is None:
            weights = _GAMMA_WEIGHTS[key] = torch.tensor([cfgs.gamma ** (n_preds - i - 1) for i in range(n_preds)],
                                                         dtype=torch.float32, device=target.device)
        return (sums * weights).sum() / mask.sum()
    if target.is_cuda and cfgs.order == 'l2-norm' and runtime.atomics_ok('sequence_loss'):
        runtime.fallback('sequence_loss', 'differentiable target or %d flow channels' % n_flow_channels)
    total = 0
    for i, pred in enumerate(flow_preds):
        loss = _masked_mean_error(pred - target[:, :n_flow_channels], mask, cfgs.order)
        total = total + cfgs.gamma ** (n_preds - i - 1) * loss
    return total


def calc_sequence_loss_2d(flow_preds, target, cfgs):
    return _sequence_loss(flow_preds, target, cfgs, 2)


def calc_sequence_loss_3d(flow_preds, target, cfgs):
    return _sequence_loss(flow_preds, target, cfgs, 3)


def calc_pyramid_loss_2d(flows, target, cfgs):
    """Weighted multi-level loss of the PWC variants (losses.py:5-33)."""
    assert len(flows) <= len(cfgs.level_weights)
    if cfgs.order not in ('robust', 'l2-norm'):
        raise NotImplementedError(cfgs.order)
    mask = target[:, 2] > 0 if target.shape[1] == 3 else torch.ones_like(target)[:, 0] > 0
    total = 0
    for pred, weight in zip(flows, cfgs.level_weights):
        assert pred.shape[1] == 2
        diff = torch.abs(resize_flow2d(pred, target.shape[2], target.shape[3]) - target[:, :2])
        total = total + weight * _masked_mean_error(diff, mask, cfgs.order)
    return total


def calc_pyramid_loss_3d(flows, target, cfgs, indices):
    """losses.py:36-61: the target is gathered at each level's FPS indices."""
    assert len(flows) <= len(cfgs.level_weights)
    if cfgs.order not in ('robust', 'l2-norm'):
        raise NotImplementedError(cfgs.order)
    total = 0
    for level, (flow, weight) in enumerate(zip(flows, cfgs.level_weights)):
        level_target = batch_indexing(target, indices[level])
        if level_target.shape[1] == 4:
            mask = level_target[:, 3, :] > 0
        else:
            mask = torch.ones_like(level_target)[:, 0, :] > 0
        total = total + weight * _masked_mean_error(flow - level_target[:, :3, :], mask, cfgs.order)
    return total


class FlowModel(nn.Module):
    """dict-in / dict-out flow model with a ``loss`` attribute and running metrics
    (base.py:6-94: ``get_loss``, ``get_metrics``, ``clear_metrics``, ``update_*_metrics``)."""

    def __init__(self):
        super().__init__()
        self.loss = None
        self.metrics = {}

    def clear_metrics(self):
        self.metrics = {}

    @torch.no_grad()
    def update_metrics(self, name, var, mask=None):
        """Accumulate [sum, count] of ``var`` (over ``mask`` if given) on the device.  No boolean
        indexing / ``.item()``: nothing here synchronises with the host, so a whole step stays
        capturable in a HIP graph."""
        var = var.float()
        if mask is None:
            if var.numel() == 0:
                return
            entry = torch.stack([var.sum(), var.new_full((), float(var.numel()))])
        else:
            keep = mask.to(var.dtype)
            entry = torch.stack([(var * keep).sum(), keep.sum()])
        self.metrics[name] = self.metrics[name] + entry if name in self.metrics else entry

    def get_metrics(self):
        if not self.metrics:
            return {}
        names = sorted(self.metrics)
        packed = torch.stack([self.metrics[n] for n in names])          # [n_metrics, 2]
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(packed)                        # one collective for everything
        packed = packed.cpu()
        return {n: (packed[i, 0] / packed[i, 1]).item() for i, n in enumerate(names) if packed[i, 1] > 0}   # empty masks are dropped (base.py:25-26)

    def get_loss(self):
        if self.loss is None:
            raise ValueError('Loss is empty.')
        return self.loss

    RANKED_BY = None        # the metric that ranks checkpoints ('epe2d' / 'epe3d'); lower is better

    @classmethod
    def is_better(cls, curr_metrics, best_metrics):
        if cls.RANKED_BY is None:
            raise RuntimeError('%s does not name the metric that ranks its checkpoints' % cls.__name__)
        return best_metrics is None or curr_metrics[cls.RANKED_BY] < best_metrics[cls.RANKED_BY]

    def supervise(self, inputs, final, terms, targets=None):
        """The supervised tail all models share (the reference repeats it in every forward: camliraft.py:70-92,
        camliraft_l.py:66-77, camlipwc.py:62-85, camlipwc_l.py, pwc.py, raft.py).  ``final`` maps 'flow_2d' / 'flow_3d'
        to the full-resolution predictions and is what forward returns.  When the inputs hold a ground truth for every
        entry, ``terms[key](target)`` is that modality's loss, ``self.loss`` their sum, and the running metrics take
        the per-modality losses, the end-point metrics and -- with an occlusion mask -- the non-occluded 3-D ones."""
        if any(key not in inputs for key in final):
            return final
        if targets is None:
            targets = {key: inputs[key].float() for key in final}
        parts = [(key, terms[key](targets[key])) for key in final]
        total = parts[0][1]
        for _, value in parts[1:]:
            total = total + value
        self.loss = total
        if len(parts) > 1:
            self.update_metrics('loss', total)
        for key, value in parts:
            self.update_metrics('loss' + key[len('flow_'):], value)
        if 'flow_2d' in final:
            self.update_2d_metrics(final['flow_2d'], targets['flow_2d'])
        if 'flow_3d' in final:
            self.update_3d_metrics(final['flow_3d'], targets['flow_3d'])
            # only the two fused models score the non-occluded subset (camliraft.py:95-96, camlipwc.py:97-98)
            if 'occ_mask_3d' in inputs and len(parts) > 1:
                self.update_3d_metrics(final['flow_3d'], targets['flow_3d'], inputs['occ_mask_3d'])
        return final

    @torch.no_grad()
    def update_2d_metrics(self, pred, target):
        if target.shape[1] == 3:
            mask, target = target[:, 2] > 0, target[:, :2]
        else:
            mask = torch.ones_like(target)[:, 0] > 0
        epe = torch.linalg.norm(pred - target, dim=1)
        self.update_metrics('epe2d', epe, mask)
        self.update_metrics('acc2d_1px', epe < 1.0, mask)
        mag = torch.linalg.norm(target, dim=1) + 1e-5
        self.update_metrics('outlier2d', torch.logical_and(epe > 3.0, epe / mag > 0.05), mask)

    @torch.no_grad()
    def update_3d_metrics(self, pred, target, occ_mask=None):
        if target.shape[1] == 4:
            mask, target = target[:, 3] > 0, target[:, :3]
        else:
            mask = torch.ones_like(target)[:, 0] > 0
        epe = torch.linalg.norm(pred - target, dim=1)
        acc = epe < 0.05
        if occ_mask is not None:
            mask = torch.logical_and(occ_mask == 0, mask)
            self.update_metrics('epe3d_noc', epe, mask)
            self.update_metrics('acc3d_5cm_noc', acc, mask)
        else:
            self.update_metrics('epe3d', epe, mask)
            self.update_metrics('acc3d_5cm', acc, mask)
