"""Model-level golden vectors: the REFERENCE's model code (models/*.py, unchanged) driven by
operators with the native kernels' index semantics (tests/refmodels.py substitutes the oracle-backed
ops for models.csrc -- the "load unchanged" arrangement of SURVEY 8b), deterministic name-hashed
weights, seeded synthetic inputs.  Stored: final flows, loss, a gradient fingerprint.

Run in the build container only:  python tests/golden/make_model_golden.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import refmodels  # noqa: E402

refmodels.install(native_semantics=True)
from modelutils import MODEL_CASES, grad_fingerprint, hashed_fill_, synthetic_inputs  # noqa: E402


def main():
    torch.set_num_threads(8)
    for name, (module, cls, cfg_fn, shape) in MODEL_CASES.items():
        ref_cls = getattr(importlib.import_module('models.' + module), cls)
        torch.manual_seed(0)
        model = hashed_fill_(ref_cls(cfg_fn()), scale=0.5)
        inputs = synthetic_inputs(*shape)
        out = {}
        for mode in ('eval', 'train'):
            getattr(model, mode)()
            model.zero_grad()
            res = model(inputs)
            loss = model.get_loss()
            for k, v in res.items():
                out['%s_%s' % (mode, k)] = v.detach().numpy()
            out['%s_loss' % mode] = np.float64(loss.item())
            if mode == 'train':
                loss.backward()
                names, values = grad_fingerprint(model)
                out['grad_names'] = np.array(names)
                out['grad_norms'] = values
        out['input_checksum'] = np.array([float(v.double().sum()) for v in inputs.values()])
        path = os.path.join(HERE, 'model_%s.npz' % name)
        np.savez_compressed(path, **out)
        print('%-14s %7.1f KB  eval loss %.6f train loss %.6f' % (name, os.path.getsize(path) / 1024, out['eval_loss'],
                                                                out['train_loss']))


if __name__ == '__main__':
    main()
