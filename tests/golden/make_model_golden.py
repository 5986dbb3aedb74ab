"""Model-level golden vectors: the REFERENCE's model code (models/*.py, unchanged) driven by
operators with the native kernels' index semantics (tests/refmodels.py substitutes the oracle-backed
ops for models.csrc -- the "load unchanged" arrangement of SURVEY 8b), deterministic name-hashed
weights, seeded synthetic inputs.  Stored: final flows, loss, a gradient fingerprint; and, in
model_<name>_core_inputs.npz, the tensors the reference handed to its ``core`` (padded + normalised images, clouds
after the IDS transform) -- the GPU tests feed exactly these to the HIP cores, so that GPU <-> reference is ONE hop
(FPS is a chain of 4096 arg-max decisions and only reproducible on bit-identical inputs; the IDS transform's log /
divide differ in the last ulp between CPU and GPU).

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
        captured = {}

        def grab(_mod, args):
            for i, a in enumerate(args):
                if torch.is_tensor(a):
                    captured['arg%d' % i] = a.detach().clone().numpy()
        hook = model.core.register_forward_pre_hook(grab)
        # the clouds every point model hands to build_pc_pyramid (= FPS) -- after the IDS transform where there is one
        patched = []
        for mod in [m for n, m in sys.modules.items() if n.startswith('models.') and hasattr(m, 'build_pc_pyramid')]:
            original = mod.build_pc_pyramid

            def recording(pc1, pc2, *a, _orig=original, **kw):
                captured.setdefault('pyr_pc1', pc1.detach().clone().numpy())
                captured.setdefault('pyr_pc2', pc2.detach().clone().numpy())
                return _orig(pc1, pc2, *a, **kw)
            mod.build_pc_pyramid = recording
            patched.append((mod, original))
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
        hook.remove()
        for mod, original in patched:
            mod.build_pc_pyramid = original
        out['input_checksum'] = np.array([float(v.double().sum()) for v in inputs.values()])
        if captured:
            np.savez_compressed(os.path.join(HERE, 'model_%s_core_inputs.npz' % name), **captured)
        path = os.path.join(HERE, 'model_%s.npz' % name)
        if os.path.exists(path) and os.environ.get('CAMLI_REWRITE_MODEL_GOLDEN') != '1':
            old = np.load(path)     # flows / losses / fingerprints are already committed: check, do not rewrite
            for k in out:
                same = np.array_equal(old[k], out[k]) if out[k].dtype.kind in 'US' else np.allclose(old[k], out[k], rtol=1e-4, atol=1e-6)   # threaded CPU reductions
                assert same, 'regenerated %s[%s] differs from the committed fixture' % (name, k)
        else:
            np.savez_compressed(path, **out)
        print('%-14s %7.1f KB  eval loss %.6f train loss %.6f' % (name, os.path.getsize(path) / 1024, out['eval_loss'],
                                                                out['train_loss']))


if __name__ == '__main__':
    main()
