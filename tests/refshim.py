"""Import shim for the upstream reference (BUILD CONTAINER ONLY; /root/reference never travels).

Recipe of SURVEY.md appendix C: bypass ``models/__init__.py`` (it imports mmdet) by registering a
bare package object, and stub ``mmdet`` / ``mmcv`` with this repo's own ResNet trunk.  Used by
tests/golden/make_golden.py (fixture generation) and by the tests marked ``needs_reference`` --
those skip automatically when /root/reference is absent (e.g. on the GPU box).
"""
import logging
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'models'))


def install():
    """Make ``import models.<x>`` resolve to the reference tree.  Idempotent."""
    if 'models' in sys.modules and getattr(sys.modules['models'], '_camli_shim', False):
        return
    if not reference_available():
        raise RuntimeError('reference tree not present')
    from camliflow_amd.cores.resnet import ResNetTrunk

    sys.path.insert(0, REFERENCE_ROOT)
    pkg = types.ModuleType('models')
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'models')]
    pkg._camli_shim = True
    sys.modules['models'] = pkg
    for name in ['mmdet', 'mmdet.models', 'mmdet.models.backbones', 'mmcv', 'mmcv.utils', 'mmcv.utils.logging']:
        sys.modules.setdefault(name, types.ModuleType(name))

    class _ResNet(ResNetTrunk):
        def __init__(self, depth, num_stages, strides, dilations, out_indices, norm_eval, with_cp, init_cfg):
            super().__init__(depth=depth, num_stages=num_stages, strides=strides, norm_eval=norm_eval)

    sys.modules['mmdet.models.backbones'].ResNet = _ResNet
    sys.modules['mmcv.utils.logging'].get_logger = lambda name: logging.getLogger(name)
