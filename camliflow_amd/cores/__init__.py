"""Host-side mirror of the reference's model cores: the orchestration that drives the operators of
``camliflow_amd.csrc``.  Model names follow ``factory.py:21-35``."""
from .camliraft import CamLiRAFT, CamLiRAFT_Core, CamLiRAFT_L  # noqa: F401
from .camlipwc import CamLiPWC, CamLiPWC_Core, CamLiPWC_L, PWC, RAFT  # noqa: F401
from .raft3d import CamLiRAFT_L_Core  # noqa: F401
from .pwc3d import CamLiPWC_L_Core  # noqa: F401
from .raft2d import RAFTCore  # noqa: F401
from .pwc2d import PWCCore  # noqa: F401

MODELS = {'pwc': PWC, 'raft': RAFT, 'camlipwc': CamLiPWC, 'camlipwc_l': CamLiPWC_L,
          'camliraft': CamLiRAFT, 'camliraft_l': CamLiRAFT_L}


def model_factory(cfgs):
    """Name -> class dispatch of the reference's ``factory.model_factory`` (factory.py:21-35)."""
    if cfgs.name not in MODELS:
        raise NotImplementedError('Unknown model: %s' % cfgs.name)
    return MODELS[cfgs.name](cfgs)
