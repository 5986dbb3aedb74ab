"""Host-side mirror of the reference's model cores: the orchestration that drives the operators of
``camliflow_amd.csrc``."""
from .camliraft import CamLiRAFT, CamLiRAFT_Core, CamLiRAFT_L  # noqa: F401
from .raft3d import CamLiRAFT_L_Core  # noqa: F401
from .raft2d import RAFTCore  # noqa: F401
