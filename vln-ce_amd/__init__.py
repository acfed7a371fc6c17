"""vlnce_amd: MI355X-native implementation of VLN-CE's per-step policy hot
path (vlnce_baselines/models/* of jacobkrantz/VLN-CE) behind the reference's
policy plugin surface.  Python host code on PyTorch-ROCm calling hand-written
CDNA4 kernels through the C ABI of libvlnce_hip.so (include/vlnce_hip.h)."""
from .aux_losses import AuxLosses  # noqa: F401
from .config import make_config, make_spaces  # noqa: F401
from .registry import baseline_registry, build_model  # noqa: F401
from .seq2seq_policy import Seq2SeqPolicy  # noqa: F401
from .cma_policy import CMAPolicy  # noqa: F401
from .waypoint_policy import WaypointPolicy  # noqa: F401
