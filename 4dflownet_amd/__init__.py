"""MI355X-native 4DFlowNet hot path (package directory `4dflownet_amd`; import it with
importlib.import_module("4dflownet_amd") or through the alias module name `fdn_amd`).

Host side in Python mirroring the reference's surface (SR4DFlowNet, TrainerController, PatchHandler3D,
PatchGenerator); every FLOP runs in hand-written HIP behind the C-ABI of include/fdn.h."""
import sys as _sys

from . import _lib, ops            # noqa: F401
from ._lib import FdnError         # noqa: F401

_sys.modules.setdefault("fdn_amd", _sys.modules[__name__])
