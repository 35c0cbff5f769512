"""transoar_amd: MI355X (gfx950) native hot path for TransOAR's 3-D detection
transformer -- the multi-scale deformable attention operator and the model
glue around it.  Importing the package loads the HIP library eagerly and
raises if it is not built (no fallbacks)."""
import os as _os

# ROCm 7.x replays HIP graphs from AQL packets it pre-records at instantiation ("graph packet
# capture").  With that on, the captured training step (train_step.TrainStep.capture) computes wrong
# gradients from the second replay on and faults around the 34th; with it off the
# same graph is exact.  The runtime reads the switch at the first HIP call of the process (measured:
# setting it after `import torch` but before any device work is early enough).
import sys as _sys

_torch = _sys.modules.get("torch")
_hip_up = _torch is not None and _torch.cuda.is_initialized()      # the switch is read lazily, at the first HIP call
if not _hip_up:
    _os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
#: False when the process initialised HIP with packet capture on: TrainStep.capture refuses then
GRAPH_REPLAY_SAFE = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"

from . import _native  # noqa: F401  (fail loudly when the .so is missing)
from . import msda as MSDA  # noqa: F401
from .ms_deform_attn import MSDeformAttn, MSDeformAttnFunction  # noqa: F401
