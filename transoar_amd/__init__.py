"""transoar_amd: MI355X (gfx950) native hot path for TransOAR's 3-D detection
transformer -- the multi-scale deformable attention operator and the model
glue around it.  Importing the package loads the HIP library eagerly and
raises if it is not built (no fallbacks)."""
import os as _os

# ROCm 7.x replays HIP graphs from AQL packets it pre-records at instantiation ("graph packet
# capture").  With that on, the captured training step (train_step.TrainStep.capture) computes wrong
# gradients from the second replay on and faults around the 34th; with it off the
# same graph is exact.  The runtime reads the switch when libamdhip64 initialises, so it only takes
# effect if this package (or the variable) comes before the first HIP call of the process.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from . import _native  # noqa: F401  (fail loudly when the .so is missing)
from . import msda as MSDA  # noqa: F401
from .ms_deform_attn import MSDeformAttn, MSDeformAttnFunction  # noqa: F401
