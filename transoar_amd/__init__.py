"""transoar_amd: MI355X (gfx950) native hot path for TransOAR's 3-D detection
transformer -- the multi-scale deformable attention operator and the model
glue around it.  Importing the package loads the HIP library eagerly and
raises if it is not built (no fallbacks)."""
from . import _native  # noqa: F401  (fail loudly when the .so is missing)
from . import msda as MSDA  # noqa: F401
from .ms_deform_attn import MSDeformAttn, MSDeformAttnFunction  # noqa: F401
