import os
import sys

import pytest

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # see transoar_amd/__init__.py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# ---- MIOpen on a fresh GPU box.  The fp32 runs of the whole model and the fp32 reference convolutions next to the
# hand-written kernels go through stock (MIOpen) convolutions; with empty user databases MIOpen searches its solvers and
# compiles kernels for every new problem: 20 s to 2 min per layer shape, 730 of the 907 s the GPU suite took in round 5.
# tests/miopen_db/ holds the user find-db and kernel cache of one full run of this suite on an MI355X (same image, so the same
# MIOpen build); each session works on a private copy.  (torch's native convolution path, cudnn.enabled = False, would need
# neither -- and returns garbage gradients for the full-resolution layers: measured, not used.)  The product's bf16 path
# never calls a stock convolution, so this is about the reference side of the tests only.
_MIOPEN_SEED = os.path.join(ROOT, "tests", "miopen_db")
_MIOPEN_TMP = None
if os.path.isdir(_MIOPEN_SEED) and "MIOPEN_USER_DB_PATH" not in os.environ and os.path.exists("/dev/kfd"):
    import shutil
    import tempfile
    _MIOPEN_TMP = tempfile.mkdtemp(prefix="transoar_miopen_")
    for _f in os.listdir(_MIOPEN_SEED):
        shutil.copy(os.path.join(_MIOPEN_SEED, _f), _MIOPEN_TMP)
    os.environ["MIOPEN_USER_DB_PATH"] = _MIOPEN_TMP
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = _MIOPEN_TMP


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- observed errors: tests call tests._observe.observe(key, value, bound) next to the assertion that bounds `value`; the
# session writes the maxima to gpurun_out/observed_errors.json, from where they are copied to profiles/ and the bounds
# are set at ~2x (round-4 VERDICT item 9: "bounds nobody has looked under")
def pytest_sessionfinish(session, exitstatus):
    from tests import _observe
    _observe.dump(os.path.join(ROOT, "gpurun_out", "observed_errors.json"))
    if _MIOPEN_TMP and os.environ.get("TRANSOAR_SAVE_MIOPEN_DB"):       # refresh tests/miopen_db from this run (copy by hand)
        import shutil
        shutil.copytree(_MIOPEN_TMP, os.path.join(ROOT, "gpurun_out", "miopen_db_final"), dirs_exist_ok=True)
