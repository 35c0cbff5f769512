import os
import sys

import pytest

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # see transoar_amd/__init__.py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
