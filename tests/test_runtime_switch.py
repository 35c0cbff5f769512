"""transoar_amd switches ROCm's graph packet capture off before HIP starts and
remembers whether it was in time; TrainStep.capture refuses otherwise."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = ("import os, transoar_amd; "
         "print(os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), transoar_amd.GRAPH_REPLAY_SAFE)")


def _run(value):
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    if value is not None:
        env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = value
    r = subprocess.run([sys.executable, "-c", PROBE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


def test_unset_is_switched_off():
    assert _run(None) == "0 True"


def test_explicit_off_is_kept():
    assert _run("0") == "0 True"


def test_explicit_on_is_respected_and_flagged():
    assert _run("1") == "1 False"
