"""ROCm's graph packet capture and the captured training step.  Importing transoar_amd does NOT
touch the process environment (round-3 VERDICT); it reports whether the process has the safe setting, offers
use_safe_graph_replay() for entry points that want captured steps, and TrainStep.capture refuses otherwise."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = ("import os, transoar_amd; "
         "print(os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), transoar_amd.graph_replay_safe())")
PROBE_OPT_IN = ("import os, transoar_amd; transoar_amd.use_safe_graph_replay(); "
                "print(os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), transoar_amd.graph_replay_safe())")


def _run(value, probe=PROBE):
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    if value is not None:
        env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = value
    r = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


def test_import_leaves_the_environment_alone():
    assert _run(None) == "None False"


def test_explicit_off_is_safe():
    assert _run("0") == "0 True"


def test_explicit_on_is_respected_and_flagged():
    assert _run("1") == "1 False"


def test_opt_in_before_hip_starts():
    assert _run(None, PROBE_OPT_IN) == "0 True"


PROBE_LATE_WRITE = ("import os, transoar_amd; os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'; "
                    "print(os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), transoar_amd.graph_replay_safe())")


def test_writing_the_variable_after_import_does_not_count():
    """The runtime reads the switch once; a string written behind the package's back says nothing about what HIP saw
    (round-4 advisor: the guard must record the opt-in, not re-read the environment)."""
    assert _run(None, PROBE_LATE_WRITE) == "0 False"
    assert _run("1", PROBE_LATE_WRITE) == "0 False"
