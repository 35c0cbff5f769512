"""Recorder of observed errors (see tests/conftest.py): one dict per process, written at session end."""
import json
import os

_OBSERVED = {}


def observe(key, value, bound=None):
    value = float(value)
    cur = _OBSERVED.get(key)
    if cur is None or value > cur["max"]:
        _OBSERVED[key] = {"max": value, "bound": None if bound is None else float(bound)}
    return value


def dump(path):
    if not _OBSERVED:
        return
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        for k, v in _OBSERVED.items():
            if k not in old or v["max"] > old[k]["max"]:
                old[k] = v
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass
