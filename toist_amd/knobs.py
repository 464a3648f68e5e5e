"""Tuning switches of the Python side, gated like the C side's (`make KNOBS=1`, csrc/common.h: tuning_knob).

A product process ignores the environment: every switch has the default that the A/B runs under tools/ settled on.  Only when
TOIST_KNOBS=1 is exported (A/B experiments inside one gpurun call) are the TOIST_* variables read -- once, at import."""
import os

ENABLED = os.environ.get("TOIST_KNOBS", "0") == "1"


def knob(name, default):
    """value of the environment variable `name` (typed like `default`) when TOIST_KNOBS=1, else `default`"""
    if not ENABLED:
        return default
    raw = os.environ.get(name)
    if raw is None:
        return default
    if isinstance(default, bool):
        return raw != "0"
    if isinstance(default, int):
        return int(raw)
    return raw
