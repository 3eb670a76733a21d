"""Loads cpecan-signal_amd/binding.py (the directory name is not an importable identifier)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def binding():
    if "cpecan_binding" in sys.modules:
        return sys.modules["cpecan_binding"]
    path = os.path.join(ROOT, "cpecan-signal_amd", "binding.py")
    spec = importlib.util.spec_from_file_location("cpecan_binding", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cpecan_binding"] = mod
    spec.loader.exec_module(mod)
    return mod


def em():
    """cpecan-signal_amd/em.py: the Baum-Welch loop driver (host control only)"""
    if "cpecan_em" in sys.modules:
        return sys.modules["cpecan_em"]
    path = os.path.join(ROOT, "cpecan-signal_amd", "em.py")
    spec = importlib.util.spec_from_file_location("cpecan_em", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cpecan_em"] = mod
    spec.loader.exec_module(mod)
    return mod
