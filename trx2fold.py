"""Importable alias of the package directory `trrosettax2-dynamics_amd` (a hyphen is not a valid identifier)."""
import importlib
import sys

_pkg = importlib.import_module("trrosettax2-dynamics_amd")
sys.modules[__name__] = _pkg
