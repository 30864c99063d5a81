"""Importable alias for the package directory `webgraph-big_amd/` (a hyphen is not a Python identifier)."""
import importlib
import sys

_pkg = importlib.import_module("webgraph-big_amd")
sys.modules[__name__] = _pkg
