"""Registers the hyphenated package directory `is-vins_amd/` as the importable module `isvins_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "is-vins_amd")


def load():
    if "isvins_amd" in sys.modules:
        return sys.modules["isvins_amd"]
    spec = importlib.util.spec_from_file_location(
        "isvins_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["isvins_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
