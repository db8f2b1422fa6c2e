"""Loader for the `leg-kilo_amd/` package directory (the hyphen makes it un-importable by name).

`import lk_pkg; lk = lk_pkg.load()` registers it as the module `legkilo_amd`, after which
`import legkilo_amd.synth` etc. work normally.
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, "leg-kilo_amd")


def load():
    if "legkilo_amd" in sys.modules:
        return sys.modules["legkilo_amd"]
    spec = importlib.util.spec_from_file_location(
        "legkilo_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules["legkilo_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
