"""Import alias for the hyphen-named package directory.

The product package lives in
``c-users-sayakdutta-self-supervised-arbitrary-scale-point-cloud-upsampling-via-snn_amd/``
(the name the build contract fixes).  Hyphens are not legal in a Python
identifier, so this loader registers that directory as the importable package
``sapcu_amd``.  Usage: ``import sapcu_amd`` from the repo root.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(
    os.path.dirname(os.path.abspath(__file__)),
    "c-users-sayakdutta-self-supervised-arbitrary-scale-point-cloud-upsampling-via-snn_amd",
)

_spec = importlib.util.spec_from_file_location(
    "sapcu_amd",
    os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR],
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sapcu_amd"] = _mod
_spec.loader.exec_module(_mod)
