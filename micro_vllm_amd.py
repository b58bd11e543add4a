"""Import shim: exposes the directory `micro-vllm_amd/` as the package `micro_vllm_amd`.

A hyphen cannot appear in a Python module name, so this file loads the package's
__init__.py under the importable name and replaces itself with it in sys.modules.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro-vllm_amd")
_spec = importlib.util.spec_from_file_location(
    "micro_vllm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["micro_vllm_amd"] = _mod
_spec.loader.exec_module(_mod)
