"""Import shim: the package directory is named ``quantumpropagators.jl_amd`` (not a
valid Python identifier), so it is loaded here under the module name ``qprop_amd``.
``import qprop_amd`` from the repo root gives the package; sub-modules resolve as
``qprop_amd.synth`` etc."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quantumpropagators.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "qprop_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["qprop_amd"] = _mod
_spec.loader.exec_module(_mod)
