"""Import alias: the package directory is ``llama.cpp_amd/`` (not a valid Python identifier), so
``import llama_cpp_amd`` loads it from there under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llama.cpp_amd")
_spec = importlib.util.spec_from_file_location(
    "llama_cpp_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["llama_cpp_amd"] = _mod
_spec.loader.exec_module(_mod)
