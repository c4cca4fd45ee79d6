"""Imports the package directory `ray-tracing_amd/` (not a valid identifier) as module `raytracing_amd`."""
import importlib.util
import os
import sys

_NAME = "raytracing_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ray-tracing_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(root, "__init__.py"),
                                                  submodule_search_locations=[root])
    module = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = module
    spec.loader.exec_module(module)
    return module
