"""Registers the package directory `physically-based-rendering_amd/` (not an importable
identifier) as module `pbr_amd`."""
import importlib.util
import os
import sys

_NAME = "pbr_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "physically-based-rendering_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    module = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = module
    try:
        spec.loader.exec_module(module)
    except BaseException:
        sys.modules.pop(_NAME, None)
        raise
    return module
