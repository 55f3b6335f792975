"""Import shim: the package directory is ``soft-rendering-toolsets_amd`` (hyphenated, as the
project layout prescribes), which is not a valid Python identifier.  ``import srt_amd`` loads it
under the module name ``soft_rendering_toolsets_amd`` and re-exports its public names."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "soft-rendering-toolsets_amd")
_NAME = "soft_rendering_toolsets_amd"

if _NAME in sys.modules:
    _mod = sys.modules[_NAME]
else:
    _spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR]
    )
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)

globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
