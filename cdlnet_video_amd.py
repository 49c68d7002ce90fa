"""Import shim: `import cdlnet_video_amd` loads the package that lives in `cdlnet-video_amd/`
(a hyphen is not importable as written)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cdlnet-video_amd")
_spec = importlib.util.spec_from_file_location(
    "cdlnet_video_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cdlnet_video_amd"] = _mod
_spec.loader.exec_module(_mod)
