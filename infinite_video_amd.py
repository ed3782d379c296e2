"""Import alias: the package lives in the directory ``infinite-video_amd/`` (the layout the
build contract names); a hyphen is not importable, so ``import infinite_video_amd`` lands
here and is redirected to that directory's ``__init__.py``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "infinite-video_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
