"""Importable name of the hyphen-named package directory `bevfusion-3d_object_detection_amd/`.

    import bevfusion_amd
    from bevfusion_amd.ops import bev_pool, Voxelization

The directory is loaded ONCE under the canonical name `bevfusion_amd`; the literal directory name
(`importlib.import_module("bevfusion-3d_object_detection_amd[.sub]")`) is redirected to the same module
objects, so there is never a second copy of a submodule (registries, workspaces, the ctypes handle).
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_DIRNAME = "bevfusion-3d_object_detection_amd"
_CANON = "bevfusion_amd"
_PKG_DIR = os.path.join(_ROOT, _DIRNAME)


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _DIRNAME or fullname.startswith(_DIRNAME + "."):
            canon = _CANON + fullname[len(_DIRNAME):]
            return importlib.util.spec_from_loader(fullname, _AliasLoader(canon))
        return None


def _load():
    this = sys.modules.get(__name__)
    if getattr(this, "__path__", None):  # already the package
        return this
    spec = importlib.util.spec_from_file_location(_CANON, os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules[_CANON] = pkg
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())
    spec.loader.exec_module(pkg)
    return pkg


_load()
