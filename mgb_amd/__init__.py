"""Importable alias of the product package ``multigridbarriermpi.jl_amd/`` (a directory name with a dot cannot be imported
by name).  The real package is loaded by the regular import machinery from its own ``__init__.py`` and takes this module's
place in ``sys.modules``: ``import mgb_amd`` and ``from mgb_amd import _lib`` resolve inside that directory."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multigridbarriermpi.jl_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_pkg, "__init__.py"), submodule_search_locations=[_pkg])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
