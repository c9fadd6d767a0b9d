"""Importable alias of the product package ``multigridbarriermpi.jl_amd/`` (a directory name with a
dot cannot be imported directly).  All code lives there; this file only redirects the package path."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multigridbarriermpi.jl_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
